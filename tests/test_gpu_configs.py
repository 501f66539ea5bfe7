"""GPU parity at the sizes BASELINE.json's `configs` state (configs[1], [3], [4] were only exercised at reduced
sizes).  Each test names the config it covers; all comparisons are bit-exact against the oracle (or, for configs[3], against the
plaintext regression the reference's own Test_Regression compares with, Regression.h:193-214)."""
import os
import subprocess

import numpy as np
import pytest

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config1_ntt_round_trip_n8192_eight_primes():
    """configs[1]: single-GPU DoubleCRT NTT, n = 2^13, the exact chain of SURVEY 8(d) item 2 = the first 8 primes = 1 mod 2^15
    descending from 2^60; forward vs oracle, inverse vs oracle, forward-then-inverse = identity, on a batch."""
    m, n, L, B = 1 << 14, 1 << 13, 8, 33
    primes, roots = P.first_primes(m, L)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    rng = np.random.default_rng(42)
    rows = P.rand_rows(rng, primes, n, B)
    rows[0, 0, :] = 0
    rows[1, 3, :] = np.uint64(primes[3] - 1)
    buf = ctx.upload(rows)
    ctx.rows_ntt_fwd(buf, B)
    got = buf.download(rows.shape)
    for c in (0, 1, B - 1):                      # first, edge and last DoubleCRT of the batch, every prime
        for i in range(L):
            assert np.array_equal(got[c, i], orc.fft_residues(i, rows[c, i])), (c, i)
    ctx.rows_ntt_inv(buf, B)
    assert np.array_equal(buf.download(rows.shape), rows)
    ev = P.rand_rows(rng, primes, n, 2)
    b2 = ctx.upload(ev)
    ctx.rows_ntt_inv(b2, 2)
    back = b2.download(ev.shape)
    for c in range(2):
        for i in range(L):
            assert np.array_equal(back[c, i], orc.cmod_ifft(i, ev[c, i])), (c, i)


def test_config4_stress_ring_several_chunks():
    """configs[4]: n = 2^15, logQ = 1024 (35 primes, 43 digits).  A batch of 35 multiplications = 3 chunks of the library at that ring
    (16-17 ciphertexts each): ciphertext 0 and the last one are checked against the oracle, and every ciphertext against the same
    pair multiplied alone (a chunk boundary must not change a result)."""
    m, logQ, p, count = 1 << 16, 1024, 65537, 35
    primes, roots = P.chain_for(m, logQ, p)
    assert len(primes) == 35
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(9)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    a[count - 1, 0, 0] = O.ints_to_limbs([-(1 << (logQ - 1))], nl)[0]
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert ctx.get_option("batch_chunk") == 0        # the library's own chunking (about 16 ciphertexts per chunk at this ring)
    for c in (0, count - 1):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    for c in (15, 16, 17, 33, 34):                   # around the chunk boundaries, each multiplied alone
        assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a[c:c + 1], b[c:c + 1])[0], got[c]), c


def test_config4_generated_keys_column_parts():
    """configs[4] with a GENERATED key-switch matrix (KeySwitchSI::Init, FHE-SI.cpp:153-226): 15 centred limbs, 129 columns.  The dot product
    takes the columns in two parts through one 80 KB tile of 8 ciphertexts (dot32_kernel2p).  A ragged batch (two tiles, the second with one
    ciphertext): first, last-of-tile and ragged ciphertexts vs the oracle, the whole batch vs the general limbs (30 of them: tiles of 4
    ciphertexts through dot32_kernel2)."""
    m, logQ, p, count = 1 << 16, 1024, 65537, 9
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    seed, pub = 0x0123456789ABCDEF, 0x0FEDCBA987654321
    one = np.zeros((n, 1), dtype=np.uint64)
    one[0, 0] = 1
    d_one = F.DoubleCRT.from_poly(ctx, one)
    t = F.DoubleCRT(ctx).sample(0, 64, seed, 1)
    t2 = t.copy()
    t2.op(t, 2)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded([d_one, t, t2], t, logQ, seed, pub, 5000, 3)
    ksm = ksk.download()
    rng = np.random.default_rng(44)
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    form, rows, _ = ksk.form()
    assert form == 1 and rows == 15 and ksk.key_bits()[0], (form, rows, ksk.key_bits())
    assert "dot32_kernel2p" in ctx_kernel_name(ctx, ksk, logQ, p, a, b, nl)
    for c in (0, 7, 8):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    ctx.set_option("ks_long_keys", 1)
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b), got)


def ctx_kernel_name(ctx, ksk, logQ, p, a, b, nl):
    """name of the dot-product kernel a device-resident call launches (read back from the library's stopwatch)"""
    da, db, dout = ctx.upload(a[:1]), ctx.upload(b[:1]), ctx.alloc(a[:1].nbytes)
    ctx.prof_enable(True)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, 1)
    ctx.sync()
    name = ctx.prof_kernel_name("dot")
    ctx.prof_enable(False)
    return name


def test_config4_1024_concurrent_mults():
    """configs[4] as BASELINE.json words it: 1024 concurrent ciphertext mults at the stress shape (n = 2^15, logQ = 1024) in ONE call on
    device-resident buffers (24 GiB of operands and results; the library takes them in launches of up to 292).  64 distinct random pairs are
    replicated on the device to fill the batch (as bench.py --workload stress does): every one of the 1024 results must equal the result of
    the same pair in a call of its own 64 -- a launch or chunk boundary, or a neighbour, must not change a ciphertext -- and pair 0 the oracle's."""
    m, logQ, p, count, uniq = 1 << 16, 1024, 65537, 1024, 64
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(1024)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    a = P.rand_limbs(rng, (uniq, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (uniq, 2, n), nl, logQ)
    ct_bytes = a.nbytes // uniq
    da, db, dout = ctx.alloc(ct_bytes * count), ctx.alloc(ct_bytes * count), ctx.alloc(ct_bytes * count)
    da.upload(a)
    db.upload(b)
    done = uniq
    while done < count:
        cnt = min(done, count - done)
        ctx.dev_copy(da.ptr.value + done * ct_bytes, da.ptr.value, cnt * ct_bytes)
        ctx.dev_copy(db.ptr.value + done * ct_bytes, db.ptr.value, cnt * ct_bytes)
        done += cnt
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
    ctx.sync()
    want = ctx.ct_mul_relin(ksk, logQ, p, a, b)          # the 64 distinct pairs in a call of their own
    assert np.array_equal(want[0], orc.ct_mul_relin(ksm, a[0], b[0], logQ, p))
    for blk in range(count // uniq):
        got = np.empty_like(want)
        import ctypes as C
        from fhe_si_amd import binding as Bd
        Bd._ck(Bd._load().fhesi_dev_download(ctx.h, got.ctypes.data_as(C.c_void_p), C.c_void_p(dout.ptr.value + blk * uniq * ct_bytes), got.nbytes))
        assert np.array_equal(got, want), blk


def test_config2_metric_ring_several_chunks():
    """configs[2] at a batch that spans three chunks of 64: first, last and chunk-boundary ciphertexts vs the oracle."""
    m, logQ, p, count = 1 << 15, 512, 23, 130
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(2)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    for c in (0, 63, 64, 128, 129):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c


@pytest.mark.parametrize("bits", [30, 29])
@pytest.mark.parametrize("m,logQ,p", [(1 << 15, 512, 23), (1006, 512, 23), (1 << 16, 1024, 65537)])
def test_rounding_boundaries_take_the_exact_pass(m, logQ, p, bits):
    """configs[2] (and the generic form on a linear-convolution ring, and configs[4]'s words of 26 bits), the tensor half's CRT (crt32_scale_kernel, kernels_tensor32.hip): it forms only the words of x from bit 392 upwards and flags a
    workgroup whose bits logQ-64 .. logQ-1 read 0x7fff...f for a second pass with every word.  Coefficients of a . b placed ON ScaleDown's rounding
    boundary (Ciphertext.cpp:205-213: x + 2^(logQ-1) within a few units of a multiple of 2^logQ) must come out as the oracle rounds them -- and must
    NOT with the second pass switched off, so the inputs really exercise it (the host-side model of the same window: tests/test_crt32_model.py)."""
    count = 1
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    ctx.set_option("tensor_bits", bits)               # (primes below 2^30, or below 2^29 with M only 1.5 bits above the bound: kappa's rounding has less room there)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(77)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    mod = 1 << logQ
    inv_p = pow(p, -1, mod)

    def centred(v):
        v %= mod
        return v - mod if v >= mod // 2 else v
    # c0 = (p a0) . b0 with b0 = 1: coefficient j of the tensor product is p * A_j, and (p A_j + 2^(logQ-1)) mod 2^logQ = delta_j
    deltas = [0, 1, -1, 2, -2, 3, -3, 5, -7, 8, 100, -100, 1 << 40, -(1 << 40), (1 << (logQ - 122)) + 12345, -(1 << (logQ - 121))]
    A = [centred((d - (mod >> 1)) * inv_p) for d in deltas]
    a = np.zeros((count, 2, n, nl), dtype=np.uint64)
    b = np.zeros((count, 2, n, nl), dtype=np.uint64)
    a0 = [0] * n
    for k, v in enumerate(A):
        a0[(k * 1021 + 5) % n] = v                   # spread over many workgroups of the CRT kernel (128 coefficients each)
    a[0, 0] = O.ints_to_limbs(a0, nl)
    a[0, 1] = P.rand_limbs(rng, (n,), nl, logQ)
    b[0, 0] = O.ints_to_limbs([1] + [0] * (n - 1), nl)
    b[0, 1] = P.rand_limbs(rng, (n,), nl, logQ)
    want = orc.ct_mul_relin(ksm, a[0], b[0], logQ, p)
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b)[0], want)
    ctx.set_option("crt_skip_cleanup", 1)
    assert not np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b)[0], want)
    ctx.set_option("crt_skip_cleanup", 0)
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b)[0], want)


def test_config2_generated_keys_several_chunks():
    """configs[2] with the key-switch matrix EVERY reference driver holds -- KeySwitchSI(secretKey) of a sampleHWt(64) key (Test_AddMul.cpp:48-52
    -> FHE-SI.cpp:153-226), generated on the device as bench.py does -- at a batch that spans three chunks of 64.  The library measures the
    matrix and runs the 7 centred limbs (the form the headline times); first, last and chunk-boundary ciphertexts vs the oracle on the
    downloaded matrix, and the whole batch again through the general limbs (option ks_long_keys) bit for bit."""
    m, logQ, p, count = 1 << 15, 512, 23, 130
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    seed, pub = 0x5EC2E7C0FFEE1234, 0x1234ABCD9876FEDC
    one = np.zeros((n, 1), dtype=np.uint64)
    one[0, 0] = 1
    d_one = F.DoubleCRT.from_poly(ctx, one)
    t = F.DoubleCRT(ctx).sample(0, 64, seed, 1)                 # sampleHWt(64)
    t2 = t.copy()
    t2.op(t, 2)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded([d_one, t, t2], t, logQ, seed, pub, 1000, 3)
    ksm = ksk.download()
    rng = np.random.default_rng(5)
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    form, rows, limb_bits = ksk.form()
    centred, key_bits = ksk.key_bits()
    assert form == 1 and rows == 7 and centred and key_bits <= logQ - 1, (form, rows, limb_bits, centred, key_bits)      # (1 = four 30-bit auxiliary primes, limbs)
    for c in (0, 63, 64, 127, 128, 129):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    # the dot product ran with the keys in LDS and the digits in registers (dot32_kernel4); the digit-tile form (dot32_kernel2) gives the same bits,
    # also on a ragged batch (the last workgroup's waves past the end) and on a batch below the new kernel's threshold
    da, db, dout = ctx.upload(a), ctx.upload(b), ctx.alloc(a.nbytes)
    ctx.prof_enable(True)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
    ctx.sync()
    assert "dot32_kernel4<7, 6" in ctx.prof_kernel_name("dot"), ctx.prof_kernel_name("dot")
    ctx.prof_enable(False)
    assert np.array_equal(dout.download((count, 2, n, nl)), got)
    for cnt in (49, 25, 9):
        ctx.set_option("dot32_k4", 1)
        ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, cnt)
        new = dout.download((cnt, 2, n, nl))
        ctx.set_option("dot32_k4", 0)
        ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, cnt)
        assert np.array_equal(dout.download((cnt, 2, n, nl)), new) and np.array_equal(new, got[:cnt]), cnt
    ctx.set_option("dot32_k4", 1)
    ctx.set_option("ks_long_keys", 1)
    again = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert ksk.form()[1] == 15 and np.array_equal(again, got)


def test_refring_generated_keys_eight_limbs():
    """The metric's multiplication on the reference drivers' own ring at its size (Test_AddMul.cpp:131: m = p - 1 = 32602, phi(m) = 16300,
    logQ = 512; padded rows of 2^15) with a generated key matrix: 8 centred limbs of 72 bits -- the dot32_kernel4<8, 4, ...> instantiation.
    A ragged batch of 25 against the digit-tile form (dot32_kernel2, option dot32_k4 = 0) bit for bit, and one ciphertext against the oracle
    (Bluestein mode: ~20 s)."""
    m, logQ, p, count = 32602, 512, 32603, 25
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    one = np.zeros((n, 1), dtype=np.uint64)
    one[0, 0] = 1
    t = F.DoubleCRT(ctx).sample(0, 64, 31, 1)
    t2 = t.copy()
    t2.op(t, 2)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded([F.DoubleCRT.from_poly(ctx, one), t, t2], t, logQ, 31, 32, 7000, 3)
    rng = np.random.default_rng(77)
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    da, db, dout = ctx.upload(a), ctx.upload(b), ctx.alloc(a.nbytes)
    ctx.prof_enable(True)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
    ctx.sync()
    assert ksk.form()[1] == 8 and ksk.key_bits()[0], (ksk.form(), ksk.key_bits())
    assert "dot32_kernel4<8, 4" in ctx.prof_kernel_name("dot"), ctx.prof_kernel_name("dot")
    ctx.prof_enable(False)
    got = dout.download((count, 2, n, nl))
    ctx.set_option("dot32_k4", 0)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
    assert np.array_equal(dout.download((count, 2, n, nl)), got)
    ctx.set_option("dot32_k4", 1)
    orc = O.Oracle(m, primes, roots)
    orc.set_bluestein_fft(True)
    assert np.array_equal(got[24], orc.ct_mul_relin(ksk.download(), a[24], b[24], logQ, p))


@pytest.mark.parametrize("logQ,limbs", [(480, 7), (536, 8)])
def test_dot32_kernel4_on_column_counts_without_a_compiled_tail(logQ, limbs):
    """dot32_kernel4 at column counts other than 66: logQ = 480 gives 60 columns (five whole chunks of 12), logQ = 536 gives 69 (five chunks + one
    padded with zero key rows: the branch-free body multiplies clamped digit loads by zeros) and 8 limbs.  Generated key matrix, a ragged batch of
    26; against the digit-tile form (option dot32_k4 = 0) bit for bit and one ciphertext against the oracle."""
    m, p, count = 1 << 15, 23, 26
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    one = np.zeros((n, 1), dtype=np.uint64)
    one[0, 0] = 1
    t = F.DoubleCRT(ctx).sample(0, 64, 91, 1)
    t2 = t.copy()
    t2.op(t, 2)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded([F.DoubleCRT.from_poly(ctx, one), t, t2], t, logQ, 91, 92, 3000, 3)
    rng = np.random.default_rng(logQ)
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    da, db, dout = ctx.upload(a), ctx.upload(b), ctx.alloc(a.nbytes)
    ctx.prof_enable(True)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
    ctx.sync()
    assert ksk.form()[1] == limbs and ksk.key_bits()[0], (ksk.form(), ksk.key_bits())
    assert f"dot32_kernel4<{limbs}," in ctx.prof_kernel_name("dot") and ctx.prof_kernel_name("dot").endswith(", 0>"), ctx.prof_kernel_name("dot")
    ctx.prof_enable(False)
    got = dout.download((count, 2, n, nl))
    ctx.set_option("dot32_k4", 0)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
    assert np.array_equal(dout.download((count, 2, n, nl)), got)
    ctx.set_option("dot32_k4", 1)
    assert np.array_equal(got[25], orc.ct_mul_relin(ksk.download(), a[25], b[25], logQ, p))


def test_config2_with_a_chain_of_50_bit_primes():
    """configs[2] as the reference's era would have built it: FHEContext.cpp:92 starts the chain at 2^NTL_SP_NBITS, 50 in NTL 5.x / 6.x,
    which gives 22 primes instead of 18.  Multiplication + relinearisation against the oracle on the first, the last and
    the sub-chunk-boundary ciphertexts; the key switch must run over the four 30-bit auxiliary primes (limb mode), not per prime."""
    m, logQ, p, count = 1 << 15, 512, 23, 66
    primes, roots = P.chain_for(m, logQ, p, 1, 50)
    assert len(primes) == 22 and max(primes).bit_length() == 50
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(50)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    a[count - 1, 0, 0] = O.ints_to_limbs([-(1 << (logQ - 1))], nl)[0]
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    form, rows, bits = ksk.form()
    assert form == 1 and rows == 15 and bits > 64, (form, rows, bits)      # the chain product has the same ~1058 bits as with 60-bit primes: 15 limbs of 74 bits
    for c in (0, 63, 64, count - 1):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    ctx.set_option("ks_direct", 1)                    # the per-prime form of the reference's own structure on the same chain
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a[:2], b[:2]), got[:2]) and ksk.form()[0] == 0
    ctx.set_option("ks_direct", 0)
    ctx.set_option("tensor32", 0)                     # ... and the tensor half over the chain itself (22 rows of 50-bit residues)
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a[:2], b[:2]), got[:2])


def test_addmul_sequence_at_the_metric_ring_with_50_bit_primes():
    """Test_AddMul's operation sequence (Test_AddMul.cpp:18-86: add, 7-fold add, mul + key switch, square + key switch, 9-fold add +
    key switch, mul, key switch) on the C++ mirror of the class surface at configs[2]'s ring, chain started at 2^50: key generation on
    the device, and every result decrypts to the plaintext value (the reference's own predicate) -- the valid-key check of this chain."""
    host = os.path.join(ROOT, "tests", "host")
    subprocess.check_call(["make", "-C", host], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(host, "test_addmul"), "512", "23", "7", "11", "--m=32768", "--sp-nbits=50", "--time"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "m=32768 phi(m)=16384 logQ=512 primes=22 first prime bits=50 ndigits=22" in r.stdout, r.stdout
    assert "Test SUCCEEDED" in r.stdout


def test_config3_regression_at_the_reference_size():
    """configs[3] at its own parameters: Test_Regression's safe prime p = 8423 (m = 8422, phi(m) = 4210, Bluestein rows with N = 2^15),
    d = 8, one data block of 4096 points (blockSize = usable slots = 4096, Test_Regression.cpp:111-121), logQ = 341 by the
    reference's noise formula (Test_Regression.cpp:100-108) => 13 primes, 15 digits, 12 automorphism keys.  RegressBatched on the
    device; theta and det decrypt to the plaintext regression (Regression.h:193-214), compared slot-wise over Z_p."""
    host = os.path.join(ROOT, "tests", "host")
    subprocess.check_call(["make", "-C", host], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(host, "test_regression"), "8423", "7", "8", "1", "1", "--batched-only", "--check=slots"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "phi(m)=4210 logQ=341 primes=13 ndigits=15 dim=8 rows=1" in r.stdout, r.stdout
    assert "automorphism keys: 12" in r.stdout
    assert "batched: decrypts to the plaintext regression: yes" in r.stdout
    assert "Test SUCCEEDED" in r.stdout


def test_config3_reference_control_flow_object_at_a_time():
    """configs[3] through the reference's OWN control flow: Regression::Regress written one Ciphertext object at a time on Matrix<Ciphertext>
    (Matrix.cpp:57-98,150-263, Regression.h:102-149,166-178; tests/host/matrix_literal.h) -- 5.5 * 10^5 partial determinants at d = 8.  The
    mirror's Ciphertext records the statements, shares equal ones and evaluates them level by level in batched device calls
    (fhe-si_amd/host/fhesi_engine.h); the ciphertexts are bit-identical to the explicit waves of RegressBatched and decrypt to the plaintext
    regression."""
    host = os.path.join(ROOT, "tests", "host")
    subprocess.check_call(["make", "-C", host], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(host, "test_regression"), "8423", "7", "8", "1", "1", "--check=slots"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "phi(m)=4210 logQ=341 primes=13 ndigits=15 dim=8 rows=1" in r.stdout, r.stdout
    assert "object at a time: decrypts to the plaintext regression: yes" in r.stdout
    assert "ciphertexts of both evaluators bit-identical: yes" in r.stdout
    assert "Test SUCCEEDED" in r.stdout
