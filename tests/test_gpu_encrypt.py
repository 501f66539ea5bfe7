"""GPU parity: FHESIPubKey::Encrypt / FHESISecKey::Decrypt batches (FHE-SI.cpp:10-36, 93-119) through the C ABI vs the C oracle
(random keys, extreme inputs) and vs the Python model with valid keys (encrypt == model bit for bit, decrypt == message)."""
import numpy as np
import pytest

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

pytestmark = pytest.mark.gpu


def dcrt_from_rows(ctx, rows):
    d = F.DoubleCRT(ctx)
    for i in range(rows.shape[0]):
        d.set_row(i, np.ascontiguousarray(rows[i]))
    return d


@pytest.mark.parametrize("m,logQ,p", [(2048, 128, 23), (46, 90, 47), (4096, 511, 65537), (64, 64, 257),
                                      (45, 90, 23), (17325, 90, 23)])      # generic m: rem Phi_m by the long division in LDS / by convolutions
def test_encrypt_decrypt_vs_oracle(m, logQ, p):
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    if m > 10000:
        orc.set_bluestein_fft(True)
    n, L, nl = ctx.phim, len(primes), (logQ + 63) // 64
    rng = np.random.default_rng(m + logQ)
    count = 3
    pk_rows = P.rand_rows(rng, primes, n, 2)                       # [2][L][n] uniform residues (parity does not need a valid key)
    pk0, pk1 = dcrt_from_rows(ctx, pk_rows[0]), dcrt_from_rows(ctx, pk_rows[1])
    rand = np.zeros((count, 3, n), dtype=np.int64)
    rand[:, 0] = rng.integers(0, 2, size=(count, n))
    rand[:, 1:] = np.rint(rng.normal(0, 3.2, size=(count, 2, n))).astype(np.int64)
    rand[0, 1, 0], rand[0, 2, 0] = -40, 40                          # far tails
    msg = rng.integers(0, p, size=(count, n)).astype(np.int64)
    msg[0, :2] = [0, p - 1]
    out = ctx.alloc(count * 2 * n * nl * 8)
    ctx.encrypt_batch(pk0, pk1, logQ, p, rand, msg, out, nl)
    got = out.download((count, 2, n, nl))
    for c in range(count):
        assert np.array_equal(got[c], orc.encrypt(pk_rows, rand[c, 0], rand[c, 1:], msg[c], logQ, p, nl)), c
    # decrypt of arbitrary ciphertexts with an arbitrary key row set, incl. the extremes of the centred range
    cts = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    cts[0, 0, 0] = O.ints_to_limbs([-(1 << (logQ - 1))], nl)[0]
    cts[0, 1, 0] = O.ints_to_limbs([(1 << (logQ - 1)) - 1], nl)[0]
    t_rows = P.rand_rows(rng, primes, n, 1)[0]
    sk1 = dcrt_from_rows(ctx, t_rows)
    dm = ctx.decrypt_batch(sk1, logQ, p, ctx.upload(cts), nl, count)
    for c in range(count):
        assert np.array_equal(dm[c], orc.decrypt(t_rows, cts[c], logQ, p)), c


@pytest.mark.parametrize("m,logQ,p", [(22, 80, 23), (64, 100, 257)])
def test_valid_keys_match_python_model_and_round_trip(m, logQ, p):
    primes, roots = P.chain_for(m, logQ, p)
    rctx = R.Ctx(m, logQ, p, primes, roots)
    ctx = F.Context(m, primes, roots)
    n, L, nl = rctx.phim, len(primes), (logQ + 63) // 64
    prng = R.SplitMix64(31 + m)
    t, pk = R.keygen(rctx, prng)
    count = 4
    msgs = [[prng.bnd(p) for _ in range(n)] for _ in range(count)]
    rand = np.zeros((count, 3, n), dtype=np.int64)
    expect = []
    for c in range(count):
        small = [prng.bnd(2) for _ in range(n)]
        noise = [R.sample_gaussian(prng, n), R.sample_gaussian(prng, n)]
        rand[c, 0], rand[c, 1], rand[c, 2] = small, noise[0], noise[1]
        expect.append(R.encrypt_with(rctx, pk, msgs[c], small, noise))
    pk0 = F.DoubleCRT.from_poly(ctx, O.ints_to_limbs(pk[0], nl))
    pk1 = F.DoubleCRT.from_poly(ctx, O.ints_to_limbs(pk[1], nl))
    out = ctx.alloc(count * 2 * n * nl * 8)
    ctx.encrypt_batch(pk0, pk1, logQ, p, rand, np.array(msgs, dtype=np.int64), out, nl)
    got = out.download((count, 2, n, nl))
    for c in range(count):
        assert [O.limbs_to_ints(got[c, r]) for r in range(2)] == expect[c], c
    sk1 = F.DoubleCRT.from_poly(ctx, O.ints_to_limbs(t, 1))
    dm = ctx.decrypt_batch(sk1, logQ, p, out, nl, count)
    assert [[int(v) for v in row] for row in dm] == msgs


@pytest.mark.parametrize("m,logQ,nsrc", [(64, 100, 3), (22, 80, 3), (4096, 128, 2), (32768, 512, 3)])
def test_keyswitch_init_batch_vs_oracle(m, logQ, nsrc):
    """KeySwitchSI::Init (FHE-SI.cpp:153-209) for all columns in one device call (fhesi_keyswitch_init_batch) against
    the C oracle with the same explicit randomness: power-of-two and Bluestein rings, the s^2 -> s shape (3 source components) and the
    automorphism shape (2), up to the metric ring (66 columns of n = 2^14, 18 primes).  The matrix produced on the device must also
    key-switch correctly: a product of two encryptions decrypts to the plaintext product when it is used (valid-key check)."""
    p = 23
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, L, nd, nl = ctx.phim, len(primes), R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(m + nsrc)
    W = L + 2
    t = np.zeros(n, dtype=np.int64)
    t[rng.choice(n, size=min(64, n), replace=False)] = rng.choice([-1, 1], size=min(64, n))
    t_l = O.ints_to_limbs([int(x) for x in t], W)
    one_l = O.ints_to_limbs([1] + [0] * (n - 1), W)
    t_rows = orc.dcrt_from_poly(t_l)
    one_rows = orc.dcrt_from_poly(one_l)
    src_rows = [one_rows, t_rows, orc.dcrt_op(t_rows, t_rows, 2)][:nsrc]
    ncol = nsrc * nd
    a = P.rand_limbs(rng, (ncol, n), nl, logQ)
    a[0, 0] = O.ints_to_limbs([-(1 << (logQ - 1))], nl)[0]
    a[0, 1] = O.ints_to_limbs([(1 << (logQ - 1)) - 1], nl)[0]
    err = np.rint(rng.normal(0.0, 3.2, size=(ncol, n))).astype(np.int64)
    err[0, 0], err[0, 1] = -40, 40
    want = orc.keyswitch_init(np.stack(src_rows), t_rows, logQ, a, err)

    def dcrt(rows):
        d = F.DoubleCRT(ctx)
        for i in range(L):
            d.set_row(i, rows[i])
        return d

    src = [dcrt(r) for r in src_rows]
    ksk = F.KeySwitchMatrix(ctx, nsrc, nd).init_batch(src, src[1], logQ, a, err)
    got = ksk.download()
    assert np.array_equal(got[1], want[1]), "A rows"
    assert np.array_equal(got[0], want[0]), "b rows"
    if nsrc == 3 and m <= 4096:
        # the reference's end-to-end predicate with this matrix: decrypt(keyswitch(c1 * c2)) = m1 * m2 (Test_AddMul.cpp:59-67,84-86)
        cx = R.Ctx(m, logQ, p, list(primes), list(roots))
        prng = R.SplitMix64(9)
        tl = [int(x) for x in t]
        c1v = R.sample_random(prng, 1 << logQ, n)
        c0v = R.sample_gaussian(prng, n)
        tc1 = R.poly_mul_mod_phi(cx, tl, c1v)
        pk = [[R.reduce_logq(x + y, logQ) for x, y in zip(c0v, tc1)], [R.reduce_logq(-x, logQ) for x in c1v]]
        m1 = [int(x) for x in rng.integers(0, p, size=n)]
        m2 = [int(x) for x in rng.integers(0, p, size=n)]
        e1, e2 = R.encrypt(cx, pk, m1, prng), R.encrypt(cx, pk, m2, prng)
        ca = np.stack([O.ints_to_limbs(x, nl) for x in e1])[None]
        cb = np.stack([O.ints_to_limbs(x, nl) for x in e2])[None]
        prod = ctx.ct_mul_relin(ksk, logQ, p, ca, cb)[0]
        dec = R.decrypt(cx, tl, [O.limbs_to_ints(prod[0]), O.limbs_to_ints(prod[1])])
        assert dec == [x % p for x in R.poly_mul_mod_phi(cx, m1, m2)]


@pytest.mark.parametrize("m,logQ,p", [(4096, 128, 23), (46, 90, 47), (32768, 512, 23)])
def test_seeded_encrypt_keygen_and_samplers_vs_oracle(m, logQ, p):
    """On-device sampling: fhesi_encrypt_batch_seeded, fhesi_keyswitch_init_batch_seeded and fhesi_dcrt_sample draw their
    randomness in HBM from the counter-based generator of philox.h.  The oracle draws the same polynomials from its own statement of the
    generator and computes Encrypt / KeySwitchSI::Init with them (FHE-SI.cpp:10-36, 153-209): the device results must be those bits, for a
    batch that does not start at index 0; the secret key from fhesi_dcrt_sample(HWt 64) equals DoubleCRT(draw); encrypt o decrypt = id."""
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, L, nd, nl = ctx.phim, len(primes), R.ndigits(logQ), (logQ + 63) // 64
    seed, first = 0x1234567890ABCDEF, 41
    W = L + 2
    # secret key t = sampleHWt(64) and a Gaussian polynomial, drawn on the device
    sk1 = F.DoubleCRT(ctx).sample(0, 64, seed, 7)
    t = orc.draw_hwt(seed, 7, 64)
    t_rows = orc.dcrt_from_poly(O.ints_to_limbs([int(x) for x in t], W))
    assert all(np.array_equal(sk1.row(i), t_rows[i]) for i in range(L))
    g = F.DoubleCRT(ctx).sample(1, 0, seed, 8)
    g_rows = orc.dcrt_from_poly(O.ints_to_limbs([int(x) for x in orc.draw_gaussian(seed, 8)], W))
    assert all(np.array_equal(g.row(i), g_rows[i]) for i in range(L))
    # public key (valid): pk0 = e + t c1, pk1 = -c1 with c1 uniform, built through the oracle's rows
    rng = np.random.default_rng(m)
    c1 = P.rand_limbs(rng, (n,), nl, logQ)
    c1_rows = orc.dcrt_from_poly(c1)
    e_rows = orc.dcrt_from_poly(O.ints_to_limbs([int(x) for x in orc.draw_gaussian(seed, 9)], W))
    pk0_rows = orc.dcrt_op(orc.dcrt_op(t_rows, c1_rows, 2), e_rows, 0)
    zero = np.zeros_like(c1_rows)
    pk1_rows = orc.dcrt_op(zero, c1_rows, 1)
    # (the reference reduces pk0 modulo 2^logQ in coefficient form; for the encrypt/decrypt identity below the unreduced rows serve as well)
    pk0, pk1 = dcrt_from_rows(ctx, pk0_rows), dcrt_from_rows(ctx, pk1_rows)
    count = 3
    msg = rng.integers(0, p, size=(count, n)).astype(np.int64)
    out = ctx.alloc(count * 2 * n * nl * 8)
    ctx.encrypt_batch_seeded(pk0, pk1, logQ, p, seed, first, msg, out, nl)
    got = out.download((count, 2, n, nl))
    pk = np.stack([pk0_rows, pk1_rows])
    for c in range(count):
        small, noise = orc.draw_encrypt(seed, first + c)
        assert np.array_equal(got[c], orc.encrypt(pk, small, noise, msg[c], logQ, p, nl)), c
    # the explicit-randomness entry point with the oracle's draws gives the same ciphertexts
    rand = np.stack([np.concatenate([orc.draw_encrypt(seed, first + c)[0][None], orc.draw_encrypt(seed, first + c)[1]]) for c in range(count)])
    out2 = ctx.alloc(count * 2 * n * nl * 8)
    ctx.encrypt_batch(pk0, pk1, logQ, p, rand, msg, out2, nl)
    assert np.array_equal(out2.download((count, 2, n, nl)), got)
    if m <= 4096:
        assert np.array_equal(ctx.decrypt_batch(sk1, logQ, p, out, nl, count), msg)
    # key-switch matrix of (1, t, t^2) -> t with the column randomness drawn on the device
    one_rows = orc.dcrt_from_poly(O.ints_to_limbs([1] + [0] * (n - 1), W))
    src_rows = [one_rows, t_rows, orc.dcrt_op(t_rows, t_rows, 2)]
    src = [dcrt_from_rows(ctx, r) for r in src_rows]
    pub_seed = seed ^ 0x5DEECE66D
    with pytest.raises(ValueError):
        F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded(src, src[1], logQ, seed, seed, 1000)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded(src, src[1], logQ, seed, pub_seed, 1000)
    kgot = ksk.download()
    ncol = 3 * nd
    cols = (0, 1, ncol - 1) if n > 8192 else range(ncol)
    a = np.zeros((ncol, n, nl), dtype=np.uint64)
    err = np.zeros((ncol, n), dtype=np.int64)
    for col in range(ncol):
        a[col], err[col] = orc.draw_keygen(pub_seed, 1000 + col, nl, logQ)[0], orc.draw_keygen(seed, 1000 + col, nl, logQ)[1]      # public a, secret error
    if n <= 8192:
        want = orc.keyswitch_init(np.stack(src_rows), t_rows, logQ, a, err)
        assert np.array_equal(kgot[1], want[1]) and np.array_equal(kgot[0], want[0])
    else:
        # the metric ring: the A rows of the sampled columns = -DoubleCRT(a) checked row by row, and the whole matrix against the device call
        # that takes the oracle's draws as explicit randomness (itself checked against the oracle in test_keyswitch_init_batch_vs_oracle)
        for col in cols:
            arows = orc.dcrt_from_poly(a[col])
            assert np.array_equal(kgot[1][col], orc.dcrt_op(np.zeros_like(arows), arows, 1)), col
        k2 = F.KeySwitchMatrix(ctx, 3, nd).init_batch(src, src[1], logQ, a, err)
        assert np.array_equal(k2.download(), kgot)
