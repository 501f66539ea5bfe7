"""GPU parity: Ciphertext::operator*= + KeySwitchSI::ApplyKeySwitch (Ciphertext.cpp:167-218, FHE-SI.cpp:241-260)
through the C ABI vs the C oracle, stage by stage and end to end.  Bit-exact."""
import numpy as np
import pytest

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

pytestmark = pytest.mark.gpu


def setup(m, logQ, p, seed, count, sp_nbits=60):
    primes, roots = P.chain_for(m, logQ, p, 1, sp_nbits)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, L = ctx.phim, len(primes)
    nd = R.ndigits(logQ)
    nl = (logQ + 63) // 64
    rng = np.random.default_rng(seed)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])     # [2][3nd][L][n] uniform residues
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    return ctx, orc, ksm, a, b, nd, nl


@pytest.mark.parametrize("m,logQ,p,count", [(32, 80, 23, 3), (256, 120, 2027, 2), (4096, 128, 23, 2), (16384, 200, 65537, 1),
                                            (64, 512, 23, 2),      # the metric chain shape: 18 primes, 17-limb product (unrolled CRT instantiation)
                                            (4096, 512, 23, 1),
                                            (32768, 512, 23, 1)])   # BASELINE.json's metric configuration itself: n = 2^14, 17 x 60-bit + one 37-bit prime
def test_mul_relin_stages_and_end_to_end(m, logQ, p, count):
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, m + logQ, count)
    n, L = ctx.phim, ctx.L
    # edge coefficients: extremes of the centred range
    a[0, 0, 0] = O.ints_to_limbs([-(1 << (logQ - 1))], nl)[0]
    a[0, 0, 1] = O.ints_to_limbs([(1 << (logQ - 1)) - 1], nl)[0]
    b[0, 1, 0] = O.ints_to_limbs([-(1 << (logQ - 1))], nl)[0]
    da, db = ctx.upload(a), ctx.upload(b)
    # stage: tensor product in DoubleCRT form
    tp = ctx.alloc(count * 3 * L * n * 8)
    ctx.ct_mul_dev(p, da, db, nl, count, tp)
    tprod = tp.download((count, 3, L, n))
    for c in range(count):
        assert np.array_equal(tprod[c], orc.ct_mul(a[c], b[c], p)), c
    # stage: key switch of the scaled-up ciphertext
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    out = ctx.alloc(count * 2 * n * nl * 8)
    ctx.apply_key_switch_dev(ksk, logQ, tp, count, out, nl)
    got = out.download((count, 2, n, nl))
    for c in range(count):
        assert np.array_equal(got[c], orc.apply_key_switch(ksm, tprod[c], logQ, nl)), c
    # end to end from host buffers
    e2e = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    for c in range(count):
        assert np.array_equal(e2e[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c


def test_valid_keys_decrypt_to_product():
    """The reference's own end-to-end predicate (Test_AddMul.cpp:59-67,84-86): decrypt(keyswitch(c1*c2)) == m1*m2."""
    m, logQ, p = 64, 100, 23
    primes, roots = P.chain_for(m, logQ, p)
    rctx = R.Ctx(m, logQ, p, primes, roots)
    prng = R.SplitMix64(2024)
    t, pk = R.keygen(rctx, prng)
    n = rctx.phim
    m1 = [prng.bnd(p) for _ in range(n)]
    m2 = [prng.bnd(p) for _ in range(n)]
    c1, c2 = R.encrypt(rctx, pk, m1, prng), R.encrypt(rctx, pk, m2, prng)
    ksm_py = R.key_switch_init_s2(rctx, t, prng)
    L, nd, nl = len(primes), R.ndigits(logQ), (logQ + 63) // 64
    ksm = np.array([[[d[i] for i in range(L)] for d in ksm_py[r]] for r in range(2)], dtype=np.uint64)
    ctx = F.Context(m, primes, roots)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    a = np.stack([O.ints_to_limbs(c, nl) for c in c1])[None]
    b = np.stack([O.ints_to_limbs(c, nl) for c in c2])[None]
    out = ctx.ct_mul_relin(ksk, logQ, p, a, b)[0]
    parts = [O.limbs_to_ints(out[r]) for r in range(2)]
    assert parts == R.ct_mul_relin(rctx, ksm_py, c1, c2)
    assert R.decrypt(rctx, t, parts) == [c % p for c in R.poly_mul_mod_phi(rctx, m1, m2)]


def test_stress_config_shape_single_ciphertext():
    """configs[4] of BASELINE.json (stress): m = 2^16 (n = 2^15 > one LDS tile: generic multi-pass NTT), fhe-si logQ = 1024,
    p = 65537  =>  35 primes, 43 digits; one ciphertext mult + relinearize, bit-exact against the oracle."""
    m, logQ, p = 1 << 16, 1024, 65537
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 5, 1)
    assert ctx.L == 35 and nd == 43
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert np.array_equal(got[0], orc.ct_mul_relin(ksm, a[0], b[0], logQ, p))


def test_sum_form_crt_undecided_coefficients(monkeypatch):
    """The metric chain shape converts through the sum-form CRT kernel (kernels_crt.hip), which hands coefficients it cannot decide
    to the exact mixed-radix kernel: values whose rounding in ScaleDown (Ciphertext.cpp:205-213) sits on the edge, and values at
    +-P/2 in toPoly's centring (DoubleCRT.cpp:375-376).  Crafted inputs put such values at known positions; the result must still
    equal the oracle's, and must NOT when the clean-up pass is switched off (so the inputs really exercise it)."""
    m, logQ, p, count = 64, 512, 23, 1
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 99, count)
    n, L = ctx.phim, ctx.L
    primes = [int(q) for q in ctx.primes]
    Pprod = 1
    for q in primes:
        Pprod *= q
    mod = 1 << logQ
    inv_p = pow(p, -1, mod)

    def centred(v):
        v %= mod
        return v - mod if v >= mod // 2 else v

    # --- ScaleDown edge: x = p * A with (x + 2^(logQ-1)) mod 2^logQ = delta, delta around 0
    deltas = [0, 1, -1, 2, -2, 3, -3, 5, -5, 7, -7, 8, -8, 100, -100]
    A = [centred((d - (mod >> 1)) * inv_p) for d in deltas]
    a0 = A + [0] * (n - len(A))
    a[0, 0] = O.ints_to_limbs(a0, nl)
    a[0, 1] = 0
    b[0, 0] = O.ints_to_limbs([1] + [0] * (n - 1), nl)
    b[0, 1] = 0
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    want = orc.ct_mul_relin(ksm, a[0], b[0], logQ, p)
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b)[0], want)
    ctx.set_option("crt_skip_cleanup", 1)
    assert not np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b)[0], want)
    ctx.set_option("crt_skip_cleanup", 0)

    # --- centring edge: make the key-switch dot product return chosen polynomials.  Scaled-down parts = (1, 0, 0), so only digit 0
    # of part 0 is non-zero (the constant 1, whose transform is the all-ones row) and the dot product is key row (r, 0) itself.
    W = L + 2
    one = O.ints_to_limbs([mod] + [0] * (n - 1), W)           # x / 2^logQ = 1 exactly
    tp = np.zeros((1, 3, L, n), dtype=np.uint64)
    tp[0, 0] = orc.dcrt_from_poly(one)
    h = (Pprod - 1) // 2
    edge = [h, -h, h + 1, h - 1, -h + 1, 0, 1, -1, Pprod - 1, h + 2, -h - 2, 12345, -(1 << 600)]
    ksm2 = ksm.copy()
    for r in range(2):
        e = edge[r:] + edge[:r] + [0] * (n - len(edge))
        ksm2[r, 0] = orc.dcrt_from_poly(O.ints_to_limbs(e, W))
    ksk2 = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm2)
    dtp = ctx.upload(tp)
    out = ctx.alloc(2 * n * nl * 8)
    ctx.apply_key_switch_dev(ksk2, logQ, dtp, 1, out, nl)
    want2 = orc.apply_key_switch(ksm2, tp[0], logQ, nl)
    assert np.array_equal(out.download((2, n, nl)), want2)
    got_int = O.limbs_to_ints(out.download((2, n, nl))[0])
    assert got_int[0] == centred(h) and got_int[1] == centred(-h) and got_int[2] == centred(-h)      # (P+1)/2 wraps to -(P-1)/2
    ctx.set_option("crt_skip_cleanup", 1)
    ctx.apply_key_switch_dev(ksk2, logQ, dtp, 1, out, nl)
    assert not np.array_equal(out.download((2, n, nl)), want2)
    ctx.set_option("crt_skip_cleanup", 0)


@pytest.mark.parametrize("m,logQ,p", [(32768, 512, 23), (1 << 16, 1024, 65537),
                                      (32768, 200, 23),          # run-time CRT window (any logQ <= 512)
                                      (8422, 341, 8423),         # the reference's Test_Regression ring: linear convolutions + fold
                                      (32602, 128, 32603),       # ... with phi(m) = 16300: padded rows of 2^15 (head = duplication, tail inside the CRT kernel)
                                      (65266, 128, 65267),       # ... with phi(m) = 32632: padded rows of 2^16 (second head stage in rns32_reduce, the tail stages a pass of their own)
                                      (101, 128, 23), (16381, 128, 23)])      # odd prime m: the three-term fold modulo X^m - 1 and Phi_m
def test_tensor_half_over_30_bit_primes_equals_the_chain(m, logQ, p):
    """At the metric ring the fused pipeline forms tProd's integers modulo 35 primes below 2^30 instead of the chain
    (kernels_tensor32.hip; Ciphertext.cpp:167-218 only ever exposes round(x / 2^logQ) mod 2^logQ of them; 70 primes and rows of 2^15 at
    the stress shape, where the head / tail stages of the row transform sit inside the two conversions).  Same bits as the chain path
    (option tensor32 = 0) and as the oracle on random inputs, on the extremes of the centred range in every coefficient (the largest
    |x| the bound allows), and on coefficients whose rounding sits on the edge: those must go through the exact second pass (the
    result is wrong when that pass is switched off)."""
    count = 4
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 4242, count)
    if (m & (m - 1)) != 0:
        orc.set_bluestein_fft(True)             # the oracle's O(N log N) form of the same transforms (bluestein.cpp:116-139)
    n = ctx.phim
    mod = 1 << logQ
    lo, hi = -(mod >> 1), (mod >> 1) - 1
    # ciphertext 1: the extremes everywhere (|x| up to 2 n p 2^1022)
    rng = np.random.default_rng(5)
    for part in range(2):
        a[1, part] = O.ints_to_limbs([lo if v else hi for v in rng.integers(0, 2, n)], nl)
        b[1, part] = O.ints_to_limbs([lo if v else hi for v in rng.integers(0, 2, n)], nl)
    a[1, 0] = O.ints_to_limbs([lo] * n, nl)
    b[1, 0] = O.ints_to_limbs([lo] * n, nl)
    # ciphertext 2: x = p A with (x + 2^(logQ-1)) mod 2^logQ = delta around 0 (ScaleDown's rounding edge, Ciphertext.cpp:205-213)
    inv_p = pow(p, -1, mod)

    def centred(v):
        v %= mod
        return v - mod if v >= mod // 2 else v

    deltas = [0, 1, -1, 2, -2, 3, -3, 5, -5, 7, -7, 8, -8, 100, -100, 1 << 64, -(1 << 64), 1 << (logQ - 112), -(1 << (logQ - 112)),
              (1 << (logQ - 64)) - 1, -(1 << (logQ - 64))]
    deltas = [d for d in deltas if abs(d) < (1 << (logQ - 2))]
    A = [centred((d - (mod >> 1)) * inv_p) for d in deltas]
    a[2, 0] = O.ints_to_limbs(A + [0] * (n - len(A)), nl)
    a[2, 1] = O.ints_to_limbs([0] * (n - len(A)) + A, nl)
    b[2, 0] = O.ints_to_limbs([1] + [0] * (n - 1), nl)
    b[2, 1] = 0
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    assert ctx.get_option("tensor32") == 1
    ctx.prof_enable(True)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    # the kernels that ran are the 32-bit ones
    assert "crt32_scale" in ctx.prof_kernel_name("crt") and "rns32_reduce_kernel" in ctx.prof_kernel_name("rns_reduce")
    ctx.prof_enable(False)
    ctx.set_option("tensor32", 0)
    chain = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    ctx.set_option("tensor32", 1)
    assert np.array_equal(got, chain)
    # ... and over the OTHER prime size (option tensor_bits: 30 = the largest primes below 2^30, 29 = below 2^29 with fewer range steps in the
    # row transforms and one or two primes more): other primes, other tables, other kernels -- the same integers
    bits = ctx.get_option("tensor_bits")
    assert bits in (29, 30)
    ctx.set_option("tensor_bits", 59 - bits)
    other = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert ctx.get_option("tensor_bits") == 59 - bits
    ctx.set_option("tensor_bits", bits)
    assert np.array_equal(got, other)
    # (general m at this size: seconds per Bluestein row in the oracle; at m = 65266 minutes per multiplication -- there the chain path above,
    # per-prime Bluestein rows checked against the oracle on the smaller rings and in test_gpu_general_m.py, stands in)
    for c in (() if m > 40000 else (2,) if ctx.phim > 10000 and (m & (m - 1)) != 0 else (1, 2)):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    ctx.set_option("crt_skip_cleanup", 1)
    bad = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    ctx.set_option("crt_skip_cleanup", 0)
    # (below logQ = 130 the first pass of the run-time CRT kernel already forms every word: nothing is left to the exact pass)
    assert np.array_equal(bad[0], got[0]) and np.array_equal(bad[3], got[3]) and (logQ < 130 or not np.array_equal(bad[2], got[2]))


@pytest.mark.parametrize("m,logQ,p", [(4096, 128, 23), (32768, 512, 23), (1 << 16, 200, 23)])
def test_key_switch_paths_agree(m, logQ, p, monkeypatch):
    """The key switch has two device paths: the dot product through the two largest chain primes (kernels_ksaux.hip, default for
    n = 2^11 .. 2^15) and the per-prime dot product of the reference's own structure (FHE-SI.cpp:251-254; FHESI_KS_DIRECT=1, and every
    shape the first does not cover).  Both must give the oracle's bits -- including after the key rows are rewritten in place
    through the device pointer (the RCCL broadcast path), which has to rebuild the derived key table."""
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 31 + m, 1)
    want = orc.ct_mul_relin(ksm, a[0], b[0], logQ, p)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b)[0], want)
    ctx.set_option("ks_direct", 1)
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b)[0], want)
    ctx.set_option("ks_direct", 0)
    # new key rows written straight into HBM
    rng = np.random.default_rng(m)
    ksm2 = np.stack([P.rand_rows(rng, [int(q) for q in ctx.primes], ctx.phim, 3 * nd) for _ in range(2)])
    stage = ctx.upload(ksm2)
    ctx.dev_copy(ksk.device_ptr, stage.ptr.value, ksm2.nbytes)
    ksk.mark_dirty()
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b)[0], orc.ct_mul_relin(ksm2, a[0], b[0], logQ, p))


@pytest.mark.parametrize("m,logQ,sp_nbits", [(4096, 512, 60), (8192, 512, 60), (32768, 512, 60),          # the metric chain (18 primes); 32768: four 30-bit auxiliary primes
                                             (4096, 128, 60), (8192, 200, 60), (16384, 256, 60), (32768, 128, 60), (32768, 341, 60), (32768, 700, 60), (65536, 300, 60),
                                             (32768, 512, 50), (65536, 300, 50), (8422, 341, 50)])      # chains of 50-bit primes (NTL_SP_NBITS = 50): 22, 14 and 15 primes
def test_key_switch_limb_mode_edge_values(m, logQ, sp_nbits):
    """For n >= 2^11 the key switch runs in limb mode (kernels_ksaux.hip + ks_recombine_kernel) for ANY chain: the limb width, the limb
    count and the auxiliary modulus are derived per chain (ks_limb_plan); the dot product is recombined as an integer and reduced modulo
    the chain product P exactly.  Shapes: the metric chain (compile-time instantiations), and chains of 5 to 24 primes through the
    run-time recombination kernel, with the two 60-bit auxiliary primes and (n = 2^14) the four 30-bit ones.  Crafted key rows make
    the integer hit the edges of the reduction: 0, +-1, +-(P-1)/2, (P+1)/2 (wraps), P-1, values just inside and outside the centring
    threshold, and large multiples; the result must equal the oracle's (toPoly + ReduceCoefficients, FHE-SI.cpp:255-256)."""
    p = 23 if m != 8422 else 8423
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 7 + m, 1, sp_nbits)
    n, L = ctx.phim, ctx.L
    primes = [int(q) for q in ctx.primes]
    assert max(primes).bit_length() == sp_nbits
    Pprod = 1
    for q in primes:
        Pprod *= q
    mod = 1 << logQ
    W = L + 2
    h = (Pprod - 1) // 2
    pb = Pprod.bit_length()
    edge = [h, -h, h + 1, h - 1, -h + 1, 0, 1, -1, Pprod - 1, h + 2, -h - 2, 12345, -(1 << (pb * 4 // 7)), (1 << (pb - 58)) + 17, -(1 << (pb - 8)) + 3]
    # scaled-down parts = (digit value d at coefficient 0, 0, 0): only digit 0 of part 0 is non-zero, so the dot product is d * key row (r, 0)
    # (at position n-1 the negacyclic wrap turns the products negative: S = -d * e_j)
    for d, pos in ((1, 0), ((1 << 24) - 1, 0), ((1 << 24) - 1, n - 1), (1, n - 1)):
        tp = np.zeros((1, 3, L, n), dtype=np.uint64)
        tp[0, 0] = orc.dcrt_from_poly(O.ints_to_limbs([0] * pos + [d * mod] + [0] * (n - 1 - pos), W))
        ksm2 = ksm.copy()
        for r in range(2):
            e = edge[r:] + edge[:r] + [0] * (n - len(edge))
            ksm2[r, 0] = orc.dcrt_from_poly(O.ints_to_limbs(e, W))
        ksk2 = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm2)
        dtp = ctx.upload(tp)
        out = ctx.alloc(2 * n * nl * 8)
        ctx.apply_key_switch_dev(ksk2, logQ, dtp, 1, out, nl)
        want = orc.apply_key_switch(ksm2, tp[0], logQ, nl)
        assert np.array_equal(out.download((2, n, nl)), want), (d, pos)
        if sp_nbits == 50:
            # a chain of 50-bit primes has no pair of 60-bit primes to serve as auxiliary modulus: the four 30-bit primes carry it
            # wherever their transforms exist (rows of 2^14 / 2^15, the safe-prime rings) -- it must not drop to the per-prime form silently
            assert ksk2.form()[0] == 1, ksk2.form()
        if n == 1 << 14 and pos == 0 and sp_nbits == 60:                 # the same chain through the two 60-bit auxiliary primes
            ctx.set_option("ks_aux60", 1)
            ksk3 = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm2)
            ctx.apply_key_switch_dev(ksk3, logQ, dtp, 1, out, nl)
            ctx.set_option("ks_aux60", 0)
            assert np.array_equal(out.download((2, n, nl)), want), (d, pos, "aux60")


def test_key_switch_paths_agree_on_a_full_batch(monkeypatch):
    """BASELINE.json's metric configuration at the bench's batch size: the three device forms of the key switch -- four 30-bit auxiliary
    primes (default at n = 2^14), two 60-bit auxiliary primes in limb mode, and the per-prime dot product of the reference's structure --
    must produce the same 2 x 64 x 16384 coefficients (the oracle is too slow for 64 ciphertexts; one of them is checked against it in
    test_mul_relin_stages_and_end_to_end)."""
    m, logQ, p, count = 32768, 512, 23, 64
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 4242, count)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    ref = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    for opt, val in (("ks_aux60", 1), ("ks_residues", 1), ("ks_direct", 1)):
        ctx.set_option(opt, val)
        ksk2 = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)        # a fresh matrix: the derived table is built for the selected form
        assert np.array_equal(ctx.ct_mul_relin(ksk2, logQ, p, a, b), ref), opt
        ctx.set_option(opt, 0)
    assert np.array_equal(ref[0], orc.ct_mul_relin(ksm, a[0], b[0], logQ, p))
    # ... and on ONE live matrix: an option that selects another form rebuilds the derived table (it used to be kept silently)
    assert ksk.form()[0] == 1
    for opt, form in (("ks_aux60", 2), ("ks_residues", 3), ("ks_direct", 0)):
        ctx.set_option(opt, 1)
        assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a[:3], b[:3]), ref[:3]), opt
        assert ksk.form()[0] == form, (opt, ksk.form())
        ctx.set_option(opt, 0)
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a[:3], b[:3]), ref[:3]) and ksk.form()[0] == 1


def test_large_launch_equals_launches_of_64():
    """The 32-bit pipelines take up to 1024 ciphertexts per launch, with the digit rows tiled per sub-chunk of 64 ciphertexts
    (ntt32_core.inc / dot32_kernel2).  130 ciphertexts = two full sub-chunks and a partial one: every output must equal what launches of
    at most 64 (option batch_chunk, one sub-chunk each) and of 7 (partial tiles everywhere) produce, and the oracle's on a sample."""
    m, logQ, p, count = 32768, 512, 23, 130
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 77, 5)
    idx = np.arange(count) % 5
    a, b = a[idx].copy(), b[idx].copy()
    a[129, 0, 3] = O.ints_to_limbs([12345], nl)[0]                    # the last ciphertext is not a copy of an earlier one
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    big = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    for chunk in (64, 7):
        ctx.set_option("batch_chunk", chunk)
        assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b), big), chunk
    ctx.set_option("batch_chunk", 0)
    for c in (0, 129):
        assert np.array_equal(big[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    for c in range(5, count - 1):
        assert np.array_equal(big[c], big[c % 5]), c


@pytest.mark.parametrize("m,logQ", [(22, 20), (46, 24)])
def test_safe_prime_ring_with_a_chain_too_narrow_for_the_limb_plan(m, logQ):
    """On m = 2 x prime the exact-integer key switch exists in limb mode only.  Whether a chain admits a limb plan is decided where the
    form is chosen (ksaux_mode), so a chain without one -- here a single prime -- takes the per-prime Bluestein form instead of failing
    in ksaux_build with 'no exact limb plan'; a two-prime chain that does admit a plan runs it.  Both must give the oracle's bits."""
    p = 23 if m == 22 else 47
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 5 + m, 2)
    assert ctx.L <= 2
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    for c in range(2):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    assert ksk.form()[0] == (0 if ctx.L == 1 else 1)


@pytest.mark.parametrize("m,logQ", [(22, 80), (46, 120), (1006, 200), (8422, 341),
                                    (32602, 120),              # p = 32603: phi(m) = 16300, the metric's size in the reference's own parameterisation -- padded rows of 2^15
                                    (65266, 120),              # p = 65267: phi(m) = 32632, 2 phi(m) - 1 = 65263 <= 2^16 -- padded rows of 2^16 (two head / tail stages: the largest ring FHEContext.cpp:89 allows a safe prime below 2^16 for)
                                    (101, 80), (16381, 120)])  # odd prime m: fold modulo X^m - 1 and Phi_m = 1 + X + ... + X^(m-1) (rows of 2^14 and 2^15)
def test_key_switch_on_safe_prime_rings(m, logQ):
    """The reference's own rings (m = p - 1 = 2 q' for a safe prime p; Test_Regression: p = 8423, logQ = 341, 13 primes) take the exact
    integer key switch as well: the digit (*) key products are LINEAR convolutions carried by the 2^14- or 2^15-point 32-bit transforms on
    zero-padded rows, folded modulo X^q' + 1 and Phi_m inside the recombination (kernels_crt.hip: ks_recombine_generic_kernel, fold_q) --
    instead of one Bluestein transform of every digit polynomial per chain prime (option ks_direct, the reference's structure,
    CModulus.cpp:90-132 + bluestein.cpp:93-144).  Odd prime m takes the same route with its own fold.  Both device paths
    against the oracle, end to end and on crafted key rows that drive the folded integer through the edges of the reduction."""
    p = 23
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 5 + m, 2)
    if m > 2000:
        orc.set_bluestein_fft(True)             # the oracle's O(N log N) form of the same transforms (bluestein.cpp:116-139)
    n, L = ctx.phim, ctx.L
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    # (the oracle's Bluestein rows of 2^16 / 2^17 points take seconds each; at m = 65266 one multiplication takes 1.5-6 minutes of CPU: that ring's
    # oracle anchor is the committed bench run profiles/r06_e_refring_p65267_oracle_bench.json -- matches_oracle true at logQ = 512 -- and here it
    # is held to the per-prime device path below, which test_gpu_general_m.py and the smaller rings pin to the oracle)
    want = [orc.ct_mul_relin(ksm, a[c], b[c], logQ, p) for c in range(0 if m > 40000 else 1 if m > 10000 else 2)]
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert ksk.form()[0] == 1, ksk.form()            # limbs over the four 30-bit auxiliary primes = the linear-convolution form ran
    for c in range(len(want)):
        assert np.array_equal(got[c], want[c]), c
    ctx.set_option("ks_direct", 1)
    ksk_d = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    got_d = ctx.ct_mul_relin(ksk_d, logQ, p, a, b)
    ctx.set_option("ks_direct", 0)
    assert np.array_equal(got_d, got)
    if m > 20000:
        # ... and with a GENERATED matrix (KeySwitchSI::Init on the device: centred limbs, the form the reference's drivers would run) against the
        # per-prime Bluestein path on the same matrix
        one = np.zeros((n, 1), dtype=np.uint64)
        one[0, 0] = 1
        t = F.DoubleCRT(ctx).sample(0, 64, 77, 1)
        t2 = t.copy()
        t2.op(t, 2)
        kg = F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded([F.DoubleCRT.from_poly(ctx, one), t, t2], t, logQ, 77, 78, 100, 3)
        got_g = ctx.ct_mul_relin(kg, logQ, p, a, b)
        assert kg.form()[0] == 1 and kg.key_bits()[0], (kg.form(), kg.key_bits())
        ctx.set_option("ks_direct", 1)
        kg_d = F.KeySwitchMatrix(ctx, 3, nd).upload(kg.download())
        got_gd = ctx.ct_mul_relin(kg_d, logQ, p, a, b)
        ctx.set_option("ks_direct", 0)
        assert kg_d.form()[0] == 0 and np.array_equal(got_gd, got_g)
    # crafted rows: scaled-down parts = (d X^pos, 0, 0), key row (r, 0) = edge polynomial e, so the dot product is d X^pos e mod Phi_m
    primes = [int(q) for q in ctx.primes]
    Pprod = 1
    for q in primes:
        Pprod *= q
    mod, W, h, pb = 1 << logQ, L + 2, (Pprod - 1) // 2, Pprod.bit_length()
    edge = [h, -h, h + 1, h - 1, 0, 1, -1, Pprod - 1, h + 2, 12345, -(1 << (pb * 4 // 7)), (1 << (pb - 8)) + 17]
    for d, pos in ((1, 0), ((1 << 24) - 1, n - 1), ((1 << 24) - 1, n // 2))[(1 if m > 10000 else 0):(2 if m > 10000 else 3)]:
        tp = np.zeros((1, 3, L, n), dtype=np.uint64)
        tp[0, 0] = orc.dcrt_from_poly(O.ints_to_limbs([0] * pos + [d * mod] + [0] * (n - 1 - pos), W))
        ksm2 = ksm.copy()
        for r in range(2):
            e = (edge[r:] + edge[:r]) * (n // len(edge) + 1)
            ksm2[r, 0] = orc.dcrt_from_poly(O.ints_to_limbs(e[:n], W))
        ksk2 = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm2)
        dtp = ctx.upload(tp)
        out = ctx.alloc(2 * n * nl * 8)
        ctx.apply_key_switch_dev(ksk2, logQ, dtp, 1, out, nl)
        if m > 20000:        # (a minute per key switch in the oracle at this size: the per-prime Bluestein device path, checked against it above, stands in)
            ctx.set_option("ks_direct", 1)
            ksk3 = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm2)
            ref = ctx.alloc(2 * n * nl * 8)
            ctx.apply_key_switch_dev(ksk3, logQ, dtp, 1, ref, nl)
            ctx.set_option("ks_direct", 0)
            assert ksk3.form()[0] == 0
            want_rows = ref.download((2, n, nl))
        else:
            want_rows = orc.apply_key_switch(ksm2, tp[0], logQ, nl)
        assert np.array_equal(out.download((2, n, nl)), want_rows), (d, pos)


@pytest.mark.parametrize("count,host_chunk", [(1, 0), (3, 0), (8, 0), (37, 5), (130, 0)])
def test_host_buffer_pipeline_equals_the_device_batch(count, host_chunk):
    """fhesi_ct_mul_relin_batch (host buffers: what a caller of Ciphertext::operator*= + ApplyKeySwitch holds, Test_AddMul.cpp:59-67)
    runs as a pipeline of stages over a pinned ring -- upload, compute and download of neighbouring stages overlapped on three streams, the
    pageable side copied by several threads.  Same bits as fhesi_ct_mul_relin_batch_dev on the same pairs and as the oracle, for a single
    ciphertext, ragged last stages, more stages than ring slots, pinned buffers (fhesi_host_alloc: no staging copy) and mixed ones,
    and when the same context is called again with another batch size."""
    m, logQ, p = 2048, 128, 23
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 77 + count, count)
    n = ctx.phim
    if host_chunk:
        ctx.set_option("host_chunk", host_chunk)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    da, db, dout = ctx.upload(a), ctx.upload(b), ctx.alloc(count * 2 * n * nl * 8)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
    want = dout.download((count, 2, n, nl))
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert np.array_equal(got, want)
    for c in (0, count - 1):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    # pinned operands and result; then pinned a, pageable b, pageable result reused
    pa, pb, po = ctx.host_array(a.shape), ctx.host_array(a.shape), ctx.host_array(a.shape)
    pa[...] = a
    pb[...] = b
    po[...] = 0
    assert ctx.ct_mul_relin(ksk, logQ, p, pa, pb, out=po) is po and np.array_equal(po, want)
    out2 = np.zeros_like(a)
    ctx.ct_mul_relin(ksk, logQ, p, pa, b, out=out2)
    assert np.array_equal(out2, want)
    # another batch size on the same context (the ring is re-sized or re-used)
    k2 = max(1, count // 2)
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a[:k2], b[:k2]), want[:k2])
    big = np.concatenate([a, a])
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, big, np.concatenate([b, b])), np.concatenate([want, want]))


@pytest.mark.parametrize("m,logQ,p", [(32768, 512, 23), (1 << 16, 300, 65537), (32602, 128, 32603)])
def test_round5_layout_switches_change_no_bit(m, logQ, p):
    """The layout switches select layouts, never values: the
    scaled-down parts as 64-bit limb rows instead of 32-bit word rows (parts_words = 0), the digit-tile dot product (dot32_k4 = 0)
    -- every combination gives the bits of the default, on rows of 2^14, of 2^15 and
    on padded rows of a linear-convolution ring; 33 ciphertexts (groups past the end, a ragged last tile)."""
    count = 33
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 606 + m, count)
    n = ctx.phim
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    da, db, dout = ctx.upload(a), ctx.upload(b), ctx.alloc(a.nbytes)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
    want = dout.download((count, 2, n, nl))
    if m == 32768:
        assert np.array_equal(want[32], orc.ct_mul_relin(ksm, a[32], b[32], logQ, p))
    for opts in ({"parts_words": 0}, {"dot32_k4": 0}, {"parts_words": 0, "dot32_k4": 0}):
        for k_, v_ in opts.items():
            ctx.set_option(k_, v_)
        ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
        assert np.array_equal(dout.download((count, 2, n, nl)), want), opts
        for k_ in opts:
            ctx.set_option(k_, 1)
    if m == 32768:
        # the uploaded matrix of uniform residues has 15 general limbs: dot32_kernel2 runs whatever dot32_k4 says.  A GENERATED matrix (7 centred
        # limbs) takes dot32_kernel4, and there the switch really selects the other kernel -- same bits
        one = np.zeros((n, 1), dtype=np.uint64)
        one[0, 0] = 1
        t = F.DoubleCRT(ctx).sample(0, 64, 17, 1)
        t2 = t.copy()
        t2.op(t, 2)
        kg = F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded([F.DoubleCRT.from_poly(ctx, one), t, t2], t, logQ, 17, 18, 500, 3)
        names = {}
        for k4 in (1, 0):
            ctx.set_option("dot32_k4", k4)
            ctx.prof_enable(True)
            ctx.ct_mul_relin_dev(kg, logQ, p, da, db, dout, nl, count)
            ctx.sync()
            names[k4] = ctx.prof_kernel_name("dot")
            ctx.prof_enable(False)
            if k4:
                want_g = dout.download((count, 2, n, nl))
            else:
                assert np.array_equal(dout.download((count, 2, n, nl)), want_g)
        ctx.set_option("dot32_k4", 1)
        assert "dot32_kernel4<7" in names[1] and "dot32_kernel2<" in names[0], names


def test_host_buffer_entry_rejects_device_memory_and_releases_its_ring():
    """The host-buffer entry copies with CPU threads: a device pointer must be refused with an error (not dereferenced); a pinned array may
    outlive its Context (fhesi_host_free does not touch the context); fhesi_host_stage_release hands the staging ring back and the next
    call allocates it again with the same results."""
    import ctypes as C
    m, logQ, p = 2048, 128, 23
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 91, 4)
    n = ctx.phim
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    want = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert np.array_equal(want[0], orc.ct_mul_relin(ksm, a[0], b[0], logQ, p))
    da = ctx.upload(a)
    lib = F.binding._load()
    out = np.zeros_like(a)
    rc = lib.fhesi_ct_mul_relin_batch(ctx.h, ksk.h, logQ, p, 3, C.c_void_p(da.ptr.value), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), nl, 4)
    assert rc != 0 and b"device memory" in lib.fhesi_last_error()
    ctx.release_host_staging()
    ctx.release_host_staging()                     # (idempotent)
    assert np.array_equal(ctx.ct_mul_relin(ksk, logQ, p, a, b), want)
    assert lib.fhesi_abi_version() == F.binding.ABI_VERSION
    # a pinned array that outlives its context
    ctx2 = F.Context(m, *P.chain_for(m, logQ, p))
    arr = ctx2.host_array((16,))
    arr[...] = 7
    ctx2.close()
    assert int(arr.sum()) == 112
    del arr
    import gc
    gc.collect()


@pytest.mark.parametrize("m,logQ,p", [(32768, 512, 23), (32768, 130, 23), (1 << 16, 300, 65537), (8422, 341, 8423), (101, 80, 23)])
def test_key_switch_centred_limbs_of_generated_matrices(m, logQ, p):
    """KeySwitchSI::Init (FHE-SI.cpp:176-204) samples its polynomial modulo 2^logQ and reduces b modulo 2^logQ: the integer coefficients of a
    generated key-switch matrix lie in [-2^(logQ-1), 2^(logQ-1)], fewer than half the bits of the chain product.  The library measures the
    coefficients of the matrix it is given; when they are that small the limbs of the exact-integer key switch are cut from the CENTRED
    integers (7 instead of 15 at the metric ring) and the dot product needs no reduction modulo the chain product (kernels_aux32.hip:
    ks32_key_bits, kernels_crt.hip: ks_recombine_centred_kernel).  Same bits as the oracle, as the general limbs (option ks_long_keys) and as
    the per-prime dot product of the reference's structure (ks_direct) -- on random coefficients of that size, on the extremes -2^(logQ-1),
    2^(logQ-1) - 1 and +2^(logQ-1) (the value -poly takes when poly = -2^(logQ-1)) in every position of some columns, on digits at their
    maximum, and on a matrix with ONE coefficient a few bits larger (one more limb, or the general form: still exact)."""
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 1234 + m, 2)
    if m > 2000 and (m & (m - 1)) != 0:
        orc.set_bluestein_fft(True)
    n, L = ctx.phim, ctx.L
    W = L + 2
    rng = np.random.default_rng(m + logQ)
    half = 1 << (logQ - 1)
    ncol = 3 * nd

    def rows_of(ints):
        return orc.dcrt_from_poly(O.ints_to_limbs(ints, W))

    # a generated-looking matrix: uniform in [-2^(logQ-1), 2^(logQ-1)), with the extremes in a few columns
    km = np.empty((2, ncol, L, n), dtype=np.uint64)
    for r in range(2):
        for c in range(ncol):
            km[r, c] = orc.dcrt_from_poly(P.rand_limbs(rng, (n,), W, logQ))
    km[0, 0] = rows_of([-half] * n)
    km[1, 0] = rows_of([half - 1] * n)
    km[0, 1] = rows_of([half if i % 2 else -half for i in range(n)])          # +2^(logQ-1): the value -poly takes when poly = -2^(logQ-1)
    km[1, ncol - 1] = rows_of([(-1) ** i * (half - i) for i in range(n)])
    km[0, ncol - 1] = rows_of([0] * n)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(km)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    form, rows, bits = ksk.form()
    centred, nb = ksk.key_bits()
    assert form == 1 and centred and nb == logQ - 1, (form, rows, bits, centred, nb)
    assert rows == -(-nb // bits)                              # ceil(nb / B) limbs
    for c in range(2):
        assert np.array_equal(got[c], orc.ct_mul_relin(km, a[c], b[c], logQ, p)), c
    # digits at their maximum against the extreme columns: scaled-down parts with every byte 0xff
    tp = np.zeros((1, 3, L, n), dtype=np.uint64)
    allones = (1 << logQ) - 1
    for part in range(3):
        tp[0, part] = rows_of([allones << logQ] * n)           # ScaleDown leaves 2^logQ - 1 (every digit at its maximum)
    dtp = ctx.upload(tp)
    out = ctx.alloc(2 * n * nl * 8)
    ctx.apply_key_switch_dev(ksk, logQ, dtp, 1, out, nl)
    want = orc.apply_key_switch(km, tp[0], logQ, nl)
    assert np.array_equal(out.download((2, n, nl)), want)
    # the general limbs and the reference's per-prime structure on the same matrix
    ctx.set_option("ks_long_keys", 1)
    long_ = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert not ksk.key_bits()[0] and ksk.form()[1] > rows
    ctx.set_option("ks_long_keys", 0)
    assert np.array_equal(long_, got)
    ctx.set_option("ks_direct", 1)
    direct = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    ctx.set_option("ks_direct", 0)
    assert np.array_equal(direct, got)
    # one coefficient a few bits larger: the measurement sees it
    km2 = km.copy()
    km2[1, 2] = rows_of([0] * (n - 1) + [-(half << 5) - 3])
    ksk2 = F.KeySwitchMatrix(ctx, 3, nd).upload(km2)
    got2 = ctx.ct_mul_relin(ksk2, logQ, p, a, b)
    assert ksk2.key_bits()[1] == logQ + 5
    assert np.array_equal(got2[0], orc.ct_mul_relin(km2, a[0], b[0], logQ, p))
    # a matrix of uniform residues (the other tests' matrices) is measured too, and takes the general limbs
    ksk3 = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    got3 = ctx.ct_mul_relin(ksk3, logQ, p, a, b)
    assert not ksk3.key_bits()[0]
    assert np.array_equal(got3[0], orc.ct_mul_relin(ksm, a[0], b[0], logQ, p))


@pytest.mark.parametrize("m,logQ,p,lin_lg", [(1006, 128, 23, 16), (46, 100, 23, 15), (1006, 128, 23, 17), (1006, 200, 23, 18), (46, 128, 47, 19), (101, 128, 23, 17), (22, 100, 23, 20),
                                             (65542, 128, 65543, 0),
                                             (65537, 128, 23, 0),
                                             (524287, 128, 23, 0)])         # the largest odd prime m on this path (2^19 - 1: 2 phi(m) - 1 = 2^20 - 5, rows of 2^20; three-term fold)          # m a Fermat prime: phi(m) = 2^16, the product's 2^17 - 1 coefficients fill the padded row to its last slot but one      # p = 65543: the first safe prime beyond 2^16 -- phi(m) = 32770, 2 phi(m) - 1 = 65539 > 2^16: rows of 2^17
def test_padded_rows_beyond_2_16_take_the_simple_path(m, logQ, p, lin_lg, monkeypatch):
    """The reference admits every m below 2^20 (FHEContext.cpp:89) and its drivers use m = p - 1 (Test_AddMul.cpp:131): for safe primes beyond
    65 537 the padded rows of the linear convolutions are 2^17 .. 2^20 long.  Those run the SIMPLE path (ntt32_core.inc): head and tail stages as
    passes of their own (ntt32_headS_kernel / ntt32_tailS_kernel, the digit polynomials through dig32_headS_kernel into plain rows), 2^S
    sub-transforms of 2^14 points in between, folds from whole rows -- instead of per-prime Bluestein rows (bluestein.cpp:93-144,
    CModulus.cpp:90-132).  FHESI_LIN_LG forces longer rows than a ring needs, so that rings the oracle finishes in seconds exercise every row
    length (2^17 .. 2^20); the ring that NEEDS rows of 2^17 (p = 65543) is held to the per-prime device path, uniform and generated keys (waves of
    sums of products on forced rows: test_gpu_ct_algebra.py::test_wave_of_products_on_linear_convolution_rings).
    The same hook at 15 and 16 puts the FUSED long-row loaders (second head stage inside rns32_reduce and the digit loader, ntt32_tail2_kernel:
    what m = 65266 runs) under the oracle for full multiplications."""
    if lin_lg:
        monkeypatch.setenv("FHESI_LIN_LG", str(lin_lg))
    count = 3
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 900 + m, count)
    n = ctx.phim
    lo, hi = -(1 << (logQ - 1)), (1 << (logQ - 1)) - 1
    rng = np.random.default_rng(3)
    a[1, 0] = O.ints_to_limbs([lo] * n, nl)                       # the extremes of the centred range everywhere
    b[1, 0] = O.ints_to_limbs([lo] * n, nl)
    a[1, 1] = O.ints_to_limbs([lo if v else hi for v in rng.integers(0, 2, n)], nl)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    da, db, dout = ctx.upload(a), ctx.upload(b), ctx.alloc(a.nbytes)
    ctx.prof_enable(True)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, count)
    ctx.sync()
    S = max(lin_lg, (2 * n - 2).bit_length(), 14) - 14              # rows of 2^(14 + S): what the ring needs, or what the hook forces
    assert f"ntt32_fwd_kernel3<true, {S}, " in ctx.prof_kernel_name("ntt_fwd_digits_main"), ctx.prof_kernel_name("ntt_fwd_digits_main")
    assert "rns32_reduce_kernel" in ctx.prof_kernel_name("rns_reduce") and ksk.form()[0] == 1
    ctx.prof_enable(False)
    got = dout.download((count, 2, n, nl))
    if m < 2000:
        for c in range(count):
            assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    # the reference's own structure on the device: tensor product over the chain, one dot product per chain prime (Bluestein rows)
    ctx.set_option("tensor32", 0)
    ctx.set_option("ks_direct", 1)
    kd = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    assert np.array_equal(ctx.ct_mul_relin(kd, logQ, p, a, b), got)
    ctx.set_option("tensor32", 1)
    ctx.set_option("ks_direct", 0)
    # a GENERATED matrix (centred limbs: the other recombination kernel), and a wave of sums of products (Matrix arithmetic of Regression)
    one = np.zeros((n, 1), dtype=np.uint64)
    one[0, 0] = 1
    t = F.DoubleCRT(ctx).sample(0, min(64, n // 2), 77, 1)
    t2 = t.copy()
    t2.op(t, 2)
    kg = F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded([F.DoubleCRT.from_poly(ctx, one), t, t2], t, logQ, 77, 78, 100, 3)
    got_g = ctx.ct_mul_relin(kg, logQ, p, a, b)
    assert kg.form()[0] == 1
    ctx.set_option("ks_direct", 1)
    kg_d = F.KeySwitchMatrix(ctx, 3, nd).upload(kg.download())
    assert np.array_equal(ctx.ct_mul_relin(kg_d, logQ, p, a, b), got_g)
    ctx.set_option("ks_direct", 0)
    if m < 2000:
        assert np.array_equal(got_g[2], orc.ct_mul_relin(kg.download(), a[2], b[2], logQ, p))


def test_the_largest_ring_the_reference_admits():
    """FHEContext.cpp:89 admits every m below 2^20; with the drivers' m = p - 1 (Test_AddMul.cpp:131) the largest is p = 1048343 (the largest safe
    prime below 2^20): phi(m) = 524170, padded rows of 2^20 = 64 sub-transforms of 2^14.  The context comes up in seconds (Phi_m by linear-time
    binomial divisions, chirp powers and twiddle tables by running products -- the quadratic division alone took minutes at this size),
    Phi_m = sum (-X)^i, and a multiplication on the fused path carries the same bits as the reference's own structure on the device (per-prime
    Bluestein rows of 2^21 points, tensor product over the chain, one dot product per chain prime)."""
    import time
    p, logQ, count = 1048343, 128, 2
    m = p - 1
    primes, roots = P.chain_for(m, logQ, p, 1, 60)
    t0 = time.time()
    ctx = F.Context(m, primes, roots)
    t_ctx = time.time() - t0
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    assert n == 524170 and t_ctx < 60, t_ctx
    assert np.array_equal(ctx.phi_m(), np.array([1 if i % 2 == 0 else -1 for i in range(n + 1)], dtype=np.int64))      # Phi_2q(X) = Phi_q(-X)
    rng = np.random.default_rng(11)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    lo = -(1 << (logQ - 1))
    a[1, 0] = O.ints_to_limbs([lo] * n, nl)                       # the extreme of the centred range everywhere
    b[1, 0] = O.ints_to_limbs([lo] * n, nl)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    ctx.prof_enable(True)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert "ntt32_fwd_kernel3<true, 6, " in ctx.prof_kernel_name("ntt_fwd_digits_main"), ctx.prof_kernel_name("ntt_fwd_digits_main")
    ctx.prof_enable(False)
    ctx.set_option("tensor32", 0)
    ctx.set_option("ks_direct", 1)
    kd = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    assert np.array_equal(ctx.ct_mul_relin(kd, logQ, p, a, b), got)


def test_the_largest_power_of_two_ring():
    """m = 2^20, the largest m FHEContext.cpp:89 admits (n = 2^19 coefficients): rows beyond the fused 32-bit path (n = 2^14, 2^15), so the
    multiplication runs over the chain with the 64-bit row kernels in their multi-pass form; against the oracle, and DoubleCRT <-> polynomial."""
    m, logQ, p, count = 1 << 20, 100, 23, 2
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 5, count)
    n = ctx.phim
    assert n == 1 << 19
    lo, hi = -(1 << (logQ - 1)), (1 << (logQ - 1)) - 1
    rng = np.random.default_rng(8)
    a[1, 0] = O.ints_to_limbs([lo] * n, nl)
    b[1, 0] = O.ints_to_limbs([lo] * n, nl)
    a[1, 1] = O.ints_to_limbs([lo if v else hi for v in rng.integers(0, 2, n)], nl)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    for c in range(count):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    limbs = P.rand_limbs(rng, (n,), nl + 1, logQ + 20)
    d = F.DoubleCRT.from_poly(ctx, limbs)
    rows = orc.dcrt_from_poly(limbs)
    assert np.array_equal(d.rows(), rows)
    W = ctx.L + 2
    assert np.array_equal(d.to_poly(W), orc.dcrt_to_poly(rows, W))


@pytest.mark.parametrize("m,logQ,p", [(22, 1280, 23), (46, 1300, 47), (1006, 1280, 23)])
def test_more_digit_columns_than_the_dot_product_tiles_hold(m, logQ, p):
    """logQ above 1264 gives more than 160 digit columns (3 x 162 at logQ = 1280, ByteDecomp digits): past what the LDS tiles of the exact
    integer dot products hold (launch_dot32).  Rings that take the 30-bit key switch otherwise (the linear-convolution rings here; n = 2^14 /
    2^15 likewise) then run the chain forms -- the call used to be refused with 'columns do not fit the LDS tile'.  Against the oracle on the
    small rings, against the reference's own structure on the device (tensor32 = 0, ks_direct = 1) everywhere."""
    count = 2
    ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 31 + m, count)
    assert 3 * nd > 160
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert ksk.form()[0] != 1                                    # not the 30-bit limb form
    if m < 100:
        for c in range(count):
            assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    ctx.set_option("tensor32", 0)
    ctx.set_option("ks_direct", 1)
    kd = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    assert np.array_equal(ctx.ct_mul_relin(kd, logQ, p, a, b), got)


@pytest.mark.parametrize("m,logQ,p,sizes", [(64, 100, 23, (1, 2, 4, 5, 7)), (22, 100, 23, (1, 2, 4, 7)), (1006, 128, 23, (1, 2, 4, 7)), (32768, 128, 23, (1, 2, 4))])
def test_digit_sizes_other_than_three_bytes(m, logQ, p, sizes):
    """FHEcontext's decompSize (FHEContext.h:86-115: bytes per ByteDecomp digit, default 3 and 3 in every driver) is a run-time argument of the
    key-switch calls: 1 and 2 bytes take the 30-bit limb form where the ring has it (other byte selectors in the digit loader, more columns),
    4 bytes and more the chain forms (a digit no longer fits a 30-bit residue).  Uniform and generated matrices, the extreme of the centred
    range in every coefficient; against the oracle on the small rings and against the reference's own structure on the device."""
    count = 3
    primes, roots = P.chain_for(m, logQ, p, 1, 60)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nl = ctx.phim, (logQ + 63) // 64
    one = np.zeros((n, 1), dtype=np.uint64)
    one[0, 0] = 1
    t = F.DoubleCRT(ctx).sample(0, min(64, n // 2), 77, 1)
    t2 = t.copy()
    t2.op(t, 2)
    for db in sizes:
        nd = R.ndigits(logQ, db)
        rng = np.random.default_rng(db)
        ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
        a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
        b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
        lo = -(1 << (logQ - 1))
        a[1, 0] = O.ints_to_limbs([lo] * n, nl)
        b[1, 0] = O.ints_to_limbs([lo] * n, nl)
        ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
        kg = F.KeySwitchMatrix(ctx, 3, nd).init_batch_seeded([F.DoubleCRT.from_poly(ctx, one), t, t2], t, logQ, 77, 78, 100, db)
        got = ctx.ct_mul_relin(ksk, logQ, p, a, b, decomp_bytes=db)
        got_g = ctx.ct_mul_relin(kg, logQ, p, a, b, decomp_bytes=db)
        limb32 = db < 4 and (ctx.phim == 1 << 14 or m in (22, 1006))
        assert (ksk.form()[0] == 1) == limb32 and (kg.form()[0] == 1) == limb32, (db, ksk.form(), kg.form())
        if m < 2000:
            for c in range(count):
                assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p, db)), (db, c)
            assert np.array_equal(got_g[1], orc.ct_mul_relin(kg.download(), a[1], b[1], logQ, p, db)), db
        ctx.set_option("tensor32", 0)
        ctx.set_option("ks_direct", 1)
        kd = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
        kgd = F.KeySwitchMatrix(ctx, 3, nd).upload(kg.download())
        assert np.array_equal(ctx.ct_mul_relin(kd, logQ, p, a, b, decomp_bytes=db), got), db
        assert np.array_equal(ctx.ct_mul_relin(kgd, logQ, p, a, b, decomp_bytes=db), got_g), db
        ctx.set_option("tensor32", 1)
        ctx.set_option("ks_direct", 0)


@pytest.mark.parametrize("m,logQ", [(64, 128), (22, 128), (1006, 160), (32768, 200)])
def test_plaintext_moduli_from_two_to_sixty_one_bits(m, logQ):
    """The plaintext modulus is a ZZ in the reference (FHEContext.h:97) and the factor of Ciphertext::operator*= and ScaleDown
    (Ciphertext.cpp:167-218); the C ABI takes 64 bits.  p = 2, a 31-, a 33-, a 41- and a 61-bit prime: the chain and the tensor half's prime
    plan are sized from p per call.  Against the oracle on the small rings, against the reference's own structure on the device everywhere."""
    count = 2
    for p in (2, (1 << 31) - 1, (1 << 32) + 15, (1 << 40) + 15, (1 << 61) - 1):
        ctx, orc, ksm, a, b, nd, nl = setup(m, logQ, p, 1, count)
        n = ctx.phim
        lo = -(1 << (logQ - 1))
        a[1, 0] = O.ints_to_limbs([lo] * n, nl)
        b[1, 0] = O.ints_to_limbs([lo] * n, nl)
        ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
        got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
        if m < 2000:
            for c in range(count):
                assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), (p, c)
        ctx.set_option("tensor32", 0)
        ctx.set_option("ks_direct", 1)
        kd = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
        assert np.array_equal(ctx.ct_mul_relin(kd, logQ, p, a, b), got), p
