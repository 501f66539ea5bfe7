"""GPU parity: the ciphertext algebra Matrix<Ciphertext> / Regression run between multiplications -- Ciphertext += , *= long,
>>= k, ApplyKeySwitch on unscaled parts (SumBatchedData, Regression.h:166-178) and the fused wave
sum-of-products + key switch (Matrix.cpp:57-79,150-174,227-263) -- through the C ABI vs the golden fixtures and the C oracle.
Bit-exact."""
import json
import os

import numpy as np
import pytest

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def I(v):
    return [int(x) for x in v]


def as_ints(a):
    return [O.limbs_to_ints(a[r]) for r in range(a.shape[0])]


def test_fixtures_add_mul_long_automorph_sum_batched():
    with open(os.path.join(G, "regression.json")) as f:
        cases = json.load(f)["ct_algebra"]
    for c in cases:           # m = 22 (Bluestein rows) and m = 32 (power-of-two rows)
        m, logQ = c["m"], c["logQ"]
        primes, roots = I(c["primes"]), I(c["roots"])
        ctx = F.Context(m, primes, roots)
        n, L, nd, nl = ctx.phim, len(primes), R.ndigits(logQ), (logQ + 63) // 64
        c1 = np.stack([O.ints_to_limbs(I(x), nl) for x in c["c1"]])
        c2 = np.stack([O.ints_to_limbs(I(x), nl) for x in c["c2"]])
        d1, d2 = ctx.upload(c1), ctx.upload(c2)
        acc = ctx.upload(c1)
        ctx.ct_add_dev(logQ, acc, d2, 2, nl, 1)
        assert as_ints(acc.download((2, n, nl))) == [I(x) for x in c["added"]]
        neg = ctx.upload(c1)
        ctx.ct_mul_long_dev(logQ, neg, -1, 2, nl, 1)
        assert as_ints(neg.download((2, n, nl))) == [I(x) for x in c["negated"]]
        tri = ctx.upload(c2)
        ctx.ct_mul_long_dev(logQ, tri, 3, 2, nl, 1)
        assert as_ints(tri.download((2, n, nl))) == [I(x) for x in c["tripled"]]
        ks = c["ks"]
        rot = ctx.alloc(2 * n * (nl + 1) * 8)
        ctx.ct_automorph_dev(ks[0], d2, 2, nl, 1, rot, nl + 1)
        assert as_ints(rot.download((2, n, nl + 1))) == [I(x) for x in c["rotated"]]
        ksks = [F.KeySwitchMatrix(ctx, 2, nd).upload(np.array([[[I(row) for row in col] for col in a[r]] for r in range(2)], dtype=np.uint64))
                for a in c["auto_ksm"]]
        sw = ctx.alloc(2 * n * nl * 8)
        ctx.ct_automorph_key_switch_dev(ksks[0], logQ, ks[0], d2, nl, 1, sw, nl)
        assert as_ints(sw.download((2, n, nl))) == [I(x) for x in c["switched"]]
        # k = 1: key switch of the already rotated ciphertext (its coefficients are centred modulo the chain, nl+1 limbs)
        sw1 = ctx.alloc(2 * n * nl * 8)
        ctx.ct_automorph_key_switch_dev(ksks[0], logQ, 1, rot, nl + 1, 1, sw1, nl)
        assert as_ints(sw1.download((2, n, nl))) == [I(x) for x in c["switched"]]
        # Regression::SumBatchedData
        cur, tmp = ctx.upload(c2), ctx.alloc(2 * n * nl * 8)
        for ksk, k in zip(ksks, ks):
            ctx.ct_automorph_key_switch_dev(ksk, logQ, k, cur, nl, 1, tmp, nl)
            ctx.ct_add_dev(logQ, cur, tmp, 2, nl, 1)
        parts = as_ints(cur.download((2, n, nl)))
        assert parts == [I(x) for x in c["summed"]]
        rctx = R.Ctx(m, logQ, c["p"], primes, roots)
        assert R.decrypt(rctx, I(c["t"]), parts) == c["summed_plain"]
        with pytest.raises(F.FhesiError):
            ctx.ct_automorph_dev(2, d2, 2, nl, 1, rot, nl + 1)         # not in Zm* (DoubleCRT.cpp:442-443)


@pytest.mark.parametrize("m,logQ,p", [(2048, 128, 23), (46, 90, 47), (4096, 300, 65537),
                                      (101, 90, 23),      # prime m: Ciphertext >>= as a gather with the Phi_m = sum X^i correction
                                      (45, 90, 23),       # a ring the gather does not cover: evaluation-form automorphism
                                      (17325, 90, 23)])   # ... and one beyond the long division in LDS (3^2 5^2 7 11: rem Phi_m by convolutions)
def test_batches_vs_oracle(m, logQ, p):
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    if m > 10000:
        orc.set_bluestein_fft(True)
    n, L, nd, nl = ctx.phim, len(primes), R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(m + logQ)
    count = 3
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    a[0, 0, 0] = O.ints_to_limbs([-(1 << (logQ - 1))], nl)[0]
    a[0, 0, 1] = O.ints_to_limbs([(1 << (logQ - 1)) - 1], nl)[0]
    b[0, 0, 0] = a[0, 0, 0]
    b[0, 0, 1] = a[0, 0, 1]
    da, db = ctx.upload(a), ctx.upload(b)
    acc = ctx.upload(a)
    ctx.ct_add_dev(logQ, acc, db, 2, nl, count)
    got = acc.download(a.shape)
    for c in range(count):
        assert np.array_equal(got[c], orc.ct_add(a[c], b[c], logQ)), c
    for l in (-1, 3, -(1 << 40) - 12345, (1 << 62) + 7):
        t = ctx.upload(a)
        ctx.ct_mul_long_dev(logQ, t, l, 2, nl, count)
        got = t.download(a.shape)
        for c in range(count):
            assert np.array_equal(got[c], orc.ct_mul_long(a[c], l, logQ)), (l, c)
    # Ciphertext >>= k and the automorphism key switch with a random matrix (timing and parity do not depend on key validity)
    ksm = np.stack([P.rand_rows(rng, primes, n, 2 * nd) for _ in range(2)])
    ksk = F.KeySwitchMatrix(ctx, 2, nd).upload(ksm)
    import math
    units = [k for k in range(2, m) if math.gcd(k, m) == 1]
    ks = units[:3] + [units[len(units) // 2], m - 1]
    for k in ks:
        rot = ctx.alloc(count * 2 * n * (nl + 1) * 8)
        ctx.ct_automorph_dev(k, da, 2, nl, count, rot, nl + 1)
        got = rot.download((count, 2, n, nl + 1))
        exp = [orc.ct_automorph(a[c], k, nl + 1) for c in range(count)]
        for c in range(count):
            assert np.array_equal(got[c], exp[c]), (k, c)
        out = ctx.alloc(count * 2 * n * nl * 8)
        ctx.ct_automorph_key_switch_dev(ksk, logQ, k, da, nl, count, out, nl)
        got = out.download((count, 2, n, nl))
        for c in range(count):
            assert np.array_equal(got[c], orc.apply_key_switch_parts(ksm, exp[c], logQ, nl)), (k, c)
        # the reference's own route (DoubleCRT::automorph on evaluation rows) gives the same bits as the coefficient gather
        ctx.set_option("automorph_rows", 1)
        ctx.ct_automorph_dev(k, da, 2, nl, count, rot, nl + 1)
        assert np.array_equal(rot.download((count, 2, n, nl + 1)), np.stack(exp)), k
        ctx.ct_automorph_key_switch_dev(ksk, logQ, k, da, nl, count, out, nl)
        assert np.array_equal(out.download((count, 2, n, nl)), got), k
        ctx.set_option("automorph_rows", 0)
    # scaled-up *= long
    tp = ctx.alloc(count * 3 * L * n * 8)
    ctx.ct_mul_dev(p, da, db, nl, count, tp)
    before = tp.download((count * 3, L, n))
    ctx.rows_mul_long_dev(tp, -1, count * 3)
    after = tp.download((count * 3, L, n))
    for i, q in enumerate(primes):
        assert np.array_equal(after[:, i, :], (np.uint64(q) - before[:, i, :]) % np.uint64(q))


@pytest.mark.parametrize("chunk,operands", [(None, None), ("2", None), (None, "4"), ("1", "2")])
def test_wave_of_products_sum_and_key_switch(chunk, operands, monkeypatch):
    """out[g] = KeySwitch(sum_t pool[a_t] * pool[b_t]) (fhesi_ct_mul_sum_relin_dev) vs the oracle composed the way Matrix.cpp does:
    operator*= per product, += on the scaled-up ciphertexts, then ApplyKeySwitch.  chunk bounds the groups per key-switch call,
    operands the distinct ciphertexts transformed per pass (small values force the piecewise accumulation of one group)."""
    m, logQ, p = 1024, 128, 23
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    if chunk:
        ctx.set_option("batch_chunk", int(chunk))
    if operands:
        ctx.set_option("wave_operands", int(operands))
    orc = O.Oracle(m, primes, roots)
    n, L, nd, nl = ctx.phim, len(primes), R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(99)
    npool = 6
    pool = P.rand_limbs(rng, (npool, 2, n), nl, logQ)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    groups = [[(0, 1)], [(2, 3), (4, 5), (1, 1)], [(5, 0), (3, 2)], [(0, 0), (1, 2), (3, 4), (5, 5), (2, 2)], [(4, 1)]]
    a_idx = [x for g in groups for x, _ in g]
    b_idx = [y for g in groups for _, y in g]
    seg = np.cumsum([0] + [len(g) for g in groups])
    dpool = ctx.upload(pool)
    out = ctx.alloc(len(groups) * 2 * n * nl * 8)
    ctx.ct_mul_sum_relin_dev(ksk, logQ, p, dpool, nl, a_idx, b_idx, seg, out)
    got = out.download((len(groups), 2, n, nl))
    for gi, g in enumerate(groups):
        tp = None
        for x, y in g:
            t = orc.ct_mul(pool[x], pool[y], p)
            if tp is None:
                tp = t
            else:
                for comp in range(3):
                    for i, q in enumerate(primes):
                        tp[comp][i] = (tp[comp][i] + t[comp][i]) % np.uint64(q)
        assert np.array_equal(got[gi], orc.apply_key_switch(ksm, tp, logQ, nl)), gi
    # the gather alone
    g_out = ctx.alloc(3 * 2 * n * nl * 8)
    ctx.ct_gather_dev(dpool, [5, 0, 5], 2 * n * nl, g_out)
    assert np.array_equal(g_out.download((3, 2, n, nl)), pool[[5, 0, 5]])


@pytest.mark.parametrize("m,logQ,p,lin_lg", [(46, 128, 47, 0), (101, 128, 23, 0), (16381, 128, 23, 0), (32602, 128, 32603, 0),
                                             (65266, 128, 65267, 0),   # padded rows of 2^16 (phi(m) = 32632): second head stage in the operand conversion, tail stages a pass of their own
                                             (1 << 16, 200, 23, 0),    # power-of-two rows of 2^15: head stage in the conversion, tail in the run-time CRT kernel
                                             # FHESI_LIN_LG: longer padded rows than the ring needs -- every group of the wave against the oracle on rows of 2^15 .. 2^18
                                             (46, 128, 47, 16), (101, 128, 23, 15), (1006, 128, 23, 17), (46, 128, 47, 18)])
def test_wave_of_products_on_linear_convolution_rings(m, logQ, p, lin_lg, monkeypatch):
    """Sums of products per group on the rings whose products run as linear convolutions over primes below 2^30 (kernels_tensor32.hip:
    m = 2q' and m an odd prime, padded rows of 2^14 or of 2^15 -- phi(m) = 16300 at p = 32603, the metric's size in the reference's own
    m = p - 1 parameterisation): the same bits as the chain path (option tensor32 = 0: per-prime Bluestein rows, CModulus.cpp:90-132 +
    bluestein.cpp:93-144) on every group, and as the oracle composed the way Matrix.cpp does where that takes seconds."""
    if lin_lg:
        monkeypatch.setenv("FHESI_LIN_LG", str(lin_lg))
    primes, roots = P.chain_for(m, logQ, p, 8)             # SetUpSIContext(xi = 8): the chain leaves room for sums of 8 products (FHEContext.cpp:83-85)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    if m > 2000:
        orc.set_bluestein_fft(True)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(m)
    npool = 5
    pool = P.rand_limbs(rng, (npool, 2, n), nl, logQ)
    lo, hi = -(1 << (logQ - 1)), (1 << (logQ - 1)) - 1
    pool[4, 0] = O.ints_to_limbs([lo if v else hi for v in rng.integers(0, 2, n)], nl)      # the extremes of the centred range everywhere
    pool[4, 1] = O.ints_to_limbs([lo] * n, nl)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    groups = [[(0, 1), (2, 3)], [(4, 4), (4, 4), (4, 4), (4, 4)], [(1, 1), (3, 0), (2, 4)]]
    a_idx = [x for g in groups for x, _ in g]
    b_idx = [y for g in groups for _, y in g]
    seg = np.cumsum([0] + [len(g) for g in groups])
    dpool = ctx.upload(pool)
    outs = []
    bits = ctx.get_option("tensor_bits")
    for t32 in (1, 0, 2):                                  # (2: the tensor half over the other prime size -- option tensor_bits 30 / 29)
        ctx.set_option("tensor32", 1 if t32 else 0)
        ctx.set_option("tensor_bits", 59 - bits if t32 == 2 else bits)
        ctx.set_option("wave_single", 0)
        out = ctx.alloc(len(groups) * 2 * n * nl * 8)
        ctx.prof_enable(True)
        ctx.ct_mul_sum_relin_dev(ksk, logQ, p, dpool, nl, a_idx, b_idx, seg, out)
        if t32:
            assert "tensor_sum32" in ctx.prof_kernel_name("tensor"), ctx.prof_kernel_name("tensor")
        ctx.prof_enable(False)
        outs.append(out.download((len(groups), 2, n, nl)))
    ctx.set_option("tensor32", 1)
    ctx.set_option("tensor_bits", bits)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    for gi in (() if m > 40000 else (0,) if m > 20000 else range(len(groups))):      # (m = 65266: minutes per oracle product; the chain path above stands in)
        tp = None
        for x, y in groups[gi]:
            t = orc.ct_mul(pool[x], pool[y], p)
            if tp is None:
                tp = t
            else:
                for comp in range(3):
                    for i, q in enumerate(primes):
                        tp[comp][i] = (tp[comp][i] + t[comp][i]) % np.uint64(q)
        assert np.array_equal(outs[0][gi], orc.apply_key_switch(ksm, tp, logQ, nl)), gi


@pytest.mark.parametrize("m,logQ,p", [(1024, 128, 23), (32768, 512, 23), (46, 90, 47)])
def test_wave_of_single_products_takes_the_batch_pipeline(m, logQ, p):
    """A wave whose groups are single products (the recorded loop `c *= d; ApplyKeySwitch(c)` of the host mirror) runs through the batch
    pipeline on gathered operands (option wave_single, default) -- same bits as the sum kernels (wave_single = 0), as
    fhesi_ct_mul_relin_batch_dev on the same pairs, and as the oracle; operands repeated, shared between groups, and squared."""
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(5)
    npool = 7
    pool = P.rand_limbs(rng, (npool, 2, n), nl, logQ)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    pairs = [(0, 1), (2, 2), (6, 0), (1, 0), (3, 4), (5, 6), (0, 1), (4, 4), (2, 5)]
    a_idx, b_idx, seg = [x for x, _ in pairs], [y for _, y in pairs], np.arange(len(pairs) + 1)
    dpool = ctx.upload(pool)
    outs = []
    for single in (1, 0):
        ctx.set_option("wave_single", single)
        out = ctx.alloc(len(pairs) * 2 * n * nl * 8)
        ctx.ct_mul_sum_relin_dev(ksk, logQ, p, dpool, nl, a_idx, b_idx, seg, out)
        outs.append(out.download((len(pairs), 2, n, nl)))
    assert np.array_equal(outs[0], outs[1])
    da, db, dout = ctx.upload(pool[a_idx]), ctx.upload(pool[b_idx]), ctx.alloc(len(pairs) * 2 * n * nl * 8)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, len(pairs), 3)
    assert np.array_equal(dout.download((len(pairs), 2, n, nl)), outs[0])
    for gi in ([0, 1, len(pairs) - 1] if n > 4096 else range(len(pairs))):
        x, y = pairs[gi]
        assert np.array_equal(outs[0][gi], orc.apply_key_switch(ksm, orc.ct_mul(pool[x], pool[y], p), logQ, nl)), gi


@pytest.mark.parametrize("m,logQ,p", [(4096, 128, 23), (2026, 120, 2027), (32768, 512, 23), (45, 100, 7), (101, 90, 7)])
def test_ct_add_const_and_mul_poly_vs_oracle(m, logQ, p):
    """Ciphertext::operator+=(const ZZX&) / operator*=(const ZZX&) on unscaled device batches (Ciphertext.cpp:147-156, 245-249 ->
    CiphertextPart::operator*=(ZZX) :29-36) against the C oracle's literal restatement (floor-divided scaled constant; integer product,
    rem Phi_m, Reduce): power-of-two rings incl. the metric ring, Test_General's ring (p = 2027, m = 2026), a composite and a prime m;
    one constant for the whole batch and one per ciphertext; messages in [0, p) (what a ZZ_pX holds) and signed ones."""
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nl, count = ctx.phim, (logQ + 63) // 64, 3
    rng = np.random.default_rng(m + logQ)
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    a[0, 0, 0] = O.ints_to_limbs([-(1 << (logQ - 1))], nl)[0]
    a[0, 1, 0] = O.ints_to_limbs([(1 << (logQ - 1)) - 1], nl)[0]
    small = n > 8192            # the oracle's schoolbook product is quadratic: one part of one ciphertext at the metric ring
    for npoly in (1, count):
        poly = rng.integers(0, p, size=(npoly, n)).astype(np.int64)
        poly[0, :3] = [-(p - 1), p - 1, -1]
        poly[-1, n // 2:] = 0
        buf = ctx.upload(a)
        ctx.ct_add_const_dev(logQ, p, buf, 2, nl, count, poly)
        got = buf.download(a.shape)
        for c in range(count):
            assert np.array_equal(got[c], orc.ct_add_const(a[c], poly[c % npoly], logQ, p)), (npoly, c)
        buf = ctx.upload(a)
        ctx.ct_mul_poly_dev(logQ, buf, 2, nl, count, poly)
        got = buf.download(a.shape)
        for c in ((count - 1,) if small else range(count)):
            want = orc.ct_mul_poly(a[c][:1] if small else a[c], poly[c % npoly], logQ)
            assert np.array_equal(got[c][:want.shape[0]], want), (npoly, c)
    # the largest machine-word constants still fit the chain (it is sized for logQ-bit x logQ-bit products, FHEContext.cpp:83-85)
    if m == 4096:
        big = np.full((1, n), -(1 << 62), dtype=np.int64)
        buf = ctx.upload(a[:1])
        ctx.ct_mul_poly_dev(logQ, buf, 2, nl, 1, big)
        assert np.array_equal(buf.download(a[:1].shape)[0], orc.ct_mul_poly(a[0], big[0], logQ))
