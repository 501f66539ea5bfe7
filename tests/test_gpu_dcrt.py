"""GPU parity: DoubleCRT object surface (DoubleCRT.h:83-365) through the C ABI vs the C oracle.  Bit-exact."""
import numpy as np
import pytest

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

pytestmark = pytest.mark.gpu


def make(m, logQ=100, p=23):
    primes, roots = P.chain_for(m, logQ, p)
    return F.Context(m, primes, roots), O.Oracle(m, primes, roots), primes


@pytest.mark.parametrize("m", [16, 64, 2048, 4096])
def test_from_poly_to_poly_roundtrip(m):
    ctx, orc, primes = make(m)
    L, n = len(primes), ctx.phim
    rng = np.random.default_rng(m)
    nl = 3
    limbs = P.rand_limbs(rng, (n,), nl, 160)
    limbs[0] = 0
    d = F.DoubleCRT.from_poly(ctx, limbs)
    rows_exp = orc.dcrt_from_poly(limbs)
    assert np.array_equal(d.rows(), rows_exp)
    W = L + 2
    assert np.array_equal(d.to_poly(W), orc.dcrt_to_poly(rows_exp, W))
    assert np.array_equal(d.to_poly(W, positive=True), orc.dcrt_to_poly(rows_exp, W, positive=True))
    # subset of the index set, and narrow output (truncation mod 2^(64 nlimbs))
    sub = [0, L - 1] if L > 1 else [0]
    assert np.array_equal(d.to_poly(W, index_set=sub), orc.dcrt_to_poly(rows_exp, W, idx=sub))
    assert np.array_equal(d.to_poly(1), orc.dcrt_to_poly(rows_exp, 1))
    # short polynomial (fewer coefficients than phi(m)) and the zero polynomial
    short = F.DoubleCRT.from_poly(ctx, limbs[:5])
    assert np.array_equal(short.rows(), orc.dcrt_from_poly(limbs[:5]))
    z = F.DoubleCRT(ctx)
    assert not z.rows().any()
    assert not z.to_poly(W).any()
    # CRT boundary values: coefficients at +-(P-1)/2 and just beyond wrap to the symmetric residue
    Pprod = 1
    for q in primes:
        Pprod *= q
    edge = [(Pprod - 1) // 2, -((Pprod - 1) // 2), (Pprod + 1) // 2, Pprod, -Pprod + 1, 1, -1, 0]
    el = O.ints_to_limbs(edge, W + 1)
    de = F.DoubleCRT.from_poly(ctx, el)
    got = O.limbs_to_ints(de.to_poly(W + 1))[:len(edge)]
    assert got == [(Pprod - 1) // 2, -((Pprod - 1) // 2), -((Pprod - 1) // 2), 0, 1, 1, -1, 0]
    assert np.array_equal(de.to_poly(W + 1), orc.dcrt_to_poly(orc.dcrt_from_poly(el), W + 1))


@pytest.mark.parametrize("m", [32, 4096])
def test_ops_scalar_automorph(m):
    ctx, orc, primes = make(m)
    n = ctx.phim
    rng = np.random.default_rng(m + 1)
    ra, rb = P.rand_rows(rng, primes, n)[0], P.rand_rows(rng, primes, n)[0]

    def dev(rows):
        d = F.DoubleCRT(ctx)
        for i in range(len(primes)):
            d.set_row(i, rows[i])
        return d

    for op in (F.OP_ADD, F.OP_SUB, F.OP_MUL):
        a, b = dev(ra), dev(rb)
        a.op(b, op)
        assert np.array_equal(a.rows(), orc.dcrt_op(ra, rb, op)), op
    big = (1 << 130) + 12345
    for num in (7, -1, big, -big, 0):
        for op in (F.OP_ADD, F.OP_SUB, F.OP_MUL, F.OP_SET):
            a = dev(ra)
            a.op_scalar(num, op)
            assert np.array_equal(a.rows(), orc.dcrt_op_scalar(ra, num, op)), (num, op)
    a = dev(ra)
    a.op_scalar(23, F.OP_DIV)
    assert np.array_equal(a.rows(), orc.dcrt_op_scalar(ra, 23, 3))
    with pytest.raises(F.FhesiError):       # divisor = 0 mod q_0
        dev(ra).op_scalar(primes[0], F.OP_DIV)
    # Exp (DoubleCRT.cpp:423-434): PowerMod per element, negative exponents invert, 0^-1 is an error
    for e in (0, 1, 2, 3, 65537, -1, -5):
        a = dev(ra)
        a.exp(e)
        assert np.array_equal(a.rows(), orc.dcrt_exp(ra, e)), e
    rz = ra.copy()
    rz[1, 7] = 0
    assert np.array_equal(dev(rz).exp(3).rows(), orc.dcrt_exp(rz, 3))
    assert np.array_equal(dev(rz).exp(0).rows(), np.ones_like(rz))       # PowerMod(0, 0) = 1
    with pytest.raises(F.FhesiError, match="inverse undefined"):
        dev(rz).exp(-2)
    for k in (3, 5, m - 1):
        a = dev(ra)
        a.automorph(k)
        assert np.array_equal(a.rows(), orc.dcrt_automorph(ra, k)), k
    with pytest.raises(F.FhesiError, match="k not in Zm"):
        dev(ra).automorph(2)
    # value semantics of copy / equality
    a = dev(ra)
    c = a.copy()
    assert c.equals(a)
    c.op_scalar(1, F.OP_ADD)
    assert not c.equals(a) and np.array_equal(a.rows(), ra)
    # out-of-range residue rejected like DoubleCRT::verify
    bad = ra[0].copy()
    bad[3] = np.uint64(primes[0])
    with pytest.raises(F.FhesiError, match="inconsistent"):
        a.set_row(0, bad)


def test_index_sets_add_remove_scrt():
    m = 128
    ctx, orc, primes = make(m, logQ=150)
    L, n = len(primes), ctx.phim
    assert L >= 4
    rng = np.random.default_rng(5)
    limbs = P.rand_limbs(rng, (n,), 2, 100)      # small enough to be exact modulo two primes
    full = orc.dcrt_from_poly(limbs)
    d = F.DoubleCRT.from_poly(ctx, limbs, index_set=[0, 2])
    assert d.index_set() == [0, 2]
    assert np.array_equal(d.rows(), full[[0, 2]])
    d.add_primes([1, 3])                        # toPoly + FFT on the new rows (DoubleCRT.cpp:142-156)
    assert d.index_set() == [0, 1, 2, 3]
    assert np.array_equal(d.rows(), full[:4])
    with pytest.raises(F.FhesiError, match="disjoint"):
        d.add_primes([1])
    d.remove_primes([0, 3])
    assert d.index_set() == [1, 2] and np.array_equal(d.rows(), full[[1, 2]])
    # mismatching index sets / contexts are errors (DoubleCRT.cpp:82-83)
    e = F.DoubleCRT(ctx)
    with pytest.raises(F.FhesiError):
        e.op(d, F.OP_ADD)
    ctx2 = F.Context(m, primes, ctx.roots)
    with pytest.raises(F.FhesiError, match="incompatible"):
        F.DoubleCRT(ctx2).assign(e)
    # SingleCRT <-> DoubleCRT (DoubleCRT.cpp:484-515)
    g = F.DoubleCRT.from_poly(ctx, limbs)
    coeff = g.to_scrt()
    for i in range(L):
        assert np.array_equal(coeff[i], orc.cmod_ifft(i, full[i]))
    h = F.DoubleCRT(ctx)
    h.from_scrt(coeff)
    assert h.equals(g)


def test_context_rejects_bad_primes():
    m = 32
    primes, roots = P.chain_for(m, 80, 23)
    with pytest.raises(F.FhesiError, match="not prime"):
        F.Context(m, [primes[0], 65 * 64 + 1], [roots[0], 3])
    with pytest.raises(F.FhesiError, match="1 mod 2m"):
        F.Context(m, [1000003], [2])
    with pytest.raises(F.FhesiError, match="already in chain"):
        F.Context(m, [primes[0], primes[0]], [roots[0], roots[0]])
    with pytest.raises(F.FhesiError, match="root"):
        F.Context(m, [primes[0]], [1])
    big = next(q for q in range((1 << 60) + 1, (1 << 60) + (1 << 20), 64) if R.is_prime(q))      # = 1 mod 2m but wider than NTL_SP_NBITS
    with pytest.raises(F.FhesiError, match="60 bits"):
        F.Context(m, [big], [R.find_root_2m(big, m)])
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    z, phi = orc.tables()
    assert np.array_equal(ctx.zms_idx(), z) and np.array_equal(ctx.phi_m(), phi)


@pytest.mark.parametrize("m,logQ", [(64, 100), (22, 100), (4096, 200), (1006, 120)])
def test_modulus_switching_vs_oracle(m, logQ):
    """DoubleCRT::addPrimesAndScale / scaleDownToSet (DoubleCRT.cpp:162-208, 518-558; SURVEY a12, kernel K11) as device calls
    (fhesi_dcrt_add_primes_and_scale / fhesi_dcrt_scale_down_to_set) against the C oracle: several target sets, coefficients at
    +-(P-1)/2 and 0, power-of-two and Bluestein rings; the error paths of the reference's asserts."""
    p = 23
    primes, roots = P.chain_for(m, logQ, p)
    L = len(primes)
    assert L >= 3
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n = ctx.phim
    rng = np.random.default_rng(m)
    Pall = 1
    for q in primes:
        Pall *= q
    W = L + 2
    limbs = P.rand_limbs(rng, (n,), W, 60 * L - 3)
    limbs[0] = O.ints_to_limbs([(Pall - 1) // 2], W)[0]
    limbs[1] = O.ints_to_limbs([-((Pall - 1) // 2)], W)[0]
    limbs[2] = 0
    rows = orc.dcrt_from_poly(limbs)
    for keep in ([0], [0, 1], list(range(L - 1)), [L - 1], [1, L - 1]):
        d = F.DoubleCRT.from_poly(ctx, limbs)
        d.scale_down_to_set(keep, p)
        assert d.index_set() == sorted(keep)
        want = orc.dcrt_scale_down_to_set(rows, range(L), keep, p)
        got = d.rows()
        for s, i in enumerate(sorted(keep)):
            assert np.array_equal(got[s], want[i]), (keep, i)
    for cur in ([0], [0, 1], [1]):
        add = [i for i in range(L) if i not in cur]
        d = F.DoubleCRT.from_poly(ctx, limbs, index_set=cur)
        lf = d.add_primes_and_scale(add, p)
        assert d.index_set() == list(range(L))
        want, wlf = orc.dcrt_add_primes_and_scale(rows, cur, add, p)
        assert np.array_equal(d.rows(), want)
        assert abs(lf - wlf) < 1e-9 * max(1.0, abs(wlf))
    d = F.DoubleCRT.from_poly(ctx, limbs)
    with pytest.raises(F.FhesiError):
        d.scale_down_to_set(range(L), p)                # nothing to drop (DoubleCRT.cpp:526)
    with pytest.raises(F.FhesiError):
        d.add_primes_and_scale([0], p)                  # not disjoint (DoubleCRT.cpp:167)
    e = F.DoubleCRT.from_poly(ctx, limbs, index_set=[0, 1])
    with pytest.raises(F.FhesiError):
        e.scale_down_to_set([2], p)                     # empty intersection (DoubleCRT.cpp:525)
    assert e.add_primes_and_scale([], p) == 0.0         # nothing to do (DoubleCRT.cpp:165)


@pytest.mark.parametrize("m,logQ", [(64, 100), (22, 100), (4096, 200)])
def test_single_crt_vs_oracle(m, logQ):
    """class SingleCRT (SingleCRT.h:41-175) through the C ABI vs the C oracle: operator=(ZZX) = PolyRed per prime, toPoly over
    the full set and over subsets (+-(P-1)/2 edges), Op(SingleCRT) add / sub, scalar add / sub (constant coefficient only), mul, /=,
    and the device-to-device conversions DoubleCRT = SingleCRT and toSingleCRT (DoubleCRT.cpp:484-515)."""
    primes, roots = P.chain_for(m, logQ, 23)
    L = len(primes)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n = ctx.phim
    rng = np.random.default_rng(m + 1)
    Pall = 1
    for q in primes:
        Pall *= q
    W = L + 2
    la, lb = P.rand_limbs(rng, (n,), W, 60 * L - 3), P.rand_limbs(rng, (n,), W, 60 * L + 40)     # b exceeds P: reduced modulo each prime
    la[0] = O.ints_to_limbs([(Pall - 1) // 2], W)[0]
    la[1] = O.ints_to_limbs([-((Pall - 1) // 2)], W)[0]
    la[2] = 0
    sa, sb = F.SingleCRT(ctx).assign_poly(la), F.SingleCRT(ctx).assign_poly(lb)
    ra, rb = orc.scrt_from_poly(la), orc.scrt_from_poly(lb)
    assert np.array_equal(sa.rows(), ra) and np.array_equal(sb.rows(), rb)
    short = F.SingleCRT(ctx).assign_poly(la[:5])                # fewer coefficients than phi(m): the rest is zero
    assert np.array_equal(short.rows(), orc.scrt_from_poly(la[:5]))
    assert np.array_equal(sa.to_poly(W), orc.scrt_to_poly(ra, W))
    for idx in ([0], [1, L - 1], list(range(L - 1))):
        assert np.array_equal(sb.to_poly(W, idx), orc.scrt_to_poly(rb, W, idx)), idx
    assert not sa.to_poly(W, []).any()
    # Op(SingleCRT, AddMod / SubMod) -- the same element-wise kernels as DoubleCRT::Op
    c = F.SingleCRT(ctx)
    c.assign(sa)
    c.op(sb, F.OP_ADD)
    assert np.array_equal(c.rows(), orc.dcrt_op(ra, rb, 0))
    c.op(sb, F.OP_SUB)
    assert c.equals(sa)
    with pytest.raises(F.FhesiError):
        c.op(sb, F.OP_MUL)                                      # no MulMod between SingleCRT objects (SingleCRT.h:127-133)
    # scalars: add / sub on the constant coefficient, mul / div on all of them
    k = (1 << 100) + 12345
    for op in (0, 1, 2, 3):
        c.assign(sa)
        c.op_scalar(k, op)
        assert np.array_equal(c.rows(), orc.scrt_op_scalar(ra, k, op)), op
    with pytest.raises(F.FhesiError):
        c.op_scalar(primes[0], 3)                               # InvMod of 0 (SingleCRT.cpp:288)
    # conversions in HBM
    d = F.DoubleCRT(ctx)
    F.dcrt_assign_scrt(d, sa)
    assert np.array_equal(d.rows(), orc.dcrt_from_poly(la))
    back = F.SingleCRT(ctx).assign_dcrt(d)
    assert back.equals(sa)
    part = F.SingleCRT(ctx).assign_dcrt(d, [0, L - 1])
    assert part.index_set() == [0, L - 1] and np.array_equal(part.rows(), ra[[0, L - 1]])
    with pytest.raises(F.FhesiError):
        d.op(sa, F.OP_ADD)                                      # mixing the two forms is an error
    with pytest.raises(F.FhesiError):
        sa.assign(d)
