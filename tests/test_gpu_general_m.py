"""GPU parity for general (non power-of-two) m -- the reference's own parameterisation m = p-1, p a safe prime
(Test_AddMul.cpp:131, README:35): device Bluestein (bluestein.cpp:93-144) + Z_m^* gather/scatter + reduction modulo
Phi_m (CModulus.cpp:90-132) through the C ABI vs the C oracle.  Bit-exact."""
import numpy as np
import pytest

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m", [3, 6, 9, 12, 15, 21, 22, 46, 101, 1006, 45])      # (3, 6: phi(m) = 2, the smallest rings that are not powers of two)
def test_rows_general_m(m):
    primes, roots = P.first_primes(m, 3)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n = ctx.phim
    rng = np.random.default_rng(m)
    count = 2
    rows = P.rand_rows(rng, primes, n, count)
    rows[0, 0, :] = 0
    buf = ctx.upload(rows)
    ctx.rows_ntt_fwd(buf, count)
    got = buf.download(rows.shape)
    for c in range(count):
        for i in range(len(primes)):
            assert np.array_equal(got[c, i], orc.fft_residues(i, rows[c, i])), (m, c, i)
    ctx.rows_ntt_inv(buf, count)
    assert np.array_equal(buf.download(rows.shape), rows)
    ev = P.rand_rows(rng, primes, n, 1)
    b2 = ctx.upload(ev)
    ctx.rows_ntt_inv(b2, 1)
    g2 = b2.download(ev.shape)
    for i in range(len(primes)):
        assert np.array_equal(g2[0, i], orc.cmod_ifft(i, ev[0, i])), (m, i)


@pytest.mark.parametrize("m", [8422, 16411, 32771])
def test_large_safe_prime_ring_single_rows(m):
    """m = 8422: config 4a of SURVEY.md section 8d (p = 8423; convolution size N = 2^15: order-free transforms with the head stage
    fused into the pre-multiplication and the tail stage into the post-processing); m = 16411 and 32771 (primes): N = 2^16 and
    2^17, order-free transforms with 2 and 3 separate head / tail stages."""
    primes, roots = P.first_primes(m, 2)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    rng = np.random.default_rng(4)
    rows = P.rand_rows(rng, primes, ctx.phim, 1)
    buf = ctx.upload(rows)
    ctx.rows_ntt_fwd(buf, 1)
    got = buf.download(rows.shape)
    assert np.array_equal(got[0, 0], orc.fft_residues(0, rows[0, 0]))
    ctx.rows_ntt_inv(buf, 1)
    assert np.array_equal(buf.download(rows.shape), rows)
    ev = P.rand_rows(rng, primes, ctx.phim, 1)
    b2 = ctx.upload(ev)
    ctx.rows_ntt_inv(b2, 1)
    assert np.array_equal(b2.download(ev.shape)[0, 1], orc.cmod_ifft(1, ev[0, 1]))


@pytest.mark.parametrize("m,logQ,p", [(3, 80, 7), (6, 80, 7), (12, 80, 13), (22, 80, 23), (46, 90, 47), (166, 120, 167)])
def test_mul_relin_reference_parameterisation(m, logQ, p):
    """README smoke parameters `80 23 7` (README:46-47) and friends: full mult + key switch on the GPU."""
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(m)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    a = P.rand_limbs(rng, (2, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (2, 2, n), nl, logQ)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    for c in range(2):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    # DoubleCRT surface on the same ring
    limbs = P.rand_limbs(rng, (n,), nl + 1, logQ + 20)
    d = F.DoubleCRT.from_poly(ctx, limbs)
    rows = orc.dcrt_from_poly(limbs)
    assert np.array_equal(d.rows(), rows)
    W = len(primes) + 2
    assert np.array_equal(d.to_poly(W), orc.dcrt_to_poly(rows, W))
    k = next(k for k in range(2, m) if R.zms_idx(m)[0][k] >= 0)
    d.automorph(k)
    assert np.array_equal(d.rows(), orc.dcrt_automorph(rows, k))


@pytest.mark.parametrize("m,hook", [(9, 1), (15, 1), (45, 1), (105, 1), (360, 1), (1155, 1), (4620, 1), (16380, 1),
                                    (17325, 0), (20480, 0), (30030, 0)])
def test_generic_m_reduction_modulo_phi_by_convolutions(m, hook, monkeypatch):
    """rem(., Phi_m) for m that is neither a power of two, a prime, nor twice a prime (CModulus.cpp:128-129; the reference admits every m below
    2^20, FHEContext.cpp:89): up to m = 16384 the device divides in LDS, above that -- and at any size under FHESI_PHI_CONV=1 -- by two exact
    convolutions, f - Phi_m * top(f * Psi_m) with Psi_m = (X^m - 1) / Phi_m (bluestein.hip).  Squarefree m with three to five prime factors,
    prime powers, m = 2^12 * 5; inverse transforms of RANDOM evaluation vectors (polynomials of full degree m - 1 before the reduction) against
    the oracle, bit for bit, and the forward / inverse round trip."""
    if hook:
        monkeypatch.setenv("FHESI_PHI_CONV", "1")
    primes, roots = P.first_primes(m, 2)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    if m > 2000:
        orc.set_bluestein_fft(True)
    n = ctx.phim
    rng = np.random.default_rng(m)
    ev = P.rand_rows(rng, primes, n, 2)
    ev[1, 0, :] = np.uint64(primes[0] - 1)                       # the largest residue everywhere
    buf = ctx.upload(ev)
    ctx.rows_ntt_inv(buf, 2)
    got = buf.download(ev.shape)
    for c in range(2):
        for i in range(len(primes)):
            assert np.array_equal(got[c, i], orc.cmod_ifft(i, ev[c, i])), (m, c, i)
    ctx.rows_ntt_fwd(buf, 2)
    assert np.array_equal(buf.download(ev.shape), ev)
    if hook:                                                     # ... and the same bits as the long division in LDS
        monkeypatch.delenv("FHESI_PHI_CONV")
        ctx2 = F.Context(m, primes, roots)
        b2 = ctx2.upload(ev)
        ctx2.rows_ntt_inv(b2, 2)
        assert np.array_equal(b2.download(ev.shape), got)


@pytest.mark.parametrize("m,phim", [(1048575, 480000), (1048574, 524286), (1048573, 1048572)])
def test_rows_at_the_largest_rings_of_each_kind(m, phim):
    """The three largest m below 2^20 happen to be one of each kind: 1048575 = 3 * 5^2 * 11 * 31 * 41 (generic: rem Phi_m by convolutions),
    1048574 = 2 * 524287 (twice a prime, beyond the padded rows of the linear-convolution path) and 1048573 (prime).  Bluestein convolutions
    of 2^21 points: the context comes up, and polynomial -> evaluations -> polynomial is the identity -- the inverse transform scatters the
    evaluations over Z_m^*, which gives a polynomial of degree m - 1 that only the reduction modulo Phi_m brings back below phi(m)."""
    primes, roots = P.first_primes(m, 2)
    ctx = F.Context(m, primes, roots)
    assert ctx.phim == phim
    rng = np.random.default_rng(1)
    rows = P.rand_rows(rng, primes, ctx.phim, 2)
    rows[1, 1, :] = np.uint64(primes[1] - 1)
    buf = ctx.upload(rows)
    ctx.rows_ntt_fwd(buf, 2)
    assert not np.array_equal(buf.download(rows.shape), rows)
    ctx.rows_ntt_inv(buf, 2)
    assert np.array_equal(buf.download(rows.shape), rows)


def test_mul_relin_on_a_composite_ring_beyond_the_lds_division():
    """The metric's multiplication (Ciphertext::operator*= + ApplyKeySwitch + ScaleDown) on m = 17325 = 3^2 5^2 7 11 (phi(m) = 7200): a ring the
    reference admits and the device refused until round 6 (generic m above 16384).  Per-prime Bluestein rows, reduction modulo Phi_m by
    convolutions; against the oracle (its Bluestein-FFT mode), and DoubleCRT <-> polynomial on the same ring."""
    m, logQ, p = 17325, 100, 23
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    orc.set_bluestein_fft(True)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    assert n == 7200
    rng = np.random.default_rng(m)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    a = P.rand_limbs(rng, (2, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (2, 2, n), nl, logQ)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    assert np.array_equal(got[1], orc.ct_mul_relin(ksm, a[1], b[1], logQ, p))
    limbs = P.rand_limbs(rng, (n,), nl + 1, logQ + 20)
    d = F.DoubleCRT.from_poly(ctx, limbs)
    rows = orc.dcrt_from_poly(limbs)
    assert np.array_equal(d.rows(), rows)
    W = len(primes) + 2
    assert np.array_equal(d.to_poly(W), orc.dcrt_to_poly(rows, W))
