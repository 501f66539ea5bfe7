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


@pytest.mark.parametrize("m", [9, 15, 22, 46, 101, 1006, 45])
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


@pytest.mark.parametrize("m,logQ,p", [(22, 80, 23), (46, 90, 47), (166, 120, 167)])
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
