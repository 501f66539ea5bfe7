"""GPU parity: Cmod::FFT / iFFT rows (power-of-two m) through the C ABI vs the C oracle.  Bit-exact."""
import numpy as np
import pytest

import fhe_si_amd as F
import oracle_lib as O
import params as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("logn", [1, 2, 3, 5, 8, 10, 11, 12, 13, 14, 15, 16, 17, 18])
def test_rows_fwd_inv_vs_oracle(logn):
    n = 1 << logn
    m = 2 * n
    L = 3 if logn < 18 else 2
    primes, roots = P.first_primes(m, L)
    if logn <= 17:     # also a small / odd-sized prime, like the last prime of a chain (FHEContext.cpp:101-107): the tile kernels
        # (logn 11..17) transform its rows modulo a 60-bit multiple of it
        small, sroots = P.first_primes(m, 1, sp_nbits=max(20, logn + (4 if logn <= 12 else 8)))
        primes, roots = primes[:2] + small, roots[:2] + sroots
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    rng = np.random.default_rng(100 + logn)
    count = 3
    rows = P.rand_rows(rng, primes, n, count)
    rows[0, 0, :] = 0                      # zero row stays zero (bluestein.cpp:96-97)
    rows[1, 1, :] = np.uint64(primes[1] - 1)   # extreme residues
    buf = ctx.upload(rows)
    ctx.rows_ntt_fwd(buf, count)
    got = buf.download(rows.shape)
    for c in range(count):
        for i in range(len(primes)):
            exp = orc.fft_residues(i, rows[c, i])
            assert np.array_equal(got[c, i], exp), (logn, c, i)
    ctx.rows_ntt_inv(buf, count)
    back = buf.download(rows.shape)
    assert np.array_equal(back, rows)
    # inverse against the oracle on fresh evaluations
    ev = P.rand_rows(rng, primes, n, 1)
    buf2 = ctx.upload(ev)
    ctx.rows_ntt_inv(buf2, 1)
    got2 = buf2.download(ev.shape)
    for i in range(len(primes)):
        assert np.array_equal(got2[0, i], orc.cmod_ifft(i, ev[0, i])), (logn, i)


@pytest.mark.parametrize("logn", [11, 13, 14])
@pytest.mark.parametrize("bits", [49, 50, 53, 57, 59, 60])
def test_tile_rows_across_prime_sizes(logn, bits):
    """The tile kernels' store reduces a lazy value below 4q + 2^32 with a quotient ESTIMATE taken from the high word (norm_fwd63_x2,
    modarith63.h; the constant depends on the bit length of q >> 32): every size of chain prime the kernels take as it is (2^48 <= q < 2^60),
    on rows of extreme residues, against the oracle (CModulus.cpp:90-132)."""
    n = 1 << logn
    m = 2 * n
    primes, roots = P.first_primes(m, 3, sp_nbits=bits)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    rng = np.random.default_rng(1000 * logn + bits)
    count = 4
    rows = P.rand_rows(rng, primes, n, count)
    for i, q in enumerate(primes):
        rows[1, i, :] = np.uint64(q - 1)                                  # every residue at the top of its range
        rows[2, i, 0::2] = np.uint64(q - 1); rows[2, i, 1::2] = 0         # alternating extremes
        rows[3, i, :] = 0; rows[3, i, n - 1] = np.uint64(q - 1)           # one coefficient
    buf = ctx.upload(rows)
    ctx.rows_ntt_fwd(buf, count)
    got = buf.download(rows.shape)
    for c in range(count):
        for i in range(len(primes)):
            assert np.array_equal(got[c, i], orc.fft_residues(i, rows[c, i])), (logn, bits, c, i)
    assert int(got.max()) < max(primes)
    ctx.rows_ntt_inv(buf, count)
    assert np.array_equal(buf.download(rows.shape), rows)


def test_cmod_fft_single_row_bigint():
    m = 64
    primes, roots = P.chain_for(m, 100, 23)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    rng = np.random.default_rng(7)
    # signed big coefficients, more coefficients than phi(m) (degree >= phi(m) folds modulo Phi_m, >= m ignored)
    for ncoeffs in (5, 32, 40, 64, 70):
        limbs = P.rand_limbs(rng, (ncoeffs,), 3, 150)
        for i in range(len(primes)):
            assert np.array_equal(ctx.cmod_fft(i, limbs), orc.cmod_fft(i, limbs)), (ncoeffs, i)
    y = P.rand_rows(rng, primes, ctx.phim)[0]
    for i in range(len(primes)):
        assert np.array_equal(ctx.cmod_ifft(i, y[i]), orc.cmod_ifft(i, y[i]))


@pytest.mark.parametrize("m", [1 << 15, 1 << 16, 1006])
def test_aux32_transforms_are_a_ring_isomorphism(m):
    """The key switch's 32-bit auxiliary transforms (kernels_aux32.hip, four primes below 2^30; rows of 2^14 elements for n = 2^14 and
    for the linear-convolution rings, rows of 2^15 = head stage + two sub-transforms for n = 2^15): monomials multiply like monomials in
    Z_p[X]/(X^N + 1) (with the sign of the wrap) and inverse(forward(x)) = x -- checked inside the library."""
    primes, roots = P.chain_for(m, 128, 23)
    ctx = F.Context(m, primes, roots)
    ctx.selftest_aux32()
