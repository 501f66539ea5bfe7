"""GPU parity over RANDOM shapes: the fused ciphertext multiplication + key switch (fhesi_ct_mul_relin_batch, Ciphertext.cpp:167-218 +
FHE-SI.cpp:241-260) against the C oracle on rings, moduli, plaintext moduli and batch sizes drawn from a seeded generator -- the fixed-shape
tests pin the configurations the reference's drivers use; this one walks the plan selection in between (limb plans of the key switch,
bounds of the 30-bit tensor half, chunking, ragged batches, launches of one)."""
import numpy as np
import pytest

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

pytestmark = pytest.mark.gpu

POW2 = [8, 16, 32, 64, 128, 256, 512, 1024, 2048]
SAFE = [22, 46, 94, 118, 166, 214]              # m = 2 q', q' prime: the reference's m = p - 1 rings
GENERAL = [9, 12, 15, 20, 21, 45, 60, 100, 101, 105, 126]      # everything else: Bluestein rows (Phi_105 has a coefficient -2)


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    general = 24 <= seed < 36 or (seed >= 36 and seed % 4 == 3)          # (seeds from 36 upwards: tools/fuzz_shapes.py)
    m = int(rng.choice(GENERAL if general else (POW2 if seed % 3 else SAFE)))
    logQ = int(rng.integers(40, 420))
    p = int(rng.choice([2, 3, 23, 257, 2027, 8423, 65537, int(rng.integers(2, 1 << 20)), (1 << 31) - 1]))
    count = int(rng.integers(1, 6))
    return m, logQ, p, count


def _sp_nbits(seed):
    return 50 if seed % 5 == 0 else 60              # where the chain starts (FHEContext.cpp:92): the NTL of the reference's era, or today's


@pytest.mark.parametrize("seed", range(36))
def test_mul_relin_on_random_shapes(seed):
    m, logQ, p, count = _case(seed)
    primes, roots = P.chain_for(m, logQ, p, 1, _sp_nbits(seed))
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(seed)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    a[0, 0, 0] = O.ints_to_limbs([-(1 << (logQ - 1))], nl)[0]          # the extremes of the centred range
    b[0, 1, 0] = O.ints_to_limbs([(1 << (logQ - 1)) - 1], nl)[0]
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    if seed % 4 == 3:
        ctx.set_option("batch_chunk", 2)                                # ragged chunks
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    form = ksk.form()
    for c in range(count):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), (m, logQ, p, count, c, form)
    # the same pairs as a wave of single products through pool indices (the recorded per-object loop of the host mirror)
    pool = np.concatenate([a, b])
    dpool = ctx.upload(pool)
    out = ctx.alloc(count * 2 * n * nl * 8)
    ctx.ct_mul_sum_relin_dev(ksk, logQ, p, dpool, nl, list(range(count)), list(range(count, 2 * count)), np.arange(count + 1), out)
    assert np.array_equal(out.download((count, 2, n, nl)), got), (m, logQ, p, count, "wave")


BIG = [32768, 32768, 1 << 16, 8422, 16381, 32602]       # rings whose key switch runs over the four 30-bit auxiliary primes at full row length (the last two: fold and tail stage of padded 2^15-point rows in the recombination's loader, m prime and m = 2q')


def _case_generated(seed):
    rng = np.random.default_rng(5000 + seed)
    m = int(BIG[seed - 12]) if 12 <= seed < 12 + len(BIG) else int(rng.choice(SAFE + [101, 107, 227] + (POW2 if seed >= 18 else [])))
    logQ = int(rng.integers(64, 500))
    p = int(rng.choice([2, 23, 257, 8423, 65537, int(rng.integers(2, 1 << 20))]))
    count = 1 if 12 <= seed < 12 + len(BIG) else int(rng.integers(1, 4))
    return m, logQ, p, count


@pytest.mark.parametrize("seed", range(18))
def test_mul_relin_on_random_shapes_with_generated_keys(seed):
    """The same walk with key-switch matrices shaped like the ones KeySwitchSI::Init produces (FHE-SI.cpp:176-204): integer coefficients
    uniform in [-2^(logQ-1), 2^(logQ-1)) plus the extremes, as residue rows.  On the rings whose key switch runs over the four 30-bit auxiliary
    primes the library measures them and cuts centred limbs (the count it reports must be ceil(nb / B)); everywhere the result is the oracle's."""
    m, logQ, p, count = _case_generated(seed)
    primes, roots = P.chain_for(m, logQ, p, 1, _sp_nbits(seed))
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    if m > 2000 and (m & (m - 1)) != 0:
        orc.set_bluestein_fft(True)
    n, nd, nl, W = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64, len(primes) + 2
    rng = np.random.default_rng(seed)
    half = 1 << (logQ - 1)
    ksm = np.empty((2, 3 * nd, len(primes), n), dtype=np.uint64)
    for r in range(2):
        for c in range(3 * nd):
            ksm[r, c] = orc.dcrt_from_poly(P.rand_limbs(rng, (n,), W, logQ))
    ksm[0, 0] = orc.dcrt_from_poly(O.ints_to_limbs([-half if i % 3 else half - 1 for i in range(n)], W))
    ksm[1, 3 * nd - 1] = orc.dcrt_from_poly(O.ints_to_limbs([half] + [0] * (n - 1), W))        # -poly for poly = -2^(logQ-1)
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    form, rows, bits = ksk.form()
    centred, nb = ksk.key_bits()
    if form == 1:
        assert centred and nb == logQ - 1 and rows == -(-nb // bits), (m, logQ, form, rows, bits, centred, nb)
    for c in range(count):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), (m, logQ, p, count, c, form, rows)
