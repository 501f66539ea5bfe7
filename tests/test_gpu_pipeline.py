"""GPU parity: Ciphertext::operator*= + KeySwitchSI::ApplyKeySwitch (Ciphertext.cpp:167-218, FHE-SI.cpp:241-260)
through the C ABI vs the C oracle, stage by stage and end to end.  Bit-exact."""
import numpy as np
import pytest

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

pytestmark = pytest.mark.gpu


def setup(m, logQ, p, seed, count):
    primes, roots = P.chain_for(m, logQ, p)
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
