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


@pytest.mark.parametrize("m,logQ,p", [(2048, 128, 23), (46, 90, 47), (4096, 511, 65537), (64, 64, 257)])
def test_encrypt_decrypt_vs_oracle(m, logQ, p):
    primes, roots = P.chain_for(m, logQ, p)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
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
