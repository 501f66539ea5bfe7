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
