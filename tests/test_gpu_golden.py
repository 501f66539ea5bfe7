"""GPU parity against the committed golden fixtures (tests/golden/*.json): the HIP path through the C ABI must
reproduce the independent Python big-int model bit for bit."""
import json
import os

import numpy as np
import pytest

import fhe_si_amd as F
import oracle_lib as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def I(v):
    return [int(x) for x in v]


def nlimbs_for(vals, extra=1):
    return (max([abs(int(v)).bit_length() for v in vals] + [1]) + 1 + 63) // 64 + extra


def pow2(m):
    return m & (m - 1) == 0


def test_transform_fixtures():
    done = 0
    for c in load("transforms.json")["cases"]:
        m, q, root = c["m"], int(c["q"]), int(c["root"])
        ctx = F.Context(m, [q], [root])
        x = I(c["x"])
        assert [int(v) for v in ctx.cmod_fft(0, O.ints_to_limbs(x, nlimbs_for(x)))] == I(c["fft"]), (m, q)
        assert [int(v) for v in ctx.cmod_ifft(0, np.array(I(c["ev"]), dtype=np.uint64))] == I(c["ifft"]), (m, q)
        done += 1
    assert done >= 10


def test_dcrt_fixtures():
    for c in load("dcrt.json")["cases"]:
        m, primes, roots = c["m"], I(c["primes"]), I(c["roots"])
        ctx = F.Context(m, primes, roots)
        L = len(primes)
        x = I(c["x"])
        d = F.DoubleCRT.from_poly(ctx, O.ints_to_limbs(x, nlimbs_for(x)))
        assert [[int(v) for v in r] for r in d.rows()] == [I(r) for r in c["rows"]]
        W = L + 2
        assert O.limbs_to_ints(d.to_poly(W)) == I(c["to_poly"])
        assert O.limbs_to_ints(d.to_poly(W, positive=True)) == I(c["to_poly_positive"])
        assert O.limbs_to_ints(d.to_poly(W, index_set=c["subset"])) == I(c["to_poly_subset"])
        y = I(c["y"])
        for name, op in (("add", F.OP_ADD), ("sub", F.OP_SUB), ("mul", F.OP_MUL)):
            e = d.copy()
            e.op(F.DoubleCRT.from_poly(ctx, O.ints_to_limbs(y, nlimbs_for(y))), op)
            assert [[int(v) for v in r] for r in e.rows()] == [I(r) for r in c[name]], name
        e = d.copy()
        e.op_scalar(int(c["scalar"]), F.OP_MUL)
        assert [[int(v) for v in r] for r in e.rows()] == [I(r) for r in c["mul_scalar"]]
        e = d.copy()
        e.op_scalar(c["p"], F.OP_DIV)
        assert [[int(v) for v in r] for r in e.rows()] == [I(r) for r in c["div_scalar"]]
        e = d.copy()
        e.automorph(c["automorph_k"])
        assert [[int(v) for v in r] for r in e.rows()] == [I(r) for r in c["automorph"]]


def test_mul_relin_fixtures():
    for c in load("ciphertext.json")["mul_relin"]:
        m, logQ, p = c["m"], c["logQ"], c["p"]
        primes, roots = I(c["primes"]), I(c["roots"])
        ctx = F.Context(m, primes, roots)
        L, nl, n = len(primes), (logQ + 63) // 64, ctx.phim
        a = np.stack([O.ints_to_limbs(I(x), nl) for x in c["c1"]])[None]
        b = np.stack([O.ints_to_limbs(I(x), nl) for x in c["c2"]])[None]
        ksm = np.array([[[I(row) for row in col] for col in c["ksm"][r]] for r in range(2)], dtype=np.uint64)
        ksk = F.KeySwitchMatrix(ctx, 3, ksm.shape[1] // 3).upload(ksm)
        da, db = ctx.upload(a), ctx.upload(b)
        tp = ctx.alloc(3 * L * n * 8)
        ctx.ct_mul_dev(p, da, db, nl, 1, tp)
        tprod = tp.download((3, L, n))
        assert [[[int(v) for v in tprod[k][i]] for i in range(L)] for k in range(3)] == [[I(r) for r in t] for t in c["tprod"]]
        out = ctx.ct_mul_relin(ksk, logQ, p, a, b)[0]
        assert [O.limbs_to_ints(out[r]) for r in range(2)] == [I(x) for x in c["result"]]
