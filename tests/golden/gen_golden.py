#!/usr/bin/env python3
"""Generates tests/golden/*.json from the independent Python big-int restatement (oracle/fhesi_pyref.py).

Run from the repo root:  python tests/golden/gen_golden.py
The reference itself cannot run here (needs NTL), so these vectors come from the exact-integer Python model; every
fixture states its primes and roots explicitly (the reference draws roots at random, NumbTh.cpp:101-115).
Big integers are stored as decimal strings.  The transform fixtures are produced by the LITERAL Bluestein restatement
(bluestein.cpp:93-144) and cross-checked here against the reference's slow definition tDFT (bluestein.cpp:149-172).
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import fhesi_pyref as R  # noqa: E402


def S(v):
    return [str(int(x)) for x in v]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def chain(m, logQ, p):
    _, phim = R.zms_idx(m)
    primes = R.add_primes_by_size(m, R.si_context_size(logQ, p, phim))
    return primes, [R.find_root_2m(q, m) for q in primes]


def gen_transforms():
    cases = []
    rng = random.Random(20241001)
    for m, logQ, p in [(22, 80, 23), (16, 80, 23), (32, 80, 23), (9, 60, 19), (15, 70, 31), (46, 90, 47), (1024, 100, 23)]:
        primes, roots = chain(m, logQ, p)
        idx, phim = R.zms_idx(m)
        for i, (q, root) in enumerate(zip(primes, roots)):
            if m == 1024 and i > 0:
                continue
            x = [rng.randrange(-(1 << (logQ + 8)), 1 << (logQ + 8)) for _ in range(phim)]
            x[0] = -1
            if phim > 3:
                x[1], x[2], x[3] = 0, q, -q - 1
            y = R.cmod_fft(x, m, q, root)                      # literal Bluestein
            if m <= 64:
                assert y == R.cmod_fft(x, m, q, root, use_bluestein=False)      # == tDFT definition
                assert R.cmod_ifft(y, m, q, root) == [c % q for c in x]
            if m & (m - 1) == 0:
                assert y == R._ntt_pow2([c % q for c in x], m // 2, q, root * root % q)
            ev = [rng.randrange(q) for _ in range(phim)]
            cases.append({"m": m, "q": str(q), "root": str(root), "x": S(x), "fft": S(y),
                          "ev": S(ev), "ifft": S(R.cmod_ifft(ev, m, q, root) if m <= 64 else R._intt_pow2(ev, m // 2, q, root * root % q))})
    # zero input -> zero row (bluestein.cpp:96-97); long input: degree >= m ignored, degrees phi(m)..m-1 fold modulo Phi_m
    m, (primes, roots) = 22, chain(22, 80, 23)
    q, root = primes[0], roots[0]
    _, phim = R.zms_idx(m)
    long_x = [rng.randrange(-(1 << 70), 1 << 70) for _ in range(m + 5)]
    cases.append({"m": m, "q": str(q), "root": str(root), "x": S([0] * phim), "fft": S(R.cmod_fft([0] * phim, m, q, root)), "ev": S([0] * phim),
                  "ifft": S([0] * phim)})
    cases.append({"m": m, "q": str(q), "root": str(root), "x": S(long_x), "fft": S(R.cmod_fft(long_x, m, q, root)), "ev": S([1] * phim),
                  "ifft": S(R.cmod_ifft([1] * phim, m, q, root))})
    dump("transforms.json", {"cases": cases})


def gen_dcrt():
    out = []
    rng = random.Random(77)
    for m, logQ, p in [(22, 80, 23), (32, 80, 23), (15, 70, 31), (128, 150, 257)]:
        primes, roots = chain(m, logQ, p)
        ctx = R.Ctx(m, logQ, p, primes, roots)
        n, L = ctx.phim, ctx.L
        P = 1
        for q in primes:
            P *= q
        half = (P - 1) // 2
        x = [rng.randrange(-P, 2 * P) for _ in range(n)]
        edge = [half, -half, half + 1, -half - 1, P, -P, 0, 1, -1, P + 5]
        for i, v in enumerate(edge[:n]):
            x[i] = v
        rows = R.dcrt_from_poly(ctx, x)
        y = [rng.randrange(-(1 << 60), 1 << 60) for _ in range(n)]
        rows_y = R.dcrt_from_poly(ctx, y)
        k = next(k for k in range(2, m) if ctx.idx[k] >= 0)
        sub = sorted({0, L - 1})
        case = {
            "m": m, "logQ": logQ, "p": p, "primes": S(primes), "roots": S(roots),
            "x": S(x), "rows": [S(rows[i]) for i in range(L)],
            "to_poly": S(R.dcrt_to_poly(ctx, rows)), "to_poly_positive": S(R.dcrt_to_poly(ctx, rows, positive=True)),
            "subset": sub, "to_poly_subset": S(R.dcrt_to_poly(ctx, rows, idxset=sub)),
            "y": S(y), "rows_y": [S(rows_y[i]) for i in range(L)],
            "add": [S(r) for r in R.dcrt_op(ctx, rows, rows_y, "add").values()],
            "sub": [S(r) for r in R.dcrt_op(ctx, rows, rows_y, "sub").values()],
            "mul": [S(r) for r in R.dcrt_op(ctx, rows, rows_y, "mul").values()],
            "scalar": str(-(1 << 70) - 3),
            "mul_scalar": [S(r) for r in R.dcrt_op_scalar(ctx, rows, -(1 << 70) - 3, "mul").values()],
            "div_scalar": [S(r) for r in R.dcrt_div_scalar(ctx, rows, p).values()],
            "automorph_k": k, "automorph": [S(r) for r in R.dcrt_automorph(ctx, rows, k).values()],
        }
        # to_poly of a product equals the polynomial product modulo (Phi_m, P), centred
        prod = R.poly_mul_mod_phi(ctx, x, y)
        cen = [((c + half) % P) - half for c in prod]
        assert R.dcrt_to_poly(ctx, R.dcrt_op(ctx, rows, rows_y, "mul")) == cen
        out.append(case)
    # intVecCRT edge cases (NumbTh.cpp:307-335): short vq, centred boundaries
    q1, q2 = 1152921504606845777, 1152921504606845161
    vp = [q1 // 2, -(q1 // 2), 0, 5, -5, q1 // 2 - 1]
    vq = [0, q2 - 1, q2 // 2, q2 // 2 + 1]          # shorter than vp: tail treated as 0 mod q
    res = list(vp)
    R.int_vec_crt(res, q1, vq, q2)
    dump("dcrt.json", {"cases": out, "intveccrt": {"p": str(q1), "q": str(q2), "vp": S(vp), "vq": S(vq), "out": S(res)}})


def gen_ciphertext():
    out = {}
    # Reduce (Util.cpp:3-26) at the range boundaries
    red = []
    for logQ in (80, 64, 100, 512):
        Q = 1 << logQ
        vals = [0, 1, -1, Q // 2 - 1, Q // 2, -Q // 2, -Q // 2 - 1, Q, -Q, Q + 7, 3 * Q + Q // 2, -5 * Q - 9]
        red.append({"logQ": logQ, "vals": S(vals), "centered": S([R.reduce_logq(v, logQ) for v in vals]),
                    "positive": S([R.reduce_logq(v, logQ, True) for v in vals])})
    out["reduce"] = red
    # ScaleDown rounding (Ciphertext.cpp:205-210): floor semantics for negatives, ties round up
    sd = []
    for logQ in (80, 128):
        Q = 1 << logQ
        vals = [0, Q // 2 - 1, Q // 2, Q // 2 + 1, -Q // 2, -Q // 2 - 1, -Q // 2 + 1, 3 * Q + Q // 2, -3 * Q - Q // 2, -Q, Q * Q // 2 + Q // 2, -(Q * Q // 2)]
        sd.append({"logQ": logQ, "vals": S(vals), "out": S([R.scale_down_coeff(v, logQ) for v in vals])})
    out["scale_down"] = sd
    # ByteDecomp digit order (Ciphertext.cpp:82-121)
    logQ, nd = 80, R.ndigits(80)
    poly = [0x0123456789ABCDEF0123, -1, 1 << 79, -(1 << 79), 0xFFFFFF, 1 << 24]
    out["byte_decomp"] = {"logQ": logQ, "nd": nd, "poly": S(poly), "digits": [S(d) for d in R.byte_decomp_part(poly, logQ, nd)]}
    # full mult + key switch with valid keys (Test_AddMul.cpp:59-67,84-86), smoke config and a power-of-two ring
    e2e = []
    for m, logQ, p, seed in [(22, 80, 23, 1), (32, 80, 23, 2), (64, 100, 257, 3)]:
        primes, roots = chain(m, logQ, p)
        ctx = R.Ctx(m, logQ, p, primes, roots)
        prng = R.SplitMix64(seed)
        t, pk = R.keygen(ctx, prng)
        n = ctx.phim
        m1 = [prng.bnd(p) for _ in range(n)]
        m2 = [prng.bnd(p) for _ in range(n)]
        c1, c2 = R.encrypt(ctx, pk, m1, prng), R.encrypt(ctx, pk, m2, prng)
        assert R.decrypt(ctx, t, c1) == m1
        ksm = R.key_switch_init_s2(ctx, t, prng)
        tprod = R.ct_mul(ctx, c1, c2)
        res = R.apply_key_switch(ctx, ksm, tprod)
        expect = [c % p for c in R.poly_mul_mod_phi(ctx, m1, m2)]
        assert R.decrypt(ctx, t, res) == expect
        e2e.append({"m": m, "logQ": logQ, "p": p, "seed": seed, "primes": S(primes), "roots": S(roots), "t": S(t), "m1": m1, "m2": m2,
                    "c1": [S(c) for c in c1], "c2": [S(c) for c in c2],
                    "ksm": [[[S(d[i]) for i in range(ctx.L)] for d in ksm[r]] for r in range(2)],
                    "tprod": [[S(tp[i]) for i in range(ctx.L)] for tp in tprod],
                    "scaled": [S(x) for x in R.ct_scale_down(ctx, tprod)],
                    "result": [S(x) for x in res], "product_mod_p": expect})
    out["mul_relin"] = e2e
    # prime chains (FHEContext.cpp:83-115) with 60-bit start
    chains = []
    for m, logQ, p in [(22, 80, 23), (1 << 14, 128, 23), (1 << 15, 512, 23), (8422, 341, 8423)]:
        _, phim = R.zms_idx(m) if m <= 1 << 15 else (None, None)
        primes = R.add_primes_by_size(m, R.si_context_size(logQ, p, phim, 8 if m == 8422 else 1))
        chains.append({"m": m, "logQ": logQ, "p": p, "xi": 8 if m == 8422 else 1, "phim": phim, "primes": S(primes), "ndigits": R.ndigits(logQ)})
    out["chains"] = chains
    dump("ciphertext.json", out)


def gen_regression():
    """Ciphertext algebra of Matrix<Ciphertext> / Regression (Ciphertext.cpp:21-27,54-59,123-145,232-243,264-275;
    FHE-SI.cpp:229-260; Regression.h:166-178) with valid keys, plus the decrypt predicates that pin the semantics."""
    cases = []
    for m, logQ, p, g, seed in [(22, 80, 23, 7, 11), (32, 80, 17, 3, 12)]:
        primes, roots = chain(m, logQ, p)
        ctx = R.Ctx(m, logQ, p, primes, roots)
        prng = R.SplitMix64(seed)
        t, pk = R.keygen(ctx, prng)
        n = ctx.phim
        # (Z/32)^* is not cyclic, so the k sequence is just a few odd exponents there; for m = 22 it is Regression's own
        ks = R.automorph_generators(m, g, R.usable_slots(m, p)) if m == 22 else [3, 9, 17]
        auto = [R.key_switch_init_automorph(ctx, t, k, prng) for k in ks]
        m1 = [prng.bnd(p) for _ in range(n)]
        m2 = [prng.bnd(p) for _ in range(n)]
        c1, c2 = R.encrypt(ctx, pk, m1, prng), R.encrypt(ctx, pk, m2, prng)
        c1[0][0] = -(1 << (logQ - 1))          # the one value whose negation wraps (Reduce, Util.cpp:3-26)
        added = R.ct_add(ctx, c1, c2)
        neg = R.ct_mul_long(ctx, c1, -1)
        tripled = R.ct_mul_long(ctx, c2, 3)
        assert R.decrypt(ctx, t, tripled) == [(3 * x) % p for x in m2]
        rot = R.ct_automorph(ctx, c2, ks[0])
        sw = R.apply_key_switch_parts(ctx, auto[0], rot)
        summed = R.sum_batched_data(ctx, auto, ks, c2)

        def pt_auto(msg, k):
            big = [0] * m
            for i, c in enumerate(msg):
                big[(i * k) % m] = (big[(i * k) % m] + c) % p
            f, df = ctx.phi, len(ctx.phi) - 1
            for i in range(m - 1, df - 1, -1):
                c = big[i]
                if c:
                    for j in range(df + 1):
                        big[i - df + j] = (big[i - df + j] - c * f[j]) % p
            return [x % p for x in big[:df]]
        assert R.decrypt(ctx, t, sw) == pt_auto(m2, ks[0])
        cur = list(m2)
        for k in ks:
            cur = [(x + y) % p for x, y in zip(cur, pt_auto(cur, k))]
        assert R.decrypt(ctx, t, summed) == cur
        cases.append({"m": m, "logQ": logQ, "p": p, "g": g, "seed": seed, "primes": S(primes), "roots": S(roots), "t": S(t), "ks": ks,
                      "m2": m2, "c1": [S(c) for c in c1], "c2": [S(c) for c in c2],
                      "auto_ksm": [[[[S(d[i]) for i in range(ctx.L)] for d in a[r]] for r in range(2)] for a in auto],
                      "added": [S(x) for x in added], "negated": [S(x) for x in neg], "tripled": [S(x) for x in tripled],
                      "rotated": [S(x) for x in rot], "switched": [S(x) for x in sw], "summed": [S(x) for x in summed],
                      "summed_plain": cur})
    dump("regression.json", {"ct_algebra": cases})


if __name__ == "__main__":
    gen_transforms()
    gen_dcrt()
    gen_ciphertext()
    gen_regression()
