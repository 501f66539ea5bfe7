#!/usr/bin/env python3
"""More of tests/test_gpu_shape_fuzz.py::test_mul_relin_on_random_shapes_with_generated_keys: random rings (m = 2q', odd prime m), moduli,
plaintext moduli, chains of 50- and 60-bit primes, key-switch matrices with coefficients bounded like generated ones (|K| <= 2^(logQ-1)) or a
few bits larger or smaller, against the C oracle.  tools/fuzz_centred.py [cases = 300] [first seed = 100]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

RINGS = [22, 46, 94, 118, 166, 214, 262, 101, 107, 227, 251, 1006, 509]
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
stats = {}
for seed in range(first, first + cases):
    rng = np.random.default_rng(seed)
    m = int(rng.choice(RINGS))
    logQ = int(rng.integers(64, 500))
    p = int(rng.choice([2, 23, 257, 8423, 65537, int(rng.integers(2, 1 << 20))]))
    count = int(rng.integers(1, 4))
    kbits = logQ + int(rng.choice([0, 0, 0, -30, -1, 1, 7, 40]))          # size of the key coefficients (logQ = what KeySwitchSI::Init produces)
    primes, roots = P.chain_for(m, logQ, p, 1, 50 if seed % 5 == 0 else 60)
    ctx = F.Context(m, primes, roots)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl, W = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64, len(primes) + 2
    ksm = np.empty((2, 3 * nd, len(primes), n), dtype=np.uint64)
    for r in range(2):
        for c in range(3 * nd):
            ksm[r, c] = orc.dcrt_from_poly(P.rand_limbs(rng, (n,), W, kbits))
    h = 1 << (kbits - 1)
    ksm[int(rng.integers(0, 2)), int(rng.integers(0, 3 * nd))] = orc.dcrt_from_poly(O.ints_to_limbs([int(rng.choice([-h, h - 1, h, 0, 1, -1])) for _ in range(n)], W))
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    got = ctx.ct_mul_relin(ksk, logQ, p, a, b)
    form, rows, bits = ksk.form()
    centred, nb = ksk.key_bits()
    key = (form, centred)
    stats[key] = stats.get(key, 0) + 1
    for c in range(count):
        if not np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)):
            bad += 1
            print("MISMATCH", seed, m, logQ, p, count, c, kbits, form, rows, bits, centred, nb, flush=True)
    del ksk, ctx
print(f"{cases} cases, {bad} mismatches; (form, centred) counts: {stats}")
sys.exit(1 if bad else 0)
