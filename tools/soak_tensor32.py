#!/usr/bin/env python3
"""Soak test (development tool, run through gpurun): the fused multiplication with the tensor half over primes below 2^30 against the
chain path (option tensor32 = 0, itself checked against the oracle by the test-suite) on many random batches and several rings.
  python tools/soak_tensor32.py [--seconds 90]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=90)
    args = ap.parse_args()
    import fhe_si_amd as F
    import params as P
    import fhesi_pyref as R
    shapes = [(32768, 512, 23, 16), (8422, 341, 8423, 16), (32768, 200, 65537, 8), (1006, 200, 23, 8), (1 << 16, 1024, 65537, 4), (32768, 512, 2, 8)]
    per = args.seconds / len(shapes)
    total = 0
    for m, logQ, p, count in shapes:
        primes, roots = P.chain_for(m, logQ, p)
        ctx = F.Context(m, primes, roots)
        n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
        rng = np.random.default_rng(m + logQ)
        ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)]))
        t0, it = time.time(), 0
        while time.time() - t0 < per:
            a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
            b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
            if it % 3 == 1:       # sparse / small operands
                a[:, :, n // 3:] = 0
                b[:, 1] = 0
            ctx.set_option("tensor32", 1)
            x = ctx.ct_mul_relin(ksk, logQ, p, a, b)
            ctx.set_option("tensor32", 0)
            y = ctx.ct_mul_relin(ksk, logQ, p, a, b)
            if not np.array_equal(x, y):
                print("MISMATCH", m, logQ, p, it)
                sys.exit(1)
            it += 1
            total += count
        print(f"m={m} logQ={logQ} p={p}: {it} batches of {count} equal")
    print("soak ok:", total, "multiplications")


if __name__ == "__main__":
    main()
