#!/usr/bin/env python3
"""Soak test (development tool, run through gpurun): the fused multiplication on the linear-convolution rings whose rows are padded to 2^15
(m = 2q' and m prime: fold and tail stage inside the loaders of the closing kernels) and on a ring with rows of 2^14, with key-switch matrices
generated on the device (centred limbs) -- against the same call with the general limbs (option ks_long_keys: generic recombination after a
separate tail pass) and with the tensor half over the chain (option tensor32 = 0: per-prime Bluestein rows), both checked against the oracle
by the test-suite.   python3 tools/soak_lin.py [--seconds 120]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    args = ap.parse_args()
    import bench as B
    import torch  # noqa: F401
    import fhe_si_amd as F
    import params as P
    import fhesi_pyref as R
    shapes = [(32602, 512, 32603, 3), (16381, 300, 23, 3), (32602, 120, 257, 5), (8422, 341, 8423, 6), (16381, 512, 65537, 2)]
    per = args.seconds / len(shapes)
    total = 0
    for m, logQ, p, count in shapes:
        primes, roots = P.chain_for(m, logQ, p)
        ctx = F.Context(m, primes, roots)
        n, nd, nl = ctx.phim, R.ndigits(logQ), (logQ + 63) // 64
        B.LOGQ, B.P_PLAIN = logQ, p
        ksm = B.KeyGen(ctx, F, n, nd).s2_matrix()
        rng = np.random.default_rng(m + logQ)
        t0, it = time.time(), 0
        while time.time() - t0 < per:
            a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
            b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
            if it % 3 == 1:
                a[:, :, n // 3:] = 0
                b[:, 1] = 0
            res = []
            for opts in ({}, {"ks_long_keys": 1}, {"tensor32": 0}):
                for k in ("ks_long_keys", "tensor32"):
                    ctx.set_option(k, opts.get(k, 0 if k == "ks_long_keys" else 1))
                ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
                res.append(ctx.ct_mul_relin(ksk, logQ, p, a, b))
                if not opts:
                    assert ksk.key_bits()[0], "centred limbs expected"
                del ksk
            if not (np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])):
                print("MISMATCH", m, logQ, p, it, [bool(np.array_equal(res[0], r)) for r in res])
                sys.exit(1)
            it += 1
        total += it
        print(f"m={m} logQ={logQ} p={p}: {it} batches of {count} agree (centred limbs = general limbs = chain tensor half)", flush=True)
    print("soak ok,", total, "batches")


main()
