#!/usr/bin/env python3
"""Where the time of Regression::Regress goes on the reference's own ring (Test_Regression: p = 8423, m = 8422, logQ = 341, d = 8):
replays the wave schedule (fhe-si_amd/regression.py) with random ciphertexts / key rows and prints the library's per-class kernel
times.  Development tool (run through gpurun); not part of the bench contract.

  python tools/reg_split.py [--m 8422] [--logq 341] [--p 8423] [--dim 8] [--steps 3] [--option NAME=VALUE ...]
"""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=8422)
    ap.add_argument("--logq", type=int, default=341)
    ap.add_argument("--p", type=int, default=8423)
    ap.add_argument("--dim", type=int, default=8)
    ap.add_argument("--rows", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--option", action="append", default=[])
    args = ap.parse_args()
    import torch
    import fhe_si_amd as F
    from fhe_si_amd import regression as G
    import params as P
    import fhesi_pyref as R
    primes, roots = P.chain_for(args.m, args.logq, args.p)
    ctx = F.Context(args.m, primes, roots)
    for o in args.option:
        k, v = o.split("=")
        ctx.set_option(k, int(v))
    n, L = ctx.phim, len(primes)
    nd, nl = R.ndigits(args.logq), (args.logq + 63) // 64
    rng = np.random.default_rng(1)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)]))
    ks = G.automorphism_exponents(args.m, 7, args.p, n)
    autos = [F.KeySwitchMatrix(ctx, 2, nd).upload(np.stack([P.rand_rows(rng, primes, n, 2 * nd) for _ in range(2)])) for _ in ks]
    d, N = args.dim, args.rows
    nin = N * (d + 1)
    X = [[i * d + j for j in range(d)] for i in range(N)]
    y = [N * d + i for i in range(N)]

    import bench
    counter = bench._CountingBackend(nin)
    G.regress_waves(counter, X, y)
    pool = G.ShardedPool(2 * n * nl, counter.used + 8, device="cuda:0", dist=None)
    be = G.DeviceBackend(ctx, args.logq, args.p, ksk, autos, ks, pool, 3)
    be.upload(P.rand_limbs(rng, (nin, 2, n), nl, args.logq))
    mark = pool.used
    stats = {}

    def step():
        pool.used = mark
        stats.update(G.regress_waves(be, X, y)[2])

    step()
    ctx.sync()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / args.steps
    prof = {k: ctx.prof_read(k) for k in F.binding.PROF_CLASSES}
    names = {k: ctx.prof_kernel_name(k) for k in F.binding.PROF_CLASSES}
    print(json.dumps({"ring": {"m": args.m, "phi": n, "L": L, "logQ": args.logq, "nd": nd}, "s_per_regress": round(dt, 4), "stats": stats,
                      "kernel_ms_per_regress": {k: round(v[2] / args.steps, 2) for k, v in prof.items() if v[0]},
                      "kernels": {k: v for k, v in names.items() if v}}))


if __name__ == "__main__":
    main()
