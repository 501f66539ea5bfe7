#!/usr/bin/env python3
"""Latency of small device-resident calls of fhesi_ct_mul_relin_batch_dev at the metric ring: per call with a synchronisation after each
(what a caller who reads every result pays) and back to back (what the device itself needs).  tools/bench_small.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench as B
import torch  # noqa: F401
import fhe_si_amd as F
m, logQ, p = 1 << 15, 512, 23
n = m // 2
primes = B.prime_chain(m, logQ, p, n)
roots = [B.root_2m(q, m) for q in primes]
nd, nl = (logQ + 23) // 24, 8
ctx = F.Context(m, primes, roots)
for kv in sys.argv[1:]:
    k, _, v = kv.partition("=")
    if v:
        ctx.set_option(k, int(v))
B.LOGQ, B.P_PLAIN = logQ, p
ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(B.KeyGen(ctx, F, n, nd).s2_matrix() if "uniform" not in sys.argv else B.rand_residue_rows(np.random.default_rng(8), primes, (2, 3 * nd), n))
rng = np.random.default_rng(1)
a, b = B.rand_coeffs(rng, (16, 2, n), nl), B.rand_coeffs(rng, (16, 2, n), nl)
da, db, dout = ctx.upload(a), ctx.upload(b), ctx.alloc(a.nbytes)
for cnt in [int(x) for x in os.environ.get("BENCH_SMALL_COUNTS", "1,2,4,8,16").split(",")]:
    for _ in range(5):
        ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, cnt)
    ctx.sync()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N):
        ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, cnt)
        ctx.sync()
    t_sync = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for _ in range(N):
        ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, cnt)
    t_issue = (time.perf_counter() - t0) / N
    ctx.sync()
    t_async = (time.perf_counter() - t0) / N
    print(f"count {cnt:3d}: {t_sync * 1e3:.3f} ms per call with a sync after each; back to back {t_async * 1e3:.3f} ms per call (host issue time {t_issue * 1e3:.3f} ms) -> {cnt / t_async:.0f} mults/s", flush=True)
