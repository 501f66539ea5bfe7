#!/usr/bin/env python3
"""Experiment: two contexts (two HIP streams) on one GPU, each running half of the batch concurrently."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import bench as B
import fhe_si_amd as F
n = B.M_RING // 2
primes = B.prime_chain(B.M_RING, B.LOGQ, B.P_PLAIN, n); roots = [B.root_2m(q, B.M_RING) for q in primes]
nd, nl = 22, 8
ksm = B.rand_residue_rows(np.random.default_rng(8), primes, (2, 3 * nd), n)
def make(batch):
    ctx = F.Context(B.M_RING, primes, roots)
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(ksm)
    rng = np.random.default_rng(7)
    a = B.rand_coeffs(rng, (batch, 2, n), nl); b = B.rand_coeffs(rng, (batch, 2, n), nl)
    da, db = ctx.upload(a), ctx.upload(b); do = ctx.alloc(a.nbytes)
    return ctx, ksk, da, db, do
def run(objs, batch, steps):
    ctx, ksk, da, db, do = objs
    for _ in range(steps): ctx.ct_mul_relin_dev(ksk, B.LOGQ, B.P_PLAIN, da, db, do, nl, batch, B.DECOMP)
    ctx.sync()
for nstreams, batch in ((1, 64), (2, 32), (2, 64), (3, 32), (3, 21), (4, 32)):
    objs = [make(batch) for _ in range(nstreams)]
    for o in objs: run(o, batch, 2)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(o, batch, 6)) for o in objs]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(nstreams, "streams x batch", batch, "->", round(nstreams * batch * 6 / dt, 1), "mults/s", flush=True)
    del objs
