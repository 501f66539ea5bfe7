#!/usr/bin/env python3
"""Host-buffer entry (fhesi_ct_mul_relin_batch) at the metric ring: mults/s for several batch sizes, pageable and pinned buffers,
and for a sweep of the options host_threads / host_chunk.  tools/bench_host.py [threads,...] [chunks,...] [batches,...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench as B
import torch  # noqa: F401  (one HIP runtime per process)
import fhe_si_amd as F
m, logQ, p = 1 << 15, 512, 23
n = m // 2
primes = B.prime_chain(m, logQ, p, n)
roots = [B.root_2m(q, m) for q in primes]
nd, nl = (logQ + 23) // 24, 8
ctx = F.Context(m, primes, roots)
ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(B.rand_residue_rows(np.random.default_rng(8), primes, (2, 3 * nd), n))
rng = np.random.default_rng(1)
a64, b64 = B.rand_coeffs(rng, (64, 2, n), nl), B.rand_coeffs(rng, (64, 2, n), nl)
threads = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
chunks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
batches = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 8, 64, 1024]
for hb in batches:
    reps = (hb + 63) // 64
    ah = np.concatenate([a64] * reps)[:hb]; bh = np.concatenate([b64] * reps)[:hb]
    oh = np.ones_like(ah)
    pa, pb, po = ctx.host_array(ah.shape), ctx.host_array(ah.shape), ctx.host_array(ah.shape)
    pa[...] = ah; pb[...] = bh
    for T in threads:
        for hc in chunks:
            ctx2 = ctx
            ctx.set_option("host_threads", T); ctx.set_option("host_chunk", hc)
            res = []
            for (x, y, o) in ((ah, bh, oh), (pa, pb, po)):
                ctx.ct_mul_relin(ksk, logQ, p, x, y, out=o)
                best = None
                for _ in range(2 if hb >= 1024 else 6):
                    t0 = time.perf_counter(); ctx.ct_mul_relin(ksk, logQ, p, x, y, out=o); d = time.perf_counter() - t0
                    best = d if best is None or d < best else best
                res.append(hb / best)
            print(f"batch {hb:5d} threads {T:3d} chunk {hc:4d}: pageable {res[0]:9.1f}/s  pinned {res[1]:9.1f}/s  ({hb / res[0] * 1e3:.3f} / {hb / res[1] * 1e3:.3f} ms per call)", flush=True)
