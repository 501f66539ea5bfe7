#!/usr/bin/env python3
"""NTT-only timing: forward / inverse row transforms for n = 2^11..2^14 (HIP events through fhesi_prof_*)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import fhe_si_amd as F
import params as P

def run(logn, L=4, rows_target=16384, iters=10):
    n = 1 << logn; m = 2 * n
    primes, roots = P.first_primes(m, L)
    ctx = F.Context(m, primes, roots)
    count = max(1, rows_target // L)
    rng = np.random.default_rng(1)
    rows = P.rand_rows(rng, primes, n, 1)
    host = np.broadcast_to(rows, (count, L, n)).copy()
    buf = ctx.upload(host)
    out = {}
    for name, fn in (("ntt_fwd", ctx.rows_ntt_fwd), ("ntt_inv", ctx.rows_ntt_inv)):
        fn(buf, count); ctx.sync()
        ctx.prof_enable(True)
        for _ in range(iters): fn(buf, count)
        l, r, ms = ctx.prof_read(name); ctx.prof_enable(False)
        gbs = r * 2 * n * 8 / (ms * 1e-3) / 1e9
        out[name] = dict(rows_per_s=r / (ms * 1e-3), GBs=gbs, frac=gbs / 8000, us_per_launch=ms / l * 1e3, rows=r / l)
    return out

if __name__ == "__main__":
    L = int(os.environ.get("NTT_L", "4"))
    logs = [int(x) for x in sys.argv[1:]] or [11, 12, 13, 14]
    for lg in logs:
        r = run(lg, L=L, rows_target=int(os.environ.get("NTT_ROWS", "16384")))
        print(lg, json.dumps({k: {a: round(b, 4) if isinstance(b, float) else b for a, b in v.items()} for k, v in r.items()}))
