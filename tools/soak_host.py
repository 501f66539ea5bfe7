#!/usr/bin/env python3
"""Soak test (development tool, run through gpurun) of the host-buffer entry fhesi_ct_mul_relin_batch: random batch sizes, random
selections of operands, pageable and pinned buffers in every combination, the options host_chunk / host_threads changed between calls --
every result compared with the device-resident batch call on the same pairs.   python3 tools/soak_host.py [--seconds 150]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150)
    args = ap.parse_args()
    import bench as B
    import torch  # noqa: F401
    import fhe_si_amd as F
    m, logQ, p = 1 << 15, 512, 23
    n = m // 2
    primes = B.prime_chain(m, logQ, p, n)
    roots = [B.root_2m(q, m) for q in primes]
    nd, nl = (logQ + 23) // 24, 8
    ctx = F.Context(m, primes, roots)
    B.LOGQ, B.P_PLAIN = logQ, p
    ksk = F.KeySwitchMatrix(ctx, 3, nd).upload(B.KeyGen(ctx, F, n, nd).s2_matrix())
    rng = np.random.default_rng(3)
    U = 48
    a, b = B.rand_coeffs(rng, (U, 2, n), nl), B.rand_coeffs(rng, (U, 2, n), nl)
    da, db, dout = ctx.upload(a), ctx.upload(b), ctx.alloc(a.nbytes)
    ctx.ct_mul_relin_dev(ksk, logQ, p, da, db, dout, nl, U)
    want = dout.download((U, 2, n, nl))
    t0, calls, mults = time.time(), 0, 0
    while time.time() - t0 < args.seconds:
        cnt = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 31, 33, 64, 70, 130]))
        idx = rng.integers(0, U, size=cnt)
        ah, bh = a[idx], b[idx]
        pin = rng.integers(0, 2, size=3)
        bufs = []
        for k, src in enumerate((ah, bh, None)):
            arr = ctx.host_array((cnt, 2, n, nl)) if pin[k] else np.empty((cnt, 2, n, nl), dtype=np.uint64)
            if src is not None:
                arr[...] = src
            else:
                arr.fill(7)
            bufs.append(arr)
        ctx.set_option("host_chunk", int(rng.choice([0, 0, 1, 3, 16, 40])))
        ctx.set_option("host_threads", int(rng.choice([0, 0, 1, 2, 5])))
        ctx.ct_mul_relin(ksk, logQ, p, bufs[0], bufs[1], out=bufs[2])
        if not np.array_equal(bufs[2], want[idx]):
            bad = [int(i) for i in range(cnt) if not np.array_equal(bufs[2][i], want[idx[i]])]
            print("MISMATCH", cnt, pin.tolist(), bad[:8])
            sys.exit(1)
        calls += 1
        mults += cnt
    print(f"soak ok: {calls} calls, {mults} multiplications from host buffers equal the device batch")


main()
