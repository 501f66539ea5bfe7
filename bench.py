#!/usr/bin/env python3
"""bench.py -- ciphertext mult + relinearize throughput of the gfx950 DoubleCRT backend (BASELINE.json's metric).

One step = one batch of B independent ciphertext mults (Ciphertext::operator*= + KeySwitchSI::ApplyKeySwitch,
Test_AddMul.cpp:59-67) per GPU at the metric configuration: m = 2^15 (n = phi(m) = 2^14), fhe-si logQ = 512, p = 23,
decompSize = 3  =>  L = 18 chain primes (60-bit rule of FHEContext.cpp:88-115), ndigits = 22.
Inputs are synthetic (uniform coefficients in [-2^511, 2^511), uniform key-switch rows) and resident in HBM before the
timed region.  N > 1: one process per GPU, ciphertext batches sharded (each rank its own B), key-switch matrix generated
on rank 0 and broadcast with RCCL; no collective inside the timed loop (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task description) including
  roofline     -- forward-NTT kernel: algorithmic bytes (2*n*8 per row) / HIP-event time of its launches, vs 8 TB/s
  cpu_baseline -- the C oracle (oracle/fhesi_oracle.c, single thread) on a bounded sample of the same workload
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M_RING = 1 << 15
LOGQ = 512
P_PLAIN = 23
DECOMP = 3
SP_NBITS = 60
HBM_PEAK_GBS = 8000.0


# ---- host-side setup (pure integer helpers; the chain rule restates FHEContext.cpp:83-115) -----------------------
def _is_prime(n):
    if n < 2:
        return False
    sp = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in sp:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in sp:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def prime_chain(m, logQ, p, phim, xi=1, sp_nbits=SP_NBITS):
    total = logQ * math.log(2.0) * 2 + math.log(p) + math.log(phim) * 2 + math.log(2) + math.log(xi)
    chain, q, two_m, last, left = [], (1 << sp_nbits) - 1, 2 * m, False, total
    q -= q % two_m
    q += two_m + 1
    while left > 0.0:
        if left < math.log(float(q)) and not last:
            last = True
            q = int(math.ceil(math.exp(left)))
            q -= (q % two_m) - 1
            two_m = -two_m
        while True:
            q -= two_m
            if _is_prime(q):
                break
        if q not in chain:
            chain.append(q)
            left -= math.log(float(q))
    return chain


def root_2m(q, m):
    e = 2 * m
    facts, t, f = [], e, 2
    while f * f <= t:
        if t % f == 0:
            facts.append(f)
            while t % f == 0:
                t //= f
        f += 1
    if t > 1:
        facts.append(t)
    for s in range(2, 1000):
        r = pow(s, (q - 1) // e, q)
        if pow(r, e, q) == 1 and all(pow(r, e // g, q) != 1 for g in facts):
            return r
    raise RuntimeError("no root")


def rand_residue_rows(rng, primes, shape_prefix, n):
    out = np.empty(tuple(shape_prefix) + (len(primes), n), dtype=np.uint64)
    for i, q in enumerate(primes):
        out[..., i, :] = rng.integers(0, q, size=tuple(shape_prefix) + (n,), dtype=np.uint64)
    return out


def rand_coeffs(rng, shape, nlimbs):
    """uniform in [-2^(64 nlimbs - 1), 2^(64 nlimbs - 1)): any limb pattern is a valid two's complement value."""
    hi = rng.integers(0, 1 << 63, size=tuple(shape) + (nlimbs,), dtype=np.uint64)
    lo = rng.integers(0, 2, size=tuple(shape) + (nlimbs,), dtype=np.uint64)
    return hi * np.uint64(2) + lo


def cpu_baseline(primes, roots, ksm, a, b, n_sample):
    """Oracle (single-thread C restatement) timed on the host cores: the reported CPU figure, never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    orc = O.Oracle(M_RING, primes, roots)
    t0 = time.perf_counter()
    done = 0
    for i in range(n_sample):
        orc.ct_mul_relin(ksm, a[i], b[i], LOGQ, P_PLAIN, DECOMP)
        done += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = time.perf_counter() - t0
    # the same oracle on all host cores: independent ciphertexts, one per thread (ctypes releases the GIL; the oracle keeps no
    # shared mutable state).  The reference itself is single-threaded; this is the generous CPU figure.
    from concurrent.futures import ThreadPoolExecutor
    cores = max(1, min(os.cpu_count() or 1, 32))          # (each oracle call holds ~0.4 GB of digit rows)
    t1 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(lambda i: orc.ct_mul_relin(ksm, a[i % n_sample], b[i % n_sample], LOGQ, P_PLAIN, DECOMP), range(cores)))
    dt_all = time.perf_counter() - t1
    return {"value": done / dt, "unit": "ciphertext-mults/s", "cores": 1, "kind": "port",
            "sample": f"{done} ciphertext mult+relin at the bench config (n=2^14, L={len(primes)}, ndigits={ksm.shape[1] // 3}) "
                      f"with the C oracle's direct negacyclic NTT (optimistic vs the reference's Bluestein over NTL), {dt:.1f} s",
            "all_cores": {"value": cores / dt_all, "unit": "ciphertext-mults/s", "cores": cores,
                          "sample": f"{cores} mults, one per thread, {dt_all:.1f} s"}}


class _CountingBackend:
    """dry run of the wave schedule: how many pool entries one Regress needs"""
    auto_ks = ()

    def __init__(self, used):
        self.used = used

    def run_wave(self, w):
        first = self.used
        self.used += w.groups
        return first

    def sum_batched(self, first, count):
        pass

    def negated(self, idx):
        first = self.used
        self.used += len(idx)
        return first


def run_regression(args, ctx, ksk, primes, n, nd, nl, rank, world, local_rank, dist, torch, F, chain_bits):
    """configs[3] (Test_Regression d=8) replayed at the metric ring: one step = one Regression::Regress
    (Regression.h:102-134) evaluated in waves (fhe-si_amd/regression.py); the groups of every wave are sharded over the ranks
    and the wave's outputs exchanged (RCCL broadcast per producing rank), so total work is fixed: strong scaling."""
    from fhe_si_amd import regression as G, shard
    d, N, L = args.reg_dim, args.reg_rows, len(primes)
    dev = f"cuda:{local_rank}"
    ks = G.automorphism_exponents(M_RING, 7, P_PLAIN, n)
    autos = []
    for i in range(len(ks)):             # KeySwitchSI(secretKey, k) matrices (2 source components), broadcast like the main one
        a = F.KeySwitchMatrix(ctx, 2, nd)
        host = rand_residue_rows(np.random.default_rng(100 + i), primes, (2, 2 * nd), n) if rank == 0 else None
        if world > 1:
            stage = shard.broadcast_key_matrix(host, a.nbytes, dist, device=dev)
            torch.cuda.synchronize()
            ctx.dev_copy(a.device_ptr, stage.data_ptr(), a.nbytes)
            del stage
        else:
            a.upload(host)
        autos.append(a)
    nin = N * (d + 1)
    X = [[i * d + j for j in range(d)] for i in range(N)]
    y = [N * d + i for i in range(N)]
    counter = _CountingBackend(nin)
    G.regress_waves(counter, X, y)
    pool = G.ShardedPool(2 * n * nl, counter.used + 8, device=dev, dist=dist)
    be = G.DeviceBackend(ctx, LOGQ, P_PLAIN, ksk, autos, ks, pool, DECOMP)
    first = be.upload(rand_coeffs(np.random.default_rng(7), (nin, 2, n), nl))      # same inputs on every rank
    assert first == 0
    mark = pool.used
    stats = {}

    def step():
        pool.used = mark
        stats.update(G.regress_waves(be, X, y)[2])

    for _ in range(max(1, args.warmup)):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = {k: ctx.prof_read(k) for k in F.binding.PROF_CLASSES}
    ctx.prof_enable(False)
    launches, rows, ms = prof["ntt_fwd_digits_main"]
    achieved = rows * 2 * n * 8 / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    if rank == 0:
        ksw = stats["key_switches"] + stats["automorph_key_switches"]
        line = {
            "metric": "key-switched ciphertext products/sec inside Regression::Regress (wave-scheduled) at n=2^14, logQ=512",
            "value": round(ksw * args.steps / dt, 2), "unit": "key-switches/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"configs[3] replay: Regression::Regress d={d}, {N} data block(s), m=2^15 n=2^14, fhe-si logQ=512, p=23, decompSize=3",
                       "L": L, "chain_bits": round(chain_bits, 1), "ndigits": nd, "automorphism_keys": len(ks), "waves": stats["waves"],
                       "products_per_regress": stats["products"], "key_switches_per_regress": stats["key_switches"],
                       "automorph_key_switches_per_regress": stats["automorph_key_switches"], "regress_per_s": round(args.steps / dt, 3),
                       "sharding": "groups of every wave sharded over ranks, outputs exchanged by RCCL broadcast" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "ntt_fwd_tile<14, true, 0, false>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "launches": launches,
                         "avg_launch_ms": round(ms / launches, 4) if launches else None},
            "cpu_baseline": None,
            "kernel_ms_per_step": {k: round(v[2] / args.steps, 3) for k, v in prof.items() if v[0] and k != "ntt_fwd_digits_main"},
        }
        print(json.dumps(line), flush=True)


def run_ntt_round_trips(args, rank, world, local_rank, dist, torch, F):
    """configs[1]: m = 2^14 (n = 8192), the first 8 primes = 1 mod 2^15 descending from 2^60, B DoubleCRTs of
    uniform residues per GPU; one step = forward + inverse transform of the whole batch (2 B L row transforms); the round trip must
    reproduce the input bit for bit, and one DoubleCRT is checked against the C oracle in both directions."""
    m, n, L = 1 << 14, 1 << 13, 8
    B = args.batch if args.batch else 1024
    primes, q = [], (1 << 60) - 1
    q -= q % (2 * m)
    q += 2 * m + 1
    while len(primes) < L:
        q -= 2 * m
        if _is_prime(q):
            primes.append(q)
    roots = [root_2m(q, m) for q in primes]
    ctx = F.Context(m, primes, roots, device=local_rank)
    host = rand_residue_rows(np.random.default_rng(42 + rank), primes, (B,), n)
    buf = ctx.upload(host)

    def step():
        ctx.rows_ntt_fwd(buf, B)
        ctx.rows_ntt_inv(buf, B)

    for _ in range(max(1, args.warmup)):
        step()
    ctx.sync()
    if dist:
        dist.barrier()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fl, frows, fms = ctx.prof_read("ntt_fwd")
    il, irows, ims = ctx.prof_read("ntt_inv")
    ctx.prof_enable(False)
    identity = bool(np.array_equal(buf.download(host.shape), host))
    cpu = None
    oracle_ok = None
    if rank == 0 and args.cpu_sample > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        orc = O.Oracle(m, primes, roots)
        one = ctx.upload(host[:1])
        ctx.rows_ntt_fwd(one, 1)
        ev = one.download((1, L, n))
        t1 = time.perf_counter()
        exp = np.stack([orc.fft_residues(i, host[0, i]) for i in range(L)])
        back = np.stack([orc.cmod_ifft(i, exp[i]) for i in range(L)])
        cdt = time.perf_counter() - t1
        oracle_ok = bool(np.array_equal(ev[0], exp) and np.array_equal(back, host[0]))
        cpu = {"value": 1.0 / cdt, "unit": "DoubleCRT round trips/s", "cores": 1, "kind": "port",
               "sample": f"1 DoubleCRT (8 rows of n=2^13) forward + inverse with the C oracle's direct negacyclic NTT, {cdt * 1e3:.1f} ms"}
    if rank == 0:
        row_bytes = 2 * n * 8
        ach = frows * row_bytes / (fms * 1e-3) / 1e9 if fms > 0 else 0.0
        line = {
            "metric": "DoubleCRT forward+inverse NTT round trips/sec at n=2^13, 8 primes", "value": round(B * args.steps * world / dt, 1),
            "unit": "DoubleCRT round trips/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[1]: DoubleCRT NTT round trip, m=2^14 n=2^13, 8 primes of 60 bits", "L": L, "batch_per_gpu": B,
                       "round_trip_is_identity": identity, "matches_oracle": oracle_ok},
            "roofline": {"bound": "hbm", "kernel": "ntt_fwd_tile<13, false, 0, false>", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches": fl, "avg_launch_ms": round(fms / fl, 4) if fl else None,
                         "row_ntts_per_s": round(frows / (fms * 1e-3), 1) if fms > 0 else None,
                         "inverse": {"kernel": "ntt_inv_tile<13, 0, false>", "achieved": round(irows * row_bytes / (ims * 1e-3) / 1e9, 1) if ims > 0 else None,
                                     "row_ntts_per_s": round(irows / (ims * 1e-3), 1) if ims > 0 else None}},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
        if not identity or oracle_ok is False:
            raise SystemExit("NTT round trip / oracle parity failed")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="ciphertext mults per GPU per step (default 64 = one chunk of the library; ntt workload: DoubleCRTs per step, default 1024)")
    ap.add_argument("--cpu-sample", type=int, default=3, help="oracle ciphertext mults timed for cpu_baseline (0 = skip)")
    ap.add_argument("--lanes", type=int, default=1, help="concurrent half-batches inside the library (FHESI_LANES); 2 gives ~+5 %% throughput but "
                    "overlapping kernels, so per-kernel durations (and the roofline line) are no longer those of a kernel running alone")
    ap.add_argument("--workload", default="metric", choices=["metric", "stress", "regression", "ntt"], help="metric = configs[2] (default, the contract line); "
                    "ntt = configs[1]: DoubleCRT forward+inverse round trips at n=2^13, 8 primes, --batch DoubleCRTs per GPU (default there: 1024); "
                    "stress = configs[4]: m=2^16 (n=2^15), logQ=1024, p=65537 (35 primes, 43 digits) -- reporting only; "
                    "regression = configs[3] replayed at the metric ring: Regression::Regress (d = --reg-dim, --reg-rows data blocks) in waves, "
                    "every wave's groups sharded over the ranks (strong scaling)")
    ap.add_argument("--reg-dim", type=int, default=8)
    ap.add_argument("--reg-rows", type=int, default=1)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the N > 1 plumbing)")
    ap.add_argument("--one-device", action="store_true", help="plumbing check on a 1-GPU box: every rank uses GPU 0 (never for measurements)")
    ap.add_argument("--ntt-rows", type=int, default=0, help="extra: rows for a standalone forward-NTT timing (0 = use pipeline launches)")
    args = ap.parse_args()

    global M_RING, LOGQ, P_PLAIN
    if args.workload == "stress":
        M_RING, LOGQ, P_PLAIN = 1 << 16, 1024, 65537
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    os.environ["FHESI_LANES"] = str(args.lanes)
    import torch
    import fhe_si_amd as F
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (HIP path only; there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    if args.workload == "ntt":
        run_ntt_round_trips(args, rank, world, local_rank, dist, torch, F)
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    n = M_RING // 2
    primes = prime_chain(M_RING, LOGQ, P_PLAIN, n)
    roots = [root_2m(q, M_RING) for q in primes]
    L = len(primes)
    nd = (LOGQ + 8 * DECOMP - 1) // (8 * DECOMP)
    nl = (LOGQ + 63) // 64
    ncol = 3 * nd
    chain_bits = sum(math.log2(q) for q in primes)
    B = args.batch if args.batch else (16 if args.workload == "stress" else 64)      # one chunk of the library (about 75k digit rows)

    ctx = F.Context(M_RING, primes, roots, device=local_rank)
    ksk = F.KeySwitchMatrix(ctx, 3, nd)

    # key-switch matrix: generated on rank 0, RCCL-broadcast over xGMI into every rank's HBM copy
    ksm_host = None
    if rank == 0:
        ksm_host = rand_residue_rows(np.random.default_rng(8), primes, (2, ncol), n)
    if world > 1:
        from fhe_si_amd import shard
        stage = shard.broadcast_key_matrix(ksm_host, ksk.nbytes, dist, device=f"cuda:{local_rank}")   # one RCCL broadcast
        torch.cuda.synchronize()
        ctx.dev_copy(ksk.device_ptr, stage.data_ptr(), ksk.nbytes)
        del stage
    else:
        ksk.upload(ksm_host)

    if args.workload == "regression":
        run_regression(args, ctx, ksk, primes, n, nd, nl, rank, world, local_rank, dist, torch, F, chain_bits)
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    rng = np.random.default_rng(7 + rank)
    a_host = rand_coeffs(rng, (B, 2, n), nl)
    b_host = rand_coeffs(rng, (B, 2, n), nl)
    da, db = ctx.upload(a_host), ctx.upload(b_host)
    dout = ctx.alloc(a_host.nbytes)

    def step():
        ctx.ct_mul_relin_dev(ksk, LOGQ, P_PLAIN, da, db, dout, nl, B, DECOMP)

    for _ in range(args.warmup):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # live per-kernel timing of the timed region (HIP events on the context's stream)
    prof = {k: ctx.prof_read(k) for k in F.binding.PROF_CLASSES}
    ctx.prof_enable(False)
    launches, rows, ms = prof["ntt_fwd_digits_main"]       # fused ByteDecomp + forward NTT of the digit polynomials
    # n = 2^14 at the metric chain shape: the digit rows are transformed modulo four 30-bit primes (kernels_aux32.hip): 4-byte residues
    aux32 = (n == (1 << 14) and L == 18 and LOGQ == 512 and not args.ntt_rows
             and not any(v in os.environ for v in ("FHESI_KS_DIRECT", "FHESI_KS_RESIDUES", "FHESI_KS_AUX60")))
    row_bytes = 2 * n * (4 if aux32 else 8)                 # row read once + written once
    if args.ntt_rows:
        # optional standalone measurement on a fixed row count
        cnt = max(1, args.ntt_rows // L)
        buf = ctx.upload(rand_residue_rows(np.random.default_rng(1), primes, (cnt,), n))
        ctx.rows_ntt_fwd(buf, cnt)
        ctx.prof_enable(True)
        for _ in range(10):
            ctx.rows_ntt_fwd(buf, cnt)
        launches, rows, ms = ctx.prof_read("ntt_fwd")
        ctx.prof_enable(False)
    achieved = rows * row_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_ntt_fwd.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            traffic = pmc.get("hbm_bytes_per_launch")
            if not launches or pmc.get("rows_per_launch") != round(rows / launches):
                traffic = None      # the counters were collected on another launch shape (batch)
        except Exception:
            traffic = None
    if args.workload != "metric" or args.ntt_rows:
        traffic = None          # the PMC passes in profiles/ were taken on the metric workload's launch shape
    # template parameters <LOGN, DIGITS, S0, CONTIG>; rows of 2^15 points: the head-fused sub-transform kernel (S0 = 1, CONTIG)
    if aux32:
        kname = "ntt32_fwd_kernel<true>"
    elif ctx.phim > (1 << 14):
        kname = "ntt_fwd_tile<14, false, 1, false>" if args.ntt_rows else "ntt_fwd_tile<14, true, 1, true>"
    else:
        kname = "ntt_fwd_tile<14, %s, 0, false>" % ("false" if args.ntt_rows else "true")
    roofline_ntt = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "launches": launches, "avg_launch_ms": round(ms / launches, 4) if launches else None,
                    "rows_per_launch": round(rows / launches, 1) if launches else None, "row_bytes": row_bytes // 2,
                    "row_ntts_per_s": round(rows / (ms * 1e-3), 1) if ms > 0 else None}
    # The dominant kernel of the pipeline is the key-switch dot product through the two auxiliary primes (kernels_ksaux.hip):
    # algorithmic bytes per launch of c ciphertexts = (digit rows c*ncol*2 + key rows 2*R*2*ncol + output rows c*2*R*2) * n * 8
    # (DESIGN.md section 6); it is bound by the VALU (64-bit multiply-adds) and the L2, not by HBM -- the contract's roof is HBM.
    aux = "FHESI_KS_DIRECT" not in os.environ      # the library's A/B switch back to the per-prime dot product
    dl, dunits, dms = prof["dot"]
    # output rows per (ciphertext, key row, auxiliary prime): the limbs of the key's integer coefficients where the library runs the
    # key switch in limb mode (ks_limb_plan: 15 at the metric chain shape, 30 at the stress shape), the L residues otherwise
    R = L if "FHESI_KS_RESIDUES" in os.environ else {(18, 512): 15, (35, 1024): 30}.get((L, LOGQ), L)
    if aux:
        dbytes = (dunits * (ncol * 2 + 2 * R * 2) + dl * (2 * R * 2 * ncol)) * n * 8
        dname = "dot32_kernel<8, 16>" if aux32 else ("dot_aux_kernel<4, 16, 1>" if ncol * 4 * 512 <= 150 * 1024 else "dot_aux_kernel<2, 16, 2>")
    else:
        dbytes = (dunits * (ncol + 2) * L + dl * (2 * ncol * L)) * n * 8
        dname = "dot_accum_kernel<2, %s>" % ("true" if ctx.phim > (1 << 14) else "false")
    dach = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
    dtraffic = None
    dpmc_path = os.path.join(ROOT, "profiles", "pmc_dot_aux.json")
    if aux and args.workload == "metric" and os.path.exists(dpmc_path):
        try:
            dp = json.load(open(dpmc_path))
            if dl and dp.get("ciphertexts_per_launch") == round(dunits / dl):
                dtraffic = dp.get("hbm_bytes_per_launch")
        except Exception:
            dtraffic = None
    roofline_dot = {"bound": "hbm", "kernel": dname, "achieved": round(dach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(dach / HBM_PEAK_GBS, 4), "traffic": dtraffic, "launches": dl,
                    "avg_launch_ms": round(dms / dl, 4) if dl else None, "ciphertexts_per_launch": round(dunits / dl, 1) if dl else None,
                    "note": "integer multiply-accumulate bound by the LDS pipe and the VALU (the key slices are re-read from L2), not by HBM; see roofline_ntt for the "
                            "DoubleCRT transform the metric's GB/s figure refers to"}
    # `roofline` is the kernel with the largest share of the step
    roofline = roofline_dot if dms >= ms else roofline_ntt

    if rank == 0:
        total_mults = B * args.steps * world
        value = total_mults / dt
        breakdown = {k: round(v[2] / args.steps, 3) for k, v in prof.items() if v[0] and k != "ntt_fwd_digits_main"}
        cpu = None
        if args.cpu_sample > 0 and world == 1:       # CPU baseline on rank 0 at N=1 only
            cpu = cpu_baseline(primes, roots, ksm_host, a_host, b_host, min(args.cpu_sample, B))
        line = {
            "metric": "homomorphic ciphertext-mults/sec (incl. relinearize) at n=2^14, logQ=512" if args.workload == "metric"
                      else "homomorphic ciphertext-mults/sec (incl. relinearize) at n=2^15, logQ=1024 (stress shape)",
            "value": round(value, 2), "unit": "ciphertext-mults/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[2]: full ciphertext mul + relinearize + scale-down, m=2^15 n=2^14, fhe-si logQ=512, p=23, decompSize=3"
                       if args.workload == "metric" else "configs[4] stress shape: m=2^16 n=2^15, fhe-si logQ=1024, p=65537, decompSize=3",
                       "L": L, "chain_bits": round(chain_bits, 1), "ndigits": nd, "batch_per_gpu": B, "lanes": args.lanes,
                       "fwd_row_ntts_per_mult": (4 * L + (4 if aux32 else 2) * ncol) if aux else (4 + ncol) * L,
                       "inv_row_ntts_per_mult": (3 * L + (8 if aux32 else 4) * R) if aux else 5 * L,
                       "key_switch": ("4 x 30-bit auxiliary primes, 15 limbs" if aux32 else "2 x 60-bit auxiliary primes") if aux else "per chain prime",
                       "sharding": "independent ciphertexts per GPU, key-switch matrix RCCL-broadcast" if world > 1 else "single GPU"},
            "roofline": roofline, "roofline_ntt": roofline_ntt, "roofline_dot": roofline_dot, "cpu_baseline": cpu, "kernel_ms_per_step": breakdown,
        }
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
