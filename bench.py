#!/usr/bin/env python3
"""bench.py -- ciphertext mult + relinearize throughput of the gfx950 DoubleCRT backend (BASELINE.json's metric).

One step = one batch of B independent ciphertext mults (Ciphertext::operator*= + KeySwitchSI::ApplyKeySwitch,
Test_AddMul.cpp:59-67) per GPU at the metric configuration: m = 2^15 (n = phi(m) = 2^14), fhe-si logQ = 512, p = 23,
decompSize = 3  =>  L = 18 chain primes (60-bit rule of FHEContext.cpp:88-115), ndigits = 22.
Inputs are synthetic (uniform coefficients in [-2^511, 2^511), uniform key-switch rows) and resident in HBM before the
timed region.  N > 1: one process per GPU, ciphertext batches sharded (each rank its own B), key-switch matrix generated
on rank 0 and broadcast with RCCL; no collective inside the timed loop (weak scaling).

N > 1 (first contact with a real multi-GPU node is the driver's run, so the path is defensive): the process group is gloo for what lives
in host memory + RCCL for what lives in HBM; before any device collective a ROLL CALL gathers every rank's GPU identity (UUID / PCI address)
and ends the run on every rank, with a message, if two ranks drive the same GPU (unless --one-device); after the timed region every rank
checks its own buffer against the oracle evaluated on its own copy of the broadcast key matrix, and all ranks recompute rank 0's first
pair and compare digests (config.multi_gpu: RCCL version, communicator size, devices, broadcast staging vs collective time, parity).

`python bench.py --gpus N` with N > 1 and no RANK in the environment launches the N ranks itself (torch.distributed.run as a child
process, started before anything in this process touches the GPU) and relays rank 0's line; under an external launcher (RANK set) it
is one of the ranks.

Test hooks (environment; tests/test_gpu_multirank.py): FHESI_BENCH_GROUP_AT_N1=1 -- ONE rank runs the whole N > 1 path (process group,
roll call, RCCL key broadcast, per-rank parity, digests) so that a 1-GPU box exercises it over real RCCL; FHESI_BENCH_NO_MIXED_GROUP=1 --
take the nccl-only fallback (roll call through the rendezvous store) as a build without gloo would.

Prints ONE JSON line on rank 0 (contract in the task description) including
  roofline       -- the kernel with the largest share of the step: algorithmic bytes (for the fused digit transform the rows
                    written + the source read once) / HIP-event time of its launches, vs 8 TB/s, with `bound` naming what binds the kernel
                    ("valu" when the committed SQ pass shows the VALU >= 90 % busy), valu_busy, valu_instr_per_butterfly and
                    issue_ceiling_frac beside the HBM fraction, frac_on_traffic (committed PMC bytes) and frac_nominal (2 x row bytes);
                    kernel names are read back from the library (fhesi_prof_kernel_name), not written here
  roofline_ntt   -- the same for the transform of the digit polynomials (fused ByteDecomp + forward NTT)
  cpu_baseline   -- the C oracle (oracle/fhesi_oracle.c) on a bounded sample of the same workload: one thread (at any N, on rank 0), and at
                    N = 1 also all cores and the reference's own Bluestein-over-FFT-primes structure (like for like)
  matches_oracle -- the first cpu-sample outputs of the timed buffer compared bit for bit with the oracle's (exit code 1 otherwise)
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


def host_cpu_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"model": model, "logical_cpus": os.cpu_count(), "usable_cpus": usable}


def cpu_baseline(primes, roots, ksm, a, b, n_sample, bluestein_sample=1, all_cores=True):
    """Oracle (C restatement) timed on the host cores: the reported CPU figure, never the product path.  Returns (record, outputs).
    all_cores = False (rank 0 of an N > 1 job, the other ranks waiting): the single-thread sample alone."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    orc = O.Oracle(M_RING, primes, roots)
    pow2 = M_RING & (M_RING - 1) == 0
    if not pow2:
        orc.set_bluestein_fft(True)            # (rows are Bluestein transforms, as in the reference: bluestein.cpp:93-144; the oracle's direct form of them is quadratic)
        bluestein_sample = 0
    info = host_cpu_info()
    t0 = time.perf_counter()
    outs = []
    for i in range(n_sample):
        outs.append(orc.ct_mul_relin(ksm, a[i], b[i], LOGQ, P_PLAIN, DECOMP))
        if time.perf_counter() - t0 > 30.0:
            break
    done = len(outs)
    dt = time.perf_counter() - t0
    # the same oracle on all host cores: independent ciphertexts, one per thread (ctypes releases the GIL; the oracle keeps no
    # shared mutable state).  The reference itself is single-threaded; this is the generous CPU figure.
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(info["usable_cpus"], 64))        # (each oracle call holds ~0.4 GB of digit rows)
    dt_all = None
    if all_cores and pow2:          # (Bluestein-mode multiplications on the reference's rings take minutes each: the single-thread sample alone)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(lambda i: orc.ct_mul_relin(ksm, a[i % n_sample], b[i % n_sample], LOGQ, P_PLAIN, DECOMP), range(threads)))
        dt_all = time.perf_counter() - t1
    rec = {"value": done / dt, "unit": "ciphertext-mults/s", "cores": 1, "kind": "port",
           "sample": (f"{done} ciphertext mult+relin at the bench config (n=2^{M_RING.bit_length() - 2}, L={len(primes)}, ndigits={ksm.shape[1] // 3}) "
                      f"with the C oracle's direct negacyclic NTT (optimistic vs the reference's Bluestein over NTL), {dt:.1f} s") if pow2 else
                     (f"{done} ciphertext mult+relin at the bench config (m={M_RING}, L={len(primes)}, ndigits={ksm.shape[1] // 3}), every row transform as Bluestein + "
                      f"3-prime FFT convolution (the reference's algorithm, bluestein.cpp:93-144), {dt:.1f} s"),
           "host": info,
           "all_cores": {"value": threads / dt_all, "unit": "ciphertext-mults/s", "cores": threads,
                         "sample": f"{threads} mults, one per thread on {info['usable_cpus']} usable logical CPUs, {dt_all:.1f} s"} if dt_all else None}
    if bluestein_sample > 0:
        # like for like: the reference evaluates every transform (power-of-two m too) as Bluestein with an N = 2^16-point cyclic
        # product through NTL's FFT primes (bluestein.cpp:116-139); the oracle's bluestein_fft mode has that structure
        orc.set_bluestein_fft(True)
        t2 = time.perf_counter()
        o = orc.ct_mul_relin(ksm, a[0], b[0], LOGQ, P_PLAIN, DECOMP)
        dt_b = time.perf_counter() - t2
        orc.set_bluestein_fft(False)
        rec["bluestein_mode"] = {"value": 1.0 / dt_b, "unit": "ciphertext-mults/s", "cores": 1, "kind": "port",
                                 "same_result": bool(np.array_equal(o, outs[0])),
                                 "sample": f"1 mult+relin with every row transform as Bluestein + 3-prime FFT convolution of 2^{(2 * M_RING - 1).bit_length()} points "
                                           f"(the reference's algorithm, bluestein.cpp:93-144), {dt_b:.1f} s"}
    return rec, np.stack(outs)


def oracle_outputs(primes, roots, ksm, a, b, n_sample):
    """the checker alone (no timing): the C oracle's results for the first n_sample pairs"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    orc = O.Oracle(M_RING, primes, roots)
    if M_RING & (M_RING - 1):
        orc.set_bluestein_fft(True)
    return np.stack([orc.ct_mul_relin(ksm, a[i], b[i], LOGQ, P_PLAIN, DECOMP) for i in range(n_sample)])


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


class KeyGen:
    """The key-switch matrices the reference's drivers hold (Test_AddMul.cpp:48-52, Regression.h:68-80), generated on the device:
    a sampleHWt(64) secret key t (FHE-SI.cpp:88-91), KeySwitchSI(secretKey) = Init((1, t, t^2) -> t) (FHE-SI.cpp:211-226) and
    KeySwitchSI(secretKey, k) = Init((1, t(X^k)) -> t) (:228-239), randomness from the seeded generator (csrc/philox.h)."""

    def __init__(self, ctx, F, n, nd):
        self.ctx, self.F, self.nd = ctx, F, nd
        self.seed, self.pub_seed, self.next = 0x5EC2E7C0FFEE1234, 0x5EC2E7C0FFEE1234 ^ 0x9E3779B97F4A7C15, 1000
        one = np.zeros((n, 1), dtype=np.uint64)
        one[0, 0] = 1
        self.one = F.DoubleCRT.from_poly(ctx, one)
        self.t = F.DoubleCRT(ctx).sample(0, 64, self.seed, 1)

    def _matrix(self, src):
        m = self.F.KeySwitchMatrix(self.ctx, len(src), self.nd).init_batch_seeded(src, self.t, LOGQ, self.seed, self.pub_seed, self.next, DECOMP)
        self.next += len(src) * self.nd                                     # (one counter for every column ever drawn from this seed)
        return m.download()

    def s2_matrix(self):
        t2 = self.t.copy()
        t2.op(self.t, 2)
        return self._matrix([self.one, self.t, t2])

    def automorph_matrix(self, k):
        tk = self.t.copy()
        tk.automorph(k)
        return self._matrix([self.one, tk])


def run_regression(args, ctx, ksk, primes, n, nd, nl, rank, world, local_rank, dist, torch, F, chain_bits, keygen=None):
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
        host = None
        if rank == 0:
            host = keygen.automorph_matrix(ks[i]) if keygen is not None else rand_residue_rows(np.random.default_rng(100 + i), primes, (2, 2 * nd), n)
        if dist is not None:
            stage = shard.broadcast_key_matrix(host, a.nbytes, dist, device=dev)
            torch.cuda.synchronize()
            a.upload_dev(stage.data_ptr())
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
    be = G.DeviceBackend(ctx, LOGQ, P_PLAIN, ksk, autos, ks, pool, DECOMP, overlap=args.reg_overlap)
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
    kname = ctx.prof_kernel_name("ntt_fwd_digits_main")
    ctx.prof_enable(False)
    launches, rows, ms = prof["ntt_fwd_digits_main"]
    # (the 32-bit digit rows have 2^14 elements, also on the zero-padded linear-convolution rings; the bytes counted are the rows' own)
    row_elems = max(n if M_RING & (M_RING - 1) == 0 else 1 << (2 * n - 2).bit_length(), 1 << 14) if kname.startswith("ntt32_") else n      # (padded rows of 2^14 / 2^15 on the linear-convolution rings)
    achieved = rows * 2 * row_elems * (4 if kname.startswith("ntt32_") else 8) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    ks_form, ks_rows, ks_limb_bits = ksk.form()
    if rank == 0:
        ksw = stats["key_switches"] + stats["automorph_key_switches"]
        line = {
            "metric": f"key-switched ciphertext products/sec inside Regression::Regress (wave-scheduled) at phi(m)={n}, logQ={LOGQ}",
            "value": round(ksw * args.steps / dt, 2), "unit": "key-switches/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 rows (tensor products and key switch over primes below 2^30), u64 coefficient limbs" if kname.startswith("ntt32_") else "u64", "data": "synthetic",
            "config": {"workload": (f"configs[3] replay: Regression::Regress d={d}, {N} data block(s), m=2^15 n=2^14, fhe-si logQ=512, p=23, decompSize=3" if M_RING == 1 << 15 else
                                    f"configs[3]: Regression::Regress d={d}, {N} data block(s) of phi(m) slots on the reference's Test_Regression ring m={M_RING} (p={P_PLAIN}), fhe-si logQ={LOGQ}, decompSize=3"),
                       "L": L, "chain_bits": round(chain_bits, 1), "sp_nbits": args.sp_nbits, "ndigits": nd,
                       "key_switch_form": {"form": F.KeySwitchMatrix.FORMS.get(ks_form, str(ks_form)), "rows": ks_rows, "limb_bits": ks_limb_bits,
                                           "centred_limbs": ksk.key_bits()[0], "key_coefficient_bits": ksk.key_bits()[1]}, "keys": args.keys, "automorphism_keys": len(ks), "waves": stats["waves"],
                       "products_per_regress": stats["products"], "key_switches_per_regress": stats["key_switches"],
                       "automorph_key_switches_per_regress": stats["automorph_key_switches"], "regress_per_s": round(args.steps / dt, 3),
                       "sharding": "groups of every wave sharded over ranks, outputs exchanged by RCCL broadcast" if dist is not None else "single GPU",
                       "exchange_overlap_chunks": args.reg_overlap, "waves_run_in_chunks": sum(1 for _, c in pool.schedule if c > 1)},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
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
    B = args.batch if args.batch else 8192          # (65536 rows per launch: at 1024 the launch is 10.7 rounds of the resident workgroups and the sub-millisecond kernels never leave the clock ramp -- profiles/r04_pmc_sq_ntt64.txt)
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
    fname, iname = ctx.prof_kernel_name("ntt_fwd"), ctx.prof_kernel_name("ntt_inv")
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
            "roofline": {"bound": "hbm", "kernel": fname, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches": fl, "avg_launch_ms": round(fms / fl, 4) if fl else None,
                         "row_ntts_per_s": round(frows / (fms * 1e-3), 1) if fms > 0 else None,
                         "inverse": {"kernel": iname, "achieved": round(irows * row_bytes / (ims * 1e-3) / 1e9, 1) if ims > 0 else None,
                                     "row_ntts_per_s": round(irows / (ims * 1e-3), 1) if ims > 0 else None}},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
        if not identity or oracle_ok is False:
            raise SystemExit("NTT round trip / oracle parity failed")


def self_launch(args):
    """`python bench.py --gpus N` without an external launcher: start the N ranks as a child torch.distributed.run (one process per GPU),
    wait, relay its output and exit code.  Nothing in THIS process has touched the GPU (no torch / library import yet), and the child
    is a subprocess, never an exec of this one."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="ciphertext mults per GPU per step (default 1024 = 16 chunks of the library, so that 20 steps time more than "
                    "a second; stress: 64; ntt workload: DoubleCRTs per step, default 8192)")
    ap.add_argument("--cpu-sample", type=int, default=3, help="oracle ciphertext mults timed for cpu_baseline and compared with the timed output buffer (0 = skip)")
    ap.add_argument("--no-bluestein-cpu", action="store_true", help="skip the like-for-like (Bluestein-mode) CPU timing, about 20 s")
    ap.add_argument("--lanes", type=int, default=1, help="concurrent half-batches inside the library (option lanes); 2 gave ~+4 %% with launches of 64 ciphertexts, +0.6 %% with today's launches of 1024; it "
                    "overlaps kernels, so per-kernel durations (and the roofline line) are no longer those of a kernel running alone")
    ap.add_argument("--workload", default="metric", choices=["metric", "stress", "regression", "ntt", "refring"], help="metric = configs[2] (default, the contract line); "
                    "refring = the same multiplication on the reference drivers' own ring at the metric's size (Test_AddMul.cpp:131: m = p - 1 for the safe prime p = 32603, "
                    "phi(m) = 16300, logQ=512: Bluestein rows in the reference, linear convolutions on padded rows of 2^15 here) -- reporting only; "
                    "ntt = configs[1]: DoubleCRT forward+inverse round trips at n=2^13, 8 primes, --batch DoubleCRTs per GPU (default there: 8192); "
                    "stress = configs[4]: m=2^16 (n=2^15), logQ=1024, p=65537 (35 primes, 43 digits) -- reporting only; "
                    "regression = configs[3] replayed at the metric ring: Regression::Regress (d = --reg-dim, --reg-rows data blocks) in waves, "
                    "every wave's groups sharded over the ranks (strong scaling)")
    ap.add_argument("--reg-dim", type=int, default=8)
    ap.add_argument("--reg-rows", type=int, default=1)
    ap.add_argument("--reg-ring", default="metric", choices=["metric", "reference"], help="regression workload: replay at the metric ring (default) or on "
                    "the reference's own Test_Regression ring (p = 8423, m = 8422, logQ = 341: configs[3] itself)")
    ap.add_argument("--reg-overlap", type=int, default=1, help="regression workload at N > 1: every wave in this many chunks, the exchange of a chunk (asynchronous RCCL broadcasts) "
                    "travelling while the next chunk is computed (1 = compute the shard, then exchange)")
    ap.add_argument("--reg-p", type=int, default=8423, help="--reg-ring reference: the safe prime p (m = p - 1); 8423 = Test_Regression's, 32603 = phi(m) 16300")
    ap.add_argument("--ref-p", type=int, default=32603, help="--workload refring: the safe prime p (m = p - 1); 32603 = phi(m) 16300 (padded rows of 2^15), 65267 = phi(m) 32632 "
                    "(padded rows of 2^16: the largest safe prime below 2^16)")
    ap.add_argument("--keys", default="generated", choices=["generated", "uniform"], help="key-switch matrix of the mult workloads: generated = KeySwitchSI::Init of a "
                    "sampleHWt(64) secret key (what every reference driver holds; default); uniform = uniform residues in every row")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the N > 1 plumbing)")
    ap.add_argument("--one-device", action="store_true", help="plumbing check on a 1-GPU box: every rank uses GPU 0 (never for measurements)")
    ap.add_argument("--ntt-rows", type=int, default=0, help="extra: rows for a standalone forward-NTT timing (0 = use pipeline launches)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="library option (fhesi_ctx_set_option), e.g. ks_direct=1, tensor32=0")
    ap.add_argument("--no-surface", dest="surface", action="store_false", help="skip the class-surface / host-buffer rates (about 15 s)")
    ap.add_argument("--sp-nbits", type=int, default=SP_NBITS, help="where the prime chain starts (FHEContext.cpp:92: 2^NTL_SP_NBITS): 60 = today's NTL (default, the contract "
                    "line), 50 = the NTL 5.x / 6.x of the reference's era (22 primes instead of 18 at the metric ring)")
    ap.add_argument("--gpu-seconds", type=float, default=5.0, help="metric / stress workloads: the block of exactly --steps timed steps is repeated until the GPU phase has "
                    "lasted about this long (every block bracketed like the first; `value` is the MEDIAN block, all blocks are listed); 0 = one block")
    return ap.parse_args()


def select_ring(args):
    """the ring, logQ and plaintext modulus of the workload (module globals: the helpers above read them)"""
    global M_RING, LOGQ, P_PLAIN
    if args.workload == "stress":
        M_RING, LOGQ, P_PLAIN = 1 << 16, 1024, 65537
    if args.workload == "refring":
        if not _is_prime(args.ref_p) or not _is_prime((args.ref_p - 1) // 2):
            raise SystemExit(f"--ref-p {args.ref_p}: the reference's rings are m = p - 1 for a safe prime p")
        M_RING, LOGQ, P_PLAIN = args.ref_p - 1, 512, args.ref_p
    if args.workload == "regression" and args.reg_ring == "reference":
        # Test_Regression.cpp:100-108: m = p - 1, logQ by its noise formula (p = 8423, d = 8: 341; p = 32603 -- phi(m) = 16300, the metric's size in
        # the reference's own parameterisation --: 377)
        pp, dim = args.reg_p, args.reg_dim
        if not _is_prime(pp) or not _is_prime((pp - 1) // 2):
            raise SystemExit(f"--reg-p {pp}: the reference's rings are m = p - 1 for a safe prime p")
        nn = (pp - 1) // 2 - 1
        lgq = 4.5 * math.log(nn) + max(1, dim - 1) * (math.log(1280) + 2 * math.log(nn) + math.log(max(args.reg_rows, dim)))
        M_RING, LOGQ, P_PLAIN = pp - 1, int(math.ceil(lgq / math.log(2) + 24.7)), pp


class Ranks:
    """this process among the ranks of the job: rank / local_rank / world, the process group (dist, None at N = 1), the devices of the roll call
    (topo) and whether barriers, stopwatches and object collectives run on the CPU backend of the group (cpu_side)"""
    rank = 0
    local_rank = 0
    world = 1
    dist = None
    topo = None
    cpu_side = False
    comm_init_s = None      # seconds of the first device collective (it creates the RCCL communicator: kept out of the key broadcast's GB/s)


def init_ranks(args, torch):
    """One process per GPU.  N > 1: gloo for what lives in host memory (roll call, barriers, stopwatches), RCCL for what lives in HBM (the key
    broadcast, the exchanges of the regression waves).  No device_id: the RCCL communicator is created by the first device collective, i.e. AFTER
    the roll call has established that every rank drives its own GPU.  A build without gloo takes the roll call through the rendezvous store
    itself (shard.roll_call_store) and only then creates the RCCL group on that store -- no communicator before the ranks' GPUs are known."""
    r = Ranks()
    r.rank = int(os.environ.get("RANK", "0"))
    r.local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    r.world = int(os.environ.get("WORLD_SIZE", "1"))
    if r.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={r.world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (HIP path only; there is no CPU fallback)")
    if torch.cuda.device_count() <= r.local_rank:
        raise SystemExit(f"bench: rank {r.rank} wants cuda:{r.local_rank} but this process sees {torch.cuda.device_count()} GPU(s) "
                         f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')}); one process per GPU, or --one-device for plumbing checks on a 1-GPU box")
    torch.cuda.set_device(r.local_rank)
    if r.world == 1 and os.environ.get("FHESI_BENCH_GROUP_AT_N1") != "1":
        return r
    if r.world == 1:
        # Test hook: ONE rank runs the whole N > 1 path -- process group over gloo + real RCCL, roll call, key broadcast, per-rank parity, digests --
        # so that a 1-GPU box exercises every call the first multi-GPU run will make (RCCL forms a communicator of one rank; it refuses two
        # ranks on one GPU, which is why the 2-rank tests of that box have to use gloo).
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(port))):
            os.environ.setdefault(k_, v_)
    import torch.distributed as dist
    from fhe_si_amd import shard
    r.dist = dist
    ident = shard.device_identity(torch, r.local_rank)
    try:
        if args.backend == "nccl":
            try:
                if os.environ.get("FHESI_BENCH_NO_MIXED_GROUP") == "1":          # (test hook: take the fallback below as a build without gloo would)
                    raise RuntimeError("mixed group disabled by FHESI_BENCH_NO_MIXED_GROUP")
                dist.init_process_group("cpu:gloo,cuda:nccl")
                r.cpu_side = True
            except Exception as e:
                print(f"bench: mixed gloo + nccl process group not available ({e}); roll call through the rendezvous store, then nccl only", file=sys.stderr)
                # (the store init_process_group("env://") itself would use: under torchrun a client of the agent's store, else rank 0 hosts it)
                store, _, _ = next(dist.rendezvous("env://", rank=r.rank, world_size=r.world))
                r.topo = shard.roll_call_store(store, ident, r.rank, r.world, allow_shared=args.one_device)
                dist.init_process_group("nccl", store=store, rank=r.rank, world_size=r.world, device_id=torch.device("cuda", r.local_rank))
        else:
            dist.init_process_group(args.backend)
            r.cpu_side = True
        if r.topo is None:
            r.topo = shard.roll_call(dist, ident, allow_shared=args.one_device)
    except RuntimeError as e:          # a shared GPU: raised on EVERY rank, nobody is left waiting in a collective
        if dist.is_initialized():
            dist.destroy_process_group()
        raise SystemExit(f"bench: {e}")
    if args.backend == "nccl":
        # the first device collective creates the RCCL communicator (seconds on a real node): done here, timed on its own, so that the key
        # broadcast's GB/s is the broadcast's
        t0 = time.perf_counter()
        w = torch.zeros(2, dtype=torch.int64, device=f"cuda:{r.local_rank}")
        dist.broadcast(w, src=0)
        torch.cuda.synchronize()
        r.comm_init_s = round(time.perf_counter() - t0, 4)
    return r


def finish(r):
    if r.dist:
        r.dist.barrier()
        r.dist.destroy_process_group()


# ---- offline counter records (profiles/): used only when they were taken on the kernel and launch shape of THIS run -------------------------------
def offline_traffic(fname, kernel, shape_key, shape_val):
    """HBM bytes per launch from a committed `rocprofv3 --pmc` pass (collected offline as the microarch guide prescribes: separate
    passes, FETCH_SIZE corrected)."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        rec = json.load(open(path))
    except Exception:
        return None, None
    if rec.get("kernel") != kernel or rec.get(shape_key) != shape_val:
        return None, None
    return rec.get("hbm_bytes_per_launch"), f"profiles/{fname} (offline rocprofv3 --pmc passes on {rec.get('kernel')}, {shape_key}={shape_val})"


def offline_sq(kernel, batch):
    """SQ counters of one kernel from the committed pass (profiles/sq_main_kernels.json, tools/pmc_sq_multi.sh: four --pmc groups of the metric
    command at --batch 1024): VALU busy fraction, VALU instructions per wave, effective shader clock under load."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "sq_main_kernels.json")))
    except Exception:
        return None
    if rec.get("batch") != batch:
        return None
    k = rec.get("kernels", {}).get(kernel)
    if k:
        k = dict(k, source=f"profiles/sq_main_kernels.json ({rec.get('command')})")
    return k


# issue cycles of one wave-butterfly: v_add / v_sub issue a wave in 2 cycles, every multiply form, v_min and v_add3 in 4 (profiles/r02_ubench_int.txt)
BFLY_ISSUE_CYCLES = {"fwd30": 24.0,                                # mul_hi + 2 mad (12), range step sub + min (6), difference (2), three-operand sum (4)
                     "fwd29": (8 * 18.0 + 6 * 24.0) / 14,          # primes below 2^29: 8 of the 14 stages without the range step
                     "inv30": 24.0,                                # sum, difference + 2p (6), range step (6), three multiplies (12)
                     "inv29": (7 * 18.0 + 7 * 24.0) / 14}          # 6 of 13 range steps gone; the last stage has none in either form


def valu_fields(sq, kind, rows_per_s, row_log2=14):
    """what binds a row-transform kernel, from the committed SQ pass: VALU busy, VALU instructions per butterfly and the fraction of the ISSUE
    ceiling -- rows/s if the kernel issued nothing but butterflies on all 1024 SIMDs at the effective clock measured under load"""
    if not sq:
        return {"valu_busy": None, "valu_instr_per_butterfly": None, "issue_ceiling_frac": None, "valu_source": None}
    lane_bfly = row_log2 * (1 << (row_log2 - 1)) / 512.0                  # butterflies per lane of a 512-thread workgroup: 224 at 2^14 points
    wave_bfly = row_log2 * (1 << (row_log2 - 1)) / 64.0                   # wave-butterflies per row: 1792
    clk = sq.get("eff_clock_ghz_median")
    ceil_rows = 1024 * clk * 1e9 / (wave_bfly * BFLY_ISSUE_CYCLES[kind]) if clk else None
    return {"valu_busy": sq.get("valu_busy"), "valu_instr_per_butterfly": round(sq["valu_instr_per_wave"] / lane_bfly, 2) if sq.get("valu_instr_per_wave") else None,
            "issue_ceiling_frac": round(rows_per_s / ceil_rows, 3) if ceil_rows and rows_per_s else None,
            "issue_ceiling": {"butterfly_issue_cycles": round(BFLY_ISSUE_CYCLES[kind], 2), "eff_clock_ghz": clk, "rows_per_s_at_ceiling": round(ceil_rows, 1) if ceil_rows else None},
            "valu_source": sq.get("source")}


def bound_of(valu_busy):
    return "valu" if valu_busy is not None and valu_busy >= 0.9 else "hbm"


class KernelClock:
    """per-kernel-class HIP-event totals of the library, read after every timed block: the average launch of each block, so that the line can
    quote min / median over blocks beside the mean"""

    def __init__(self, ctx, classes):
        self.ctx, self.classes, self.last, self.blocks = ctx, classes, {k: (0, 0.0, 0.0) for k in classes}, {k: [] for k in classes}

    def mark(self):
        for k in self.classes:
            l, u, ms = self.ctx.prof_read(k)
            l0, u0, ms0 = self.last[k]
            if l > l0:
                self.blocks[k].append((ms - ms0) / (l - l0))
            self.last[k] = (l, u, ms)

    def stats(self, k):
        """(min, median) over the timed blocks of the average launch of class k, in ms"""
        b = self.blocks[k]
        if not b:
            return None, None
        b = sorted(b)
        return round(b[0], 4), round(b[(len(b) - 1) // 2], 4)


def make_inputs(ctx, rng, B, n, nl):
    """`uniq` distinct random ciphertext pairs per GPU, repeated to fill the batch (every pair is an independent multiplication; repeating
    keeps host generation and upload of multi-GiB batches short without making the data any less random)"""
    uniq = min(B, 64)
    a_host = rand_coeffs(rng, (uniq, 2, n), nl)
    b_host = rand_coeffs(rng, (uniq, 2, n), nl)
    ct_bytes = a_host.nbytes // uniq
    da, db, dout = ctx.alloc(ct_bytes * B), ctx.alloc(ct_bytes * B), ctx.alloc(ct_bytes * B)
    da.upload(a_host)
    db.upload(b_host)
    done = uniq
    while done < B:        # device-to-device replication
        cnt = min(done, B - done)
        ctx.dev_copy(da.ptr.value + done * ct_bytes, da.ptr.value, cnt * ct_bytes)
        ctx.dev_copy(db.ptr.value + done * ct_bytes, db.ptr.value, cnt * ct_bytes)
        done += cnt
    return uniq, a_host, b_host, ct_bytes, da, db, dout


def surface_rates(args, ctx, F, ksk, primes, n, nd, nl, B, uniq, a_host, b_host, da, db, dout):
    """What a caller of the kept class surface gets (missing 1): the same multiplication (a) from HOST buffers through
    fhesi_ct_mul_relin_batch at several batch sizes (upload + compute + download, pageable and pinned memory), (b) with a matrix of uniform
    residues, (c) one Ciphertext object at a time through the C++ mirror of Ciphertext::operator*= + KeySwitchSI::ApplyKeySwitch
    (tests/host/test_addmul --time, tests/host/test_lazy --time; host big-integer conversions included).  Reported beside `value`, never as it."""
    ncol = 3 * nd
    surface = {"device_resident_batch": None, "host_buffers": {}, "class_surface": None, "host_buffers_pinned": {}}
    first_out = None
    for hb in (1, 8, 64, 1024):
        try:
            reps = (hb + uniq - 1) // uniq
            ah = np.concatenate([a_host] * reps)[:hb] if hb > uniq else a_host[:hb]
            bh = np.concatenate([b_host] * reps)[:hb] if hb > uniq else b_host[:hb]
            oh = np.zeros_like(ah)
            oh.fill(1)                                                    # (touch the pages: a caller's result buffer exists before the call)
            ctx.ct_mul_relin(ksk, LOGQ, P_PLAIN, ah, bh, DECOMP, out=oh)  # (first call of a shape allocates)
            if hb == 8:
                first_out = oh.copy()
            best = None
            for _ in range(2 if hb >= 1024 else 5):
                t0 = time.perf_counter()
                ctx.ct_mul_relin(ksk, LOGQ, P_PLAIN, ah, bh, DECOMP, out=oh)
                d = time.perf_counter() - t0
                best = d if best is None or d < best else best
            surface["host_buffers"][str(hb)] = round(hb / best, 1)
            # the same from buffers the caller allocated pinned (fhesi_host_alloc): no staging copy
            pa, pb2, po = ctx.host_array(ah.shape), ctx.host_array(ah.shape), ctx.host_array(ah.shape)
            pa[...] = ah; pb2[...] = bh
            ctx.ct_mul_relin(ksk, LOGQ, P_PLAIN, pa, pb2, DECOMP, out=po)
            best = None
            for _ in range(2 if hb >= 1024 else 5):
                t0 = time.perf_counter()
                ctx.ct_mul_relin(ksk, LOGQ, P_PLAIN, pa, pb2, DECOMP, out=po)
                d = time.perf_counter() - t0
                best = d if best is None or d < best else best
            surface["host_buffers_pinned"][str(hb)] = round(hb / best, 1)
            surface["host_buffers_pinned_equals_pageable"] = bool(np.array_equal(po, oh)) and surface.get("host_buffers_pinned_equals_pageable", True)
            del ah, bh, oh, pa, pb2, po
        except MemoryError:
            surface["host_buffers"][str(hb)] = None
    if first_out is not None:
        # ... and the host-buffer results are those of the device-resident call on the same pairs
        surface["host_buffers_equal_device_batch"] = bool(np.array_equal(first_out, dout.download((8, 2, n, nl)))) if uniq >= 8 else None
    if args.keys == "generated":
        # the same timed step with a matrix of uniform residues (what rounds 1-3 measured): no key generation produces such rows, the
        # library measures the coefficients and runs its general limbs (15 instead of 7 at the metric ring)
        ku = F.KeySwitchMatrix(ctx, 3, nd).upload(rand_residue_rows(np.random.default_rng(8), primes, (2, ncol), n))
        ctx.ct_mul_relin_dev(ku, LOGQ, P_PLAIN, da, db, dout, nl, B, DECOMP)
        ctx.sync()
        ctx.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.ct_mul_relin_dev(ku, LOGQ, P_PLAIN, da, db, dout, nl, B, DECOMP)
        ctx.sync()
        t_u = time.perf_counter() - t0
        prof_u = {k: ctx.prof_read(k) for k in F.binding.PROF_CLASSES}
        names_u = {k: ctx.prof_kernel_name(k) for k in ("dot", "ntt_fwd_digits_main")}
        ctx.prof_enable(False)
        surface["uniform_key_matrix"] = {"value": round(5 * B / t_u, 1), "rows": ku.form()[1], "centred_limbs": ku.key_bits()[0],
                                         "kernel_ms_per_step": {k: round(v[2] / 5, 3) for k, v in prof_u.items() if v[0] and k != "ntt_fwd_digits_main"},
                                         "digits_ms_per_launch": round(prof_u["ntt_fwd_digits_main"][2] / max(1, prof_u["ntt_fwd_digits_main"][0]), 3),
                                         "kernels": names_u}
        ctx.ct_mul_relin_dev(ksk, LOGQ, P_PLAIN, da, db, dout, nl, B, DECOMP)      # (the timed buffer's contents again, for the oracle check)
        ctx.sync()
        del ku
    exe = os.path.join(ROOT, "tests", "host", "test_addmul")
    if os.path.exists(exe):
        import re
        import subprocess
        try:
            r = subprocess.run([exe, str(LOGQ), str(P_PLAIN), "7", "11", f"--m={M_RING}", f"--sp-nbits={args.sp_nbits}", "--time"], capture_output=True, text=True, timeout=600)
            cs = {"ok": r.returncode == 0 and "Test SUCCEEDED" in r.stdout}
            mm = re.search(r"second use\): .*? = ([0-9.]+) ciphertext-mults/s", r.stdout)
            if mm:
                cs["object_at_a_time"] = round(float(mm.group(1)), 1)
            surface["class_surface"] = cs
            # the per-object statements `c *= d; keySwitch.ApplyKeySwitch(c)` in a loop over 1024 (and 64) ciphertexts, results read
            # afterwards: the mirror records them and runs one device call (fhe-si_amd/host/fhesi_engine.h); tests/host/test_lazy --time
            lz = os.path.join(ROOT, "tests", "host", "test_lazy")
            if os.path.exists(lz):
                for cnt in (1024, 64):
                    r = subprocess.run([lz, "--time", str(cnt), str(M_RING), str(LOGQ), str(P_PLAIN), "7"], capture_output=True, text=True, timeout=600)
                    rates = [float(x) for x in re.findall(r"recorded: .*? = ([0-9.]+) per second", r.stdout)]
                    if r.returncode == 0 and rates:
                        cs[f"per_object_loop_{cnt}"] = round(max(rates[1:] or rates), 1)
                    mm = re.search(r"at once: .*? = ([0-9.]+) per second", r.stdout)
                    if mm and cnt == 64:
                        cs["per_object_statements_at_once"] = round(float(mm.group(1)), 1)
                    every = [float(x) for x in re.findall(r"result asked after every object: .*? = ([0-9.]+) per second", r.stdout)]
                    if every and cnt == 64:          # (the reference's semantics, Test_AddMul.cpp:59-86: the result of every statement is looked at before the next)
                        cs["result_read_after_every_object"] = round(max(every), 1)
        except Exception as e:          # the surface figures are extras: never fail the contract line over them
            surface["class_surface"] = {"ok": False, "error": str(e)[:200]}
    return surface


def ranks_parity(args, r, ctx, ksk, ksm_host, primes, roots, n, nl, B, uniq, ct_bytes, a_host, b_host, dout):
    """N > 1: parity of EVERY rank's timed buffer (the reference's predicate is per ciphertext, Test_AddMul.cpp:84-86), and of the replicas of
    the key matrix: (i) each rank checks its own first pair (and the batch's last) against the C oracle evaluated on THIS rank's HBM copy of the
    broadcast matrix; (ii) every rank recomputes rank 0's first pair on its own GPU with its own copy and the ranks compare 64-bit digests.
    all_ok = all ranks equal rank 0 and every rank equals the oracle (exit code 1 otherwise, on every rank)."""
    from fhe_si_amd import shard
    mine_ok = None
    first_chunk = dout.download((uniq, 2, n, nl))
    if args.cpu_sample > 0:
        ksm_local = ksm_host if r.rank == 0 else ksk.download()
        want1 = oracle_outputs(primes, roots, ksm_local, a_host, b_host, 1)
        last = np.frombuffer(ctx_download_tail(ctx, dout, B, ct_bytes), dtype=np.uint64).reshape(2, n, nl)
        mine_ok = bool(np.array_equal(first_chunk[0], want1[0])) and bool(np.array_equal(last, first_chunk[(B - 1) % uniq]))
        del ksm_local
    if r.rank == 0:
        a0, b0 = a_host[:1], b_host[:1]
    else:                       # rank 0's generator (seed 7 + 0), drawn in its order: a, then b
        rng0 = np.random.default_rng(7)
        a0 = rand_coeffs(rng0, (uniq, 2, n), nl)[:1].copy()
        b0 = rand_coeffs(rng0, (uniq, 2, n), nl)[:1].copy()
    d1a, d1b, d1o = ctx.upload(a0), ctx.upload(b0), ctx.alloc(ct_bytes)
    ctx.ct_mul_relin_dev(ksk, LOGQ, P_PLAIN, d1a, d1b, d1o, nl, 1, DECOMP)
    common = d1o.download((1, 2, n, nl))
    del d1a, d1b, d1o
    agree, digests = shard.all_ranks_agree(r.dist, shard.digest64(common))
    same_as_batch = bool(np.array_equal(common[0], first_chunk[0])) if r.rank == 0 else True      # (one ciphertext per call = the batch's first)
    all_ok, per_rank_ok = shard.all_ranks_ok(r.dist, mine_ok is not False and agree and same_as_batch)
    return {"all_ok": all_ok, "per_rank_ok": per_rank_ok, "oracle_checked_on_every_rank": args.cpu_sample > 0,
            "rank0_first_pair_digest_by_rank": [f"{d:016x}" for d in digests], "digests_equal": agree}


def run_mult(args, r, torch, F, options):
    """metric (configs[2], the contract line), stress (configs[4]) and refring: one step = one batch of B ciphertext mult + relinearize per GPU"""
    rank, world, local_rank, dist = r.rank, r.world, r.local_rank, r.dist
    grouped = dist is not None          # (N > 1 -- or one rank with the process group of an N > 1 run: FHESI_BENCH_GROUP_AT_N1)
    n = sum(1 for k in range(1, M_RING) if math.gcd(k, M_RING) == 1) if M_RING & (M_RING - 1) else M_RING // 2       # phi(m)
    primes = prime_chain(M_RING, LOGQ, P_PLAIN, n, 1, args.sp_nbits)
    roots = [root_2m(q, M_RING) for q in primes]
    L = len(primes)
    nd = (LOGQ + 8 * DECOMP - 1) // (8 * DECOMP)
    nl = (LOGQ + 63) // 64
    ncol = 3 * nd
    chain_bits = sum(math.log2(q) for q in primes)
    B = args.batch if args.batch else 1024          # (configs[4] asks for 1024 concurrent mults at the stress shape as well: 26 GB of operands and results)

    ctx = F.Context(M_RING, primes, roots, device=local_rank)
    for k, v in options.items():
        ctx.set_option(k, v)
    ksk = F.KeySwitchMatrix(ctx, 3, nd)

    # key-switch matrix: generated on rank 0, RCCL-broadcast over xGMI into every rank's HBM copy.  --keys generated (default): what the
    # reference's drivers hold -- KeySwitchSI(secretKey) (Test_AddMul.cpp:48-52 -> FHE-SI.cpp:153-226) of a sampleHWt(64) secret key, built on
    # the device by the seeded batch key generation; --keys uniform: uniform residues in every row (no matrix a key generation can produce:
    # the library then runs its general limbs, reported beside the line as surface.uniform_key_matrix)
    ksm_host = None
    keygen = None
    if rank == 0:
        if args.keys == "generated":
            keygen = KeyGen(ctx, F, n, nd)
            ksm_host = keygen.s2_matrix()
        else:
            ksm_host = rand_residue_rows(np.random.default_rng(8), primes, (2, ncol), n)
    bcast_s, bcast_t = None, None
    if grouped:
        from fhe_si_amd import shard
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter()
        bcast_t = {}
        stage = shard.broadcast_key_matrix(ksm_host, ksk.nbytes, dist, device=f"cuda:{local_rank}", timings=bcast_t)   # one RCCL broadcast
        torch.cuda.synchronize()
        bcast_s = round(time.perf_counter() - tb, 4)           # (staging + collective: split in config.multi_gpu.key_broadcast)
        ksk.upload_dev(stage.data_ptr())
        del stage
    else:
        ksk.upload(ksm_host)

    if args.workload == "regression":
        run_regression(args, ctx, ksk, primes, n, nd, nl, rank, world, local_rank, dist, torch, F, chain_bits, keygen)
        return True

    uniq, a_host, b_host, ct_bytes, da, db, dout = make_inputs(ctx, np.random.default_rng(7 + rank), B, n, nl)

    def step():
        ctx.ct_mul_relin_dev(ksk, LOGQ, P_PLAIN, da, db, dout, nl, B, DECOMP)

    for _ in range(max(1, args.warmup)):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ctx.prof_enable(True)
    clock = KernelClock(ctx, ("ntt_fwd_digits_main", "ntt_fwd", "ntt_inv", "dot", "rns_reduce", "crt"))
    tdev = "cpu" if r.cpu_side else f"cuda:{local_rank}"

    def timed_block():
        """exactly --steps steps between barrier + synchronize on both sides; the maximum over the ranks"""
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ctx.sync()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        d = time.perf_counter() - t0
        own = d
        if dist:
            tt = torch.tensor([d], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        clock.mark()
        return d, own

    t_phase = time.perf_counter()
    block_dt, own_dt = [], []
    d, own = timed_block()
    block_dt.append(d); own_dt.append(own)
    # the same number of blocks on every rank: decided from the (all-reduced) first block
    nblocks = max(1, min(64, int(math.ceil(args.gpu_seconds / max(d, 1e-6))))) if args.gpu_seconds > 0 else 1
    for _ in range(nblocks - 1):
        d, own = timed_block()
        block_dt.append(d); own_dt.append(own)
    gpu_phase_s = time.perf_counter() - t_phase
    dt = sorted(block_dt)[(len(block_dt) - 1) // 2]          # the median block (lower median for an even count)
    per_rank = None
    if dist:
        mine = torch.tensor([B * args.steps / sorted(own_dt)[(len(own_dt) - 1) // 2]], dtype=torch.float64, device=tdev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [round(float(v.item()), 1) for v in allv]
    nblk = len(block_dt)

    # live per-kernel timing of the timed region (HIP events on the context's stream); kernel names as the library launched them
    prof = {k: ctx.prof_read(k) for k in F.binding.PROF_CLASSES}
    names = {k: ctx.prof_kernel_name(k) for k in F.binding.PROF_CLASSES}
    ctx.prof_enable(False)

    # ---- roofline of the fused ByteDecomp + forward transform of the digit polynomials ------------------------------------------------------------
    launches, rows, ms = prof["ntt_fwd_digits_main"]
    kname = names["ntt_fwd_digits_main"]
    aux32 = kname.startswith("ntt32_")                      # digit rows transformed modulo four 30-bit primes (kernels_aux32.hip): 4-byte residues
    row_elems = (n if M_RING & (M_RING - 1) == 0 else max(1 << 14, 1 << (2 * n - 2).bit_length())) if aux32 else n      # (linear-convolution rings: padded rows of 2^14 / 2^15 / 2^16)
    elem = 4 if aux32 else 8
    row_bytes = 2 * row_elems * elem                        # row read once + written once (the nominal figure)
    standalone = bool(args.ntt_rows)
    if standalone:
        # optional standalone measurement on a fixed row count
        cnt = max(1, args.ntt_rows // L)
        buf = ctx.upload(rand_residue_rows(np.random.default_rng(1), primes, (cnt,), n))
        ctx.rows_ntt_fwd(buf, cnt)
        ctx.prof_enable(True)
        for _ in range(10):
            ctx.rows_ntt_fwd(buf, cnt)
        launches, rows, ms = ctx.prof_read("ntt_fwd")
        kname = ctx.prof_kernel_name("ntt_fwd")
        row_bytes = 2 * n * 8
        ctx.prof_enable(False)
    nominal = rows * row_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    # SURVEY 8(d) K2': the fused loader never reads a row -- it reads the scaled-down parts ONCE (3 polynomials of nl 64-bit limbs per
    # ciphertext) and writes the rows: algorithmic bytes = rows written + source read once
    cts_timed = B * args.steps * nblk
    k2_bytes = rows * row_elems * elem + cts_timed * 3 * nl * 8 * n if not standalone else rows * row_bytes
    achieved = k2_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic, traffic_src = (None, None)
    if args.workload == "metric" and not standalone and launches:
        traffic, traffic_src = offline_traffic("pmc_ntt_fwd.json", kname, "rows_per_launch", round(rows / launches))
    on_traffic = traffic * launches / (ms * 1e-3) / 1e9 if traffic and ms > 0 else None
    rows_per_s = rows / (ms * 1e-3) if ms > 0 else None
    sq_d = offline_sq(kname, B) if args.workload == "metric" and not standalone else None
    vf = valu_fields(sq_d, "fwd30", rows_per_s) if aux32 else valu_fields(None, "fwd30", None)
    bmin, bmed = clock.stats("ntt_fwd_digits_main") if not standalone else (None, None)
    roofline_ntt = {"bound": bound_of(vf["valu_busy"]), "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "basis": "SURVEY 8(d) K2': rows written + the scaled-down parts read once (the fused loader never reads a row)" if not standalone else "algorithmic 2 x row bytes",
                    "algorithmic_bytes_per_launch": round(k2_bytes / launches) if launches else None,
                    "frac_on_traffic": round(on_traffic / HBM_PEAK_GBS, 4) if on_traffic else None, "achieved_on_traffic": round(on_traffic, 1) if on_traffic else None,
                    "frac_nominal": round(nominal / HBM_PEAK_GBS, 4), "achieved_nominal": round(nominal, 1),
                    "traffic": traffic, "traffic_source": traffic_src, "traffic_measured": False,
                    "valu_busy": vf["valu_busy"], "valu_instr_per_butterfly": vf["valu_instr_per_butterfly"], "issue_ceiling_frac": vf["issue_ceiling_frac"],
                    "issue_ceiling": vf.get("issue_ceiling"), "valu_source": vf["valu_source"],
                    "launches": launches, "avg_launch_ms": round(ms / launches, 4) if launches else None, "min_block_launch_ms": bmin, "median_block_launch_ms": bmed,
                    "rows_per_launch": round(rows / launches, 1) if launches else None, "row_bytes": row_bytes // 2,
                    "row_ntts_per_s": round(rows_per_s, 1) if rows_per_s else None}

    # ---- the tensor half's forward rows (kernels_tensor32.hip: class ntt_fwd minus the digit rows), when it runs over the small primes -------------
    roofline_ntt_tensor = None
    tl, trows, tms = prof["ntt_fwd"]
    if names["ntt_fwd"].startswith("ntt32_") and aux32 and not standalone and trows > rows and tms > ms:
        t_rows, t_ms, t_l = trows - rows, tms - ms, max(1, tl - launches)
        t_ach = t_rows * 2 * row_elems * 4 / (t_ms * 1e-3) / 1e9
        ttr, ttr_src = offline_traffic("pmc_t32_fwd.json", names["ntt_fwd"], "rows_per_launch", round(t_rows / t_l)) if args.workload == "metric" else (None, None)
        sq_t = offline_sq(names["ntt_fwd"], B) if args.workload == "metric" else None
        vt = valu_fields(sq_t, "fwd29" if ctx.get_option("tensor_bits") == 29 else "fwd30", t_rows / (t_ms * 1e-3))
        roofline_ntt_tensor = {"bound": bound_of(vt["valu_busy"]), "kernel": names["ntt_fwd"], "achieved": round(t_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(t_ach / HBM_PEAK_GBS, 4), "traffic": ttr, "traffic_source": ttr_src, "traffic_measured": False,
                               "valu_busy": vt["valu_busy"], "valu_instr_per_butterfly": vt["valu_instr_per_butterfly"], "issue_ceiling_frac": vt["issue_ceiling_frac"],
                               "issue_ceiling": vt.get("issue_ceiling"), "valu_source": vt["valu_source"], "launches": t_l,
                               "avg_launch_ms": round(t_ms / t_l, 4), "rows_per_launch": round(t_rows / t_l, 1), "row_bytes": row_elems * 4,
                               "row_ntts_per_s": round(t_rows / (t_ms * 1e-3), 1)}

    # ---- the key-switch dot product through the auxiliary primes (kernels_ksaux.hip / kernels_aux32.hip): algorithmic bytes per launch of c
    # ciphertexts = (digit rows c*ncol*2 + key rows 2*R*2*ncol + output rows c*2*R*2) * n * 8
    dname = names["dot"]
    aux = not dname.startswith("dot_accum")      # dot_accum_kernel = the per-chain-prime dot product (option ks_direct)
    dl, dunits, dms = prof["dot"]
    # output rows per (ciphertext, key row, auxiliary prime): the limbs of the key's integer coefficients where the library runs the
    # key switch in limb mode, the L residues otherwise
    ks_form, ks_rows, ks_limb_bits = ksk.form()       # which exact form of the dot product ran, and its rows (limbs or residues) per key coefficient
    R = ks_rows if ks_rows > 0 else L
    if aux:
        dbytes = (dunits * (ncol * 2 + 2 * R * 2) + dl * (2 * R * 2 * ncol)) * n * 8
    else:
        dbytes = (dunits * (ncol + 2) * L + dl * (2 * ncol * L)) * n * 8
    dach = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
    dtraffic, dtraffic_src = (None, None)
    if aux and args.workload == "metric" and dl:
        dtraffic, dtraffic_src = offline_traffic("pmc_dot_aux.json", dname, "ciphertexts_per_launch", round(dunits / dl))
    sq_dot = offline_sq(dname, B) if args.workload == "metric" else None
    dmin, dmed = clock.stats("dot")
    roofline_dot = {"bound": bound_of(sq_dot.get("valu_busy") if sq_dot else None), "kernel": dname, "achieved": round(dach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(dach / HBM_PEAK_GBS, 4), "traffic": dtraffic, "traffic_source": dtraffic_src, "traffic_measured": False,
                    "valu_busy": sq_dot.get("valu_busy") if sq_dot else None, "valu_source": sq_dot.get("source") if sq_dot else None, "launches": dl,
                    "avg_launch_ms": round(dms / dl, 4) if dl else None, "min_block_launch_ms": dmin, "median_block_launch_ms": dmed,
                    "ciphertexts_per_launch": round(dunits / dl, 1) if dl else None,
                    "ms_per_64_ciphertexts": round(dms / dunits * 64, 4) if dunits else None,
                    "note": "exact integer dot product over four 30-bit primes: the digit rows stream from HBM once, the key block sits in LDS (dot32_kernel4) or streams from L2 "
                            "(dot32_kernel2); its algorithmic bytes per ciphertext shrink with the launch size (the key rows are read once per launch), so frac is comparable only "
                            "at equal ciphertexts_per_launch -- ms_per_64_ciphertexts is; see roofline_ntt for the DoubleCRT transform the metric's GB/s figure refers to"}
    # `roofline` is the kernel with the largest share of the step
    roofline = roofline_dot if dms >= ms else roofline_ntt

    surface = None
    if rank == 0 and not grouped and args.workload == "metric" and args.surface:
        surface = surface_rates(args, ctx, F, ksk, primes, n, nd, nl, B, uniq, a_host, b_host, da, db, dout)

    parity = ranks_parity(args, r, ctx, ksk, ksm_host, primes, roots, n, nl, B, uniq, ct_bytes, a_host, b_host, dout) if grouped else None
    ok = True if parity is None else parity["all_ok"]
    if rank == 0:
        total_mults = B * args.steps * world
        value = total_mults / dt
        if surface is not None:
            surface["device_resident_batch"] = round(value, 1)
        breakdown = {k: round(v[2] / (args.steps * nblk), 3) for k, v in prof.items() if v[0] and k != "ntt_fwd_digits_main"}
        cpu, matches = None, None
        if args.cpu_sample > 0:
            # The CPU baseline runs on rank 0 at ANY N (the reference's path is timed per ciphertext, Test_Regression.cpp:24-64): the full record
            # (all cores, Bluestein mode) at N = 1, the single-thread sample alone at N > 1 while the other ranks wait at the closing barrier.
            ns = min(args.cpu_sample, uniq) if args.workload != "refring" else 1      # (a Bluestein-mode oracle multiplication takes ~20 s)
            cpu, want = cpu_baseline(primes, roots, ksm_host, a_host, b_host, ns, 0 if (args.no_bluestein_cpu or args.workload != "metric" or grouped) else 1,
                                     all_cores=not grouped)
            if not grouped:                      # its outputs check the timed buffer (N > 1: every rank was checked above)
                got = dout.download((want.shape[0], 2, n, nl))
                last = np.frombuffer(ctx_download_tail(ctx, dout, B, ct_bytes), dtype=np.uint64).reshape(2, n, nl)
                # the batch repeats the `uniq` pairs: the last ciphertext of the batch equals output (B-1) % uniq of the first chunk
                first_chunk = dout.download((uniq, 2, n, nl))
                matches = bool(np.array_equal(got, want)) and bool(np.array_equal(last, first_chunk[(B - 1) % uniq]))
                ok = matches
        if parity is not None:
            matches = parity["all_ok"] if args.cpu_sample > 0 else None      # (without the oracle only the agreement of the ranks was checked: multi_gpu.parity)
        multi_gpu = None
        if grouped:
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:
                rccl = f"unavailable ({e})"
            try:
                nr = dist.group.WORLD._get_backend(torch.device("cuda")).size() if args.backend == "nccl" else dist.get_world_size()
            except Exception:
                nr = dist.get_world_size()
            multi_gpu = {"backend": args.backend + (" (cuda) + gloo (cpu side)" if args.backend == "nccl" and r.cpu_side else ""), "rccl_version": rccl,
                         "communicator_nranks": nr, "communicator_init_s": r.comm_init_s, "devices": r.topo, "distinct_devices": len({(t["host"], t["id"]) for t in r.topo}) if r.topo else None,
                         "one_device_plumbing_mode": bool(args.one_device),
                         "key_broadcast": dict(bcast_t, GBps=round(bcast_t["bytes"] / bcast_t["collective_s"] / 1e9, 2) if bcast_t and bcast_t.get("collective_s") else None) if bcast_t else None,
                         "parity": parity}
        line = {
            "metric": "homomorphic ciphertext-mults/sec (incl. relinearize) at n=2^14, logQ=512" if args.workload == "metric"
                      else f"homomorphic ciphertext-mults/sec (incl. relinearize) at phi(m)={n} (m={M_RING}), logQ={LOGQ} (the reference drivers' own ring)" if args.workload == "refring"
                      else "homomorphic ciphertext-mults/sec (incl. relinearize) at n=2^15, logQ=1024 (stress shape)",
            "value": round(value, 2), "unit": "ciphertext-mults/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (f"u32 (every transformed row is a row of 4-byte residues modulo a prime below 2^30: tensor half over primes of {ctx.get_option('tensor_bits')} bits, key switch over four of 30; coefficients are u64 limbs)"
                      if roofline_ntt_tensor else "u64 (chain-prime rows u64; key-switch rows u32 modulo four 30-bit primes)") if aux32 else "u64",
            "data": f"synthetic ({uniq} distinct uniform ciphertext pairs per GPU repeated to the batch; key-switch matrix " +
                    ("generated by KeySwitchSI::Init from a sampleHWt(64) secret key, FHE-SI.cpp:153-226)" if args.keys == "generated" else "of uniform residues)"),
            "config": {"workload": "configs[2]: full ciphertext mul + relinearize + scale-down, m=2^15 n=2^14, fhe-si logQ=512, p=23, decompSize=3"
                       if args.workload == "metric" else f"Test_AddMul's parameterisation at the metric's size: m = p - 1 = {M_RING} for the safe prime p = {P_PLAIN}, phi(m) = {n}, fhe-si logQ={LOGQ}, decompSize=3 "
                       f"(rows are Bluestein transforms in the reference; here linear convolutions on padded rows of 2^{(2 * n - 2).bit_length()} over 30-bit primes)" if args.workload == "refring"
                       else "configs[4] stress shape: m=2^16 n=2^15, fhe-si logQ=1024, p=65537, decompSize=3",
                       "L": L, "chain_bits": round(chain_bits, 1), "sp_nbits": args.sp_nbits, "ndigits": nd,
                       "key_switch_form": {"form": F.KeySwitchMatrix.FORMS.get(ks_form, str(ks_form)), "rows": ks_rows, "limb_bits": ks_limb_bits,
                                           "centred_limbs": ksk.key_bits()[0], "key_coefficient_bits": ksk.key_bits()[1]}, "keys": args.keys, "batch_per_gpu": B,
                       "options": {k: ctx.get_option(k) for k in ("lanes", "ks_direct", "ks_residues", "ks_aux60", "tensor32", "tensor_bits", "batch_chunk")},
                       "timed_region_s": round(dt, 3), "blocks": nblk, "block_values": [round(B * args.steps * world / d, 1) for d in block_dt],
                       "gpu_phase_s": round(gpu_phase_s, 3), "per_rank_value": per_rank, "n1_equivalent": per_rank[0] if per_rank else round(value, 1),
                       "key_broadcast_s": bcast_s,
                       "sharding": "independent ciphertexts per GPU, key-switch matrix RCCL-broadcast" if grouped else "single GPU",
                       "multi_gpu": multi_gpu},
            "matches_oracle": matches,
            "surface": surface,
            "roofline": roofline, "roofline_ntt": roofline_ntt, "roofline_ntt_tensor": roofline_ntt_tensor, "roofline_dot": roofline_dot, "cpu_baseline": cpu, "kernel_ms_per_step": breakdown,
            "kernel_ms_per_launch_blocks": {k: {"min": clock.stats(k)[0], "median": clock.stats(k)[1]} for k in clock.classes if clock.blocks[k]},
            "kernels": {k: v for k, v in names.items() if v},
        }
        print(json.dumps(line), flush=True)
    if not ok:
        finish(r)
        raise SystemExit("bench: the timed output buffer differs from the oracle" if parity is None else f"bench: parity failed on rank {rank} or another (per rank: {parity['per_rank_ok']}, digests equal: {parity['digests_equal']})")
    return True


def main():
    args = parse_args()
    select_ring(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import fhe_si_amd as F
    r = init_ranks(args, torch)
    options = {"lanes": args.lanes}
    for kv in args.option:
        k, _, v = kv.partition("=")
        options[k] = int(v)
    if args.workload == "ntt":
        run_ntt_round_trips(args, r.rank, r.world, r.local_rank, r.dist, torch, F)
    else:
        run_mult(args, r, torch, F, options)
    finish(r)


def ctx_download_tail(ctx, buf, B, ct_bytes):
    """bytes of the last ciphertext of a device batch"""
    import ctypes as C
    out = np.empty(ct_bytes, dtype=np.uint8)
    from fhe_si_amd import binding as Bd
    Bd._ck(Bd._load().fhesi_dev_download(ctx.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(buf.ptr.value + (B - 1) * ct_bytes), ct_bytes))
    return out.tobytes()


if __name__ == "__main__":
    main()
