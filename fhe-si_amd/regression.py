"""Regression::Regress (Regression.h:102-149) as waves of independent ciphertext products over a device-resident pool, with
the groups of every wave sharded across ranks (BASELINE.json configs[3]: "ciphertext batches sharded 8xMI355X, keys
RCCL-broadcast").  Python twin of `fhe-si_amd/host/fhesi_matrix.h::Regression::RegressBatched`: same expression DAG, same
memoisation of the minors the Laplace recursion (Matrix.cpp:227-263) would recompute, hence bit-identical ciphertexts.

Plumbing only: every ciphertext operation is a C-ABI call (`fhesi_ct_mul_sum_relin_dev`, `fhesi_ct_automorph_key_switch_dev`,
`fhesi_ct_add_dev`, `fhesi_ct_gather_dev`, `fhesi_ct_mul_long_dev`); torch supplies the pool's device memory and the
collective.  The one real exchange step of this workload is the all-gather of a wave's outputs: the next wave's products
read ciphertexts produced by every rank.

The scheduler talks to a *backend* with five methods (see `DeviceBackend`); the CPU tests plug in a checker-backed one to
exercise the N>1 schedule under gloo.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from .shard import shard_bounds


def total_slots(m: int, p: int, phim: int) -> int:
    """PlaintextSpace::GetTotalSlots (PlaintextSpace.cpp:29-31): phi(m) / ord_m(p) factors of Phi_m modulo p."""
    d, x = 1, p % m
    while x != 1:
        x = (x * p) % m
        d += 1
    return phim // d


def automorphism_exponents(m: int, g: int, p: int, phim: int) -> List[int]:
    """k = g, g^2, g^4, ... mod m, one per halving of the usable slot count (Regression.h:71-80)."""
    usable, t = 1, total_slots(m, p, phim)
    while t > 1:
        usable <<= 1
        t >>= 1
    ks, k = [], g
    while usable > 1:
        ks.append(k)
        usable >>= 1
        k = (k * k) % m
    return ks


class Wave:
    """Independent groups  out[g] = KeySwitch(sum_t pool[a_t] * pool[b_t])."""

    def __init__(self):
        self.a: List[int] = []
        self.b: List[int] = []
        self.seg: List[int] = [0]

    def product(self, ai: int, bi: int):
        self.a.append(ai)
        self.b.append(bi)

    def end_group(self) -> int:
        self.seg.append(len(self.a))
        return len(self.seg) - 2

    @property
    def groups(self) -> int:
        return len(self.seg) - 1

    def slice(self, lo: int, hi: int) -> Tuple[List[int], List[int], List[int]]:
        """groups [lo, hi) as their own (a, b, seg)."""
        t0, t1 = self.seg[lo], self.seg[hi]
        return self.a[t0:t1], self.b[t0:t1], [s - t0 for s in self.seg[lo:hi + 1]]


class ShardedPool:
    """Pool of unscaled 2-part ciphertexts [capacity][words] as a torch int64 tensor (device memory on GPUs, host memory in the
    CPU tests); entries are written once.  `exchange` makes the entries [first, first+count) -- of which this rank produced its
    `shard_bounds` slice -- identical on every rank."""

    def __init__(self, words: int, capacity: int, device=None, dist=None):
        import torch
        self.torch, self.dist = torch, dist
        self.words, self.used = words, 0
        self.t = torch.zeros((capacity, words), dtype=torch.int64, device=device if device is not None else "cpu")
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self._pending = []
        self.schedule = []          # (entries, chunks) of every sharded step since the pool was made

    def alloc(self, count: int) -> int:
        if self.used + count > self.t.shape[0]:
            raise RuntimeError("ciphertext pool exhausted: %d + %d > %d" % (self.used, count, self.t.shape[0]))
        first = self.used
        self.used += count
        return first

    def my_span(self, count: int) -> Tuple[int, int]:
        return shard_bounds(count, self.rank, self.world)

    def exchange(self, first: int, count: int):
        if self.dist is None or count == 0:          # (a process group of ONE rank still runs its collectives: bench.py's FHESI_BENCH_GROUP_AT_N1)
            return
        for r in range(self.world):           # shards differ by at most one entry: one broadcast per producing rank
            lo, hi = shard_bounds(count, r, self.world)
            if hi > lo:
                self.dist.broadcast(self.t[first + lo:first + hi], src=r)

    def exchange_begin(self, first: int, count: int):
        """the same broadcasts issued asynchronously (RCCL's own stream on GPUs): they travel while the caller computes its next chunk"""
        if self.dist is None or count == 0:
            return
        for r in range(self.world):
            lo, hi = shard_bounds(count, r, self.world)
            if hi > lo:
                self._pending.append(self.dist.broadcast(self.t[first + lo:first + hi], src=r, async_op=True))

    def exchange_end(self):
        for w in self._pending:
            w.wait()
        self._pending = []

    def run_sharded(self, first: int, count: int, compute, overlap: int = 1, sync_out=None, sync_in=None):
        """Entries [first, first + count) of a wave: `compute(lo, hi)` produces the entries first + lo .. first + hi of this rank, then the
        ranks exchange.  overlap > 1 cuts the wave into that many chunks (each sharded over ALL ranks) and lets the exchange of chunk k travel
        while chunk k + 1 is computed -- a wave's outputs are read by the NEXT wave only (Regression.h:102-149 over Matrix.cpp:182-263).  Which
        rank computes which entry changes with the chunking; the entries do not.  sync_out / sync_in: the stream hand-over between the
        compute side (C ABI) and the collective side (torch)."""
        C = max(1, min(int(overlap), count // self.world)) if self.dist is not None else 1
        for c in range(C):
            c0, c1 = shard_bounds(count, c, C)
            lo, hi = shard_bounds(c1 - c0, self.rank, self.world)
            if hi > lo:
                compute(c0 + lo, c0 + hi)
            if sync_out:
                sync_out()
            if C == 1:
                self.exchange(first, count)
            else:
                self.exchange_begin(first + c0, c1 - c0)
        self.exchange_end()
        if sync_in:
            sync_in()
        self.schedule.append((count, C))


class DeviceBackend:
    """The five operations of the schedule on an MI355X through the C ABI."""

    def __init__(self, ctx, logQ: int, p: int, ksk, auto_ksks: Sequence, auto_ks: Sequence[int], pool: ShardedPool, decomp_bytes: int = 3, overlap: int = 1):
        from .binding import DevBuf  # noqa: F401  (documentation of what .ptr means)
        self.ctx, self.logQ, self.p, self.ksk, self.auto_ksks, self.auto_ks, self.pool = ctx, logQ, p, ksk, list(auto_ksks), list(auto_ks), pool
        self.decomp_bytes = decomp_bytes
        self.overlap = overlap          # chunks per wave whose exchange overlaps the next chunk's compute (1 = none)
        self.nl = (logQ + 63) // 64
        assert pool.words == 2 * ctx.phim * self.nl
        self._tmp = None

    class _Raw:
        def __init__(self, ptr: int):
            self.ptr = ptr

    def _at(self, idx: int):
        return self._Raw(self.pool.t.data_ptr() + idx * self.pool.words * 8)

    def _sync_in(self):       # torch -> C ABI: the collective (torch's stream) must have landed
        if self.pool.t.is_cuda:
            self.pool.torch.cuda.synchronize()

    def _sync_out(self):      # C ABI -> torch
        self.ctx.sync()

    def upload(self, cts: np.ndarray) -> int:
        """cts: [count][2][phim][nl] uint64 on the host (the same on every rank)."""
        cts = np.ascontiguousarray(cts, dtype=np.uint64)
        first = self.pool.alloc(cts.shape[0])
        self.pool.t[first:first + cts.shape[0]].copy_(self.pool.torch.from_numpy(cts.view(np.int64).reshape(cts.shape[0], -1)))
        self._sync_in()
        return first

    def download(self, idx: int) -> np.ndarray:
        self._sync_out()
        return self.pool.t[idx].cpu().numpy().view(np.uint64).reshape(2, self.ctx.phim, self.nl)

    def run_wave(self, w: Wave) -> int:
        first = self.pool.alloc(w.groups)

        def compute(lo, hi):
            a, b, seg = w.slice(lo, hi)
            self.ctx.ct_mul_sum_relin_dev(self.ksk, self.logQ, self.p, self._at(0), self.nl, a, b, seg, self._at(first + lo), self.decomp_bytes)
        self.pool.run_sharded(first, w.groups, compute, self.overlap, self._sync_out, self._sync_in)
        return first

    def sum_batched(self, first: int, count: int):
        """Regression::SumBatchedData (Regression.h:166-178) on `count` consecutive entries, in place, each rank on its slice."""
        def compute(lo, hi):
            n = hi - lo
            if not self.auto_ksks:
                return
            if self._tmp is None or self._tmp.shape[0] < n:
                self._tmp = self.pool.torch.empty((n, self.pool.words), dtype=self.pool.torch.int64, device=self.pool.t.device)
            tmp = self._Raw(self._tmp.data_ptr())
            for ksk, k in zip(self.auto_ksks, self.auto_ks):
                self.ctx.ct_automorph_key_switch_dev(ksk, self.logQ, k, self._at(first + lo), self.nl, n, tmp, self.nl, self.decomp_bytes)
                self.ctx.ct_add_dev(self.logQ, self._at(first + lo), tmp, 2, self.nl, n)
        self.pool.run_sharded(first, count, compute, self.overlap, self._sync_out, self._sync_in)

    def negated(self, idx: Sequence[int]) -> int:
        """new entries = -1 * pool[idx] (Ciphertext::operator*=(long), Ciphertext.cpp:232-237); cheap, done by every rank"""
        if not len(idx):
            return self.pool.used
        first = self.pool.alloc(len(idx))
        self.ctx.ct_gather_dev(self._at(0), idx, self.pool.words, self._at(first))
        self.ctx.ct_mul_long_dev(self.logQ, self._at(first), -1, 2, self.nl, len(idx))
        return first


def regress_waves(backend, X: Sequence[Sequence[int]], y: Sequence[int]) -> Tuple[List[int], int, Dict[str, int]]:
    """Pool indices of (theta[0..d), det) for the data matrix X (N rows of d pool indices) and labels y, without the
    GenerateNoise masking (Regression.h:136-148 needs the slot embedding).  Also returns the work counts."""
    N, d = len(X), len(X[0])
    stats = {"waves": 0, "products": 0, "key_switches": 0, "automorph_key_switches": 0}

    def run(w: Wave) -> int:
        stats["waves"] += 1
        stats["products"] += len(w.a)
        stats["key_switches"] += w.groups
        return backend.run_wave(w)

    # wave 1: last = X^T y (Matrix.cpp:81-98) and the upper triangle of X^T X (Matrix.cpp:150-174), key switch, SumBatchedData
    w1 = Wave()
    for j in range(d):
        for i in range(N):
            w1.product(X[i][j], y[i])
        w1.end_group()
    for i in range(d):
        for j in range(i, d):
            for k in range(N):
                w1.product(X[k][i], X[k][j])
            w1.end_group()
    first1 = run(w1)
    backend.sum_batched(first1, w1.groups)
    stats["automorph_key_switches"] += w1.groups * len(getattr(backend, "auto_ks", []))
    last = [first1 + j for j in range(d)]
    A = [[-1] * d for _ in range(d)]
    g = d
    for i in range(d):
        for j in range(i, d):
            A[i][j] = A[j][i] = first1 + g
            g += 1
    if d == 1:
        return [last[0]], A[0][0], stats
    # negated copies of the entries: the `tmp *= -1` of the expansion acts on the unscaled entry (Matrix.cpp:245)
    negA = backend.negated([A[i][j] for i in range(d) for j in range(d)])

    def entry(r: int, c: int, neg: bool) -> int:
        return negA + r * d + c if neg else A[r][c]

    # minors needed by Invert (Matrix.cpp:182-200), memoised on (used rows, used columns); level[s] = partial determinants of size s
    level: List[Dict[Tuple[int, int], int]] = [dict() for _ in range(d)]

    def first_unused(mask: int) -> int:
        r = 0
        while mask >> r & 1:
            r += 1
        return r

    def need(R: int, C: int, dim: int):
        if (R, C) in level[dim]:
            return
        level[dim][(R, C)] = -1
        if dim == 1:
            return
        row = first_unused(R)
        for col in range(d):
            if not C >> col & 1:
                need(R | 1 << row, C | 1 << col, dim - 1)

    for i in range(d):
        for j in range(d):
            need(1 << i, 1 << j, d - 1)
    for (R, C) in level[1]:
        level[1][(R, C)] = A[first_unused(R)][first_unused(C)]     # size 1: the entry itself, no reduce (Matrix.cpp:238-241)
    for s in range(2, d):
        w, order = Wave(), []
        for (R, C) in sorted(level[s]):
            row, negative = first_unused(R), False
            for col in range(d):
                if C >> col & 1:
                    continue
                w.product(entry(row, col, negative), level[s - 1][(R | 1 << row, C | 1 << col)])
                negative = not negative
            w.end_group()
            order.append((R, C))
        first = run(w)
        for gi, key in enumerate(order):
            level[s][key] = first + gi
    # adjugate: adj(j,i) = (-1)^(i+j) minor(i,j) (Matrix.cpp:192-199)
    adj = [[0] * d for _ in range(d)]
    to_neg = []
    for i in range(d):
        for j in range(d):
            adj[j][i] = level[d - 1][(1 << i, 1 << j)]
            if (i + j) % 2 == 1:
                to_neg.append(adj[j][i])
    first_neg = backend.negated(to_neg)
    gi = 0
    for i in range(d):
        for j in range(d):
            if (i + j) % 2 == 1:
                adj[j][i] = first_neg + gi
                gi += 1
    # det = sum_i A(0,i) adj(i,0) (Matrix.cpp:202-212); theta = adj * last + MapAll key switch (Matrix.cpp:57-79, Regression.h:131-134)
    wf = Wave()
    for i in range(d):
        wf.product(A[0][i], adj[i][0])
    wf.end_group()
    for i in range(d):
        for k in range(d):
            wf.product(adj[i][k], last[k])
        wf.end_group()
    first_f = run(wf)
    return [first_f + 1 + i for i in range(d)], first_f, stats
