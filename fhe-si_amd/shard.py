"""Multi-GPU partitioning of the ciphertext path: independent ciphertexts are data-parallel,
every rank holds the full context tables and a replica of the key-switch matrix, and the only collective is ONE
broadcast of that matrix (RCCL over xGMI on GPUs, gloo in the CPU tests).  No collective sits inside the data path."""
from __future__ import annotations

import numpy as np


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous, balanced shard [lo, hi) of `total` independent units for `rank` (first `total % world` ranks get one more)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_key_matrix(ksm_host, nbytes: int, dist, device=None, src: int = 0):
    """Rank `src` passes the [2][ncomp*nd][L][phim] uint64 key-switch matrix (KeySwitchSI::keySwitchMatrix,
    FHE-SI.cpp:206-208); every rank returns a torch tensor (int64 view, `nbytes` bytes) holding the broadcast copy on
    `device` (None = CPU for the gloo tests)."""
    import torch
    t = torch.empty(nbytes // 8, dtype=torch.int64, device=device if device is not None else "cpu")
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(ksm_host).view(np.int64).reshape(-1)))
    dist.broadcast(t, src=src)
    return t


def gather_to_rank0(local: np.ndarray, total: int, dist):
    """Collect the per-rank result shards (first axis = units of this rank) on rank 0 in unit order.
    Shards differ by at most one unit, so every rank pads to the largest shard (gather needs equal sizes)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    spans = [shard_bounds(total, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in spans)
    padded = np.zeros((cap,) + local.shape[1:], dtype=local.dtype)
    padded[:local.shape[0]] = local
    send = torch.from_numpy(padded.view(np.int64))
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, gather_list=bufs, dst=0)
    if rank != 0:
        return None
    out = np.empty((total,) + local.shape[1:], dtype=local.dtype)
    for (lo, hi), b in zip(spans, bufs):
        out[lo:hi] = b.numpy().view(local.dtype)[:hi - lo]
    return out
