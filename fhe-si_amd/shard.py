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


def broadcast_key_matrix(ksm_host, nbytes: int, dist, device=None, src: int = 0, timings: dict | None = None):
    """Rank `src` passes the [2][ncomp*nd][L][phim] uint64 key-switch matrix (KeySwitchSI::keySwitchMatrix,
    FHE-SI.cpp:206-208); every rank returns a torch tensor (int64 view, `nbytes` bytes) holding the broadcast copy on
    `device` (None = CPU for the gloo tests).  `timings` (optional) receives staging_s (rank src: host -> device copy of the matrix)
    and collective_s (the broadcast itself, synchronised) -- two different costs that a single stopwatch would add up."""
    import time
    import torch
    on_gpu = device is not None and str(device) != "cpu"
    t = torch.empty(nbytes // 8, dtype=torch.int64, device=device if device is not None else "cpu")
    t0 = time.perf_counter()
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(ksm_host).view(np.int64).reshape(-1)))
    if on_gpu:
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    dist.broadcast(t, src=src)
    if on_gpu:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    if timings is not None:
        timings["staging_s"] = round(t1 - t0, 4)
        timings["collective_s"] = round(t2 - t1, 4)
        timings["bytes"] = int(nbytes)
    return t


def device_identity(torch, index: int) -> dict:
    """What tells the GPU behind `cuda:index` of THIS process apart from any other GPU of the node: its UUID where the runtime
    reports one, else its PCI address; plus the name and the host.  (LOCAL_RANK alone proves nothing: with HIP_VISIBLE_DEVICES set
    per process, index 0 is a different GPU in every rank -- or the same one in all of them.)"""
    import socket
    props = torch.cuda.get_device_properties(index)
    parts = []
    u = getattr(props, "uuid", None)
    if u is not None and str(u).strip("0-") != "":
        parts.append("uuid:" + str(u))
    pci = [getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    if any(v is not None for v in pci):
        parts.append("pci:" + ":".join("?" if v is None else f"{int(v):x}" for v in pci))
    # (both where the runtime reports both: two GPUs differ in at least one, one GPU agrees with itself in both)
    ident = "|".join(parts) if parts else f"index:{index}"      # (index: nothing better -- distinct only within one process's view)
    return {"host": socket.gethostname(), "device_index": int(index), "id": ident, "name": getattr(props, "name", None),
            "visible": __import__("os").environ.get("HIP_VISIBLE_DEVICES") or __import__("os").environ.get("CUDA_VISIBLE_DEVICES")}


def roll_call(dist, ident: dict, allow_shared: bool = False) -> list:
    """Every rank's device identity, gathered on every rank through the CPU side of the process group BEFORE any device collective
    runs.  Two ranks naming the same (host, device id) raise RuntimeError on EVERY rank (all of them have left the gather, so none
    is left waiting): an RCCL communicator over a shared GPU hangs or fails in ways that are hard to read, and a run with shared
    devices measures nothing.  allow_shared = the plumbing mode of a 1-GPU box (bench.py --one-device)."""
    world = dist.get_world_size()
    got = [None] * world
    dist.all_gather_object(got, dict(ident, rank=dist.get_rank()))
    return _check_shared(got, allow_shared)


def _check_shared(got: list, allow_shared: bool) -> list:
    seen = {}
    for g in got:
        seen.setdefault((g["host"], g["id"]), []).append(g["rank"])
    shared = {k: v for k, v in seen.items() if len(v) > 1}
    if shared and not allow_shared:
        raise RuntimeError("ranks share a GPU: " + "; ".join(f"{k[1]} on {k[0]} <- ranks {v}" for k, v in shared.items()) +
                           " (one process per GPU is required: check LOCAL_RANK / HIP_VISIBLE_DEVICES; --one-device is the 1-GPU plumbing mode)")
    return got


def roll_call_store(store, ident: dict, rank: int, world: int, allow_shared: bool = False) -> list:
    """The roll call through the rendezvous key-value store itself (torch.distributed.TCPStore), for a build whose process group has no CPU
    backend: every rank posts its identity under its own key and reads everyone's (a blocking get) BEFORE any process group -- and with it
    any RCCL communicator -- exists.  Same verdict on every rank as roll_call."""
    import json
    import time
    store.set(f"fhesi_roll_call/{rank}", json.dumps(dict(ident, rank=rank)))
    got = [json.loads(bytes(store.get(f"fhesi_roll_call/{r}")).decode()) for r in range(world)]
    # Nobody leaves before everybody has read everything: the rank that hosts the store may be about to end the run (a shared GPU), and a
    # rank still reading would see a dead connection instead of the verdict.
    try:
        store.add("fhesi_roll_call/done", 1)
        t0 = time.time()
        while store.add("fhesi_roll_call/done", 0) < world and time.time() - t0 < 60.0:
            time.sleep(0.01)
    except Exception:          # (the host of the store left after seeing every rank done: this rank has all it needs)
        pass
    return _check_shared(got, allow_shared)


def digest64(arr: np.ndarray) -> int:
    """64-bit digest of an array's bytes (blake2b): what the ranks compare instead of shipping whole results"""
    import hashlib
    return int.from_bytes(hashlib.blake2b(np.ascontiguousarray(arr).tobytes(), digest_size=8).digest(), "little")


def all_ranks_agree(dist, value: int):
    """(True if every rank holds rank 0's value, the list of values by rank) -- an object collective on the CPU side"""
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, int(value))
    return all(v == got[0] for v in got), got


def all_ranks_ok(dist, ok: bool):
    """(True if `ok` on every rank, the list by rank)"""
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, bool(ok))
    return all(got), got


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
