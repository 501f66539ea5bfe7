"""CPU, world_size 2, gloo: the multi-GPU partitioning of bench.py / fhe-si_amd/shard.py -- ciphertext batches sharded
across ranks, key-switch matrix broadcast once from rank 0, results gathered in unit order.  The per-rank compute is the
C oracle here (checker standing in for the HIP kernels, which need a GPU); what is under test is the N>1 plumbing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, outdir):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import fhe_si_amd as F
    from fhe_si_amd import shard
    import fhesi_pyref as R
    import oracle_lib as O
    import params as P
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, logQ, p = 64, 100, 23
    primes, roots = P.chain_for(m, logQ, p)
    orc = O.Oracle(m, primes, roots)
    n, nd, nl, L = orc.phim, R.ndigits(logQ), (logQ + 63) // 64, len(primes)
    nbytes = 2 * 3 * nd * L * n * 8
    ksm = None
    if rank == 0:
        ksm = np.stack([P.rand_rows(np.random.default_rng(8), primes, n, 3 * nd) for _ in range(2)])
    t = shard.broadcast_key_matrix(ksm, nbytes, dist)
    ksm_local = t.numpy().view(np.uint64).reshape(2, 3 * nd, L, n)
    # every rank derives the same global inputs from the seed and works on its shard only
    rng = np.random.default_rng(7)
    a = P.rand_limbs(rng, (total, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (total, 2, n), nl, logQ)
    lo, hi = shard.shard_bounds(total, rank, world)
    local = np.stack([orc.ct_mul_relin(ksm_local, a[i], b[i], logQ, p) for i in range(lo, hi)]) if hi > lo else np.zeros((0, 2, n, nl), np.uint64)
    res = shard.gather_to_rank0(local, total, dist)
    if rank == 0:
        ref = np.stack([orc.ct_mul_relin(ksm, a[i], b[i], logQ, p) for i in range(total)])
        np.save(os.path.join(outdir, "ok.npy"), np.array([int(np.array_equal(res, ref))]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 2])
def test_two_rank_sharding(tmp_path, total):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    assert int(np.load(tmp_path / "ok.npy")[0]) == 1


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from fhe_si_amd import shard
    for total in (0, 1, 7, 8, 1023):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _roll_call_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fhe_si_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    # distinct devices: every rank gets the whole list
    got = shard.roll_call(dist, {"host": "h", "id": f"uuid:{rank}", "device_index": rank, "name": "x", "visible": None})
    res["distinct"] = [g["rank"] for g in got] == list(range(world)) and len({g["id"] for g in got}) == world
    # a shared device: refused on EVERY rank (nobody is left waiting), with the ranks named
    try:
        shard.roll_call(dist, {"host": "h", "id": "uuid:same", "device_index": 0, "name": "x", "visible": None})
        res["shared_refused"] = False
    except RuntimeError as e:
        res["shared_refused"] = "ranks share a GPU" in str(e) and "[0, 1]" in str(e)
    # ... unless the caller asked for the one-GPU plumbing mode; the same id on two HOSTS is two devices
    res["shared_allowed"] = len(shard.roll_call(dist, {"host": "h", "id": "uuid:same", "device_index": 0, "name": "x", "visible": None}, allow_shared=True)) == world
    res["two_hosts"] = len(shard.roll_call(dist, {"host": f"h{rank}", "id": "uuid:same", "device_index": 0, "name": "x", "visible": None})) == world
    # digests: equal arrays agree, a single differing bit on one rank is seen by all
    a = np.arange(1000, dtype=np.uint64)
    ok, vals = shard.all_ranks_agree(dist, shard.digest64(a))
    res["agree"] = ok and len(vals) == world
    if rank == 1:
        a[500] ^= 1
    ok, vals = shard.all_ranks_agree(dist, shard.digest64(a))
    res["disagree_seen"] = (not ok) and vals[0] != vals[1]
    ok_all, per = shard.all_ranks_ok(dist, rank != 1)
    res["ok_flags"] = (not ok_all) and per == [True, False]
    # the broadcast reports its two costs separately
    t = {}
    m = np.arange(64, dtype=np.uint64).reshape(2, 4, 2, 4) if rank == 0 else None
    out = shard.broadcast_key_matrix(m, 64 * 8, dist, timings=t)
    res["timings"] = set(t) == {"staging_s", "collective_s", "bytes"} and t["bytes"] == 512 and out.numpy().view(np.uint64).tolist() == list(range(64))
    np.save(os.path.join(outdir, f"r{rank}.npy"), np.array([int(all(res.values()))] + [int(v) for v in res.values()]))
    dist.barrier()
    dist.destroy_process_group()


def test_roll_call_digests_and_broadcast_timings(tmp_path):
    """What bench.py does before and after its timed region at N > 1 (CPU side of the process group): the device roll call refuses shared
    GPUs on every rank, result digests are compared across ranks, ok flags are combined, the key broadcast splits staging from collective."""
    port = _free_port()
    mp.spawn(_roll_call_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        v = np.load(tmp_path / f"r{r}.npy")
        assert v[0] == 1, (r, v.tolist())


def _store_roll_call_worker(rank, world, port, outdir, shared):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import datetime
    import torch.distributed as dist
    from fhe_si_amd import shard
    store = dist.TCPStore("127.0.0.1", port, world, rank == 0, timeout=datetime.timedelta(seconds=60))
    ident = {"host": "box", "id": "uuid:same" if shared else f"uuid:gpu{rank}", "device_index": 0, "name": "test", "visible": None}
    res = {}
    try:
        topo = shard.roll_call_store(store, ident, rank, world)
        res["refused"] = False
        res["topo"] = [t["rank"] for t in topo] == list(range(world)) and len({t["id"] for t in topo}) == world
    except RuntimeError as e:
        res["refused"] = "ranks share a GPU" in str(e)
        res["topo"] = True
    if not shared:
        # the process group is created on the SAME store afterwards (bench.py's fallback: nccl there, gloo here) and works
        dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
        ok, per = shard.all_ranks_ok(dist, True)
        res["group"] = ok and per == [True] * world
        dist.barrier()
        dist.destroy_process_group()
    np.save(os.path.join(outdir, f"s{rank}.npy"), np.array([int(res["refused"]), int(res["topo"]), int(res.get("group", True))]))


@pytest.mark.parametrize("shared", [False, True])
def test_roll_call_through_the_rendezvous_store(tmp_path, shared):
    """bench.py's fallback when the mixed gloo + nccl group is not available: the roll call goes through the TCPStore BEFORE any process group
    (and so any RCCL communicator) exists; shared GPUs are refused on every rank, distinct ones go on to a group built on that store."""
    port = _free_port()
    mp.spawn(_store_roll_call_worker, args=(2, port, str(tmp_path), shared), nprocs=2, join=True)
    for r in range(2):
        v = np.load(tmp_path / f"s{r}.npy").tolist()
        assert v == [int(shared), 1, 1], (r, v)
