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
