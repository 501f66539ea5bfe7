"""CPU: the wave schedule of fhe_si_amd.regression (the Python twin of Regression::RegressBatched) against the literal
control flow of Matrix.cpp / Regression.h restated in the Python model, single rank and sharded over 2 gloo ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import fhesi_pyref as R
import wave_backends as WB
from fhe_si_amd import regression as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_waves(case, dist=None, overlap=1, pool_out=None):
    c = case["ctx"]
    nl = (c.logQ + 63) // 64
    N, d = len(case["X"]), len(case["X"][0])
    pool = G.ShardedPool(2 * c.phim * nl, 8 * (N * (d + 1) + 6 * d * d + 2 ** d * d + 16), dist=dist)
    if pool_out is not None:
        pool_out.append(pool)
    be = WB.PyrefBackend(c, case["ksm"], case["auto"], case["ks"], pool, overlap)
    first = be.upload([case["X"][i][j] for i in range(N) for j in range(d)] + case["y"])
    X = [[first + i * d + j for j in range(d)] for i in range(N)]
    y = [first + N * d + i for i in range(N)]
    theta, det, stats = G.regress_waves(be, X, y)
    return [be.download(i) for i in theta], be.download(det), stats


@pytest.mark.parametrize("d,N", [(1, 2), (2, 2), (3, 1)])
def test_waves_equal_literal_control_flow(d, N):
    case = WB.regression_case(d=d, N=N, seed=70 + d)
    c = case["ctx"]
    theta_ref, det_ref = R.regress(c, case["ksm"], case["auto"], case["ks"], case["X"], case["y"])
    theta, det, stats = run_waves(case)
    assert det == det_ref and theta == theta_ref
    assert stats["key_switches"] > 0 and stats["waves"] == (1 if d == 1 else d)        # inner products, minors of size 2..d-1, final
    if d == 2:      # decrypts to the regression over the plaintext ring Z_p[X]/Phi_m with SumBatchedData = sum of automorphisms
        p, m = c.p, c.m

        def pt_auto(msg, k):
            big = [0] * m
            for i, v in enumerate(msg):
                big[(i * k) % m] = (big[(i * k) % m] + v) % p
            f, df = c.phi, len(c.phi) - 1
            for i in range(m - 1, df - 1, -1):
                v = big[i]
                if v:
                    for j in range(df + 1):
                        big[i - df + j] = (big[i - df + j] - v * f[j]) % p
            return [x % p for x in big[:df]]

        def summed(msg):
            cur = list(msg)
            for k in case["ks"]:
                cur = [(x + y) % p for x, y in zip(cur, pt_auto(cur, k))]
            return cur

        mul = lambda a, b: [v % p for v in R.poly_mul_mod_phi(c, a, b)]
        add = lambda a, b: [(x + y) % p for x, y in zip(a, b)]
        neg = lambda a: [(-x) % p for x in a]
        mx, my = case["msgX"], case["msgY"]
        A = [[None] * 2 for _ in range(2)]
        for i in range(2):
            for j in range(2):
                acc = [0] * c.phim
                for k in range(N):
                    acc = add(acc, mul(mx[k][i], mx[k][j]))
                A[i][j] = summed(acc)
        last = []
        for i in range(2):
            acc = [0] * c.phim
            for k in range(N):
                acc = add(acc, mul(mx[k][i], my[k]))
            last.append(summed(acc))
        det_pt = add(mul(A[0][0], A[1][1]), neg(mul(A[0][1], A[1][0])))
        adj = [[A[1][1], neg(A[0][1])], [neg(A[1][0]), A[0][0]]]
        theta_pt = [add(mul(adj[i][0], last[0]), mul(adj[i][1], last[1])) for i in range(2)]
        assert R.decrypt(c, case["t"], det) == det_pt
        assert [R.decrypt(c, case["t"], th) for th in theta] == theta_pt


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, outdir, overlap):
    for q in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, q)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = WB.regression_case(d=3, N=1, seed=73)
    pools = []
    theta, det, stats = run_waves(case, dist=dist, overlap=overlap, pool_out=pools)
    if rank == 0:
        import pickle
        with open(os.path.join(outdir, "res.pkl"), "wb") as f:
            pickle.dump((theta, det, stats, pools[0].schedule), f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [1, 2, 3])
def test_two_rank_sharded_waves_equal_single_rank(tmp_path, overlap):
    """Each rank evaluates its shard of every wave's groups and the outputs are exchanged (ShardedPool.run_sharded); the result must not
    depend on the number of ranks -- nor on the exchange / compute overlap (overlap > 1: a wave in chunks, the asynchronous exchange of chunk k
    issued before chunk k + 1 is computed, all of them awaited at the end of the wave)."""
    import pickle
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), overlap), nprocs=2, join=True)
    with open(os.path.join(str(tmp_path), "res.pkl"), "rb") as f:
        theta2, det2, stats2, schedule = pickle.load(f)
    case = WB.regression_case(d=3, N=1, seed=73)
    theta1, det1, stats1 = run_waves(case)
    assert det2 == det1 and theta2 == theta1 and stats2 == stats1
    assert all(c == max(1, min(overlap, n // 2)) for n, c in schedule) and (overlap == 1 or any(c > 1 for _, c in schedule)), schedule
