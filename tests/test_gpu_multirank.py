"""GPU: the multi-GPU side of the C ABI on ONE GPU: (i) real RCCL with a one-rank communicator (proves librccl is
found and the collectives run on the library's streams), (ii) two ranks sharing GPU 0 -- a loopback group, because RCCL refuses
duplicate devices -- driven by one host thread per rank, through the key broadcast, the exchange of sharded wave outputs, the exact
all-reduce of partial scaled-up sums, and the C++ wave evaluator (GroupExecutor) whose results must be bit-identical to one GPU.
The 8-GPU scaling itself is measured by the driver with bench.py."""
import os
import subprocess
import threading

import numpy as np
import pytest

import fhe_si_amd as F
import fhesi_pyref as R
import oracle_lib as O
import params as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")


def _threads(n, fn):
    errs = []

    def run(r):
        try:
            fn(r)
        except Exception as e:      # noqa: BLE001
            errs.append((r, e))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_key_broadcast_exchange_and_allreduce(devices):
    m, logQ, p = 4096, 128, 23
    primes, roots = P.chain_for(m, logQ, p)
    G = len(devices)
    ctxs = [F.Context(m, primes, roots, device=d) for d in devices]
    orc = O.Oracle(m, primes, roots)
    n, L, nd, nl = ctxs[0].phim, len(primes), R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(3)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    a = P.rand_limbs(rng, (1, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (1, 2, n), nl, logQ)
    want = orc.ct_mul_relin(ksm, a[0], b[0], logQ, p)
    comms = F.Comm.init_all(devices)
    assert [c.rank for c in comms] == list(range(G)) and all(c.size == G for c in comms)
    ksks = [F.KeySwitchMatrix(ctxs[r], 3, nd) for r in range(G)]
    ksks[0].upload(ksm)
    for r in range(1, G):                                   # stale tables on the receivers must be rebuilt after the broadcast
        ksks[r].upload(np.zeros_like(ksm))
        ctxs[r].ct_mul_relin(ksks[r], logQ, p, a, b)
    _threads(G, lambda r: comms[r].ksk_broadcast(ksks[r], 0))
    for r in range(G):
        assert np.array_equal(ksks[r].download(), ksm)
        assert np.array_equal(ctxs[r].ct_mul_relin(ksks[r], logQ, p, a, b)[0], want), r
    # exchange: rank r owns a slice of a buffer of 7 "ciphertexts"
    words, total = 2 * n * nl, 7
    from fhe_si_amd import shard
    bounds = [shard.shard_bounds(total, r, G) for r in range(G)]
    full = rng.integers(0, 1 << 63, size=(total, words), dtype=np.uint64)
    bufs = []
    for r in range(G):
        mine = np.zeros_like(full)
        lo, hi = bounds[r]
        mine[lo:hi] = full[lo:hi]
        bufs.append(ctxs[r].upload(mine))
    off = [bounds[0][0] * words] + [hi * words for _, hi in bounds]
    _threads(G, lambda r: comms[r].exchange(ctxs[r], bufs[r], off))
    for r in range(G):
        assert np.array_equal(bufs[r].download(full.shape), full), r
    # the same exchange in two chunks through begin / end (the communicator's own stream behind an event of the compute stream): chunk 0 =
    # entries [0, 4), chunk 1 = entries [4, 7), each sharded over all ranks; work queued on the compute stream in between
    bufs2 = []
    chunks = [(0, 4), (4, 7)]
    for r in range(G):
        mine = np.zeros_like(full)
        for c0, c1 in chunks:
            lo, hi = shard.shard_bounds(c1 - c0, r, G)
            mine[c0 + lo:c0 + hi] = full[c0 + lo:c0 + hi]
        bufs2.append(ctxs[r].upload(mine))

    def overlapped(r):
        for c0, c1 in chunks:
            bb = [shard.shard_bounds(c1 - c0, q, G) for q in range(G)]
            comms[r].exchange_begin(ctxs[r], bufs2[r], [(c0 + bb[0][0]) * words] + [(c0 + hi) * words for _, hi in bb])
            ctxs[r].ct_mul_relin(ksks[r], logQ, p, a, b)          # (compute between the begins: the exchange travels meanwhile)
        comms[r].exchange_end(ctxs[r])
        comms[r].exchange_end(ctxs[r])                             # (idempotent: nothing pending)
    _threads(G, overlapped)
    for r in range(G):
        assert np.array_equal(bufs2[r].download(full.shape), full), r
    # exact all-reduce of partial scaled-up sums: 2 DoubleCRTs per rank
    parts = [P.rand_rows(np.random.default_rng(50 + r), primes, n, 2) for r in range(G)]
    parts[0][0, 0, :] = np.uint64(primes[0] - 1)
    dev = [ctxs[r].upload(parts[r]) for r in range(G)]
    _threads(G, lambda r: comms[r].allreduce_rows(ctxs[r], dev[r], 2))
    tot = np.zeros((2, L, n), dtype=object)
    for r in range(G):
        tot = tot + parts[r].astype(object)
    for i, q in enumerate(primes):
        tot[:, i, :] %= q
    for r in range(G):
        assert np.array_equal(dev[r].download(parts[0].shape).astype(object), tot), r
    for c in comms:
        c.destroy()


@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_metric_key_matrix_through_the_broadcast(devices):
    """The set-up collective at its real size: configs[2]'s key-switch matrix (2 x 66 DoubleCRTs of 18 x 2^14 words = 297 MiB) goes through
    fhesi_ksk_broadcast -- real RCCL with one rank, and rank 0 -> rank 1 in a group sharing the GPU -- into a replica that had already
    served a key switch (its derived tables are stale and must be rebuilt); the receiving rank's multiplication + key switch then equals
    the oracle's on the first and the last ciphertext of a batch, and equal exchange shards take the single all-gather."""
    m, logQ, p, count = 1 << 15, 512, 23, 9
    primes, roots = P.chain_for(m, logQ, p)
    G = len(devices)
    ctxs = [F.Context(m, primes, roots, device=d) for d in devices]
    orc = O.Oracle(m, primes, roots)
    n, nd, nl = ctxs[0].phim, R.ndigits(logQ), (logQ + 63) // 64
    rng = np.random.default_rng(297)
    ksm = np.stack([P.rand_rows(rng, primes, n, 3 * nd) for _ in range(2)])
    assert ksm.nbytes == 2 * 66 * 18 * 16384 * 8
    a = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    b = P.rand_limbs(rng, (count, 2, n), nl, logQ)
    comms = F.Comm.init_all(devices)
    ksks = [F.KeySwitchMatrix(ctxs[r], 3, nd) for r in range(G)]
    ksks[0].upload(ksm)
    for r in range(1, G):
        ksks[r].upload(np.ascontiguousarray(ksm[::-1]))      # another matrix, used once: stale derived tables on the receiver
        ctxs[r].ct_mul_relin(ksks[r], logQ, p, a[:1], b[:1])
    _threads(G, lambda r: comms[r].ksk_broadcast(ksks[r], 0))
    last = G - 1
    got = ctxs[last].ct_mul_relin(ksks[last], logQ, p, a, b)
    assert ksks[last].form()[0] == 1
    for c in (0, count - 1):
        assert np.array_equal(got[c], orc.ct_mul_relin(ksm, a[c], b[c], logQ, p)), c
    if G > 1:
        assert np.array_equal(ctxs[0].ct_mul_relin(ksks[0], logQ, p, a, b), got)
    # equal shards: 2 G ciphertexts, G ranks
    words, total = 2 * n * nl, 2 * G
    full = rng.integers(0, 1 << 63, size=(total, words), dtype=np.uint64)
    bufs = []
    for r in range(G):
        mine = np.zeros_like(full)
        mine[2 * r:2 * r + 2] = full[2 * r:2 * r + 2]
        bufs.append(ctxs[r].upload(mine))
    off = [2 * r * words for r in range(G + 1)]
    _threads(G, lambda r: comms[r].exchange(ctxs[r], bufs[r], off))
    for r in range(G):
        assert np.array_equal(bufs[r].download(full.shape), full), r
    for c in comms:
        c.destroy()


@pytest.mark.parametrize("overlap", [1, 2, 3])
@pytest.mark.parametrize("args,devices", [(("23", "7", "3", "2", "2"), "0,0"), (("257", "3", "4", "2", "5"), "0,0,0"), (("47", "5", "3", "1", "6"), "0"),
                                          (("47", "5", "5", "1", "6"), "0,0,0,0,0,0,0,0")])
def test_wave_evaluator_sharded_over_ranks_is_bit_identical(args, devices, overlap):
    """fhesi::Regression::RegressBatchedMultiGpu (fhe-si_amd/host/fhesi_matrix.h: GroupExecutor, one host thread per rank, keys
    broadcast from rank 0, every wave's groups sharded by shard_bounds, outputs exchanged) against the one-GPU waves and the plaintext
    regression.  `0` = one rank through real RCCL; `0,0` / `0,0,0` / eight zeros = ranks sharing GPU 0 (loopback group).
    overlap > 1: every wave in that many chunks, the exchange of chunk k (fhesi_comm_exchange_begin, the communicator's own stream) travelling
    while chunk k + 1 is computed, fhesi_comm_exchange_end closing the wave -- the same ciphertexts bit for bit, and the schedule is printed."""
    subprocess.check_call(["make", "-C", HOST, "-j8"], stdout=subprocess.DEVNULL)
    extra = [f"--overlap={overlap}"] if overlap > 1 else []
    r = subprocess.run([os.path.join(HOST, "test_regression"), *args, "--batched-only", f"--devices={devices}", *extra], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "multi-rank ciphertexts bit-identical to one GPU: yes" in r.stdout
    assert "batched: decrypts to the plaintext regression: yes" in r.stdout
    if overlap > 1:
        assert "overlapped-exchange ciphertexts bit-identical to one GPU: yes" in r.stdout, r.stdout[-3000:]
        if devices != "0":          # (one rank: nothing to exchange, every wave is one chunk)
            import re
            m = re.search(r"waves run in chunks: (\d+) of (\d+)", r.stdout)
            assert m and int(m.group(1)) >= 1 and "overlaps the next chunk's compute" in r.stdout, r.stdout[-3000:]


def test_config3_sharded_over_eight_ranks():
    """configs[3] as BASELINE.json words it -- Test_Regression d = 8, one block of 4096 points on the reference's ring (p = 8423, m = 8422,
    logQ = 341), the ciphertext batches of every wave sharded over EIGHT ranks with the keys broadcast from rank 0 -- on the one GPU of this
    box: eight ranks in a group sharing GPU 0 (fhesi_comm_init_all's loopback group; RCCL itself needs eight devices and is exercised by the
    driver's 8-GPU run).  The sharded waves must produce the ciphertexts of the one-GPU evaluation bit for bit and decrypt to the plaintext
    regression (Regression.h:193-214)."""
    subprocess.check_call(["make", "-C", HOST, "-j8"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(HOST, "test_regression"), "8423", "7", "8", "1", "1", "--batched-only", "--check=slots", "--devices=0,0,0,0,0,0,0,0"],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "phi(m)=4210 logQ=341 primes=13 ndigits=15 dim=8 rows=1" in r.stdout, r.stdout
    assert "multi-rank ciphertexts bit-identical to one GPU: yes" in r.stdout
    assert "batched: decrypts to the plaintext regression: yes" in r.stdout


@pytest.mark.parametrize("workload,extra", [("metric", ["--batch", "64"]), ("regression", ["--reg-dim", "3"]), ("ntt", ["--batch", "64"]),
                                            ("regression", ["--reg-dim", "4", "--reg-overlap", "2"]),      # waves in chunks, exchange of a chunk overlapped with the next chunk's compute
                                            ("regression", ["--reg-dim", "3", "--reg-ring", "reference"])])      # configs[3] on the reference's own ring
def test_bench_launches_its_own_ranks(workload, extra):
    """`python bench.py --gpus 2` without an external launcher (how a single-command driver starts it): the parent spawns the ranks
    as a child torch.distributed.run before touching the GPU and relays rank 0's JSON line.  On this 1-GPU box both ranks use GPU 0 and
    the collectives run over gloo (--one-device --backend gloo: plumbing only, never a measurement)."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--backend", "gloo", "--workload", workload,
                        "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--gpu-seconds", "0.5", *extra], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    assert d["roofline"]["kernel"] and d["roofline"]["frac"] > 0
    if "--reg-overlap" in extra:
        assert d["config"]["exchange_overlap_chunks"] == 2 and d["config"]["waves_run_in_chunks"] >= 1, d["config"]
    if workload == "metric":
        assert len(d["config"]["per_rank_value"]) == 2 and d["config"]["blocks"] >= 1 and d["config"]["key_broadcast_s"] > 0
        mg = d["config"]["multi_gpu"]
        assert mg["communicator_nranks"] == 2 and len(mg["devices"]) == 2 and mg["distinct_devices"] == 1 and mg["one_device_plumbing_mode"]
        assert mg["key_broadcast"]["staging_s"] >= 0 and mg["key_broadcast"]["collective_s"] > 0 and mg["key_broadcast"]["bytes"] > 0
        par = mg["parity"]          # every rank recomputed rank 0's first pair with its own copy of the key matrix
        assert par["all_ok"] and par["digests_equal"] and len(set(par["rank0_first_pair_digest_by_rank"])) == 1 and par["per_rank_ok"] == [True, True]
        assert d["matches_oracle"] is None          # (--cpu-sample 0: the ranks' agreement only)


def test_bench_two_ranks_checks_every_rank_against_the_oracle():
    """N > 1 with the oracle on: every rank holds its own timed buffer to the oracle evaluated on ITS copy of the broadcast key matrix,
    all ranks agree on rank 0's first pair, and the line says so (matches_oracle true instead of null)."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "1", "--no-bluestein-cpu", "--gpu-seconds", "0", "--batch", "16"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["matches_oracle"] is True and d["config"]["multi_gpu"]["parity"]["oracle_checked_on_every_rank"] is True
    # every N > 1 line is a complete measurement record: rank 0 times the single-thread oracle sample while the other ranks wait
    cb = d["cpu_baseline"]
    assert cb is not None and cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] == "port" and cb["all_cores"] is None
    rf = d["roofline"]
    assert rf["kernel"] and rf["frac"] > 0 and rf["bound"] in ("hbm", "valu") and "valu_busy" in rf and rf["avg_launch_ms"] > 0
    assert d["config"]["n1_equivalent"] == d["config"]["per_rank_value"][0] > 0


@pytest.mark.parametrize("mixed_group", [True, False])
def test_bench_refuses_ranks_that_share_a_gpu(mixed_group):
    """Two ranks driving the same GPU without --one-device (a launcher that hands every rank LOCAL_RANK=0) must END with a message, on
    every rank, before any RCCL communicator exists -- not hang in a collective.  mixed_group = False: the fallback of a build without gloo
    (FHESI_BENCH_NO_MIXED_GROUP: the roll call goes through the rendezvous store, shard.roll_call_store, before the nccl-only group is made)."""
    import socket
    import sys
    extra_env = {} if mixed_group else {"FHESI_BENCH_NO_MIXED_GROUP": "1"}
    for attempt in range(3):          # (the port is found by bind-then-close: another process of a shared box may take it in between -- try again then)
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        procs = []
        for rank in range(2):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **extra_env)
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--batch", "8"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
        outs = [p.communicate(timeout=600) for p in procs]
        if not any("address already in use" in se.lower() for _, se in outs):
            break
    for p, (so, se) in zip(procs, outs):
        assert p.returncode != 0 and "ranks share a GPU" in se, se[-1500:]
        assert not any(ln.startswith("{") for ln in so.splitlines())


@pytest.mark.parametrize("args", [["--devices=0,0,0"], ["46", "90", "47", "5", "--devices=0,0"]])
def test_recorded_ciphertext_operations_on_a_group_of_ranks(args):
    """The recording evaluator of the C++ mirror's Ciphertext (fhe-si_amd/host/fhesi_engine.h) with EnableCiphertextGroup: arena and keys
    replicated on every rank (RCCL broadcast), key-switch levels sharded, outputs exchanged.  Loopback group on one GPU; every flow of
    tests/host/test_lazy.cpp must give the ciphertexts of the statements run at once, and switching the group mid-life must not change them."""
    subprocess.check_call(["make", "-C", HOST, "-j8"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(HOST, "test_lazy"), *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Test SUCCEEDED" in r.stdout and "FAIL" not in r.stdout and "GPU rank(s)" in r.stdout


def test_reference_regress_control_flow_on_a_group_of_ranks():
    """Regression::Regress written one object at a time (Regression.h:102-149), recorded, evaluated on a loopback group of three ranks:
    bit-identical to the explicit waves on one GPU and to the statements run at once."""
    subprocess.check_call(["make", "-C", HOST, "-j8"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(HOST, "test_regression"), "47", "5", "4", "1", "6", "--at-once", "--literal-devices=0,0,0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "recorded operations run on 3 GPU rank(s)" in r.stdout
    assert "ciphertexts of both evaluators bit-identical: yes" in r.stdout and "recorded and at-once ciphertexts bit-identical: yes" in r.stdout


@pytest.mark.parametrize("mixed_group", [True, False])
def test_bench_one_rank_runs_the_whole_multi_rank_path_over_real_rccl(mixed_group):
    """RCCL refuses two ranks on one GPU, so the 2-rank bench tests of a 1-GPU box run over gloo.  FHESI_BENCH_GROUP_AT_N1 makes ONE rank take
    every step of an N > 1 run over the real thing: the gloo + nccl process group (or, mixed_group = False, the nccl-only fallback with its roll
    call through the rendezvous store), the device roll call, the RCCL broadcast of the key matrix, barriers and stopwatches, per-rank parity
    against the oracle, digests, the multi_gpu record and the single-thread cpu_baseline of an N > 1 line."""
    import json
    import sys
    env = dict(os.environ, FHESI_BENCH_GROUP_AT_N1="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if not mixed_group:
        env["FHESI_BENCH_NO_MIXED_GROUP"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-sample", "1", "--no-bluestein-cpu",
                        "--gpu-seconds", "0", "--batch", "16"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    mg = d["config"]["multi_gpu"]
    assert mg is not None and mg["communicator_nranks"] == 1 and mg["distinct_devices"] == 1 and not mg["one_device_plumbing_mode"], mg
    assert mg["backend"].startswith("nccl") and ("gloo" in mg["backend"]) == mixed_group, mg["backend"]
    assert "unavailable" not in mg["rccl_version"] and mg["key_broadcast"]["collective_s"] >= 0 and mg["key_broadcast"]["bytes"] > 0, mg
    assert mg["communicator_init_s"] > 0, mg      # (the first device collective, timed apart from the key broadcast)
    assert mg["parity"]["all_ok"] and mg["parity"]["oracle_checked_on_every_rank"] and mg["parity"]["per_rank_ok"] == [True]
    assert d["matches_oracle"] is True and d["n_gpus"] == 1 and d["value"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["all_cores"] is None and d["config"]["n1_equivalent"] == d["config"]["per_rank_value"][0]
    if mixed_group:
        # ... and the regression workload with its waves in chunks: the asynchronous broadcasts of ShardedPool.exchange_begin over RCCL
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "regression", "--reg-dim", "4", "--reg-overlap", "2",
                            "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert d["value"] > 0 and d["config"]["exchange_overlap_chunks"] == 2 and d["config"]["waves_run_in_chunks"] >= 1, d["config"]
