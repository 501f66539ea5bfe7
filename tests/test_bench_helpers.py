"""CPU: the helpers behind bench.py's roofline record -- what the line says about the binding roof must follow from the committed counter files
by arithmetic a reader can redo (issue ceiling, K2' bytes, min / median over blocks), and must degrade to nulls, not guesses, when the committed
pass is not the run's kernel or batch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B


def test_issue_ceiling_and_bound():
    sq = {"valu_busy": 0.946, "valu_instr_per_wave": 1706.0, "eff_clock_ghz_median": 1.891, "source": "x"}
    f = B.valu_fields(sq, "fwd30", 33.6e6)
    # 2^14-point row: 14 stages x 8192 butterflies = 1792 wave-butterflies of 24 issue cycles on 1024 SIMDs
    assert abs(f["issue_ceiling"]["rows_per_s_at_ceiling"] - 1024 * 1.891e9 / (1792 * 24.0)) < 1.0
    assert f["valu_instr_per_butterfly"] == round(1706.0 / 224, 2) and abs(f["issue_ceiling_frac"] - 33.6e6 / (1024 * 1.891e9 / (1792 * 24.0))) < 1e-3
    assert B.bound_of(0.946) == "valu" and B.bound_of(0.73) == "hbm" and B.bound_of(None) == "hbm"
    none = B.valu_fields(None, "fwd30", 1.0)
    assert none["valu_busy"] is None and none["issue_ceiling_frac"] is None
    # primes below 2^29: 8 of 14 stages without the 6-cycle range step
    assert abs(B.BFLY_ISSUE_CYCLES["fwd29"] - (8 * 18 + 6 * 24) / 14) < 1e-9 and B.BFLY_ISSUE_CYCLES["fwd29"] < B.BFLY_ISSUE_CYCLES["fwd30"]


def test_committed_counter_files_match_the_kernels_of_the_metric_step():
    """profiles/sq_main_kernels.json and profiles/pmc_*.json are what the driver-run line quotes: they must name the kernels the library launches
    today (a renamed template parameter silently turns every field into null) and carry the fields bench.py reads."""
    sq = json.load(open(os.path.join(ROOT, "profiles", "sq_main_kernels.json")))
    assert sq["batch"] == 1024 and len(sq["kernels"]) >= 8
    names = list(sq["kernels"])
    for needle in ("ntt32_fwd_kernel3<true, 0, false, Aux32Primes, true, true, 30>", "ntt32_fwd_kernel3<false, 0, false, T32Primes, true, false, 30>", "dot32_kernel4<7, 6, 12, 3, 8, 1, 6>"):
        assert needle in names, (needle, names)
        k = sq["kernels"][needle]
        assert 0.5 < k["valu_busy"] < 1.1 and k["valu_instr_per_wave"] > 100 and 1.0 < k["eff_clock_ghz_median"] < 2.5
    assert B.offline_sq(names[0], 1024)["source"].startswith("profiles/sq_main_kernels.json") and B.offline_sq(names[0], 64) is None and B.offline_sq("no_such_kernel", 1024) is None
    t, src = B.offline_traffic("pmc_ntt_fwd.json", "ntt32_fwd_kernel3<true, 0, false, Aux32Primes, true, true, 30>", "rows_per_launch", 270336)
    assert t and 2.0e10 < t < 3.0e10 and "pmc_ntt_fwd.json" in src
    assert B.offline_traffic("pmc_ntt_fwd.json", "ntt32_fwd_kernel3<true, 0, false, Aux32Primes, true, true, 30>", "rows_per_launch", 1000) == (None, None)


class _FakeCtx:
    def __init__(self):
        self.t = {"dot": (0, 0.0, 0.0)}

    def prof_read(self, k):
        return self.t[k]


def test_kernel_clock_block_statistics():
    ctx = _FakeCtx()
    clk = B.KernelClock(ctx, ("dot",))
    for launches, ms in ((20, 100.0), (40, 196.0), (60, 300.0)):      # blocks of 20 launches: 5.0, 4.8, 5.2 ms per launch
        ctx.t["dot"] = (launches, float(launches), ms)
        clk.mark()
    assert clk.stats("dot") == (4.8, 5.0)
    clk2 = B.KernelClock(ctx, ("dot",))
    assert clk2.stats("dot") == (None, None)
