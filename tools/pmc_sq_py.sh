#!/bin/bash
# SQ counters of SEVERAL kernels of the metric step from ONE set of --pmc passes (run ON THE GPU BOX through gpurun):
#   PMC_ARGS="tools/bench_ntt.py 13 14" tools/pmc_sq_py.sh "<kernel substring>" ...   (any python script of the repo; paths relative to the repo root)
# Four groups of four SQ counters, one rocprofv3 pass each, every pass with --kernel-trace for the durations (--pmc is never combined with
# -s/-r or the hip/hsa/memory-copy trace domains, which gpurun refuses; tools/profile_round.sh takes ONE counter per pass -- the slower,
# safer form -- when a group of four does not come back).  A pass that times out or fails is reported as such in summary.txt.  Per kernel: per-launch averages, the kernel's
# duration under the pass, the effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and
# VALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8).  Output: gpurun_out/pmc_sq_multi/summary.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_sq_py
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; PMC_ARGS=${PMC_ARGS:?set PMC_ARGS to the python script and its arguments}; case "$PMC_ARGS" in /*) ;; *) PMC_ARGS="$R/$PMC_ARGS";; esac; cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$O/g$i" -o pmc -- python3 $PMC_ARGS > /dev/null 2> "$O/g$i.log"
  rc=$?
  if [ $rc -eq 124 ]; then echo "pass $i ($grp): TIMED OUT after 240 s" | tee -a "$O/failed_passes.txt"; elif [ $rc -ne 0 ]; then echo "pass $i ($grp): rocprofv3 exit code $rc" | tee -a "$O/failed_passes.txt"; fi
done
{ [ -f "$O/failed_passes.txt" ] && cat "$O/failed_passes.txt"; python3 - "$O" "$@" <<'PY'
import csv, glob, sys, collections
O, kernels = sys.argv[1], sys.argv[2:]
for k in kernels:
    print("==", k)
    tot = {}
    for g in range(1, 5):
        d = f"{O}/g{g}"
        cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
        if not cc: print("  no counter file in", d); continue
        dur = {}
        if kt:
            for r in csv.DictReader(open(kt[0])):
                if k in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        acc = collections.defaultdict(list); clk = []
        for r in csv.DictReader(open(cc[0])):
            if k in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur: clk.append(float(r["Counter_Value"]) / 8 / dur[r["Dispatch_Id"]])
        for c, v in sorted(acc.items()):
            tot[c] = sum(v) / len(v)
            print(f"  {c:24s} launches={len(v)} avg_per_launch={tot[c]:.6g}")
        if dur: print(f"  kernel duration under pass {g}: avg {sum(dur.values())/len(dur)/1e6:.3f} ms")
        if clk: print("  effective shader clock per launch (GHz):", " ".join(f"{c:.3f}" for c in clk))
    if "SQ_ACTIVE_INST_VALU" in tot and "GRBM_GUI_ACTIVE" in tot:
        print(f"  VALU busy = {tot['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / (tot['GRBM_GUI_ACTIVE'] / 8):.3f};  VALU instructions per wave = {tot.get('SQ_INSTS_VALU', 0) / max(tot.get('SQ_WAVES', 1), 1):.0f}")
PY
} | tee "$O/summary.txt"
find "$O" -name '*.db' -delete; find "$O" -name '*.csv' -delete
