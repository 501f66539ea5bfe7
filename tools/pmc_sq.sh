#!/bin/bash
# SQ counters of the dominant kernel, one --pmc pass per group (run ON THE GPU BOX through gpurun).  Prints per-launch averages,
# the effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration) and VALUBusy.  Output: gpurun_out/pmc_sq/summary.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_sq
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
K=${1:-"ntt_fwd_tile<14, true, 0, false>"}
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$O/g$i" -o pmc -- python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-sample 0 --batch 1024 --no-surface --gpu-seconds 0 > /dev/null 2> "$O/g$i.log"
  python3 - "$O/g$i" "$K" <<'PY' | tee -a "$O/summary.txt"
import csv, glob, sys, collections
d, k = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not cc: print("no counter file in", d); sys.exit()
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        if k in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(list); clk = []
for r in csv.DictReader(open(cc[0])):
    if k in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur: clk.append(float(r["Counter_Value"]) / 8 / dur[r["Dispatch_Id"]])
for c, v in sorted(acc.items()): print(f"{c:24s} launches={len(v)} avg_per_launch={sum(v)/len(v):.6g}")
if dur: print(f"kernel duration under the counter pass: avg {sum(dur.values())/len(dur)/1e6:.3f} ms")
if clk: print("effective shader clock per launch (GHz):", " ".join(f"{c:.3f}" for c in clk))
PY
done
find "$O" -name '*.db' -delete; find "$O" -name '*.csv' -delete
