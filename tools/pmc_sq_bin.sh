#!/bin/bash
# SQ counters of one kernel of a standalone program (run ON THE GPU BOX through gpurun), one --pmc pass per counter group.
#   tools/pmc_sq_bin.sh <kernel name substring> <tag> <program> [args...]        output: gpurun_out/pmc_sq_<tag>/summary.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; TAG=$2; shift 2
PROG=$(realpath "$1"); shift
# the profiler's preloaded library initialises the GPU before the program starts: anything that execs another program behind `--` (a script
# with an interpreter line, a launcher) is the forbidden exec-after-GPU-init -- only ELF binaries are profiled here
if [ "$(head -c 4 "$PROG" | od -An -c | tr -d ' ')" != '177ELF' ]; then echo "pmc_sq_bin.sh: $PROG is not an ELF binary (scripts and launchers are refused)"; exit 2; fi
O=$R/gpurun_out/pmc_sq_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$O/g$i" -o pmc -- "$PROG" "$@" > "$O/g$i.out" 2> "$O/g$i.log"
  rc=$?
  if [ $rc -eq 124 ]; then echo "pass $i ($grp): TIMED OUT after 180 s" | tee -a "$O/summary.txt"; elif [ $rc -ne 0 ]; then echo "pass $i ($grp): exit code $rc" | tee -a "$O/summary.txt"; fi
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
for c, v in sorted(acc.items()): print(f"{c:28s} launches={len(v)} avg_per_launch={sum(v)/len(v):.6g}")
if dur: print(f"kernel duration under the counter pass: avg {sum(dur.values())/len(dur)/1e6:.3f} ms")
if clk: print("effective shader clock per launch (GHz):", " ".join(f"{c:.3f}" for c in clk))
PY
done
find "$O" -name '*.db' -delete; find "$O" -name '*.csv' -delete
