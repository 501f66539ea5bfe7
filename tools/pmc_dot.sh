#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_dot
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum" "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$O/g$i" -o pmc -- python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$O/g$i.log"
  python3 - "$O/g$i" "dot_aux_kernel" <<'PY' | tee -a "$O/summary.txt"
import csv, glob, sys, collections
d, k = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not cc: print("no counter file", d); sys.exit()
acc = collections.defaultdict(list)
for r in csv.DictReader(open(cc[0])):
    if k in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()): print(f"{c:24s} launches={len(v)} avg_per_launch={sum(v)/len(v):.6g}")
PY
  tail -2 "$O/g$i.log" | cut -c1-160 >> "$O/log_tail.txt"
done
find "$O" -name '*.db' -delete; find "$O" -name '*.csv' -delete
