#!/bin/bash
# SQ counters of SEVERAL kernels of the metric step from ONE set of --pmc passes (run ON THE GPU BOX through gpurun):
#   tools/pmc_sq_multi.sh "<kernel substring>" "<kernel substring>" ...
# Four groups of four SQ counters, one rocprofv3 pass each, every pass with --kernel-trace for the durations (--pmc is never combined with
# -s/-r or the hip/hsa/memory-copy trace domains, which gpurun refuses; tools/profile_round.sh takes ONE counter per pass -- the slower,
# safer form -- when a group of four does not come back).  A pass that times out or fails is reported as such in summary.txt.  Per kernel: per-launch averages, the kernel's
# duration under the pass, the effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and
# VALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8).  Output: gpurun_out/pmc_sq_multi/summary.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_sq_multi
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$O/g$i" -o pmc -- python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-sample 0 --batch 1024 --no-surface --gpu-seconds 0 > /dev/null 2> "$O/g$i.log"
  rc=$?
  if [ $rc -eq 124 ]; then echo "pass $i ($grp): TIMED OUT after 240 s" | tee -a "$O/failed_passes.txt"; elif [ $rc -ne 0 ]; then echo "pass $i ($grp): rocprofv3 exit code $rc" | tee -a "$O/failed_passes.txt"; fi
done
{ [ -f "$O/failed_passes.txt" ] && cat "$O/failed_passes.txt"; python3 - "$O" "$@" <<'PY'
import csv, glob, sys, collections, json, re
O, kernels = sys.argv[1], sys.argv[2:]
def norm(name):          # "void kernel<args>(params)" -> "kernel<args>" (what fhesi_prof_kernel_name returns)
    name = name[5:] if name.startswith("void ") else name
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0: return name[:i]
    return name
out_json = {"command": "rocprofv3 --kernel-trace --pmc <4 SQ counters per pass> -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --batch 1024 --no-surface --gpu-seconds 0 (tools/pmc_sq_multi.sh)",
            "batch": 1024, "kernels": {}}
for k in kernels:
    print("==", k)
    tot = {}
    full, clks, durs = None, [], []
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
                full = full or norm(r["Kernel_Name"])
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur: clk.append(float(r["Counter_Value"]) / 8 / dur[r["Dispatch_Id"]])
        for c, v in sorted(acc.items()):
            tot[c] = sum(v) / len(v)
            print(f"  {c:24s} launches={len(v)} avg_per_launch={tot[c]:.6g}")
        if dur: print(f"  kernel duration under pass {g}: avg {sum(dur.values())/len(dur)/1e6:.3f} ms"); durs += [v / 1e6 for v in dur.values()]
        if clk: print("  effective shader clock per launch (GHz):", " ".join(f"{c:.3f}" for c in clk)); clks += clk
    if "SQ_ACTIVE_INST_VALU" in tot and "GRBM_GUI_ACTIVE" in tot:
        busy = tot['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / (tot['GRBM_GUI_ACTIVE'] / 8)
        ipw = tot.get('SQ_INSTS_VALU', 0) / max(tot.get('SQ_WAVES', 1), 1)
        print(f"  VALU busy = {busy:.3f};  VALU instructions per wave = {ipw:.0f}")
        if full:
            sc, sd = sorted(clks), sorted(durs)
            out_json["kernels"][full] = {"valu_busy": round(busy, 3), "valu_instr_per_wave": round(ipw, 1), "waves_per_launch": tot.get("SQ_WAVES"),
                                         "valu_wave_instr_per_launch": tot.get("SQ_INSTS_VALU"), "salu_wave_instr_per_launch": tot.get("SQ_INSTS_SALU"),
                                         "lds_wave_instr_per_launch": tot.get("SQ_INSTS_LDS"), "lds_bank_conflict_cycles": tot.get("SQ_LDS_BANK_CONFLICT"),
                                         "wait_inst_lds_wave_cycles": tot.get("SQ_WAIT_INST_LDS"), "vmem_rd_wave_instr": tot.get("SQ_INSTS_VMEM_RD"), "vmem_wr_wave_instr": tot.get("SQ_INSTS_VMEM_WR"),
                                         "eff_clock_ghz": [round(c, 3) for c in clks], "eff_clock_ghz_median": round(sc[(len(sc) - 1) // 2], 3) if sc else None,
                                         "duration_ms_min": round(sd[0], 3) if sd else None, "duration_ms_median": round(sd[(len(sd) - 1) // 2], 3) if sd else None}
json.dump(out_json, open(O + "/sq_main_kernels.json", "w"), indent=1)
PY
} | tee "$O/summary.txt"
find "$O" -name '*.db' -delete; find "$O" -name '*.csv' -delete
