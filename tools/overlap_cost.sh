mkdir -p gpurun_out/r06_w
for c in 1 2 4; do FHESI_BENCH_GROUP_AT_N1=1 python bench.py --workload regression --reg-overlap $c --steps 5 --warmup 2 > gpurun_out/r06_w/reg_overlap_$c.json 2>/dev/null; done
for c in 1 4; do FHESI_BENCH_GROUP_AT_N1=1 python bench.py --workload regression --reg-ring reference --reg-overlap $c --steps 5 --warmup 2 > gpurun_out/r06_w/regref_overlap_$c.json 2>/dev/null; done
python bench.py --workload regression --steps 5 --warmup 2 > gpurun_out/r06_w/reg_nogroup.json 2>/dev/null
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_w/*.json")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("waves_run_in_chunks"), d["config"].get("waves"), d["config"].get("sharding"))
PY
