#!/bin/bash
# A/B of library variants on the row-level NTT workload (configs[1]): tools/ab_ntt.sh <outdir> <variants...>
out=$1; shift
mkdir -p gpurun_out/$out
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset FHESI_LIB; else export FHESI_LIB=$PWD/fhe-si_amd/csrc/variants/lib_$v.so; fi
    python bench.py --workload ntt --steps 10 --warmup 2 > gpurun_out/$out/${v}_$rep.json 2> gpurun_out/$out/${v}_$rep.err
    python - "$v" $rep gpurun_out/$out/${v}_$rep.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    r = d["roofline"]; ri = d.get("roofline_inverse") or {}
    print(sys.argv[1], sys.argv[2], d["value"], "fwd", r.get("frac"), r.get("avg_launch_ms"), "inv", ri.get("frac"), ri.get("avg_launch_ms"), d.get("matches_oracle"), d.get("round_trip_is_identity"))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
PY
  done
done | tee gpurun_out/$out/summary.txt
