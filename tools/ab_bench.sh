#!/bin/bash
# A/B of library variants ON THE GPU BOX (through gpurun): tools/ab_bench.sh <outdir> <variant names...>  ("base" = the regular build).
# Each variant runs the metric bench twice (interleaved, so box drift affects all alike); prints value and the per-kernel breakdown.
out=$1; shift
mkdir -p gpurun_out/$out
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset FHESI_LIB; else export FHESI_LIB=$PWD/fhe-si_amd/csrc/variants/lib_$v.so; fi
    python bench.py --cpu-sample 1 --no-bluestein-cpu --no-surface --gpu-seconds 2 > gpurun_out/$out/${v}_$rep.json 2> gpurun_out/$out/${v}_$rep.err
    python - "$v" $rep gpurun_out/$out/${v}_$rep.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    print(sys.argv[1], sys.argv[2], d["value"], d["matches_oracle"], d["kernel_ms_per_step"])
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
PY
  done
done | tee gpurun_out/$out/summary.txt
