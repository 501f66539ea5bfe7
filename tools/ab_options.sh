#!/bin/bash
# A/B of library OPTIONS on the GPU box (one library, same box, interleaved): tools/ab_options.sh <outdir> "<name>:<bench args>" ...
# e.g. tools/ab_options.sh ab_grp "new:" "limbrows:--option parts_words=0" "b29:--option tensor_bits=29"
# a spec whose arguments start with @name runs the library variant fhe-si_amd/csrc/variants/lib_name.so (csrc/build_variant.sh)
out=$1; shift
mkdir -p gpurun_out/$out
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; args=${spec#*:}
    unset FHESI_LIB
    if [ "${args:0:1}" = "@" ]; then v=${args%% *}; v=${v#@}; export FHESI_LIB=$PWD/fhe-si_amd/csrc/variants/lib_$v.so; if [ "$args" = "@$v" ]; then args=""; else args=${args#* }; fi; fi
    python bench.py --cpu-sample 1 --no-bluestein-cpu --no-surface --gpu-seconds 2 $args > gpurun_out/$out/${name}_$rep.json 2> gpurun_out/$out/${name}_$rep.err
    python - "$name" $rep gpurun_out/$out/${name}_$rep.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    print(sys.argv[1], sys.argv[2], d["value"], d["matches_oracle"], d["kernel_ms_per_step"], "digits", d["roofline_ntt"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
PY
  done
done | tee gpurun_out/$out/summary.txt
