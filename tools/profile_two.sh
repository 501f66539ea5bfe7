#!/bin/bash
# The `stress` and `ntt` lines of tools/profile_round.sh alone (run ON THE GPU BOX through gpurun): kernel-trace statistics of the two workloads whose
# kernels changed after the last full profile round; results in gpurun_out/prof_r05d/, copied into profiles/r05_d_* by hand.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05d
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o "$name" -- python3 "$R/bench.py" "$@" > "$O/${name}_bench.json" 2> "$O/${name}.log"
  cp "$O/$name"/*/"${name}_kernel_stats.csv" "$O/${name}_kernel_stats.csv" 2>/dev/null || cp "$O/$name/${name}_kernel_stats.csv" "$O/${name}_kernel_stats.csv" 2>/dev/null
  tail -1 "$O/${name}_bench.json" | cut -c1-300
  find "$O/$name" -name '*.db' -delete; find "$O/$name" -name '*trace.csv' -delete
}
run stress --workload stress --steps 3 --warmup 1 --cpu-sample 0 --gpu-seconds 0
run ntt --workload ntt --steps 10 --warmup 2
