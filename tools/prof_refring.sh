#!/bin/bash
# kernel times of `bench.py --workload refring` (the reference drivers' ring at the metric's size) under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refring_prof; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o refring -- python3 bench.py --workload refring --steps 5 --gpu-seconds 0 --cpu-sample 0 "$@" > $O/bench.json 2> $O/log.txt
head -16 $O/refring_kernel_stats.csv | cut -c1-150
