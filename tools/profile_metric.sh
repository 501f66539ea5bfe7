#!/bin/bash
# The `metric` line of tools/profile_round.sh alone (run ON THE GPU BOX through gpurun): rocprofv3 --kernel-trace --stats of the default workload at the
# current library; results in gpurun_out/prof_metric/, copied into profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_metric
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/metric" -o metric -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-bluestein-cpu --no-surface --gpu-seconds 0 > "$O/metric_bench.json" 2> "$O/metric.log"
cp "$O/metric"/*/metric_kernel_stats.csv "$O/metric_kernel_stats.csv" 2>/dev/null || cp "$O/metric/metric_kernel_stats.csv" "$O/metric_kernel_stats.csv" 2>/dev/null
find "$O/metric" -name '*.db' -delete; find "$O/metric" -name '*trace.csv' -delete
tail -1 "$O/metric_bench.json" | cut -c1-200
head -10 "$O/metric_kernel_stats.csv" | cut -c1-150
