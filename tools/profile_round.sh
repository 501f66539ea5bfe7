#!/bin/bash
# Profiles of one round, run ON THE GPU BOX (via gpurun): kernel-trace statistics of the bench workloads and two separate PMC
# passes (FETCH_SIZE, WRITE_SIZE) of the metric workload.  Results land in gpurun_out/prof_$1/ and are copied into profiles/ by hand.
set -u
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() {   # name, then bench.py arguments
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o "$name" -- python3 "$R/bench.py" "$@" > "$O/${name}_bench.json" 2> "$O/${name}.log"
  cp "$O/$name"/*/"${name}_kernel_stats.csv" "$O/${name}_kernel_stats.csv" 2>/dev/null || cp "$O/$name/${name}_kernel_stats.csv" "$O/${name}_kernel_stats.csv" 2>/dev/null
  # min / median / mean per kernel from the trace (the stats file has no median, and its mean includes the cold first launch)
  python3 - "$O/$name" "$O/${name}_kernel_times.csv" <<'PY'
import csv, glob, sys, collections
tr = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if tr:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    with open(sys.argv[2], "w") as f:
        f.write("kernel,calls,min_ms,median_ms,mean_ms,max_ms,total_ms\n")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            v.sort()
            f.write('"%s",%d,%.4f,%.4f,%.4f,%.4f,%.3f\n' % (k.replace('"', "'"), len(v), v[0], v[(len(v) - 1) // 2], sum(v) / len(v), v[-1], sum(v)))
PY
  tail -1 "$O/${name}_bench.json" | cut -c1-400
}
if [ -z "${PROFILE_ONLY_PMC:-}" ]; then
run metric --steps 5 --warmup 2 --no-bluestein-cpu --no-surface --gpu-seconds 0
run stress --workload stress --steps 3 --warmup 1 --cpu-sample 0 --gpu-seconds 0
run regression --workload regression --steps 3 --warmup 1
run regression_ref --workload regression --reg-ring reference --steps 5 --warmup 2
run regression_ref_p32603 --workload regression --reg-ring reference --reg-p 32603 --steps 5 --warmup 2
run ntt --workload ntt --steps 10 --warmup 2
run refring --workload refring --steps 5 --warmup 2 --cpu-sample 0 --gpu-seconds 0
run refring_p65267 --workload refring --ref-p 65267 --steps 3 --warmup 1 --cpu-sample 0 --gpu-seconds 0 --batch 512
run refring_p65543 --workload refring --ref-p 65543 --steps 3 --warmup 1 --cpu-sample 0 --gpu-seconds 0 --batch 256
fi
for c in FETCH_SIZE WRITE_SIZE; do        # one counter per pass (combining them has hung the profiler on this pool)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_$c" -o pmc -- python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-sample 0 --batch 1024 --no-surface --gpu-seconds 0 > /dev/null 2> "$O/pmc_$c.log"
done
F=$(find "$O/pmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
W=$(find "$O/pmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
python3 "$R/tools/pmc_traffic.py" "$F" "$W" "dot32_kernel4<7, 6, 12, 3, 8, 1, 6>" "$O/pmc_dot_aux.json" ciphertexts_per_launch=1024
python3 "$R/tools/pmc_traffic.py" "$F" "$W" "ntt32_fwd_kernel3<true, 0, false, Aux32Primes, true, true, 30>" "$O/pmc_ntt_fwd.json" rows_per_launch=270336
python3 "$R/tools/pmc_traffic.py" "$F" "$W" "ntt32_fwd_kernel3<false, 0, false, T32Primes, true, false, 30>" "$O/pmc_t32_fwd.json" rows_per_launch=143360
python3 "$R/tools/pmc_traffic.py" "$F" "$W" "ntt32_inv_kernel3<false, true, T32Primes, 30>" "$O/pmc_t32_inv.json" rows_per_launch=107520
python3 "$R/tools/pmc_traffic.py" "$F" "$W" "rns32_reduce_kernel<8, 0, true>" "$O/pmc_t32_rns.json" ciphertexts_per_launch=1024
python3 "$R/tools/pmc_traffic.py" "$F" "$W" "crt32_scale_kernel<512, false, 28, 38, 0>" "$O/pmc_t32_crt.json" ciphertexts_per_launch=1024
python3 "$R/tools/pmc_traffic.py" "$F" "$W" "ntt32_inv_kernel3<true, false, Aux32Primes, 30>" "$O/pmc_ntt_inv.json" rows_per_launch=57344
python3 "$R/tools/pmc_traffic.py" "$F" "$W" "ks_recombine_centred_kernel" "$O/pmc_recombine.json" ciphertexts_per_launch=1024
find "$O" -name '*.db' -delete; find "$O" -name '*agent_info.csv' -delete; find "$O" -name '*kernel_trace.csv' -delete; find "$O" -name '*counter_collection.csv' -delete
ls -la "$O"
