cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
: # (set FHESI_LIB to a variant library here)
mkdir -p gpurun_out/small_t
timeout 600 python3 tools/bench_small.py > gpurun_out/small_t/plain.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/small_t/prof -o small -- python3 tools/bench_small.py > gpurun_out/small_t/prof.txt 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/small_t/prof/**/*kernel_trace.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
d=collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name']].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items(), key=lambda kv:-len(kv[1]))[:10]:
    v2=sorted(v)
    print(f"{k[:90]:90s} n={len(v):6d} min={v2[0]/1e3:8.1f} p10={v2[len(v2)//10]/1e3:8.1f} med={v2[len(v2)//2]/1e3:8.1f} us")
PY
head -3 gpurun_out/small_t/plain.txt
