#!/bin/bash
# VGPRs / scratch / occupancy per kernel of one HIP source with extra compiler flags: tools/res.sh <file> <name pattern> [flags...]
f=$1; pat=$2; shift 2
cd "$(dirname "$f")"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c "$(basename "$f")" -o /dev/null 2>&1 | PAT="$pat" python3 -c "
import re,sys,subprocess,os
cur=None; rows={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)', line)
    if m:
        cur=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip().split('(')[0]; rows[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+)', line)
    if m and cur: rows[cur][m.group(1).strip()]=int(m.group(2))
for k,v in rows.items():
    if os.environ['PAT'] in k: print(f\"{k:60s} vgpr={v.get('VGPRs')} scratch={v.get('ScratchSize [bytes/lane]')} occ={v.get('Occupancy [waves/SIMD]')}\")
"
