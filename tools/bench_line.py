#!/usr/bin/env python3
"""Pretty-print the interesting fields of bench.py's JSON line read from stdin (dev helper)."""
import json, sys
for line in sys.stdin:
    if '"metric"' in line:
        d = json.loads(line)
        print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], d["roofline"]["row_ntts_per_s"], d["kernel_ms_per_step"])
