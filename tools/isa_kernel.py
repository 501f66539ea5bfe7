#!/usr/bin/env python3
"""Instruction histogram (and optionally the text) of one kernel in a `hipcc -S --cuda-device-only` listing:
   tools/isa_kernel.py listing.s <mangled-name prefix> [--dump] [--grep PATTERN]"""
import collections, re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(' + re.escape(name) + r'\S*):', s, re.M)
start = m.start()
end = s.index('s_endpgm', start)
lines = s[start:end].splitlines()
ins = [l for l in lines if l.strip() and l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
print("instructions:", len(ins))
cnt = collections.Counter(l.split()[0] for l in ins)
print(" ".join(f"{k}:{v}" for k, v in cnt.most_common(60)))
if "--dump" in sys.argv:
    print("\n".join(lines))
if "--grep" in sys.argv:
    pat = sys.argv[sys.argv.index("--grep") + 1]
    for i, l in enumerate(lines):
        if re.search(pat, l): print(i, l)
