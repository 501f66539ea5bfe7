#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch for kernels matching a substring."""
import csv, collections, glob, sys
def summarise(path, needle):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: (len(x), sum(x) / len(x)) for c, x in v.items()})
if __name__ == "__main__":
    for f in glob.glob(sys.argv[1]):
        summarise(f, sys.argv[2] if len(sys.argv) > 2 else "ntt_fwd_tile<14, true, 0, false>")
