#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: one counter per pass, FETCH_SIZE doubled on gfx950 (it reports half
of the bytes of wide coalesced streaming reads), WRITE_SIZE as counted (calibrated in round 1 on a kernel with a known byte count:
fetch x1.97, write x0.98).  Writes the JSON bench.py reads for `roofline.traffic`.

  pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv "kernel substring" out.json key=value ...
"""
import csv, json, sys


def mean_counter(path, needle, counter):
    vals, name = [], None
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
            name = r["Kernel_Name"]
    return (sum(vals) / len(vals) if vals else None), len(vals), name


def short_name(full):
    """'void k<a, b>(args)' -> 'k<a, b>' (what fhesi_prof_kernel_name returns)"""
    s = full[5:] if full.startswith("void ") else full
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s[:i]
    return s


if __name__ == "__main__":
    fpath, wpath, needle, out = sys.argv[1:5]
    extra = dict(kv.split("=", 1) for kv in sys.argv[5:])
    f, nf, name = mean_counter(fpath, needle, "FETCH_SIZE")
    w, nw, _ = mean_counter(wpath, needle, "WRITE_SIZE")
    if f is None or w is None:
        sys.exit(f"no launches of '{needle}' in the counter files")
    rec = {"kernel": short_name(name), "launches_averaged": [nf, nw],
           "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
           "correction": "gfx950: FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE x 1; round-1 calibration on a kernel with a known byte count: x1.97 / x0.98",
           "hbm_bytes_per_launch": int(f * 1024 * 2 + w * 1024)}
    for k, v in extra.items():
        rec[k] = int(v) if v.lstrip("-").isdigit() else v
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))
