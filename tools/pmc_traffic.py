#!/usr/bin/env python3
"""HBM traffic per launch and kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately
as MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B?  -- rocprofv3 reports KB (1000 B)
... the guide's calibration: a coalesced 16 B/lane stream shows exactly half its bytes in FETCH_SIZE on gfx950, so
fetch is doubled; WRITE_SIZE is taken as reported (uncalibrated).  Output: JSON {kernel: {launches, fetch_bytes, write_bytes}}
with per-launch averages, keyed by the short kernel name plus grid size."""
import csv, sys, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "awmk::" not in name:
                continue
            short = name.replace("(anonymous namespace)::", "").split("awmk::")[1].split("(")[0]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, ctrs in sorted(acc.items()):
    e = {"launches": max(len(v) for v in ctrs.values())}
    if "FETCH_SIZE" in ctrs:
        v = ctrs["FETCH_SIZE"]
        e["fetch_bytes_per_launch"] = round(2 * 1024 * sum(v) / len(v))     # KiB, x2 (gfx950 calibration)
    if "WRITE_SIZE" in ctrs:
        v = ctrs["WRITE_SIZE"]
        e["write_bytes_per_launch"] = round(1024 * sum(v) / len(v))
    out[k] = e
json.dump(out, sys.stdout, indent=1)
print()
