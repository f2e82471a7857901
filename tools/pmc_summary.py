#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per awmk kernel (per launch)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "awmk::" not in name:
                continue
            short = name.replace("(anonymous namespace)::", "").split("awmk::")[1].split("(")[0]
            acc[(short, row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (k, grid), ctrs in sorted(acc.items()):
    print(f"{k} grid={grid} launches={len(next(iter(ctrs.values())))}")
    for c, v in sorted(ctrs.items()):
        print(f"    {c:28s} {sum(v)/len(v):16.1f}")
