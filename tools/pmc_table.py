#!/usr/bin/env python3
"""per-kernel averages of every counter in rocprofv3 --pmc counter_collection CSVs (awmk kernels): pmc_table.py <csv> ..."""
import collections
import csv
import sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "awmk::" not in n:
            continue
        short = n.replace("(anonymous namespace)::", "").split("awmk::")[1].split("(")[0][:40]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
print("%-42s %6s " % ("kernel", "calls") + " ".join("%22s" % n for n in names))
for k, ctr in sorted(acc.items(), key=lambda kv: -sum(kv[1].get(names[0], [0]))):
    calls = max(len(v) for v in ctr.values())
    print("%-42s %6d " % (k, calls) + " ".join("%22.0f" % (sum(ctr[n]) / len(ctr[n])) if n in ctr else "%22s" % "-" for n in names))
