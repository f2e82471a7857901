#!/usr/bin/env python3
"""Concurrency profile of one timed bench step from a rocprofv3 kernel trace: per 100 us bin, which kernels ran for how many
microseconds (summed over concurrent instances).  usage: step_timeline.py <kernel_trace.csv> [index of the add_mix launch from the end, default 7]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void awmk::", "").replace("awmk::", "")[:34]) for r in rows)
idx = [i for i, k in enumerate(ks) if "add_mix" in k[2]]
i0, i1 = idx[-back], idx[-back + 1]
t0 = ks[i0][0]
seg = ks[i0:i1]
end = max(k[1] for k in seg)
print("step wall %.1f us, %d launches" % ((end - t0) / 1e3, len(seg)))
bins = collections.defaultdict(collections.Counter)
for s, e, n in seg:
    for b in range(int((s - t0) / 1e5), int((e - t0) / 1e5) + 1):
        lo, hi = max(s, t0 + b * 100000), min(e, t0 + (b + 1) * 100000)
        if hi > lo:
            bins[b][n] += (hi - lo) / 1e3
for b in sorted(bins):
    print("%5.1f ms: " % (b / 10) + "  ".join("%s %.0f" % (n, v) for n, v in bins[b].most_common(4)))
