"""HIP API time inside each roctx range of tools/gpu_first_call_trace.py: python tools/first_call_summary.py <dir with the rocprofv3 CSVs>"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
def rows(pattern):
    for p in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(p, newline="") as f:
            yield from csv.DictReader(f)
ranges = [(r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows("*marker_api_trace.csv")]
api = [(r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows("*hip_api_trace.csv")]
ker = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows("*kernel_trace.csv")]
for name, a, b in sorted(ranges, key=lambda r: r[1]):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f, s, e in api:
        if a <= s and e <= b:
            tot[f][0] += 1
            tot[f][1] += (e - s) / 1e6
    kt = sum((e - s) / 1e6 for _, s, e in ker if a <= s and e <= b)
    print("%s: %.2f ms wall, kernels %.2f ms (sum of durations), HIP API calls %.2f ms" % (name, (b - a) / 1e6, kt, sum(v[1] for v in tot.values())))
    for f, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:8]:
        print("    %-34s %6d calls %9.3f ms" % (f, c, ms))
    slow = sorted(((e - s_) / 1e6, f, (s_ - a) / 1e6) for f, s_, e in api if a <= s_ and e <= b)[-3:]
    print("    slowest single calls:", ["%s %.2f ms at +%.1f ms" % (f, d_, at) for d_, f, at in reversed(slow)])
