#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of the clip batch: per kernel launches / total time per clip, GPU busy fraction
(union of kernel intervals over the wall time of the last batch).  usage: clip_trace_summary.py <kernel_trace.csv> <n_clips>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n_clips = int(sys.argv[2])
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# the last batch = after the last long gap (> 2 ms)
cut = 0
for i in range(1, len(ks)):
    if ks[i][0] - max(k[1] for k in ks[max(0, i - 50):i]) > 2_000_000:
        cut = i
ks = ks[cut:]
t0, t1 = ks[0][0], max(k[1] for k in ks)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in ks:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in ks:
    short = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("awmk::", "")[:48]
    agg[short][0] += 1
    agg[short][1] += e - s
print(f"last batch: {len(ks)} launches = {len(ks)/n_clips:.1f} per clip, wall {(t1-t0)/1e6:.3f} ms = {(t1-t0)/1e6/n_clips:.3f} ms per clip, "
      f"GPU busy (union) {busy/1e6:.3f} ms = {100*busy/(t1-t0):.0f} %, sum of kernel durations {sum(v[1] for v in agg.values())/1e6:.3f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:50s} {c/n_clips:6.2f} launches/clip  {t/1e3/n_clips:8.2f} us/clip  avg {t/1e3/c:8.2f} us")
