#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 kernel-trace CSV (gaps = host/launch time)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-44:]) for r in rows)
idx = [i for i, k in enumerate(ks) if 'add_mix' in k[2]]
i0 = idx[-1]
t0 = ks[i0][0]
prev = t0
busy = 0
for s, e, n in ks[i0:]:
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}  gap {(s-prev)/1e3:7.1f}  {n}")
    busy += e - s
    prev = e
print(f"total {(prev-t0)/1e3:.1f} us, busy {busy/1e3:.1f} us, idle {(prev-t0-busy)/1e3:.1f} us")
