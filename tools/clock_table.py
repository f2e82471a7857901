#!/usr/bin/env python3
"""effective shader clock per kernel: GRBM_GUI_ACTIVE (cycles the graphics engine was busy during a dispatch, from its own --pmc pass)
divided by the kernel's stand-alone duration (rocprofv3 --kernel-trace --stats of the one-lane pass) -- the chip clocks to its power
budget (MI355X_MICROARCH.md, DVFS give-back), so a dense kernel runs below the 2.4 GHz nameplate.
usage: clock_table.py <kernel_stats_one_lane.csv> <pmc_GRBM_GUI_ACTIVE.txt>"""
import csv, re, sys
dur = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "awmk::" not in n:
        continue
    short = n.replace("(anonymous namespace)::", "").split("awmk::")[1].split("(")[0][:40]
    dur[short] = float(r["AverageNs"])
XCDS = 8      # the counter is summed over the 8 XCDs of the chip
print("# effective clock = GRBM_GUI_ACTIVE / 8 XCDs / stand-alone duration; the counter also ticks while a dispatch starts and drains")
print("# (a few us): meaningful for the kernels that run for hundreds of microseconds, too high for the short ones")
print("%-42s %10s %16s %8s" % ("kernel", "avg us", "GRBM_GUI_ACTIVE", "GHz"))
for line in open(sys.argv[2]):
    m = re.match(r"^(\S.*?)\s+(\d+)\s+(\d+)\s*$", line)
    if not m or m.group(1).startswith("kernel"):
        continue
    k, cycles = m.group(1).strip(), float(m.group(3))
    if k in dur and dur[k] > 0:
        print("%-42s %10.1f %16.0f %8.2f" % (k, dur[k] / 1e3, cycles, cycles / XCDS / dur[k]))
