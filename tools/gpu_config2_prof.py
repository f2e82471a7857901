import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""BASELINE configs[2] (60 min stereo 48 kHz, add at 48 kHz, replay 2 % fast, get --detect-speed) as bench.py measures it, alone in a
process: for `rocprofv3 --kernel-trace --stats` (-> profiles/rNN/rocprofv3_kernel_stats_config2_detect_speed.csv) and host timing."""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import audiowmark_amd as awm
import bench
ctx = awm.Context(0)
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
print(json.dumps(bench.detect_speed_config(torch, awm, ctx, None, bench.PAYLOAD, minutes)))
