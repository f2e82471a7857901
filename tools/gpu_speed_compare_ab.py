import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""A / B of K14 (speed_compare_kernel): all 11 relative speeds of a centre in one thread (the centre's matrix gathered once) against
groups of six (round 5), inside BASELINE configs[2] (`get --detect-speed` of 60 min stereo 48 kHz replayed 2 % fast): bench.py's
detect_speed_config with the forms taking turns; patterns and speeds must not change."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = sys.argv[:1]
import torch
import bench
import audiowmark_amd as awm
ctx = awm.Context(0)
first = None
for wide in (2, 0, 2, 0, 1, 0):
    awm.lib.awm_debug_set_speed_compare_wide(1 if wide == 1 else 0)
    awm.lib.awm_debug_set_speed_compare_fold(1 if wide == 2 else 0)
    r = bench.detect_speed_config(torch, awm, ctx, None, bench.PAYLOAD, 60.0, 4)
    k14 = [k for k in (r.get("kernels_one_lane") or []) if "speed_compare" in k.get("scope", "")]
    sig = (r.get("patterns"), r.get("payload_matches"), r.get("detected_speeds"))
    if first is None:
        first = sig
    print("form %d (0 groups of six, 1 all speeds in one thread, 2 groups of six folded onto one XCD): get --detect-speed %.3f ms, first call %s, K14 %s, same results as the first run: %s"
          % (wide, r.get("get_detect_speed_ms", -1), (r.get("first_call_ms") or {}).get("get_detect_speed"), json.dumps(k14)[:300], sig == first), flush=True)
awm.lib.awm_debug_set_speed_compare_wide(0)
awm.lib.awm_debug_set_speed_compare_fold(1)
