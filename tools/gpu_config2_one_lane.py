import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""BASELINE configs[2] with its kernels ONE AFTER THE OTHER (one chunk lane, the plain decode after the speed part instead of beside it):
`get --detect-speed` of 60 min stereo 48 kHz replayed at 1.02, repeated -- for `rocprofv3 --kernel-trace --stats` (stand-alone kernel
durations -> profiles/rNN/rocprofv3_kernel_stats_config2_one_lane.csv; the same pass as bench.py's detect_speed_config.kernels_one_lane)
and for the separate `--pmc` passes."""
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import audiowmark_amd as awm

PAY = "0123456789abcdef0011223344556677"
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rate = 48000
ctx = awm.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(4711)
x = torch.rand((60 * 60 * rate, 2), generator=g, device="cuda") * 2 - 1
w = ctx.add_watermark(None, PAY, x, sample_rate=rate)
del x
fast = ctx.resample_ratio(w, 1 / 1.02, rate=rate)
del w
awm.set_speed_params(detect_speed=True)
awm.lib.awm_ctx_set_chunk_lanes(ctx._h, 1)
awm.lib.awm_debug_set_speed_overlap(0)
for _ in range(calls):
    pats = ctx.get_watermark(None, ctx.resample(fast, rate, 44100))
torch.cuda.synchronize()
print("patterns with the payload:", sum(p["bits"] == PAY for p in pats), "speeds", sorted({round(p["speed"], 5) for p in pats if p["bits"] == PAY}))
