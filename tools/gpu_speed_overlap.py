import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""`get --detect-speed` of BASELINE configs[2] with the plain decode beside the speed part (default) and after it
(awm_debug_set_speed_overlap), alternating in one process; pattern lists must be identical.

  python tools/gpu_speed_overlap.py [minutes = 60]        ->  copy the output to profiles/rNN/speed_overlap.txt"""
import ctypes, json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import audiowmark_amd as awm
import bench

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = awm.Context(0)
lib = awm.lib
lib.awm_debug_set_speed_overlap.argtypes = [ctypes.c_int]
lib.awm_debug_set_speed_overlap.restype = None
rate, speed = 48000, 1.02
g = torch.Generator(device="cuda"); g.manual_seed(4711)
x = torch.rand((int(minutes * 60 * rate), 2), generator=g, device="cuda", dtype=torch.float32) * 2 - 1
w = ctx.add_watermark(None, bench.PAYLOAD, x, sample_rate=rate)
del x
fast = ctx.resample_ratio(w, 1 / speed, rate=rate)
del w
awm.set_speed_params(detect_speed=True)
ref = None
for rep in range(3):
    for mode in (1, 0):
        lib.awm_debug_set_speed_overlap(mode)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pats = ctx.get_watermark(None, ctx.resample(fast, rate, 44100))
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        if ref is None:
            ref = pats
        assert pats == ref, "pattern lists differ between the two orders of work"
        print("plain decode %s the speed part: get --detect-speed %.2f ms  (%d patterns, %d with the payload)" %
              ("beside" if mode else "after ", best * 1e3, len(pats), sum(p["bits"] == bench.PAYLOAD for p in pats)), flush=True)
lib.awm_debug_set_speed_overlap(1)
