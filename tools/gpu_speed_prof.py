import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the library leaves the environment alone)
"""detect_speed on a replayed 30 s stereo clip, a few times: for rocprofv3 --kernel-trace --stats and host timing."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import audiowmark_amd as awm
import _oracle

key = bytes(range(16))
C = 2
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(seconds * 44100)
x = _oracle.gen_noise(key, n * C)
ctx = awm.Context()
xd = torch.from_numpy(x.reshape(-1, C)).cuda()
yd = ctx.add_watermark(key, "0123456789abcdef0011223344556677", xd)
zd = ctx.resample_ratio(yd, 1 / 0.9764)
for patient in (False, True):
    ctx.detect_speed(key, zd, patient)
    torch.cuda.synchronize()
    ts = []
    for i in range(5):
        t = time.perf_counter(); r = ctx.detect_speed(key, zd, patient); ts.append(time.perf_counter() - t)
    print("detect_speed patient=%d: %s  min %.2f ms  median %.2f ms  (%.0f s stereo)" % (patient, r, min(ts) * 1e3, sorted(ts)[2] * 1e3, seconds))
r = ctx.resample_ratio(zd, 0.9764); torch.cuda.synchronize()
t = time.perf_counter(); r = ctx.resample_ratio(zd, 0.9764); torch.cuda.synchronize()
print("resample_ratio of the whole input: %.2f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); loc = ctx.speed_clip_location(key, zd, 25.0)
print("clip location: %.2f ms" % ((time.perf_counter() - t) * 1e3))
del r
awm.set_speed_params(detect_speed=True)
ctx.get_watermark(key, zd)
ts = []
for i in range(3):
    t = time.perf_counter(); p = ctx.get_watermark(key, zd); ts.append(time.perf_counter() - t)
awm.set_speed_params()
t = time.perf_counter(); p0 = ctx.get_watermark(key, zd); t0 = time.perf_counter() - t
print("get --detect-speed: %.2f ms (%d patterns), plain get %.2f ms" % (min(ts) * 1e3, len(p), t0 * 1e3))
