import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""K12 variants (awm_debug_set_resample_var_mode): the stretched copy of a 25 min chunk alone, and get --detect-speed of configs[2] as a whole.
  python tools/gpu_resample_var.py        ->  profiles/rNN/resample_var_modes.txt"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm
import bench
ctx = awm.Context(0)
rate = 48000
g = torch.Generator(device="cuda"); g.manual_seed(4711)
x = torch.rand((60 * 60 * rate, 2), generator=g, device="cuda", dtype=torch.float32) * 2 - 1
w = ctx.add_watermark(None, bench.PAYLOAD, x, sample_rate=rate)
del x
fast = ctx.resample_ratio(w, 1 / 1.02, rate=rate)
del w
chunk = fast[: 25 * 60 * 44100].contiguous()
awm.set_speed_params(detect_speed=True)
ref = ref_out = None
def best_of(fn, n=5):
    b = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return r, b
for mode in (3, 0, 1, 2, 3, 0):
    awm.lib.awm_debug_set_resample_var_mode(mode)
    out, t_one = best_of(lambda: ctx.resample_ratio(chunk, 1.02, rate=44100))
    pats, t_get = best_of(lambda: ctx.get_watermark(None, ctx.resample(fast, rate, 44100)), 3)
    if ref is None:
        ref, ref_out = pats, out
    print("mode %d (window in LDS %d, table kept %d): stretched copy of 25 min %.3f ms, get --detect-speed %.2f ms, same output %s, same patterns %s" %
          (mode, mode & 1, (mode >> 1) & 1, t_one * 1e3, t_get * 1e3, bool(torch.equal(out, ref_out)), pats == ref), flush=True)
awm.lib.awm_debug_set_resample_var_mode(3)
