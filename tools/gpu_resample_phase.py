import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""K10: the phase-per-thread kernel for stereo 48 <-> 44.1 kHz against the generic kernel (awm_debug_set_resample_phase): 60 min of stereo noise
down and up, outputs compared bit for bit, best of 5, alternating.   -> profiles/rNN/resample_phase.txt"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm
ctx = awm.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(7)
x48 = torch.rand((60 * 60 * 48000 + 12345, 2), generator=g, device="cuda", dtype=torch.float32) * 2 - 1
def best_of(fn, n=5):
    b = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return r, b
ref = {}
for mode in (1, 0, 1, 0):
    awm.lib.awm_debug_set_resample_phase(mode)
    down, t_d = best_of(lambda: ctx.resample(x48, 48000, 44100))
    up, t_u = best_of(lambda: ctx.resample(down, 44100, 48000))
    same = ""
    if mode in ref or (1 - mode) in ref:
        o = ref.get(1 - mode) or ref.get(mode)
        same = ", equal to the other kernel's output: %s / %s" % (bool(torch.equal(down, o[0])), bool(torch.equal(up, o[1])))
    ref[mode] = (down, up)
    gb = (x48.numel() + down.numel()) * 4 / 1e9
    print("%s kernel: 60 min stereo 48 -> 44.1 kHz %.3f ms (%.2f TB/s), 44.1 -> 48 kHz %.3f ms%s" %
          ("phase-per-thread" if mode else "generic         ", t_d * 1e3, gb / t_d / 1e3, t_u * 1e3, same), flush=True)
# short and ragged inputs through both kernels (ends of the stream, fewer outputs than a tile)
for n in (1, 17, 146, 147, 148, 2351, 2352, 2353, 100003):
    y = x48[:n].contiguous()
    outs = []
    for mode in (1, 0):
        awm.lib.awm_debug_set_resample_phase(mode)
        outs.append((ctx.resample(y, 48000, 44100), ctx.resample(y, 44100, 48000)))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), n
print("short / ragged inputs: equal")
awm.lib.awm_debug_set_resample_phase(1)
