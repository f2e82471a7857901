import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""`add` of 60 min stereo in slabs (awm_debug_set_add_slab_mb): fused add of a slab, then the limiter for everything whose look-ahead
second is complete, so that the limiter may find the slab in the 256 MB memory-side cache.  Prints, per slab size, the time of the
add + limiter inside an add + get step (the bench's mix of kernels: a loop of nothing but `add` runs the chip hotter) and the two
scopes' stand-alone times; the output must be bit-identical to the whole-stream add.   -> profiles/rNN/add_slab_sweep.txt"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm
ctx = awm.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.rand((60 * 60 * 44100, 2), generator=g, device="cuda") * 2 - 1
out = torch.empty_like(x)
P = "0123456789abcdef0011223344556677"
awm.lib.awm_prof_name.restype = C.c_char_p
ctx.add_watermark(None, P, x, out=out)
ref = out.clone()

def scopes(steps):
    res = {}
    for i in range(awm.lib.awm_prof_count()):
        ms, n, b = C.c_double(), C.c_long(), C.c_double()
        awm.lib.awm_prof_read(ctx._h, i, C.byref(ms), C.byref(n), C.byref(b))
        if n.value: res[awm.lib.awm_prof_name(i).decode()] = (ms.value / steps, n.value / steps)
    return res

sizes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 32, 64, 96, 192, 384, 0]
for mb in sizes:
    awm.lib.awm_debug_set_add_slab_mb(mb)
    ctx.add_watermark(None, P, x, out=out)
    same = bool(torch.equal(out, ref))
    def step():
        ctx.add_watermark(None, P, x, out=out)
        return ctx.get_watermark(None, out)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / 10 * 1e3
    # the add alone, stream order, with the per-scope events
    awm.lib.awm_prof_reset(ctx._h); awm.lib.awm_prof_enable(ctx._h, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ctx.add_watermark(None, P, x, out=out)
    torch.cuda.synchronize()
    t_add = (time.perf_counter() - t0) / 10 * 1e3
    awm.lib.awm_prof_enable(ctx._h, 0)
    s = scopes(10)
    print("slab %4d MB: add + get %.3f ms per step | add alone %.3f ms (add_mix %.3f ms in %d launches, limiter %.3f ms in %d) | bit-identical to the whole-stream add: %s"
          % (mb, t_step, t_add, s["add_mix_kernel"][0], s["add_mix_kernel"][1], s["limiter_kernel"][0], s["limiter_kernel"][1], same), flush=True)
awm.lib.awm_debug_set_add_slab_mb(0)
