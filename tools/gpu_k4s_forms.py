import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""A / B of the forms of K4s (awm_debug_set_refine_form: 0 / 3 rounds 2 - 5, 4 restructured with the same arithmetic, 5 update term in
float): outputs of the kernel alone compared bit for bit, stand-alone durations from the HIP events of a one-lane pass of `get` over
60 min stereo, the patterns and qualities of every form, and the add + get step (four lanes) with the forms taking turns."""
import ctypes as C, os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import audiowmark_amd as awm

ctx = awm.Context(0)
lib = awm.lib
lib.awm_prof_name.restype = C.c_char_p
P = "0123456789abcdef0011223344556677"
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
forms = [int(f) for f in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3, 4, 5]
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.rand((int(minutes * 60 * 44100), 2), generator=g, device="cuda") * 2 - 1
out = torch.empty_like(x)

# ---- the kernel alone
rng = np.random.default_rng(1)
bases = rng.integers(0, x.shape[0] - 1024 - 8 * 65, 12750).astype(np.int64)
res = {}
for f in forms:
    lib.awm_debug_set_refine_form(f)
    res[f] = ctx.sync_db_sliding(x, bases, 65)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ref = res[forms[0]]
for f in forms[1:]:
    d = (res[f] - ref).abs()
    print("kernel alone, form %d vs %d: equal %s, max |d dB| %.3g, mean %.3g" % (f, forms[0], bool(torch.equal(res[f], ref)), d.max().item(), d.mean().item()))
del res, ref


def prof(fn, steps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    lib.awm_prof_reset(ctx._h); lib.awm_prof_enable(ctx._h, 1)
    for _ in range(steps): r = fn()
    torch.cuda.synchronize()
    lib.awm_prof_enable(ctx._h, 0)
    ms_of = {}
    for i in range(lib.awm_prof_count()):
        ms, n, b = C.c_double(), C.c_long(), C.c_double()
        lib.awm_prof_read(ctx._h, i, C.byref(ms), C.byref(n), C.byref(b))
        if n.value: ms_of[lib.awm_prof_name(i).decode()] = (ms.value / steps, n.value / steps, b.value / steps)
    return r, ms_of


ctx.add_watermark(None, P, x, out=out)
w = out.clone()
lib.awm_ctx_set_chunk_lanes(ctx._h, 1)
first = None
report = {}
for f in forms + forms[:1]:
    lib.awm_debug_set_refine_form(f)
    pats, r = prof(lambda: ctx.get_watermark(None, w), 3)
    k = [(p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"]) for p in pats]
    if first is None: first = k
    same = [a[:4] for a in k] == [a[:4] for a in first]
    dq = max(abs(a[4] - b[4]) for a, b in zip(k, first)) if len(k) == len(first) else -1
    ms, n, b = r["sync_db_kernel(refine)"]
    report[f] = {"ms_per_step": round(ms, 4), "launches": n, "us_per_launch": round(ms / n * 1e3, 1), "algorithmic_GBps": round(b / ms / 1e6, 1)}
    print("form %d: K4s %.4f ms per step (%.0f launches, %.1f us each, %.0f GB/s algorithmic), K5g %.4f; patterns %d, positions equal to form %d: %s, max |dq| %.3g"
          % (f, ms, n, ms / n * 1e3, b / ms / 1e6, r["sync_scan_kernel(refine)"][0], len(k), forms[0], same, dq))
lib.awm_ctx_set_chunk_lanes(ctx._h, 4)


def step():
    ctx.add_watermark(None, P, x, out=out)
    return ctx.get_watermark(None, out)


for f in forms + forms:
    lib.awm_debug_set_refine_form(f)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pats = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    report.setdefault(f, {}).setdefault("step_ms", []).append(round(ms, 3))
    print("form %d: %.3f ms per step (add + get, four lanes), %d patterns" % (f, ms, len(pats)))
print(json.dumps(report))
lib.awm_debug_set_refine_form(4)
