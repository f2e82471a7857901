import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""A / B of kernel variants behind debug toggles: K4s (2 bins x 2
channels on 42 lanes | 3 bins x 1 channel on 56 lanes); per-kernel times from the HIP events of a one-lane pass, results compared"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm
ctx = awm.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.rand((60 * 60 * 44100, 2), generator=g, device="cuda") * 2 - 1
out = torch.empty_like(x)
P = "0123456789abcdef0011223344556677"
awm.lib.awm_prof_name.restype = C.c_char_p
awm.lib.awm_debug_dependent_launch_us.restype = C.c_double
print("dependent launch probe: %.2f us per empty launch -> one-launch Viterbi in use by default: %d" % (awm.lib.awm_debug_dependent_launch_us(), awm.lib.awm_debug_viterbi_one_launch_in_use()))

def prof(fn, steps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    awm.lib.awm_prof_reset(ctx._h); awm.lib.awm_prof_enable(ctx._h, 1)
    for _ in range(steps): r = fn()
    torch.cuda.synchronize()
    awm.lib.awm_prof_enable(ctx._h, 0)
    res = {}
    for i in range(awm.lib.awm_prof_count()):
        ms, n, b = C.c_double(), C.c_long(), C.c_double()
        awm.lib.awm_prof_read(ctx._h, i, C.byref(ms), C.byref(n), C.byref(b))
        if n.value: res[awm.lib.awm_prof_name(i).decode()] = ms.value / steps
    return r, res

ctx.add_watermark(None, P, x, out=out)
ref = out.clone()
awm.lib.awm_ctx_set_chunk_lanes(ctx._h, 1)
first = None
for v in (1, 0):
    awm.lib.awm_debug_set_sliding3(v)
    pats, r = prof(lambda: ctx.get_watermark(None, ref), 3)
    k = [(p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"]) for p in pats]
    if first is None: first = k
    dq = max(abs(a[4] - b[4]) for a, b in zip(k, first)) if len(k) == len(first) else -1
    print("K4s variant %d: refine db %.4f ms  refine scan %.4f  viterbi %.4f  patterns %d  positions equal %s  max |dq| %.3g" % (v, r["sync_db_kernel(refine)"], r["sync_scan_kernel(refine)"],
          r["viterbi_kernel"], len(k), [a[:4] for a in k] == [a[:4] for a in first], dq))
awm.lib.awm_debug_set_sliding3(1)
base = None
for v in (1, 0, 1):
    awm.lib.awm_debug_set_viterbi_super(v)
    pats, r = prof(lambda: ctx.get_watermark(None, ref), 3)
    k = [(p["sync_index"], p["type"], p["block_type"], p["bits"], p["decode_error"]) for p in pats]
    if base is None: base = k
    print("viterbi super %d: viterbi %.4f ms per step  equal to first: %s" % (v, r["viterbi_kernel"], k == base))
awm.lib.awm_debug_set_viterbi_super(1)
for v in (1, 0, 1, 0):
    awm.lib.awm_debug_set_viterbi_persistent(v)
    pats, r = prof(lambda: ctx.get_watermark(None, ref), 5)
    k = [(p["sync_index"], p["type"], p["block_type"], p["bits"], p["decode_error"]) for p in pats]
    print("viterbi one launch %d: viterbi scope %.4f ms per step (one lane)  equal to first: %s" % (v, r["viterbi_kernel"], k == base))
awm.lib.awm_debug_set_viterbi_persistent(-1)
# the Viterbi jobs of a stream's chunks as one batch at the end (1) | per chunk (0): the timed configuration (chunks on concurrent lanes)
import time
awm.lib.awm_ctx_set_chunk_lanes(ctx._h, 4)
def step():
    ctx.add_watermark(None, P, x, out=out)
    return ctx.get_watermark(None, out)
for v in (1, 0, 1, 0, 1, 0):
    awm.lib.awm_debug_set_viterbi_persistent(v)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pats = step()
    torch.cuda.synchronize()
    k = [(p["sync_index"], p["type"], p["block_type"], p["bits"], p["decode_error"]) for p in pats]
    print("viterbi one launch %d: %.3f ms per step (add + get, four lanes), patterns %d, equal: %s" % (v, (time.perf_counter() - t0) / 20 * 1e3, len(pats), k == base))
awm.lib.awm_debug_set_viterbi_persistent(-1)
for v in (0, 1, 0, 1):
    awm.lib.awm_debug_set_merge_decodes(v)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pats = step()
    torch.cuda.synchronize()
    k = [(p["sync_index"], p["type"], p["block_type"], p["bits"], p["decode_error"]) for p in pats]
    print("decodes merged %d: %.3f ms per step (add + get, four lanes), patterns %d, equal to the per-chunk result: %s" % (v, (time.perf_counter() - t0) / 20 * 1e3, len(pats), k == base))
awm.lib.awm_debug_set_merge_decodes(0)
