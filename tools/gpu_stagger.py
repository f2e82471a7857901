import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""Phase offset between the chunk lanes of `get` (awm_debug_set_chunk_stagger): add + get of 60 min (and get of 8 h) with the chunks
started together (0) and one dB kernel apart (1), alternating in one process; results compared.  -> profiles/rNN/chunk_stagger.txt"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm
ctx = awm.Context(0)
P = "0123456789abcdef0011223344556677"
g = torch.Generator(device="cuda"); g.manual_seed(7)
for minutes, reps in ((60, 20), (480, 4)):
    x = torch.rand((minutes * 60 * 44100, 2), generator=g, device="cuda") * 2 - 1
    out = torch.empty_like(x)
    ctx.add_watermark(None, P, x, out=out)
    base = None
    for mode in (0, 1, 2, 0, 1, 2):
        awm.lib.awm_debug_set_chunk_stagger(mode)
        def step():
            ctx.add_watermark(None, P, x, out=out)
            return ctx.get_watermark(None, out)
        for _ in range(2): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): pats = step()
        torch.cuda.synchronize()
        t_step = (time.perf_counter() - t0) / reps * 1e3
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): ctx.get_watermark(None, out)
        torch.cuda.synchronize()
        t_get = (time.perf_counter() - t0) / reps * 1e3
        k = [(p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"]) for p in pats]
        if base is None: base = k
        print("%3d min, chunk stagger %d: add + get %.3f ms per step, get alone %.3f ms, patterns %d, equal: %s" % (minutes, mode, t_step, t_get, len(k), k == base), flush=True)
    del x, out
awm.lib.awm_debug_set_chunk_stagger(-1)
