import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the library leaves the environment alone)
"""16 clips decoded one at a time (no concurrency between clips) for a rocprofv3 kernel trace: stand-alone kernel durations of a clip's `get`"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm
dev = torch.device("cuda", 0)
ctx = awm.Context(0)
awm.lib.awm_ctx_set_chunk_lanes(ctx._h, 1)
n = 30 * 44100
gen = torch.Generator(device=dev); gen.manual_seed(5)
P = "0123456789abcdef0011223344556677"
outs = [ctx.add_watermark(None, P, torch.rand((n, 2), generator=gen, device=dev, dtype=torch.float32) * 2 - 1) for _ in range(16)]
for rep in range(2):
    torch.cuda.synchronize(); time.sleep(0.01); t0 = time.perf_counter()
    for o in outs:
        ctx.get_watermark(None, o)
    torch.cuda.synchronize(); print("one by one %.3f ms per clip" % ((time.perf_counter() - t0) * 1e3 / 16))
