import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the library leaves the environment alone)
"""a batch of CLIPS (default 256) watermarked 30 s clips through get_watermark_batch, for a rocprofv3 kernel trace:
cd /tmp && rocprofv3 --kernel-trace --stats -d <dir> -- python tools/gpu_clip_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm
dev = torch.device("cuda", 0)
ctx = awm.Context(0)
n = int(os.environ.get("SECONDS_PER_CLIP", "30")) * 44100
N = int(os.environ.get("CLIPS", "256"))
gen = torch.Generator(device=dev); gen.manual_seed(5)
P = "0123456789abcdef0011223344556677"
outs = [ctx.add_watermark(None, P, torch.rand((n, 2), generator=gen, device=dev, dtype=torch.float32) * 2 - 1) for _ in range(N)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = ctx.get_watermark_batch(None, outs)
    torch.cuda.synchronize(); print("batch of %d: %.3f ms per clip" % (N, (time.perf_counter() - t0) * 1e3 / N))
