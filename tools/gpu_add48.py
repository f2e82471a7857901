import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the library leaves the environment alone)
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import audiowmark_amd as awm
dev = torch.device("cuda", 0)
ctx = awm.Context(0)
n = 60 * 60 * 48000
x = torch.rand((n, 2), device=dev) * 2 - 1
out = torch.empty_like(x)
P = "0123456789abcdef0011223344556677"
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.add_watermark(None, P, x, out=out, sample_rate=48000)
    torch.cuda.synchronize(); print("add 48k: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
