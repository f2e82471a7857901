"""one staged batch of 32 clips (2 groups of 16 on one host thread) for a rocprofv3 kernel trace"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("AWM_STAGED_THREADS", "1")
import torch
import audiowmark_amd as awm
dev = torch.device("cuda", 0)
ctx = awm.Context(0)
n = 30 * 44100
gen = torch.Generator(device=dev); gen.manual_seed(5)
P = "0123456789abcdef0011223344556677"
outs = [ctx.add_watermark(None, P, torch.rand((n, 2), generator=gen, device=dev, dtype=torch.float32) * 2 - 1) for _ in range(32)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = ctx.get_watermark_batch(None, outs)
    torch.cuda.synchronize(); print("batch %.3f ms" % ((time.perf_counter() - t0) * 1e3))
