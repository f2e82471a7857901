"""staged clip batch with 1 / 2 / 4 / 8 host threads (AWM_STAGED_THREADS): ms per 30 s stereo clip"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm
dev = torch.device("cuda", 0)
ctx = awm.Context(0)
n = 30 * 44100
gen = torch.Generator(device=dev); gen.manual_seed(5)
N = int(os.environ.get("CLIPS", "128"))
P = "0123456789abcdef0011223344556677"
outs = [ctx.add_watermark(None, P, torch.rand((n, 2), generator=gen, device=dev, dtype=torch.float32) * 2 - 1) for _ in range(N)]
ref = None
for threads in (1, 2, 4, 8, 1, 2, 4, 8):
    os.environ["AWM_STAGED_THREADS"] = str(threads)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = ctx.get_watermark_batch(None, outs)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    if ref is None:
        ref = res
    print(f"staged batch, {threads} host thread(s): {1e3*best/N:.3f} ms/clip  identical: {res == ref}  -> get alone {30*N/best:.0f} xRT")
