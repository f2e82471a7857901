import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the library leaves the environment alone)
"""add + get of 30 s stereo clips (BASELINE config 5 shape) on one GPU: clips per second, one clip at a time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm
dev = torch.device("cuda", 0)
ctx = awm.Context(0)
n = 30 * 44100
gen = torch.Generator(device=dev); gen.manual_seed(5)
N = int(os.environ.get("CLIPS", "64"))
clips = [torch.rand((n, 2), generator=gen, device=dev, dtype=torch.float32) * 2 - 1 for _ in range(N)]
P = "0123456789abcdef0011223344556677"
outs = [torch.empty_like(c) for c in clips]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for c, o in zip(clips, outs):
        ctx.add_watermark(None, P, c, out=o)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ok = 0
    for o in outs:
        pats = ctx.get_watermark(None, o)
        ok += any(p["bits"] == P for p in pats)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep {rep}: add {1e3*(t1-t0)/N:.2f} ms/clip  get {1e3*(t2-t1)/N:.2f} ms/clip  recovered {ok}/{N}  -> {30*N/(t2-t0):.0f} xRT")
ref = [ctx.get_watermark(None, o) for o in outs]
for threads in (1, 2, 4, 8, 16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = ctx.get_watermark_batch(None, outs, n_threads=threads)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    same = all(a == b for a, b in zip(res, ref))
    print(f"batch get, {threads:2d} lanes: {1e3*(t1-t0)/N:.3f} ms/clip  identical to one-by-one: {same}  -> get alone {30*N/(t1-t0):.0f} xRT")
