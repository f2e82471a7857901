import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the library leaves the environment alone)
import time, torch, sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import audiowmark_amd as awm
dev = torch.device('cuda', 0)
ctx = awm.Context(0)
n = 60 * 60 * 44100
gen = torch.Generator(device=dev); gen.manual_seed(7)
x = torch.rand((n, 2), generator=gen, device=dev, dtype=torch.float32) * 2 - 1
out = torch.empty_like(x)
P = "0123456789abcdef0011223344556677"
for i in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.add_watermark(None, P, x, out=out)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pats = ctx.get_watermark(None, out)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {i}: add {1e3*(t1-t0):.2f} ms  get {1e3*(t2-t1):.2f} ms  total {1e3*(t2-t0):.2f}")
