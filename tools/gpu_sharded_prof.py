"""Where does the time of the sharded (torch.distributed) code path go?  One rank (world 1, RCCL), 60 min stereo; every phase
bracketed by device synchronisation.  usage: python tools/gpu_sharded_prof.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist
import audiowmark_amd as awm
from audiowmark_amd import sharded

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ctx = awm.Context(0)
P = "0123456789abcdef0011223344556677"
n = 60 * 60 * 44100 // 1024 * 1024
x = torch.rand((n, 2), device=dev) * 2 - 1
out = torch.empty_like(x)
pipe = sharded.ShardedStream(ctx, dist, n_frames_local=n, n_channels=2)
acc = {}
def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return r
for rep in range(6):
    if rep == 1:
        acc.clear()
    timed("sharded add", lambda: pipe.add_watermark(None, P, x, out))
    timed("sharded get", lambda: pipe.get_watermark(None, out))
    timed("plain add", lambda: ctx.add_watermark(None, P, x, out=out))
    timed("plain get", lambda: ctx.get_watermark(None, out))
    # pieces of the sharded add
    start, _ = pipe.part.span(0)
    b, a = timed(" add: edge frames", lambda: sharded.exchange_edge_frames(dist, x, 2, pipe.part.lengths))
    bm = timed(" add: block_max alloc+init", lambda: (lambda t: (ctx.add_init_block_max(t), t)[1])(torch.empty(n // 44100 + 2, dtype=torch.float32, device=dev)))
    fm = pipe._frame_mod(None, P)
    timed(" add: add_mix (host table upload + kernel)", lambda: ctx.add_mix(x, out, fm, 0.01, 0, b, a, bm))
    timed(" add: all_reduce max", lambda: dist.all_reduce(bm, op=dist.ReduceOp.MAX))
    timed(" add: limit", lambda: ctx.add_limit(out, 0, bm))
    lo, hi, mine = pipe.part.chunk_range(0)
    rel = [(c[0], c[1]) for _, c in mine]
    pw = timed(" get: decode_chunks_raw", lambda: ctx.decode_chunks_raw(None, out, rel, True))
    found = {ci: pw[0][pw[1] == i] for i, (ci, _) in enumerate(mine)}
    timed(" get: gather_and_merge", lambda: sharded.gather_and_merge(dist, pipe.part, None, found))
for k, v in acc.items():
    print(f"{k:50s} {1e3 * v / 5:8.3f} ms")
dist.destroy_process_group()
