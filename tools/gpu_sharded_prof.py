import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""The multi-GPU code path with ONE rank (RCCL world 1) and with N contexts of one process on the same device (awm_multi_*):
what the protocol of host/wmshard.cc costs beside the plain single-GPU calls.  usage: gpu_sharded_prof.py [minutes]"""
import sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
import audiowmark_amd as awm
from audiowmark_amd import sharded
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ctx = awm.Context(0)
n = int(minutes * 60 * 44100)
g = torch.Generator(device=dev); g.manual_seed(7)
x = torch.rand((n, 2), generator=g, device=dev) * 2 - 1
out = torch.empty_like(x)
P = "0123456789abcdef0011223344556677"
pipe = sharded.ShardedStream(ctx, dist, n_frames_local=n, n_channels=2)

def timed(what, fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-52s %8.3f ms" % (what, best * 1e3))
    return r

timed("plain add", lambda: ctx.add_watermark(None, P, x, out=out))
a = timed("plain get", lambda: ctx.get_watermark(None, out))
timed("sharded add, world 1 (RCCL transport)", lambda: pipe.add_watermark(None, P, x, out))
b = timed("sharded get, world 1 (RCCL transport)", lambda: pipe.get_watermark(None, out))
print("patterns equal:", [(p["sync_index"], p["bits"]) for p in a] == [(p["sync_index"], p["bits"]) for p in b], len(a))
for k in (2, 4):
    cuts = [n * i // k // 1024 * 1024 for i in range(k)] + [n]
    spans = [out[s:e] for s, e in zip(cuts[:-1], cuts[1:])]
    ctxs = [ctx] + [awm.Context(0) for _ in range(k - 1)]
    c = timed("multi get, %d contexts on one device (threads)" % k, lambda: sharded.multi_get(ctxs, None, spans))
    print("patterns equal:", [(p["sync_index"], p["bits"]) for p in a] == [(p["sync_index"], p["bits"]) for p in c])
dist.destroy_process_group()
