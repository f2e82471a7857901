import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import audiowmark_amd as awm
from audiowmark_amd import sharded
from test_gpu_parity import noise, PAY1, pkey
awm.set_params(chunk_size_min=10.0)
ctx = awm.Context(0)
minutes, cuts = 45, [0.2, 0.45, 0.7]
total = minutes * 60 * 44100 + 777
marked = ctx.add_watermark(None, PAY1, torch.from_numpy(noise(131 + minutes, total, 2)).cuda())
chunks = awm.plan_chunks(total)
first, count, off = chunks[1]
cw = marked[first:first + count].contiguous()
gi, gq, gb = ctx.sync_search(None, cw)
soft, ok = ctx.block_soft_bits(None, cw, gi)
w = 1 + np.arange(858) % 7
print("single chunk 1:", [(int(i), round(float(q), 4), round(float((s * w).sum()), 3)) for i, q, s in zip(gi, gq, soft)])
edges = [0] + [int(total * c) // 1024 * 1024 for c in cuts] + [total]
spans = [marked[a:b].contiguous() for a, b in zip(edges[:-1], edges[1:])]
ctxs = [ctx] + [awm.Context(0) for _ in spans[1:]]
os.environ["AWM_SHARD_DEBUG"] = "1"
got = sharded.multi_get(ctxs, None, spans)
want = ctx.get_watermark(None, marked)
print("equal:", [pkey(p) for p in got] == [pkey(p) for p in want], len(got), len(want))
