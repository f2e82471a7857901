import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""debug: scores of a chunk computed in parts == scores of the whole chunk?  and the failing multi-context geometry"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import audiowmark_amd as awm
from audiowmark_amd import sharded
from test_gpu_parity import noise, PAY1, pkey
awm.set_params(chunk_size_min=10.0)
ctx = awm.Context(0)
minutes, cuts = 21, [0.31, 0.34, 0.8]
total = minutes * 60 * 44100 + 777
whole = torch.from_numpy(noise(131 + minutes, total, 2)).cuda()
marked = ctx.add_watermark(None, PAY1, whole)
chunks = awm.plan_chunks(total)
first, count, off = chunks[2]
cw = marked[first:first + count].contiguous()
idx, raw, mean = ctx.search_approx(None, cw)
S = count // 1024 - 1 - 2226
print("chunk 2: first", first, "count", count, "S", S, "scores", len(idx))
for a, b in ((0, 1079), (1079, 3308), (3308, S), (500, 501), (0, 1)):
    last = b == S
    n = (count // 1024 - a) * 1024 if last else (b - a + 2227) * 1024
    sub = cw[a * 1024:a * 1024 + n].contiguous()
    i2, r2, m2 = ctx.search_approx(None, sub)
    want = raw[4 * a:4 * b]
    print("part [%d, %d): %d scores (want %d), raw identical: %s, max |d| %.3g" % (a, b, len(r2), len(want), np.array_equal(r2, want),
          np.abs(r2 - want).max() if len(r2) == len(want) else -1))
gi, gq, gb = ctx.sync_search(None, cw)
print("single sync_search:", list(zip(gi.tolist(), np.round(gq, 6).tolist(), gb.tolist())))
edges = [0] + [int(total * c) // 1024 * 1024 for c in cuts] + [total]
spans = [marked[a:b].contiguous() for a, b in zip(edges[:-1], edges[1:])]
ctxs = [ctx] + [awm.Context(0) for _ in spans[1:]]
got = sharded.multi_get(ctxs, None, spans)
want = ctx.get_watermark(None, marked)
t0 = off
g2 = sorted((p["sync_index"], round(p["sync_quality"], 6), p["block_type"]) for p in got if p["type"] == 0 and p["block_type"] < 2 and abs(p["time"] - t0 - p["sync_index"] / 44100) < 1e-6)
w2 = sorted((p["sync_index"], round(p["sync_quality"], 6), p["block_type"]) for p in want if p["type"] == 0 and p["block_type"] < 2 and abs(p["time"] - t0 - p["sync_index"] / 44100) < 1e-6)
print("multi  chunk 2 blocks:", g2)
print("single chunk 2 blocks:", w2)
print("plan:", [e for e in sharded.plan([s.shape[0] for s in spans]) if e[3]])
for _ in range(3):
    again = sharded.multi_get(ctxs, None, spans)
    print("repeat equal to first multi:", [pkey(p) for p in again] == [pkey(p) for p in got], " equal to single:", [pkey(p) for p in again] == [pkey(p) for p in want])
