import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the library leaves the environment alone)
"""How close the GPU path is to the oracle on one 115 s stereo stream: PCM, sync qualities, soft bits."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import audiowmark_amd as awm
import _oracle as orc
P = "0123456789abcdef0011223344556677"
ctx = awm.Context(0)
n = 115 * 44100
x = np.random.default_rng(60).uniform(-1, 1, (n, 2)).astype(np.float32)
want = orc.add(None, x, 2, P).reshape(n, 2)
got = ctx.add_watermark(None, P, torch.from_numpy(x).cuda()).cpu().numpy()
d = got.astype(np.float64) - want
print("add: rms %.3g  max %.3g  identical samples %.4f" % (np.sqrt(np.mean(d * d)), np.abs(d).max(), np.mean(got == want)))
w = torch.from_numpy(want).cuda()
gi, graw, gmean = ctx.search_approx(None, w)
oi, oraw, omean = orc.search_approx(None, want, 2)
print("approx: index equal %s  raw max diff %.3g  identical %.4f" % (np.array_equal(gi, oi), np.abs(graw - oraw).max(), np.mean(graw == oraw)))
g = ctx.sync_search(None, w)
o = orc.sync_search(None, want, 2)
print("search: index equal %s  quality max diff %.3g" % (g[0].tolist() == o[0].tolist(), np.abs(np.array(g[1]) - np.array(o[1])).max()))
