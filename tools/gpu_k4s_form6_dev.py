import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""K4s form 6 (update term on the matrix cores) against form 4 on the material of test_refinement_kernel_forms: where do the values differ,
and what do the two cost (kernel alone, 12 750 streams of 65 offsets)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import audiowmark_amd as awm
ctx = awm.Context(0)
lib = awm.lib
rng = np.random.default_rng(4242)
g = np.random.default_rng(901)
x = (g.random((40 * 44100, 2), dtype=np.float32) * 2 - 1)
x[5 * 44100:5 * 44100 + 30000] = 0
x[11 * 44100:11 * 44100 + 9000, 1] = 0
x[17 * 44100:17 * 44100 + 5000] = 0
x[17 * 44100 + 2500, 0] = 0.25
x *= np.linspace(1.0, 1e-3, len(x), dtype=np.float32)[:, None]
xd = torch.from_numpy(x).cuda()
bases = np.concatenate([rng.integers(0, len(x) - 1024 - 8 * 65, 300), 5 * 44100 + np.arange(-1600, 31000, 997),
                        11 * 44100 + np.arange(-1200, 9500, 511), 17 * 44100 + np.arange(-1100, 5200, 333)]).astype(np.int64)
print("streams", len(bases))
for count in (65, 64, 17, 1):
    outs = {}
    for form in (4, 6):
        lib.awm_debug_set_refine_form(form)
        outs[form] = ctx.sync_db_sliding(xd, bases, count)
    a, b = outs[4], outs[6]
    ne = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
    n = int(ne.sum())
    print("count %d: shape %s, values that differ %d of %d, max |d| %.3g, finite %s" % (count, tuple(a.shape), n, a.numel(), float((a - b).abs().nan_to_num(0).max()), bool(torch.isfinite(b).all())))
    if n:
        idx = ne.nonzero()[:12].tolist()
        for s_, r_, c_ in idx:
            print("   stream %d row %d offset %d: form 4 %.9g form 6 %.9g" % (s_, r_, c_, float(a[s_, r_, c_]), float(b[s_, r_, c_])))
        per_stream = ne.any(dim=2).any(dim=1).nonzero().flatten().tolist()
        print("   streams with differences:", len(per_stream), per_stream[:20], " offsets:", sorted(set(ne.nonzero()[:, 2].tolist()))[:20],
              " rows:", sorted(set(ne.nonzero()[:, 1].tolist()))[:30])
lib.awm_debug_set_refine_form(4)
