import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""K4s alone (awm_debug_sync_db_sliding_d): duration against the number of fine offsets -- the fixed part of a row (first transform in
double, set-up) and the cost of a step -- for the forms given.   usage: tools/gpu_k4s_alone.py [forms=3,4,5] [streams=12750] [counts=1,17,33,65]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import audiowmark_amd as awm

forms = [int(f) for f in (sys.argv[1] if len(sys.argv) > 1 else "3,4,5").split(",")]
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 12750
counts = [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "1,17,33,65").split(",")]
ctx = awm.Context(0)
lib = awm.lib
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.rand((20 * 60 * 44100, 2), generator=g, device="cuda") * 2 - 1
bases = torch.from_numpy(np.random.default_rng(1).integers(0, x.shape[0] - 1024 - 8 * 65, n_streams).astype(np.int64)).cuda()
LD = int(os.environ.get("K4S_LD", "72"))               # 64: the refinement's gathered layout (60 rows of 64 offsets + the 65th values apart)
out = torch.zeros((n_streams, 81, 72), dtype=torch.float32, device="cuda")
import ctypes as C


def run(count, reps):
    for _ in range(reps):
        rc = lib.awm_debug_sync_db_sliding_d(ctx._h, C.c_void_p(x.data_ptr()), x.shape[0], 2, C.c_void_p(bases.data_ptr()), n_streams, count, LD, C.c_void_p(out.data_ptr()))
        assert rc == 0


abls = [int(a) for a in (sys.argv[4] if len(sys.argv) > 4 else "0").split(",")]
for f, ab in [(f, ab) for f in forms for ab in abls]:
    lib.awm_debug_set_refine_form(f)
    lib.awm_debug_set_k4s_ablate(ab)
    line = []
    for c in counts:
        run(c, 3); ctx.synchronize()
        t0 = time.perf_counter()
        run(c, 20); ctx.synchronize()
        line.append((c, (time.perf_counter() - t0) / 20 * 1e6))
    (c0, t0_), (c1, t1_) = line[0], line[-1]
    per_step = (t1_ - t0_) / (c1 - c0) if c1 != c0 else 0
    print("form %d ablate %d: " % (f, ab) + "  ".join("%d offsets %.1f us" % l for l in line) + "   -> %.2f us per offset, %.1f us fixed" % (per_step, t0_ - per_step * c0))
lib.awm_debug_set_refine_form(4)
