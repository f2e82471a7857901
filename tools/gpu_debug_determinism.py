import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import audiowmark_amd as awm
ctx = awm.Context(0)
P = "0123456789abcdef0011223344556677"
n = 60 * 60 * 44100
g = torch.Generator(device="cuda"); g.manual_seed(11)
x = torch.rand((n, 2), generator=g, device="cuda") * 2 - 1
w = ctx.add_watermark(None, P, x)
key = lambda ps: [(p["sync_index"], p["type"], p["block_type"], p["bits"], p["decode_error"]) for p in ps]
for mode in (1, 0, 1):
    awm.lib.awm_debug_set_viterbi_super(mode)
    first = None; diff = 0
    for i in range(25):
        ps = ctx.get_watermark(None, w)
        k = key(ps)
        if first is None: first = k
        elif k != first:
            diff += 1
            bad = [(a, b) for a, b in zip(k, first) if a != b][:2]
            print("  run", i, "differs:", len(k), len(first), bad)
    print("super =", mode, ": runs differing from the first:", diff, " patterns", len(first), " with payload", sum(p[3] == P for p in first))
    if mode == 0: ref = first
print("super result equals plain result:", first == ref)
w2 = ctx.add_watermark(None, P, x)
print("add deterministic:", bool(torch.equal(w, w2)))
