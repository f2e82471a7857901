import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""ONE case of tools/gpu_fuzz.py again (same seed, case number and max_seconds: the generator is replayed), with the three pattern lists side
by side -- this library, the oracle, the compiled reference -- and a mark where ours and the reference's differ.
  usage: tools/gpu_fuzz_case.py <seed> <case> <max_seconds>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import audiowmark_amd as awm
import _oracle as orc
import _ref as ref
seed, target, max_seconds = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rng = np.random.default_rng(seed)
ctx = awm.Context()
PAY = "0123456789abcdef0011223344556677"
for case in range(target + 1):
    ch = int(rng.choice([1, 2, 2, 2, 3]))
    seconds = float(rng.uniform(8, max_seconds))
    n = int(seconds * 44100)
    x = rng.uniform(-1, 1, (n, ch)).astype(np.float32) * float(rng.choice([1.0, 0.3, 0.05]))
    marked = rng.random() < 0.8
    if marked and case == target:
        x = orc.add(None, x, ch, PAY).reshape(-1, ch)
    lead = int(rng.choice([0, 0, 1, 1000, 44100, 5 * 44100]))
    trail = int(rng.choice([0, 0, 1, 777, 3 * 44100]))
    x = np.concatenate([np.zeros((lead, ch), np.float32), x, np.zeros((trail, ch), np.float32)])
    for _ in range(int(rng.integers(1, 4)) if rng.random() < 0.3 else 0):
        a = int(rng.integers(0, len(x) - 44100))
        x[a:a + int(rng.integers(1, 44100))] = 0
    if rng.random() < 0.2:
        x[:, ch - 1] = 0
print("case", target, "ch", ch, "len", len(x), "marked", marked)
got = ctx.get_watermark(None, torch.from_numpy(np.ascontiguousarray(x)).cuda())
want = orc.get(None, x, ch)
theirs = ref.get(None, x, ch)
f = lambda p: "%10.6f %9d t%d b%d %s q %.9f e %.6f" % (p["time"], p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"])
for i, (g, w, r) in enumerate(zip(got, want, theirs)):
    flag = "" if (g["sync_index"], g["type"], g["block_type"], g["bits"]) == (r["sync_index"], r["type"], r["block_type"], r["bits"]) else "   <<<<"
    print("%2d gpu %s\n   orc %s\n   ref %s%s" % (i, f(g), f(w), f(r), flag))
print(len(got), len(want), len(theirs))
