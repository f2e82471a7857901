"""A/B on one box: `get` of 1024 clips of 30 s (BASELINE configs[4], one key per clip) with the padded copies written as whole slices
(margin = a slice) and with 2048 frames of zeros on either side of a clip (the default), alternating.  Prints ms per call."""
import os
import sys
import time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import audiowmark_amd as awm
import bench

PAY = "0123456789abcdef0011223344556677"
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = awm.Context(0)
n = 30 * 44100
keys = [awm.test_key(k) for k in range(1, n_clips + 1)]
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 8)) as pool:
    clips = [torch.from_numpy(c).cuda() for c in pool.map(lambda k: bench.quantise16(np, awm.binding.gen_noise(k, 2 * n)).reshape(n, 2), keys)]
outs = [torch.empty_like(c) for c in clips]
ctx.add_watermark_batch_keys(keys, PAY, clips, outs)
torch.cuda.synchronize()
res = {}
ref = None
for rep in range(6):
    for name, margin in (("whole_slices", 1 << 20), ("margin_2048", 0)):
        awm.lib.awm_debug_set_clip_pad_margin(margin)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pats = ctx.get_watermark_batch_keys(keys, outs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if rep:
            res.setdefault(name, []).append(dt)
        if ref is None:
            ref = pats
        assert pats == ref
awm.lib.awm_debug_set_clip_pad_margin(0)
for name, v in res.items():
    print(name, "ms per get of %d clips: median %.2f  min %.2f  all %s" % (n_clips, sorted(v)[len(v) // 2], min(v), [round(x, 2) for x in v]))
print("clips with the payload:", sum(any(p["bits"] == PAY for p in c) for c in ref))
