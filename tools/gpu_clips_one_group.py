import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""BASELINE configs[4], ONE group of 64 clips of 30 s (= one lane of `get`; `add` runs one clip per lane), every clip with its own key, repeated --
for `rocprofv3 --kernel-trace --stats` (-> profiles/rNN/rocprofv3_kernel_stats_clips_one_group.csv; the pass behind bench.py --config clips:
clip_batch_config.kernels_one_group_of_64_clips)."""
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import audiowmark_amd as awm
import bench

PAY = "0123456789abcdef0011223344556677"
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = awm.Context(0)
n = 30 * 44100
keys = [awm.test_key(k) for k in range(1, 65)]
clips = [torch.from_numpy(bench.quantise16(np, awm.binding.gen_noise(k, 2 * n)).reshape(n, 2)).cuda() for k in keys]
outs = [torch.empty_like(c) for c in clips]
for _ in range(calls):
    ctx.add_watermark_batch_keys(keys, PAY, clips, outs)
    pats = ctx.get_watermark_batch_keys(keys, outs)
torch.cuda.synchronize()
print("clips with the payload:", sum(any(p["bits"] == PAY for p in c) for c in pats))
