import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""Where the per-clip keys of a clip batch cost time: add and get of 1024 clips of 30 s, each phase with one key for all clips and with
a key per clip (awm_*_batch_keys_d).   usage: python tools/gpu_clip_keys.py [clips]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import audiowmark_amd as awm
from concurrent.futures import ThreadPoolExecutor
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = "0123456789abcdef0011223344556677"
ctx = awm.Context(0)
n_clip = 30 * 44100
keys = [awm.test_key(k) for k in range(1, N + 1)]
def make(k):
    x = awm.binding.gen_noise(awm.test_key(k), 2 * n_clip)
    return (np.round(x * 32768.0).clip(-32768, 32767) / 32768.0).astype(np.float32).reshape(n_clip, 2)
with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 8)) as pool:
    clips = [torch.from_numpy(c).cuda() for c in pool.map(make, range(1, N + 1))]
outs = [torch.empty_like(c) for c in clips]
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
a1, _ = timed(lambda: ctx.add_watermark_batch(keys[0], P, clips, outs))
g1, _ = timed(lambda: ctx.get_watermark_batch(keys[0], outs))
ak, _ = timed(lambda: ctx.add_watermark_batch_keys(keys, P, clips, outs))
gk, r = timed(lambda: ctx.get_watermark_batch_keys(keys, outs))
ok = sum(1 for pats in r if any(p["bits"] == P for p in pats))
print("clips %d   add: one key %.1f ms, key per clip %.1f ms   get: one key %.1f ms, key per clip %.1f ms   payloads %d" % (N, a1, ak, g1, gk, ok))
print("per clip: add %.4f -> %.4f ms, get %.4f -> %.4f ms" % (a1 / N, ak / N, g1 / N, gk / N))
