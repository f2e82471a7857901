"""What ONE KEY PER CLIP costs on each side (BASELINE configs[4]): 1024 clips of 30 s, all watermarked with the same key; `add` and `get`
through the one-key batch calls and through the per-clip-key calls given 1024 times that key -- the device decodes the same patterns either
way, the difference is the per-key machinery (`add`: K16 builds the frame_mod tables on the device; `get`: the host builds a group's sync /
mix / bit-order tables while the device works on the previous group, uploads them, the kernels index tables per slice)."""
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
key = awm.test_key(7)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 8)) as pool:
    clips = [torch.from_numpy(c).cuda() for c in pool.map(lambda k: bench.quantise16(np, awm.binding.gen_noise(awm.test_key(k), 2 * n)).reshape(n, 2), range(1, n_clips + 1))]
outs = [torch.empty_like(c) for c in clips]
keys = [key] * n_clips
legs = {
    "add, one key": lambda: ctx.add_watermark_batch(key, PAY, clips, outs),
    "add, key per clip": lambda: ctx.add_watermark_batch_keys(keys, PAY, clips, outs),
    "add, key per clip, tables beside the clips": lambda: (awm.lib.awm_debug_set_add_batched(1), ctx.add_watermark_batch_keys(keys, PAY, clips, outs), awm.lib.awm_debug_set_add_batched(2))[1],
    "get, one key": lambda: ctx.get_watermark_batch(key, outs),
    "get, key per clip": lambda: ctx.get_watermark_batch_keys(keys, outs),
    "get, key per clip, the tables of all keys first": lambda: (awm.lib.awm_debug_set_key_tables_on_device(2), ctx.get_watermark_batch_keys(keys, outs), awm.lib.awm_debug_set_key_tables_on_device(1))[1],
}
res, first, host = {}, {}, {}
import ctypes as C
threads = [int(a) for a in sys.argv[2:]] or [2]
legs = {"%s, %d host threads" % (name, t): (lambda fn=fn, t=t: (awm.lib.awm_debug_set_staged_threads(t), fn())[1]) for t in threads for name, fn in legs.items()
        if t == threads[0] or name.startswith("get")}
for rep in range(6):
    for name, fn in legs.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        tk = (C.c_double * 3)()
        awm.lib.awm_debug_clip_key_timing(tk)
        if rep and tk[2]:
            host.setdefault(name, []).append((tk[0] / 1e3, tk[1] / 1e3, int(tk[2])))
        if rep:
            res.setdefault(name, []).append(dt)
        if name.startswith("get"):
            first.setdefault("get", r)
            assert r == first["get"]
for name, v in res.items():
    print("%-60s ms per call of %d clips: median %.2f  min %.2f  (%.4f ms per clip)" % (name, n_clips, sorted(v)[len(v) // 2], min(v), sorted(v)[len(v) // 2] / n_clips))
print("clips with the payload:", sum(any(p["bits"] == PAY for p in c) for c in first["get"]))
for name, v in host.items():
    print("%-36s host, summed over the lane threads: waiting for a group's tables %.1f ms, packing + upload %.1f ms, %d groups" % (name, v[-1][0], v[-1][1], v[-1][2]))
