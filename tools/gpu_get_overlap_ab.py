import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""A / B of the file level `get`: the whole stream loaded before the first chunk starts (default) against the chunks started while the
stream is still crossing PCIe (awm_debug_set_get_overlap (1)), alternating on ONE watermarked file in the page cache, for 60 min and 8 h
of s16 stereo.   usage: tools/gpu_get_overlap_ab.py [minutes ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm

P = "0123456789abcdef0011223344556677"
ctx = awm.Context(0)
awm.lib.awm_set_quiet(1)
for minutes in [float(a) for a in sys.argv[1:]] or [60.0, 480.0]:
    n = int(minutes * 60 * 44100)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.rand((n, 2), generator=g, device="cuda") * 2 - 1
    w = ctx.add_watermark(None, P, x)
    del x
    raw = ctx.pcm_encode(w.reshape(-1), 16, 0, False, True).cpu().numpy()
    del w
    path = "/dev/shm/awm_overlap_ab_%d.raw" % os.getpid()
    raw.tofile(path)
    del raw
    rf = awm.binding.RawFormat(2, 44100, 16, 0, 0)
    key = lambda p: (p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"])
    res = {0: [], 1: []}
    first = None
    try:
        for rep in range(6):
            for mode in (0, 1):
                awm.lib.awm_debug_set_get_overlap(mode)
                t0 = time.perf_counter()
                pats = ctx.get_watermark_file(None, path, rf)
                res[mode].append((time.perf_counter() - t0) * 1e3)
                k = [key(p) for p in pats]
                first = first or k
                assert k == first
    finally:
        awm.lib.awm_debug_set_get_overlap(0)
        os.unlink(path)
    fmt = lambda v: " ".join("%.1f" % t for t in v)
    print("%g min: whole stream first  %s  (min %.2f ms)" % (minutes, fmt(res[0]), min(res[0][1:])))
    print("%g min: chunks during load  %s  (min %.2f ms)   %d patterns, identical every time" % (minutes, fmt(res[1]), min(res[1][1:]), len(first)), flush=True)
