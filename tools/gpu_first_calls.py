import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""First calls on a fresh context: wall time and allocation census (awm_debug_alloc_stats) of calls 1, 2, 3, ... of
get (60 min resident), get --detect-speed (60 min 48 kHz replayed at 1.02) and the 8 h get -- what a command line user pays.

  python tools/gpu_first_calls.py [8h = 1]   ->  gpurun_out/first_calls.json
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
PAY = "0123456789abcdef0011223344556677"


def stats(awm):
    a, b, c, d = C.c_long(), C.c_double(), C.c_long(), C.c_double()
    awm.lib.awm_debug_alloc_stats(C.byref(a), C.byref(b), C.byref(c), C.byref(d))
    return [a.value, round(b.value, 2), c.value, round(d.value, 2)]


def calls(torch, awm, fn, n):
    out = []
    for _ in range(n):
        s0 = stats(awm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        s1 = stats(awm)
        out.append({"ms": round(dt, 2), "dev_allocs": s1[0] - s0[0], "dev_alloc_ms": round(s1[1] - s0[1], 2), "pinned_allocs": s1[2] - s0[2],
                    "pinned_alloc_ms": round(s1[3] - s0[3], 2)})
    return out


def main():
    do_8h = (int(sys.argv[1]) if len(sys.argv) > 1 else 1) != 0
    import torch
    import audiowmark_amd as awm
    res = {}
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    # 60 min plain
    ctx = awm.Context(0)
    n = 60 * 60 * 44100
    x = torch.rand((n, 2), generator=g, device="cuda") * 2 - 1
    out = torch.empty_like(x)
    res["add_60min"] = calls(torch, awm, lambda: ctx.add_watermark(None, PAY, x, out=out), 4)
    res["get_60min"] = calls(torch, awm, lambda: ctx.get_watermark(None, out), 6)
    ctx.close()
    del x, out
    # configs[2]
    ctx = awm.Context(0)
    rate = 48000
    n = 60 * 60 * rate
    x = torch.rand((n, 2), generator=g, device="cuda") * 2 - 1
    w = ctx.add_watermark(None, PAY, x, sample_rate=rate)
    del x
    fast = ctx.resample_ratio(w, 1 / 1.02, rate=rate)
    del w
    ctx.close()
    ctx = awm.Context(0)
    awm.set_speed_params(detect_speed=True)
    try:
        res["get_detect_speed_60min_48k"] = calls(torch, awm, lambda: ctx.get_watermark(None, ctx.resample(fast, rate, 44100)), 6)
    finally:
        awm.set_speed_params()
    ctx.close()
    del fast
    if do_8h:
        ctx = awm.Context(0)
        n = 8 * 3600 * 44100
        x = torch.rand((n, 2), generator=g, device="cuda") * 2 - 1
        out = torch.empty_like(x)
        ctx.add_watermark(None, PAY, x, out=out)
        del x
        res["get_8h"] = calls(torch, awm, lambda: ctx.get_watermark(None, out), 8)
        ctx.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "first_calls.json"), "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        print(k, [(c["ms"], c["dev_allocs"], c["dev_alloc_ms"], c["pinned_allocs"], c["pinned_alloc_ms"]) for c in v], flush=True)


if __name__ == "__main__":
    main()
