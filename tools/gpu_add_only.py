import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the library leaves the environment alone)
"""`add` of 60 min stereo alone, N times, with the per-kernel HIP event timing: the command behind the K2 before / after numbers
(also the one the --pmc SQ_INSTS_VALU passes wrap).  usage: gpu_add_only.py [steps] [fft_pair values, e.g. 1,0,1,0]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audiowmark_amd as awm
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = awm.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.rand((60 * 60 * 44100, 2), generator=g, device="cuda") * 2 - 1
out = torch.empty_like(x)
P = "0123456789abcdef0011223344556677"
awm.lib.awm_prof_name.restype = C.c_char_p
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1]     # awm_debug_set_fft_pair values, e.g. 1,0,1,0
for v in variants:
    awm.lib.awm_debug_set_fft_pair(v)
    for _ in range(3):
        ctx.add_watermark(None, P, x, out=out)
    torch.cuda.synchronize()
    awm.lib.awm_prof_reset(ctx._h); awm.lib.awm_prof_enable(ctx._h, 1)
    for _ in range(steps):
        ctx.add_watermark(None, P, x, out=out)
    torch.cuda.synchronize()
    awm.lib.awm_prof_enable(ctx._h, 0)
    for i in range(awm.lib.awm_prof_count()):
        ms, n, b = C.c_double(), C.c_long(), C.c_double()
        awm.lib.awm_prof_read(ctx._h, i, C.byref(ms), C.byref(n), C.byref(b))
        if n.value:
            print("pair %d  %-28s %8.4f ms per launch  %7.1f GB/s algorithmic  (%.1f %% of 8 TB/s)" % (v, awm.lib.awm_prof_name(i).decode(), ms.value / n.value,
                  b.value / ms.value / 1e6, b.value / ms.value / 1e6 / 80))
    print("pair %d  checksum %.12e" % (v, float(out.double().abs().sum())))
