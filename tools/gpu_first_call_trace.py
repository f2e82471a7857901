import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""What a FIRST `get` pays beyond the steady state: run under `rocprofv3 --hip-trace --marker-trace --kernel-trace --output-format csv`;
roctx ranges bracket get call 1, 2, 3 of a fresh context (60 min stereo resident).  tools/first_call_summary.py sums the HIP API time per
call and range from the CSVs."""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import audiowmark_amd as awm

tx = None
for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
    try:
        tx = ctypes.CDLL(name)
        break
    except OSError:
        pass
PAY = "0123456789abcdef0011223344556677"
ctx = awm.Context(0)
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(minutes * 60 * 44100)
g = torch.Generator(device="cuda"); g.manual_seed(5)
x = torch.rand((n, 2), generator=g, device="cuda") * 2 - 1
out = torch.empty_like(x)
ctx.add_watermark(None, PAY, x, out=out)
torch.cuda.synchronize()
for i in range(calls):
    if tx:
        tx.roctxRangePushA(("get_call_%d" % (i + 1)).encode())
    ctx.get_watermark(None, out)
    torch.cuda.synchronize()
    if tx:
        tx.roctxRangePop()
print("done", bool(tx))
