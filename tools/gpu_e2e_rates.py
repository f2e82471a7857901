"""Ad-hoc measurement (not a pytest file): PCIe-inclusive and file-to-file rates for DESIGN.md."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import audiowmark_amd as awm
PAY = "0123456789abcdef0011223344556677"
n = 60 * 60 * 44100
ctx = awm.Context(0)
host = torch.rand((n, 2), dtype=torch.float32).mul_(2).sub_(1).pin_memory()
out_host = torch.empty_like(host).pin_memory()
dev_in = torch.empty((n, 2), dtype=torch.float32, device="cuda")
dev_out = torch.empty_like(dev_in)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev_in.copy_(host, non_blocking=True)
    ctx.add_watermark(None, PAY, dev_in, out=dev_out)
    out_host.copy_(dev_out, non_blocking=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    dev_in.copy_(out_host, non_blocking=True)
    pats = ctx.get_watermark(None, dev_in)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("PCIe inclusive (pinned host): add %.1f ms get %.1f ms -> %.0f xRT, matches %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, 3600 / (t2 - t0), sum(p["bits"] == PAY for p in pats)))
A = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audiowmark_amd", "audiowmark")
raw = "/tmp/in60.raw"
(host * 32767).to(torch.int16).numpy().tofile(raw)
args = ["--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16"]
t0 = time.perf_counter()
subprocess.run([A, "add", "-q", "--format", "raw"] + args + [raw, "/tmp/out60.raw", PAY], check=True)
t1 = time.perf_counter()
r = subprocess.run([A, "cmp", "-q", "--input-format", "raw"] + args + ["/tmp/out60.raw", PAY], stdout=subprocess.PIPE)
t2 = time.perf_counter()
print("CLI file->file 60 min stereo s16 raw: add %.2f s, cmp %.2f s -> %.0f xRT; %s" % (t1 - t0, t2 - t1, 3600 / (t2 - t0), r.stdout.decode().splitlines()[-2:]))
