import os; os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, audiowmark_amd as awm
PAY = "0123456789abcdef0011223344556677"
for hours, lanes in ((8, 4), (8, 2), (2, 4)):
    ctx = awm.Context(0)
    awm.lib.awm_ctx_set_chunk_lanes(ctx._h, lanes)
    n = int(hours * 3600 * 44100)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.rand((n, 2), generator=g, device="cuda") * 2 - 1
    out = torch.empty_like(x)
    ctx.add_watermark(None, PAY, x, out=out)
    del x
    ts = []
    for i in range(14):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.get_watermark(None, out); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    print(hours, "h", lanes, "lanes:", ts, flush=True)
    ctx.close(); del out
