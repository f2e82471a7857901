import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the library leaves the environment alone)
"""Development check of the speed detection path on a GPU box: every stage against the oracle, with timings."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import audiowmark_amd as awm
import _oracle

key = bytes(range(16))
C = 2
n = 30 * 44100
payload = "0123456789abcdef0011223344556677"
x = _oracle.gen_noise(key, n * C)
ctx = awm.Context()
dev = torch.device("cuda:0")
xd = torch.from_numpy(x.reshape(-1, C)).to(dev)
yd = ctx.add_watermark(key, payload, xd)
y = yd.cpu().numpy().ravel()

for speed in (0.9764, 1.01):
    print("=== speed", speed)
    t = time.time(); z_o = _oracle.resample_ratio(y, C, 1 / speed); t_o = time.time() - t
    zd = ctx.resample_ratio(yd, 1 / speed); torch.cuda.synchronize()
    t = time.time(); zd = ctx.resample_ratio(yd, 1 / speed); torch.cuda.synchronize(); t_g = time.time() - t
    z_g = zd.cpu().numpy().ravel()
    d = np.abs(z_g - z_o)
    print("resample_ratio: frames %d / %d, max |diff| %.3g, differing %d of %d, oracle %.3fs gpu %.4fs"
          % (len(z_g) // C, len(z_o) // C, d.max(), int((d > 0).sum()), d.size, t_o, t_g))
    zd = torch.from_numpy(z_o.reshape(-1, C)).to(dev)      # same input for the stages below
    l_o = _oracle.speed_clip_location(key, z_o, C, 25.0)
    l_g = ctx.speed_clip_location(key, zd, 25.0)
    print("clip location", l_o, l_g)
    m_o = _oracle.speed_mags(key, z_o, C, l_o, 0.98, 25.0)
    m_g = ctx.speed_mags(key, zd, l_o, 0.98, 25.0)
    print("mags", m_o.shape, m_g.shape, "max |diff| %.3g of max %.1f" % (np.abs(m_o - m_g).max(), np.abs(m_o).max()))
    s_o, q_o = _oracle.speed_scan(key, z_o, C, l_o, 25.0, 1.0007, 5, 2, [0.98])
    s_g, q_g = ctx.speed_scan(key, zd, l_o, 25.0, 1.0007, 5, 2, [0.98])
    print("scan: speeds equal", np.array_equal(s_o, s_g), "max |dq| %.3g, best %.4f at %.6f" % (np.abs(q_o - q_g).max(), q_g.max(), s_g[q_g.argmax()]))
    for patient in (False, True):
        t = time.time(); d_o = _oracle.detect_speed(key, z_o, C, patient); t_o = time.time() - t
        d_g = ctx.detect_speed(key, zd, patient)
        t = time.time(); d_g = ctx.detect_speed(key, zd, patient); t_g = time.time() - t
        print("detect patient=%d: oracle %s (%.2fs)  gpu %s (%.4fs)" % (patient, d_o, t_o, d_g, t_g))
    _oracle.set_speed_params(True, False, -1)
    awm.set_speed_params(True, False, -1, -1)
    t = time.time(); p_o = _oracle.decode_chunk(key, z_o, C, True); t_o = time.time() - t
    p_g = ctx.decode_chunk(key, zd, True)
    t = time.time(); p_g = ctx.decode_chunk(key, zd, True); t_g = time.time() - t
    _oracle.set_speed_params(False, False, -1)
    awm.set_speed_params(False, False, -1, -1)
    same = len(p_o) == len(p_g) and all(a["bits"] == b["bits"] and a["type"] == b["type"] and abs(a["speed"] - b["speed"]) < 3e-6
                                        and a["sync_index"] == b["sync_index"] for a, b in zip(p_o, p_g))
    print("decode with --detect-speed: %d / %d patterns, identical %s, oracle %.2fs gpu %.4fs" % (len(p_o), len(p_g), same, t_o, t_g))
    hits = [p for p in p_g if p["bits"] == payload]
    print("  payload hits:", [(p["type"], round(p["speed"], 6), round(p["sync_quality"], 3)) for p in hits])
    if not same:
        for a, b in zip(p_o, p_g):
            print("   ", a["type"], a["bits"][:8], a["speed"], a["sync_index"], "|", b["type"], b["bits"][:8], b["speed"], b["sync_index"])
