import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""debug: the 200 s scenario of tests/test_cli_gpu.py through the binding, our get vs the compiled reference, full precision"""
import os, sys, subprocess, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, audiowmark_amd as awm, _ref
AWM = os.path.join(ROOT, "audiowmark_amd", "audiowmark")
PAY = "f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0"
noise = subprocess.run([AWM, "test-gen-noise", "-", "200", "44100"], stdout=subprocess.PIPE).stdout
open("/tmp/n.wav", "wb").write(noise)
for who, cmd in (("ours", [AWM, "add", "--format", "wav-pipe", "/tmp/n.wav", "-", PAY]), ("ref", [_ref.BIN, "add", "--format", "wav-pipe", "/tmp/n.wav", "-", PAY])):
    data = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    pos = data.index(b"data") + 8
    s = (np.frombuffer(data[pos:pos + (len(data) - pos) // 2 * 2], dtype="<i2").astype(np.float32) / 32768.0)
    ctx = awm.Context(0)
    a = ctx.get_watermark(None, torch.from_numpy(s.reshape(-1, 2)).cuda())
    b = _ref.get(None, s, 2)
    print(who, "marked file:", len(a), len(b))
    for p, q in zip(a, b):
        flag = "" if (p["sync_index"], p["bits"]) == (q["sync_index"], q["bits"]) and abs(p["decode_error"] - q["decode_error"]) < 1e-6 else "   <-- differs"
        print("  %8.3f %9d %.9f %.6f t%d b%d | %9d %.9f %.6f%s" % (p["time"], p["sync_index"], p["sync_quality"], p["decode_error"], p["type"], p["block_type"],
                                                                     q["sync_index"], q["sync_quality"], q["decode_error"], flag))
