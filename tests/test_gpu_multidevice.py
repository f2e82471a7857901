"""Multi-GPU paths on DISTINCT devices.  These tests switch themselves on: on a box with one GPU (every gpurun box so far) they skip,
on a node with two or more they are the first run of
  * hipMemcpyPeer between different devices and the helper contexts of awm_multi_* (host/capi_shard.cc),
  * the RCCL transport with world > 1 (audiowmark_amd/sharded.py: batch_isend_irecv / all_reduce ON the context's stream),
  * `bench.py --gpus 2` as the driver launches it, and AWM_DEVICES=0,1 through the command line.
Every result must equal the single-GPU result bit for bit (PCM) / line by line (pattern lists incl. quality and error values)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAY1 = "0123456789abcdef0011223344556677"


def _device_count():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs (skips on the one-GPU boxes)")


def noise(seed, n, ch):
    return np.random.default_rng(seed).uniform(-1, 1, (n, ch)).astype(np.float32)


def pkey(p):
    return (round(p["time"], 9), p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"])


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@needs_two
@pytest.mark.parametrize("n_dev", [2, 4])
def test_multi_context_on_distinct_devices_equals_single(n_dev):
    """awm_multi_add_d / awm_multi_get_d with context i on device i: spans cut inside chunks (10 minute chunks), PCM and pattern list of the
    single-GPU calls"""
    import torch
    import audiowmark_amd as awm
    from audiowmark_amd import sharded
    if torch.cuda.device_count() < n_dev:
        pytest.skip(f"needs {n_dev} GPUs")
    awm.set_params(chunk_size_min=10.0)
    ctxs = [awm.Context(d) for d in range(n_dev)]
    try:
        total = 33 * 60 * 44100 + 777
        whole_h = noise(171, total, 2)
        cuts = [0] + [int(total * (i + 1) / n_dev * (0.93 if i % 2 else 1.04)) // 1024 * 1024 for i in range(n_dev - 1)] + [total]
        spans = [torch.from_numpy(whole_h[a:b].copy()).to(f"cuda:{d}") for d, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
        outs = [torch.empty_like(s) for s in spans]
        sharded.multi_add(ctxs, None, PAY1, spans, outs)
        whole = torch.from_numpy(whole_h).to("cuda:0")
        want = ctxs[0].add_watermark(None, PAY1, whole)
        got = torch.cat([o.to("cuda:0") for o in outs])
        assert torch.equal(got, want)
        pats = sharded.multi_get(ctxs, None, outs)
        want_pats = ctxs[0].get_watermark(None, want)
        assert [pkey(p) for p in pats] == [pkey(p) for p in want_pats]
        assert sum(p["bits"] == PAY1 for p in pats) >= 30
    finally:
        awm.set_params()
        for c in ctxs:
            c.close()


@needs_two
def test_clip_batches_on_distinct_devices_equal_single():
    import torch
    import audiowmark_amd as awm
    from audiowmark_amd import sharded
    ctxs = [awm.Context(0), awm.Context(1)]
    try:
        n = 24
        owner = [i % 2 for i in range(n)]
        host = [noise(700 + i, 25 * 44100 + 11 * i, 2) for i in range(n)]
        keys = [awm.test_key(i + 1) for i in range(n)]
        clips = [torch.from_numpy(h).to(f"cuda:{owner[i]}") for i, h in enumerate(host)]
        on0 = [torch.from_numpy(h).to("cuda:0") for h in host]
        want_pcm = ctxs[0].add_watermark_batch_keys(keys, PAY1, on0)
        got_pcm = sharded.multi_add_batch(ctxs, keys, PAY1, clips, owner)
        assert all(torch.equal(a.to("cuda:0"), b) for a, b in zip(got_pcm, want_pcm))
        want = ctxs[0].get_watermark_batch_keys(keys, want_pcm)
        got = sharded.multi_get_batch(ctxs, keys, got_pcm, owner)
        assert got == want
        assert sum(any(p["bits"] == PAY1 for p in g) for g in got) >= 20
    finally:
        for c in ctxs:
            c.close()


def _rccl_worker(rank, world, port, lengths, q):
    import torch
    import torch.distributed as dist
    import audiowmark_amd as awm
    from audiowmark_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        awm.set_params(chunk_size_min=10.0)
        whole = noise(197, sum(lengths), 2)
        s = sum(lengths[:rank])
        local = torch.from_numpy(whole[s:s + lengths[rank]].copy()).cuda()
        # (a side stream as the context's stream: the transport must order its transfers on THAT stream, not on torch's current one)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            ctx = awm.Context(rank)
        pipe = sharded.ShardedStream(ctx, dist, lengths[rank], 2)
        out = torch.empty_like(local)
        torch.cuda.synchronize()
        pipe.add_watermark(None, PAY1, local, out)
        pats = pipe.get_watermark(None, out)
        torch.cuda.synchronize()
        q.put((rank, "ok", out.cpu().numpy(), pats, pipe.comm.host_group is not None))
    except Exception:
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


@needs_two
def test_rccl_transport_two_ranks_equal_single():
    """one process per GPU over RCCL (the bench's N > 1 path): PCM bit for bit, merged pattern list identical"""
    import torch
    import torch.multiprocessing as mp
    import audiowmark_amd as awm
    lengths = [14 * 60 * 44100 // 1024 * 1024, 9 * 60 * 44100 + 333]
    port = free_port()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_rccl_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=900) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[2]
    awm.set_params(chunk_size_min=10.0)
    ctx = awm.Context(0)
    try:
        whole = torch.from_numpy(noise(197, sum(lengths), 2)).cuda()
        want = ctx.add_watermark(None, PAY1, whole)
        got = np.concatenate([results[0][2], results[1][2]])
        assert np.array_equal(got, want.cpu().numpy())
        want_pats = ctx.get_watermark(None, want)
        assert [pkey(p) for p in results[0][3]] == [pkey(p) for p in want_pats]
        assert results[1][3] is None
        assert results[0][4] == results[1][4]                      # both ranks agree on whether the gloo side group is in use
    finally:
        awm.set_params()
        ctx.close()


@needs_two
def test_bench_two_gpus_over_rccl():
    """the driver's literal command for N = 2, three steps: two ranks seen, every watermark block of the 2 x 60 min stream found"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2 and line["value"] > 0
    assert line["config"]["payload_matches"] >= 200                      # 2 h of audio: 139 blocks + AB pairs + the chunks' "all" patterns
    assert line["config"].get("rccl_version"), "the two ranks must have talked over RCCL"
    plan = line["config"]["shard_plan"]                                  # what every rank held, worked on and sent (the last timed step)
    assert len(plan["per_rank"]) == 2 and plan["chunks_shared_by_several_ranks"] >= 1 and "RCCL" in plan["transport"]
    assert all(r["start_frames"] > 0 and r["sent_in_the_last_step"]["device_bytes_sent"] > 0 for r in plan["per_rank"])


@needs_two
def test_cli_get_over_two_devices(tmp_path):
    """AWM_DEVICES=0,1: the command line spreads a long stream over both GPUs; same report as on one"""
    awm_bin = os.path.join(ROOT, "audiowmark_amd", "audiowmark")
    raw = ["--input-format", "raw", "--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16"]

    def run(cmd, **kw):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)
        assert r.returncode == 0, r.stderr
        return r

    wav = run([awm_bin, "test-gen-noise", "-", "780", "44100"]).stdout
    samples = wav[wav.index(b"data") + 8:]                                           # 13 min: > 4 blocks per GPU
    marked = tmp_path / "m.raw"
    marked.write_bytes(run([awm_bin, "add", "-q", "--format", "raw", "--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16",
                            "-", "-", PAY1], input=samples).stdout)
    one = run([awm_bin, "cmp"] + raw + [str(marked), PAY1]).stdout.decode().splitlines()
    assert sum(l.startswith("pattern") and PAY1 in l for l in one) >= 20
    two = run([awm_bin, "cmp"] + raw + [str(marked), PAY1], env=dict(os.environ, AWM_DEVICES="0,1")).stdout.decode().splitlines()
    assert two == one
