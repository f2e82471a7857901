#!/usr/bin/env python3
"""bench.py -- audio seconds watermarked + decoded per wall-second (xRT), 44.1 kHz stereo.

One step = `add` (STFT -> band edit -> inverse -> overlap-add -> mix -> limiter) followed by `get`
(chunked SyncFinder search + refine, soft-bit extraction, Viterbi, merge) over one synthetic
60-minute stereo stream that is already resident in HBM (BASELINE.json configs[1]).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With N > 1 the stream is N x 60 minutes long and sharded across the ranks (audiowmark_amd.sharded:
frame spans for `add` with a 1-frame halo exchange and an all-reduce(max) of the limiter maxima,
reference chunks for `get` with an overlap exchange and a gather of the found patterns) -> weak scaling.

Rank 0 prints ONE JSON line.  `roofline` is for the kernel with the largest share of GPU time,
timed live with HIP events on the context's stream; `cpu_baseline` times the compiled reference
(oracle/_ref, kind "reference") or the restatement (oracle/, kind "port") on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

# `get` runs the chunks of a stream on concurrent HIP streams (lanes).  The runtime multiplexes all streams of the process
# onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue run back to back; with the collective
# library's own streams in the process two lanes ended up on one queue (multi-GPU step 9.0 ms instead of 7.5 ms).  Must be
# set before the HIP runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PAYLOAD = "0123456789abcdef0011223344556677"
RATE = 44100
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(sample_seconds):
    """Reference CPU path on a bounded sample of the same workload (stereo white noise)."""
    import numpy as np
    try:
        import _ref
        have_ref = _ref.available()
    except Exception:
        have_ref = False
    kind = None
    if have_ref:
        impl, kind = _ref, "reference"
    else:
        try:
            import _oracle
            impl, kind = _oracle, "port"
        except Exception:
            return None
    rng = np.random.default_rng(7)
    n = int(sample_seconds * RATE)
    x = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    t0 = time.perf_counter()
    w = impl.add(None, x, 2, PAYLOAD)
    t1 = time.perf_counter()
    pats = impl.get(None, w, 2)
    t2 = time.perf_counter()
    ok = any(p["bits"] == PAYLOAD for p in pats)
    cores = os.cpu_count() if kind == "reference" else 1
    return {"value": round(sample_seconds / (t2 - t0), 2), "unit": "xRT", "cores": cores, "kind": kind,
            "sample": f"{sample_seconds:.0f} s stereo 44.1 kHz white noise, add {t1 - t0:.2f} s (1 thread) + get {t2 - t1:.2f} s "
                      f"({cores} threads), payload recovered={ok}; FFTW replaced by the oracle's double-precision FFT"}


# HIP-event scope (awm_prof_name) -> device kernel whose PMC counters profiles/r01/traffic.json holds
PROF_TO_KERNEL = {
    "add_mix_kernel": "add_mix_kernel_w3<2>", "limiter_kernel": "limiter_apply_kernel<2>", "sync_db_kernel(approx)": "sync_db_kernel<2, false>",
    "sync_scan_kernel(approx)": "sync_scan_window_kernel", "local_mean_kernel": "local_mean_kernel",
    "sync_db_kernel(refine)": "sync_db_sliding_kernel<2>", "sync_scan_kernel(refine)": "sync_scan_gathered_kernel<false>",
    "sync_db_kernel(block)": "sync_db_kernel<2, true>", "soft_bits_kernel": "soft_bits_kernel", "viterbi_kernel": "viterbi_kernel",
}


def pmc_traffic(prof_name):
    """HBM bytes per launch of the kernel behind a profiling scope: FETCH_SIZE (x2, gfx950 calibration) + WRITE_SIZE from the
    separate rocprofv3 --pmc passes of this very command (tools/pmc_traffic.py -> profiles/r01/traffic.json; counters cannot
    be read from inside the process).  None if the summary is not there or was taken for another workload."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        e = t[PROF_TO_KERNEL[prof_name]]
        return int(e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"])
    except Exception:
        return None


def detect_speed_config(torch, awm, ctx, key, payload, minutes):
    """BASELINE.json configs[2] (reported next to the headline number, never part of `value`): `minutes` of stereo 48 kHz,
    watermarked at 48 kHz (timed), replayed 2 % fast (untimed: that is the attacker's part), then what `get --detect-speed`
    does with the 48 kHz file (timed): loader resampling to 44.1 kHz, speed search per 30-minute chunk, decode of the stream
    stretched back to speed 1 and of the original stream.  Everything resident in HBM."""
    rate, speed = 48000, 1.02
    n = int(minutes * 60 * rate)
    g = torch.Generator(device="cuda")
    g.manual_seed(4711)
    x = torch.rand((n, 2), generator=g, device="cuda", dtype=torch.float32) * 2 - 1

    def timed(fn, reps=3):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return out, best

    w, t_add = timed(lambda: ctx.add_watermark(key, payload, x, sample_rate=rate))
    del x
    fast = ctx.resample_ratio(w, 1 / speed, rate=rate)
    del w
    awm.set_speed_params(detect_speed=True)
    try:
        pats, t_get = timed(lambda: ctx.get_watermark(key, ctx.resample(fast, rate, 44100)))
    finally:
        awm.set_speed_params()
    hits = [p for p in pats if p["bits"] == payload]
    speeds = sorted({round(p["speed"], 6) for p in hits})
    seconds = fast.shape[0] / rate
    return {"workload": "%g min stereo 48 kHz: add at 48 kHz, replay at speed %.2f, get --detect-speed (loader resampling, speed "
                        "search per chunk, decode of the stretched + the plain stream)" % (minutes, speed),
            "value": round((n / rate + seconds) / 2 / (t_add + t_get), 1), "unit": "xRT",
            "add_ms": round(t_add * 1e3, 3), "get_detect_speed_ms": round(t_get * 1e3, 3),
            "payload_matches": len(hits), "detected_speeds": speeds}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--minutes", type=float, default=60.0, help="audio minutes per GPU")
    ap.add_argument("--cpu-sample-seconds", type=float, default=200.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-detect-speed-config", action="store_true", help="skip the 48 kHz --detect-speed configuration (BASELINE configs[2])")
    ap.add_argument("--sharded", action="store_true",
                    help="debug: take the multi-GPU (ShardedStream / torch.distributed) code path even with one process")
    args = ap.parse_args()

    import torch
    import audiowmark_amd as awm

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the watermark path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    sharded_path = world > 1 or args.sharded
    if sharded_path:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    n = int(args.minutes * 60 * RATE)
    if sharded_path:
        n -= n % 1024            # spans of a sharded stream are whole frames (only the last one may be ragged)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    x = torch.rand((n, 2), generator=gen, device=dev, dtype=torch.float32) * 2 - 1   # test-gen-noise distribution
    out = torch.empty_like(x)
    ctx = awm.Context(local_rank)

    if sharded_path:
        from audiowmark_amd import sharded
        pipe = sharded.ShardedStream(ctx, dist, n_frames_local=n, n_channels=2)

        def step():
            pipe.add_watermark(None, PAYLOAD, x, out)
            return pipe.get_watermark(None, out)
    else:
        def step():
            ctx.add_watermark(None, PAYLOAD, x, out=out)
            return ctx.get_watermark(None, out)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    awm.lib.awm_prof_reset(ctx._h)
    awm.lib.awm_prof_enable(ctx._h, 1)
    sync()
    t0 = time.perf_counter()
    pats = None
    trace = os.environ.get("AWM_BENCH_TRACE")
    for i in range(args.steps):
        ts = time.perf_counter()
        pats = step()
        if trace:
            torch.cuda.synchronize(dev)
            print(f"step {i}: {1e3 * (time.perf_counter() - ts):.2f} ms", file=sys.stderr)
    sync()
    elapsed = time.perf_counter() - t0
    awm.lib.awm_prof_enable(ctx._h, 0)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel HIP-event times of the timed region (this rank).  `get` runs the chunks of the stream on concurrent
    # lanes, so these durations overlap (their sum exceeds the wall time) and each is stretched by the kernels it
    # shares the GPU with.
    import ctypes as C
    awm.lib.awm_prof_name.restype = C.c_char_p

    def read_prof():
        prof = []
        for i in range(awm.lib.awm_prof_count()):
            ms, launches, nbytes = C.c_double(), C.c_long(), C.c_double()
            awm.lib.awm_prof_read(ctx._h, i, C.byref(ms), C.byref(launches), C.byref(nbytes))
            if launches.value:
                prof.append((awm.lib.awm_prof_name(i).decode(), ms.value, launches.value, nbytes.value))
        return prof

    prof = read_prof()
    # the same kernels one after the other (AWM_ONE_LANE: a single stream), outside the timed region: the duration a
    # kernel has when it owns the GPU -- the number a roofline is about
    serial_steps = 3
    os.environ["AWM_ONE_LANE"] = "1"
    step()
    awm.lib.awm_prof_reset(ctx._h)
    awm.lib.awm_prof_enable(ctx._h, 1)
    for _ in range(serial_steps):
        step()
    sync()
    awm.lib.awm_prof_enable(ctx._h, 0)
    serial = {p[0]: p for p in read_prof()}
    del os.environ["AWM_ONE_LANE"]

    if rank == 0:
        audio_seconds = n * world / RATE
        matches = sum(1 for p in (pats or []) if p["bits"] == PAYLOAD)
        total_ms = sum(p[1] for p in prof) or 1.0
        dom = max(prof, key=lambda p: p[1]) if prof else None
        roofline = None
        if dom:
            name, ms, launches, nbytes = dom
            achieved = nbytes / (ms * 1e-3) / 1e9        # algorithmic GB/s: sum(bytes)/sum(time) == per-launch bytes / avg duration
            roofline = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(name) if args.minutes == 60.0 else None,
                        "launches": launches, "avg_ms": round(ms / launches, 4),
                        "share_of_gpu_time": round(ms / total_ms, 3)}
            if name in serial:
                _, sms, sl, sb = serial[name]
                roofline["alone_avg_ms"] = round(sms / sl, 4)            # not sharing the GPU with the other lanes' kernels
                roofline["alone_achieved"] = round(sb / (sms * 1e-3) / 1e9, 1)
                roofline["alone_frac"] = round(sb / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                if name.startswith("sync_scan_kernel(approx)"):
                    # What limits this kernel is not HBM: every candidate start gathers its 510 sync rows x 60 bands from the
                    # dB tile in LDS (ds_read_b32, ~75 TB/s for the whole chip per the microarchitecture guide).  Candidates per
                    # launch from the algorithmic bytes (324 B per frame and shift; a candidate needs a whole block after it).
                    frame_shifts = sb / 324.0
                    candidates = max(frame_shifts - 4 * 2226 * sl, 0.0)
                    lds_tbps = candidates * 510 * 60 * 4 / (sms * 1e-3) / 1e12
                    roofline["lds_gather"] = {"achieved": round(lds_tbps, 1), "peak": 75.0, "unit": "TB/s", "frac": round(lds_tbps / 75.0, 3),
                                              "note": "sequential float sums in the reference's order: 60 gathered adds per sync row and candidate"}
        res = {
            "metric": "audio seconds watermarked+decoded per wall-second (xRT), 44.1 kHz stereo",
            "value": round(audio_seconds * args.steps / elapsed, 1),
            "unit": "xRT",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.minutes:g} min stereo 44.1 kHz white noise per GPU resident in HBM, add+get incl. payload decode, "
                                   f"payload {PAYLOAD}, strength 10", "parallelism": f"stream sharded over {world} GPU(s)" if world > 1 else "1 GPU",
                       "patterns": len(pats or []), "payload_matches": matches},
            "roofline": roofline,
            # the batched STFT north_star puts the 40 % HBM target on: the fused add kernel (STFT + band edit + inverse +
            # overlap-add), 8192 algorithmic bytes per frame-channel, stand-alone duration
            "stft_roofline": ({"kernel": "add_mix_kernel", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                               "achieved": round(serial["add_mix_kernel"][3] / (serial["add_mix_kernel"][1] * 1e-3) / 1e9, 1),
                               "frac": round(serial["add_mix_kernel"][3] / (serial["add_mix_kernel"][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "avg_ms": round(serial["add_mix_kernel"][1] / serial["add_mix_kernel"][2], 4),
                               "traffic": pmc_traffic("add_mix_kernel") if args.minutes == 60.0 else None}
                              if "add_mix_kernel" in serial else None),
            "kernels_ms_per_step": {p[0]: round(p[1] / args.steps, 3) for p in prof},
            # one stream, kernels back to back (untimed extra pass): true per-kernel cost; their sum is what a step
            # would take without the concurrent lanes
            "kernels_ms_per_step_alone": {k: round(v[1] / serial_steps, 3) for k, v in serial.items()},
            # algorithmic GB/s (SURVEY.md 8d bytes / HIP-event time) of every kernel, same definition as roofline.achieved
            "kernels_algorithmic_GBps": {k: round(v[3] / (v[1] * 1e-3) / 1e9, 1) for k, v in serial.items() if v[1] > 0},
        }
        if world == 1 and not args.no_detect_speed_config:
            try:
                res["detect_speed_config"] = detect_speed_config(torch, awm, ctx, None, PAYLOAD, args.minutes)
            except Exception as e:                       # reported, never fatal for the headline line
                res["detect_speed_config"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.cpu_sample_seconds)
        else:
            res["cpu_baseline"] = None
        # RCCL prints its version banner through C stdio at exit; flush it first so that the JSON is the last line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
