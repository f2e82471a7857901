#!/usr/bin/env python3
"""bench.py -- audio seconds watermarked + decoded per wall-second (xRT), 44.1 kHz stereo.

One step = `add` (STFT -> band edit -> inverse -> overlap-add -> mix -> limiter) followed by `get` (chunked SyncFinder
search + refine, soft-bit extraction, Viterbi, merge) over synthetic white noise that is already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W                     BASELINE.json configs[1]: 60 min stereo per GPU
  python bench.py --gpus N --steps K --warmup W                     N > 1 without a launcher: bench.py starts its N ranks itself
                                                                    (re-executes under torch.distributed.run, 127.0.0.1 rendezvous)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W     (ranks from the launcher)
  ... bench.py --gpus N --config 8h                                 configs[3]: ONE 8 h stream over the N ranks (strong scaling)
  ... bench.py --gpus N --config clips                              configs[4]: 1024 clips of 30 s over the N ranks (replicas)

--config 60min with N > 1: the stream is N x 60 minutes long and split BY POSITION into one span per rank (awm_sharded_add_d /
awm_sharded_get_d behind audiowmark_amd.sharded.TorchComm): `add` exchanges one frame with each neighbour
and max-reduces the per-second limiter maxima; `get` keeps the reference's 30-minute chunks as the unit of meaning, every rank
scores / refines / decodes the start frames inside its span, the ranks that share a chunk exchange one block of overlap, their
score segments, the refined candidates and the raw soft bits, rank 0 merges the pattern records -> weak scaling.

Rank 0 prints ONE JSON line.  At N = 1 (60min) it also carries:
  roofline       the single DEVICE KERNEL with the largest stand-alone share of GPU time (what `rocprofv3 --kernel-trace --stats` of
                 the one-lane pass ranks first): algorithmic bytes per launch / its average launch duration when the chunks run
                 one after the other (awm_ctx_set_chunk_lanes (1), untimed extra pass), HBM traffic per launch from the PMC passes
                 of the same command.  Scopes that bracket several launches (limiter, local mean, refinement scan, Viterbi) are
                 listed separately in scopes_ms_per_step and never picked
  e2e            file -> file through the bounded-memory path (s16 raw in the page cache), PCIe-staged, CLI resident set size
  cpu_baseline   the compiled reference (oracle/_ref) on the host cores, on a 30 min sample, and on the same sample
  parity         the HIP path's PCM / pattern list against that reference run ("parity_checked_against": "reference")
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
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


def newest_profile_dir():
    """profiles/rNN with the highest NN that holds a traffic.json (the PMC passes of tools/profile_bench.sh)"""
    base = os.path.join(ROOT, "profiles")
    best = None
    for d in sorted(os.listdir(base)) if os.path.isdir(base) else []:
        if d.startswith("r") and d[1:].isdigit() and os.path.exists(os.path.join(base, d, "traffic.json")):
            best = os.path.join(base, d)
    return best


PROFILE_DIR = newest_profile_dir()

# scopes whose two HIP events bracket exactly ONE launch of ONE device kernel: only these can be the `roofline` kernel
SINGLE_KERNEL_SCOPES = ("add_mix_kernel", "sync_db_kernel(approx)", "sync_scan_kernel(approx)", "sync_db_kernel(refine)", "sync_db_kernel(block)",
                        "soft_bits_kernel")
# launches per scope of the others (the limiter: per-second table + apply; local mean + peak selection; refinement scan: chains +
# qualities; Viterbi: decoder input preparation + the chain of 14 launches or the one-launch kernel, see viterbi_form)
SCOPE_LAUNCHES = {"limiter_kernel": 2, "local_mean_kernel": 3, "sync_scan_kernel(refine)": 2, "viterbi_kernel": "15 (chain) | 2 (one launch)"}

# HIP-event scope (awm_prof_name) -> (device kernel in the rocprofv3 summaries, what actually limits it)
KERNELS = {
    "add_mix_kernel": ("add_mix_pair_kernel", "HBM latency <-> FP32 issue (1 450 VALU instructions per stereo frame, 4 waves / SIMD; both channels' transforms pipelined over one LDS tile)"),
    "limiter_kernel": ("limiter_apply_kernel<2>", "HBM"),
    "sync_db_kernel(approx)": ("sync_db_kernel<2, false, 33>", "FP32 issue and LDS round trips in turn (VALU issues 55 - 60 % of the time; both channels' transforms pipelined over one LDS tile; the 4 shifts of a tile share one XCD's L2: PCM read once)"),
    "sync_scan_kernel(approx)": ("sync_scan_stream_kernel<false>", "works out of LDS, not HBM: 30 ds_read_b128 gathers + 120 float additions per sync frame and wave in the reference's summation order (LDS active 51 %, VALU issuing 42 % of the cycles); 3.3 rounds of tiles"),
    "local_mean_kernel": ("local_mean_kernel", "latency"),
    "sync_db_kernel(refine)": ("sync_db_sliding4_kernel", "VALU issue (half of it FP64: 69 of ~115 instructions per wave and fine offset) + the first transform in double per row (a third of the duration); three bins of one channel per lane, 56 of 64 lanes busy, 3 waves / SIMD"),
    "sync_scan_kernel(refine)": ("sync_scan_gathered_kernel<false>", "HBM latency (300 single-wave workgroups, 60 loads in flight each; rows of 64 offsets = two whole cache lines)"),
    "sync_db_kernel(block)": ("sync_db_kernel<2, true, 33>", "FP32 issue and LDS round trips in turn"),
    "soft_bits_kernel": ("soft_bits_wave_kernel", "latency of scattered reads + sequential double precision sums (four bits per wave)"),
    "resample_kernel": ("resample_phase_kernel<147, 160, 18> / <160, 147, 16> (stereo 48 <-> 44.1 kHz) | resample_kernel",
                        "LDS window reads <-> VALU issue: a thread owns one phase; per output 18 ds_read2_b64 + 146 VALU (zita's 8 unfused operations per tap pair)"),
    "resample_var_kernel": ("resample_var_kernel", "VALU issue <-> LDS coefficient reads: 17 VALU (zita's 14 roundings + 3) and 3 LDS instructions per stereo tap pair"),
    "speed_mags_kernel": ("speed_mags_kernel", "LDS gathers in the reference's summation order (60 per time step and sync frame) + load latency"),
    "speed_compare_kernel": ("speed_compare_kernel", "VALU issue: 7 instructions per (column, speed); the matrix is served by L2 (11 relative speeds per centre share it)"),
    "frame_mod_table_kernel": ("frame_mod_table_kernel (add: K16) | clip_key_table_kernel (get: K16g)", "the serial swap chain of the 51 480-entry shuffle: LDS round trips (64 / 16 / 4 / 1 swaps per trip)"),
    "viterbi_kernel": ("viterbi_super_kernel<0> (chain of 14 launches) | viterbi_persistent_kernel (one launch): see viterbi_form",
                       "latency: 143 dependent trellis steps, a chunk's ~37 decodes are one wave per SIMD; 8 workgroups per decode exchange their metrics every 12 steps "
                       "-- through 14 dependent launches where launches are cheap on the host, through per-decode counters inside ONE launch where they are not"),
}


def _code_only(text):
    """C / C++ source without comments and with runs of white space collapsed (string and character literals kept as they are)"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def hip_sources_digest():
    """sha256 over the CODE of the device sources (audiowmark_amd/csrc/hip/*, comments and white space aside): what
    `profiles/rNN/HIP_SOURCES_SHA256` records when the PMC passes are taken (tools/gpu_final.sh), so that a traffic figure is never
    reported for kernels that changed afterwards (no git on the GPU box)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "audiowmark_amd", "csrc", "hip")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".hh")):
            h.update(name.encode())
            with open(os.path.join(d, name), "r", encoding="utf-8", errors="replace") as f:
                h.update(_code_only(f.read()).encode())
    return h.hexdigest()


def traffic_is_stale():
    """True if the newest profile directory's PMC passes were taken with other device sources than the ones in the tree (or does not say)"""
    if not PROFILE_DIR:
        return True
    try:
        recorded = open(os.path.join(PROFILE_DIR, "HIP_SOURCES_SHA256")).read().split()[0]
    except Exception:
        return True
    return recorded != hip_sources_digest()


def pmc_traffic(prof_name, minutes):
    """HBM bytes per launch of the kernel behind a profiling scope: FETCH_SIZE (x2, gfx950 calibration) + WRITE_SIZE from the
    separate rocprofv3 --pmc passes of this very command (tools/pmc_traffic.py -> profiles/r02/traffic.json; counters cannot be
    read from inside the process).  None if the summary is not there or was taken for another workload."""
    if minutes != 60.0 or not PROFILE_DIR or traffic_is_stale():
        return None
    try:
        with open(os.path.join(PROFILE_DIR, "traffic.json")) as f:
            t = json.load(f)
        e = t[KERNELS[prof_name][0]]
        return int(e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"])
    except Exception:
        return None


def traffic_provenance():
    """which profile directory roofline.traffic comes from and the commit its PMC passes were taken at (profiles/rNN/COMMIT)"""
    if not PROFILE_DIR:
        return None
    try:
        commit = open(os.path.join(PROFILE_DIR, "COMMIT")).read().split()[0]
    except Exception:
        commit = None
    stale = traffic_is_stale()
    return {"profile_dir": os.path.relpath(PROFILE_DIR, ROOT), "traffic_from_commit": commit, "traffic_stale": stale,
            "note": ("the device sources changed since these PMC passes (or the directory does not record their digest): every `traffic` is null"
                     if stale else "HBM bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE x 2, WRITE_SIZE) of this command with these device sources")}


def quantise16(np, x):
    """what reading back the 16 bit WAV of `audiowmark test-gen-noise` gives (truncation towards zero, / 32768)"""
    return (np.clip(np.trunc(x.astype(np.float64) * 32768.0), -32768, 32767) / 32768.0).astype(np.float32)


def cpu_baseline_and_parity(torch, awm, ctx, sample_seconds):
    """The compiled reference (oracle/_ref: unmodified reference sources, `add` single threaded, `get` on all host cores --
    exactly what `audiowmark add` / `audiowmark get` do) on a bounded sample of the workload (test-gen-noise input), timed;
    then the HIP path on the same sample, compared with it (untimed)."""
    import numpy as np
    try:
        import _ref
        have_ref = _ref.available()
    except Exception:
        have_ref = False
    if have_ref:
        impl, kind = _ref, "reference"
    else:
        try:
            import _oracle as impl
            kind = "port"
        except Exception:
            return None, None
    n = int(sample_seconds * RATE)
    x = quantise16(np, awm.binding.gen_noise(None, 2 * n))
    visible = os.cpu_count() or 1
    # what the process may actually use: the box gives a container a CPU quota (cgroup cpu.max, e.g. "1600000 100000" = 16 cores)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    usable = visible if quota is None else max(1, min(visible, int(round(quota))))

    def timed_run(upstream_threads_too):
        """add (single threaded, as upstream) + get; get with one worker per core the process may USE (oracle/ref_shim/threads_shim.cc; the
        reference's sources are unchanged) and -- once -- as upstream starts its pool: one worker per VISIBLE core (threadpool.cc:58-63)"""
        impl.add(None, x[:4 * RATE], 2, PAYLOAD)            # (untimed: the library is loaded and its FFT set up, as the GPU side's warm-up steps)
        t0 = time.perf_counter()
        w = impl.add(None, x, 2, PAYLOAD)
        t_add = time.perf_counter() - t0
        r = {"t_add": t_add, "w": w}
        if kind == "reference" and usable < visible:
            os.environ["AWM_REF_THREADS"] = str(usable)
        try:
            t1 = time.perf_counter()
            r["pats"] = impl.get(None, w, 2)
            r["t_get"] = time.perf_counter() - t1
        finally:
            os.environ.pop("AWM_REF_THREADS", None)
        r["threads"] = (usable if usable < visible else visible) if kind == "reference" else 1
        if upstream_threads_too and kind == "reference" and usable < visible:
            t1 = time.perf_counter()
            pats_u = impl.get(None, w, 2)
            r["t_get_upstream"] = time.perf_counter() - t1
            r["upstream_same"] = [(p["sync_index"], p["bits"]) for p in pats_u] == [(p["sync_index"], p["bits"]) for p in r["pats"]]
        return r

    def line(r, fft):
        ok = sum(p["bits"] == PAYLOAD for p in r["pats"])
        return {"value": round(sample_seconds / (r["t_add"] + r["t_get"]), 2), "unit": "xRT", "cores": min(r["threads"], usable), "kind": kind,
                "fft": fft,
                "sample": f"{sample_seconds / 60:.0f} min stereo 44.1 kHz test-gen-noise" + (" (= the whole of configs[1])" if sample_seconds == 3600 else "") +
                          f", add {r['t_add']:.2f} s (1 thread) + get {r['t_get']:.2f} s "
                          f"({r['threads']} threads" + (f", CPU quota of the process {quota:g} cores" if quota is not None else "") +
                          f"), {ok} of {len(r['pats'])} patterns carry the payload; FFT behind fftw3.h: {fft}"}

    # the build the parity bars are pinned to: FFTW (absent from the image) replaced by the oracle's double-precision FFT
    run_d = timed_run(upstream_threads_too=True)
    base_d = line(run_d, "double-precision FFT stand-in, rounded once (oracle/ref_shim/fftw_shim.cc)")
    base = base_d
    # the headline baseline: the same unmodified sources with a FLOAT FFT behind fftw3.h, which is what FFTW's fftwf_* is
    # (reference fft.cc:63,85): MKL's single-precision FFTW3 wrapper (oracle/_ref/libawm_ref_mkl.so, `make -C oracle ref_mkl`)
    if kind == "reference" and os.path.exists(_ref.PATH_MKL):
        os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
        try:
            _ref.use_backend("mkl")
            run_m = timed_run(upstream_threads_too=False)
            base = line(run_m, "MKL single-precision FFTW3 wrapper (float, like FFTW's fftwf_*)")
            base["with_double_fft_stand_in"] = {k: base_d[k] for k in ("value", "cores", "sample")}
            same = [(p["sync_index"], p["type"], p["block_type"], p["bits"]) for p in run_m["pats"]] == \
                   [(p["sync_index"], p["type"], p["block_type"], p["bits"]) for p in run_d["pats"]]
            base["patterns_equal_to_the_double_fft_build"] = bool(same)
        except Exception as e:                             # (MKL missing on the box: the double build stays the baseline)
            base["float_fft_build"] = f"unavailable: {e}"
        finally:
            _ref.use_backend("double")
    if "t_get_upstream" in run_d:
        base["with_upstream_thread_count"] = {"value": round(sample_seconds / (run_d["t_add"] + run_d["t_get_upstream"]), 2),
                                              "get_s": round(run_d["t_get_upstream"], 2), "threads": visible, "fft": "double-precision stand-in",
                                              "same_patterns": bool(run_d["upstream_same"]),
                                              "note": "one worker per visible core, as upstream starts them (oversubscribed under the quota)"}
    w, pats = run_d["w"], run_d["pats"]
    # parity of the HIP path against this very run
    xd = torch.from_numpy(x.reshape(n, 2)).cuda()
    wg = ctx.add_watermark(None, PAYLOAD, xd).cpu().numpy().ravel()
    d = wg.astype(np.float64) - w.astype(np.float64)
    got = ctx.get_watermark(None, torch.from_numpy(w.reshape(n, 2)).cuda())
    key = lambda p: (round(p["time"], 6), p["sync_index"], p["type"], p["block_type"])
    # STRICT: every pattern at the reference's sync index.  A refinement tie (neighbouring fine offsets whose qualities differ by
    # less than the float pipelines' rounding) would show as refinement_ties > 0 AND positions_equal false.
    tie = lambda a, b: (key(a) != key(b) and (a["type"], a["block_type"], a["bits"]) == (b["type"], b["block_type"], b["bits"])
                        and abs(int(a["sync_index"]) - int(b["sync_index"])) <= 8)
    ties = sum(tie(a, b) for a, b in zip(got, pats)) if len(got) == len(pats) else 0
    same_pos = len(got) == len(pats) and all(key(a) == key(b) for a, b in zip(got, pats))
    watermark_bits_equal = same_pos and all(a["bits"] == b["bits"] for a, b in zip(got, pats) if b["decode_error"] < 0.6)
    parity = {"parity_checked_against": kind + (" (double-precision FFT build: the one the test bars are pinned to)" if kind == "reference" else ""),
              "sample": base["sample"].split(",")[0],
              "pcm_rms_diff": float(np.sqrt(np.mean(d * d))), "pcm_max_abs_diff": float(np.abs(d).max()),
              "patterns": len(pats), "pattern_positions_and_types_equal": bool(same_pos), "refinement_ties": int(ties),
              "payload_bits_equal_for_every_watermark": bool(watermark_bits_equal),
              "noise_patterns_with_other_bits": int(sum(a["bits"] != b["bits"] for a, b in zip(got, pats))) if same_pos else None,
              "max_abs_sync_quality_diff": max((abs(a["sync_quality"] - b["sync_quality"]) for a, b in zip(got, pats)), default=0.0) if same_pos else None}
    return base, parity


def detect_speed_config(torch, awm, ctx, key, payload, minutes, lanes=4):
    """BASELINE.json configs[2] (reported next to the headline number, never part of `value`): `minutes` of stereo 48 kHz,
    watermarked at 48 kHz (timed), replayed 2 % fast (untimed: that is the attacker's part), then what `get --detect-speed`
    does with the 48 kHz file (timed): loader resampling to 44.1 kHz, speed search per 30-minute chunk, decode of the stream
    stretched back to speed 1 and of the original stream.  Everything resident in HBM.  The zita-resampler stages follow a
    restatement of the library (absent from the reference tree): parity with the real library is unpinned."""
    rate, speed = 48000, 1.02
    n = int(minutes * 60 * rate)
    g = torch.Generator(device="cuda")
    g.manual_seed(4711)
    x = torch.rand((n, 2), generator=g, device="cuda", dtype=torch.float32) * 2 - 1

    first = {}

    def timed(fn, reps=3, what=None):
        for i in range(3):                                 # untimed, like the headline's warm-up steps; the FIRST call is reported on its
            torch.cuda.synchronize()                       # own (first_call_ms: workspaces are sized from the stream's upper bounds when a
            t0 = time.perf_counter()                       # context first sees a stream length, so it should be close to the steady state)
            fn()
            torch.cuda.synchronize()
            if i == 0 and what:
                first[what] = round((time.perf_counter() - t0) * 1e3, 3)
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return out, best

    w, t_add = timed(lambda: ctx.add_watermark(key, payload, x, sample_rate=rate), what="add_48k")
    del x
    fast = ctx.resample_ratio(w, 1 / speed, rate=rate)
    del w
    awm.set_speed_params(detect_speed=True)
    kernels = None
    try:
        pats, t_get = timed(lambda: ctx.get_watermark(key, ctx.resample(fast, rate, 44100)), what="get_detect_speed")
        # stand-alone durations: ONE lane, the plain decode after the speed part instead of beside it, per-kernel HIP events
        # (the rocprofv3 summary of the same pass: profiles/rNN/rocprofv3_kernel_stats_config2_one_lane.csv)
        calls = 2
        awm.lib.awm_ctx_set_chunk_lanes(ctx._h, 1)
        awm.lib.awm_debug_set_speed_overlap(0)
        try:
            ctx.get_watermark(key, ctx.resample(fast, rate, 44100))
            awm.lib.awm_prof_reset(ctx._h)
            awm.lib.awm_prof_enable(ctx._h, 1)
            for _ in range(calls):
                ctx.get_watermark(key, ctx.resample(fast, rate, 44100))
            torch.cuda.synchronize()
            awm.lib.awm_prof_enable(ctx._h, 0)
            kernels = scope_table(read_prof(awm, ctx), calls)
        finally:
            awm.lib.awm_debug_set_speed_overlap(1)
            awm.lib.awm_ctx_set_chunk_lanes(ctx._h, lanes)
            awm.lib.awm_prof_reset(ctx._h)
    finally:
        awm.set_speed_params()
    hits = [p for p in pats if p["bits"] == payload]
    speeds = sorted({round(p["speed"], 6) for p in hits})
    seconds = fast.shape[0] / rate
    return {"workload": "%g min stereo 48 kHz: add at 48 kHz, replay at speed %.2f, get --detect-speed (loader resampling, speed "
                        "search per chunk, decode of the stretched + the plain stream)" % (minutes, speed),
            "value": round((n / rate + seconds) / 2 / (t_add + t_get), 1), "unit": "xRT",
            "add_ms": round(t_add * 1e3, 3), "get_detect_speed_ms": round(t_get * 1e3, 3),
            "first_call_ms": first,
            "payload_matches": len(hits), "detected_speeds": speeds, "zita_resampler": "restated (parity with the library unpinned)",
            "kernels_one_lane": kernels}


def e2e_leg(torch, awm, ctx, x, resident_ms):
    """End to end for the same 60 min stream: (1) s16 raw file in the page cache -> watermarked s16 raw file -> decoded, through
    the file-level entry points (tile loop / chunked staging, bounded host memory); (2) the same with the files replaced by
    page-locked host buffers (PCIe both ways, no file system); (3) the command line binary on the same file: wall time including
    process start and HIP initialisation, and its peak resident set size."""
    import numpy as np
    n = x.shape[0]
    seconds = n / RATE
    d = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    src, dst, dst2 = (os.path.join(d, "awm_bench_%d_%s.raw" % (os.getpid(), s)) for s in ("in", "out", "cli"))
    out = {"workload": "%g min stereo 44.1 kHz s16 raw, file -> file (add), file -> patterns (get), files in %s" % (seconds / 60, d)}
    try:
        raw = ctx.pcm_encode(x.reshape(-1), 16, 0, False, True)
        raw.cpu().numpy().tofile(src)
        rf = awm.binding.RawFormat(2, RATE, 16, 0, 0)
        awm.lib.awm_set_quiet(1)
        best_add = best_get = None
        import ctypes
        where = (ctypes.c_double * 8)()
        where_names = ["setup", "wait_input", "wait_output_slot", "queue_gpu_work", "final_gpu_wait", "final_writer_wait", "teardown", "hand_on_output"]
        for _ in range(4):
            if os.path.exists(dst):
                os.unlink(dst)                             # (freeing the previous output's 635 MB of page cache is not part of `add`: 70 ms on tmpfs)
            t0 = time.perf_counter()
            ctx.add_watermark_file(None, PAYLOAD, src, dst, rf, rf)
            t1 = time.perf_counter()
            pats = ctx.get_watermark_file(None, dst, rf)
            t2 = time.perf_counter()
            if best_add is None or t1 - t0 < best_add:
                awm.lib.awm_debug_file_timing(where)
                out["add_file_calling_thread_ms"] = {n: round(where[i], 2) for i, n in enumerate(where_names)}
            best_add = t1 - t0 if best_add is None else min(best_add, t1 - t0)
            best_get = t2 - t1 if best_get is None else min(best_get, t2 - t1)
        out["file_to_file_xRT"] = round(seconds / (best_add + best_get), 1)
        out["add_file_ms"] = round(best_add * 1e3, 2)
        out["get_file_ms"] = round(best_get * 1e3, 2)
        # the file level `get` with the whole stream loaded before the first chunk starts (rounds 1 - 5) and with the chunks started while the
        # stream is still crossing PCIe (default), ALTERNATING on the file in the page cache (the `get` behind a fresh `add` above is slower
        # either way: the file's pages were created a moment ago); and "watermark, then verify" as ONE call: the output is never read back
        key = lambda p: (p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"])
        alt = {0: [], 1: []}
        same = True
        for _ in range(4):
            for mode in (0, 1):
                awm.lib.awm_debug_set_get_overlap(mode)
                t1 = time.perf_counter()
                pats_m = ctx.get_watermark_file(None, dst, rf)
                alt[mode].append(time.perf_counter() - t1)
                same = same and [key(p) for p in pats_m] == [key(p) for p in pats]
        awm.lib.awm_debug_set_get_overlap(1)
        out["get_file_ms_alternating"] = {"whole_stream_first": round(min(alt[0]) * 1e3, 2), "chunks_started_during_the_load": round(min(alt[1]) * 1e3, 2),
                                          "same_patterns": bool(same)}
        best_both = None
        for _ in range(3):
            if os.path.exists(dst2):
                os.unlink(dst2)
            t0 = time.perf_counter()
            pats_both = ctx.add_get_watermark_file(None, PAYLOAD, src, dst2, rf, rf)
            best_both = min(best_both or 1e9, time.perf_counter() - t0)
        out["add_get_file_as_one_call_ms"] = round(best_both * 1e3, 2)
        out["add_get_file_as_one_call_xRT"] = round(seconds / best_both, 1)
        out["add_get_file_as_one_call_same_file_and_patterns"] = bool(open(dst, "rb").read() == open(dst2, "rb").read()
                                                                      and [key(p) for p in pats_both] == [key(p) for p in pats])
        out["file_bytes"] = int(raw.numel())
        out["payload_matches"] = sum(p["bits"] == PAYLOAD for p in pats)
        # (2) page-locked host buffers instead of files
        host_in = torch.empty(raw.numel(), dtype=torch.uint8).pin_memory()
        host_in.copy_(raw.cpu())
        host_out = torch.empty_like(host_in).pin_memory()
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            xin = ctx.pcm_decode(host_in.cuda(non_blocking=True), 16, 0, False).reshape(n, 2)
            w = ctx.add_watermark(None, PAYLOAD, xin)
            host_out.copy_(ctx.pcm_encode(w.reshape(-1), 16, 0, False, True), non_blocking=True)
            torch.cuda.synchronize()
            back = ctx.pcm_decode(host_out.cuda(non_blocking=True), 16, 0, False).reshape(n, 2)
            ctx.get_watermark(None, back)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["pcie_staged_xRT"] = round(seconds / best, 1)
        out["pcie_staged_ms"] = round(best * 1e3, 2)
        out["resident_ms"] = round(resident_ms, 3)
        # (3) the command line binary
        cli = os.path.join(ROOT, "audiowmark_amd", "audiowmark")
        if os.path.exists(cli):
            fmt = ["--format", "raw", "--raw-rate", str(RATE), "--raw-channels", "2", "--raw-bits", "16"]
            # measured from a FRESH small interpreter: a child forked from this process (torch: several GB resident) would
            # report the parent's resident set as its own peak (ru_maxrss covers the moment between fork and exec)
            helper = ("import os, sys, time, json\n"
                      "def child(cmd):\n"
                      "    t0 = time.perf_counter(); r, w = os.pipe(); pid = os.fork()\n"
                      "    if pid == 0:\n"
                      "        os.dup2(w, 1); os.dup2(os.open(os.devnull, os.O_WRONLY), 2); os.close(r); os.execv(cmd[0], cmd)\n"
                      "    os.close(w); text = b''\n"
                      "    while True:\n"
                      "        b = os.read(r, 1 << 16)\n"
                      "        if not b: break\n"
                      "        text += b\n"
                      "    _, status, ru = os.wait4(pid, 0)\n"
                      "    return time.perf_counter() - t0, ru.ru_maxrss / 1024.0, os.waitstatus_to_exitcode(status), text.decode(errors='replace')\n"
                      "cmds = json.loads(sys.argv[1]); print(json.dumps([child(c) for c in cmds]))\n")
            # (the queue count this process asked the HIP runtime for is its own business: the command line picks its own)
            child_env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_SDMA")}
            r = subprocess.run([sys.executable, "-c", helper, json.dumps([[cli, "add", "-q"] + fmt + [src, dst2, PAYLOAD], [cli, "get"] + fmt + [dst2]])],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=child_env)
            (ta, rss_a, rc_a, _), (tg, rss_g, rc_g, text) = json.loads(r.stdout.decode())
            text = text.encode()
            out["cli"] = {"add_s": round(ta, 3), "get_s": round(tg, 3), "xRT_incl_process_start": round(seconds / (ta + tg), 1),
                          "rc": [rc_a, rc_g], "peak_rss_mb": {"add": round(rss_a, 1), "get": round(rss_g, 1)},
                          "output_identical_to_in_process": bool(rc_a == 0 and open(dst, "rb").read() == open(dst2, "rb").read()),
                          "patterns_with_payload": text.decode(errors="replace").count(PAYLOAD)}
    finally:
        for f in (src, dst, dst2):
            try:
                os.unlink(f)
            except OSError:
                pass
    return out


def viterbi_form(awm):
    """which form of K8 ran: chosen per process from the measured cost of a dependent launch on this host (hip/viterbi.hip)"""
    import ctypes as C
    try:
        awm.lib.awm_debug_dependent_launch_us.restype = C.c_double
        us = awm.lib.awm_debug_dependent_launch_us()
        one = bool(awm.lib.awm_debug_viterbi_one_launch_in_use())
        return {"one_launch_kernel": one, "dependent_launch_us_probe": round(us, 2), "forced": False,
                "note": "chain of 14 launches where a dependent launch is cheap on this host, the one-launch kernel (8 resident workgroups per decode, "
                        "per-decode counters) above 9 us per launch; bits and error values identical"}
    except Exception:
        return None


def scope_table(rows, calls):
    """HIP-event scopes of a profiled one-lane pass -> per-kernel roofline objects, largest share first: stand-alone average duration,
    algorithmic bytes per scope (SURVEY.md 8(d) / DESIGN.md section 3 figures x the units the scope processed), the same ratio
    against 8 TB/s as `roofline.frac`, and what actually limits the kernel.  Reproducible from the rocprofv3 --kernel-trace --stats
    summary of the same pass (profiles/rNN): avg_ms there == avg_ms here."""
    total = sum(r[1] for r in rows) or 1.0
    out = []
    for name, ms, launches, nbytes in sorted(rows, key=lambda r: -r[1]):
        gbps = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out.append({"kernel": KERNELS.get(name, (name, ""))[0], "scope": name, "single_launch_scope": name in SINGLE_KERNEL_SCOPES or name in SPEED_SINGLE_SCOPES,
                    "scopes_per_call": round(launches / calls, 2), "avg_ms": round(ms / launches, 4), "ms_per_call": round(ms / calls, 4),
                    "share_of_gpu_time_alone": round(ms / total, 3), "algorithmic_bytes_per_scope": int(nbytes / launches),
                    "achieved_GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBS, 4), "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "limited_by": KERNELS.get(name, ("", "?"))[1]})
    return out


SPEED_SINGLE_SCOPES = ("resample_kernel", "resample_var_kernel", "speed_mags_kernel", "speed_compare_kernel", "frame_mod_table_kernel")


def read_prof(awm, ctx):
    import ctypes as C
    awm.lib.awm_prof_name.restype = C.c_char_p
    prof = []
    for i in range(awm.lib.awm_prof_count()):
        ms, launches, nbytes = C.c_double(), C.c_long(), C.c_double()
        awm.lib.awm_prof_read(ctx._h, i, C.byref(ms), C.byref(launches), C.byref(nbytes))
        if launches.value:
            prof.append((awm.lib.awm_prof_name(i).decode(), ms.value, launches.value, nbytes.value))
    return prof


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def ensure_ranks(args, argv):
    """`--gpus N` is the number of ranks of the job.  Under a launcher (WORLD_SIZE set) the two must agree; without one and
    N > 1 this process REPLACES itself by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same
    arguments>` (one rank per GPU, rendezvous on 127.0.0.1), so that the plain command `python bench.py --gpus N` yields a line
    with n_gpus == ranks_seen == N.  Fails loudly when the node has fewer than N GPUs."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks")
        return
    if args.gpus <= 1:
        return
    if not args.rank_check_only and not args.same_device:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node has {have} visible GPU(s)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def _shard_plan(args, sharded_path, pipe, dist, world, rate):
    """one stream over several ranks: what every rank held, worked on and sent in the LAST timed step (plan: awm_sharded_plan, pure host;
    bytes: counted by the transport), so that a scaling curve explains itself"""
    if pipe is None:
        return None
    from audiowmark_amd import sharded as _sh
    lengths = list(pipe.part.lengths)
    entries = _sh.plan(lengths)
    per_chunk = {}
    for c, r, _, k in entries:
        if k:
            per_chunk.setdefault(c, set()).add(r)
    mine_stats = dict(pipe.comm.stats)
    rows = [None] * world
    if dist is not None and world > 1:
        dist.all_gather_object(rows, mine_stats)
    else:
        rows = [mine_stats]
    return {"span_frames": lengths, "chunks": len(per_chunk),
            "chunks_shared_by_several_ranks": sum(1 for v in per_chunk.values() if len(v) > 1),
            "per_rank": [{"rank": r, "span_seconds": round(lengths[r] / rate, 1),
                          "start_frames": sum(k for _, rr, _, k in entries if rr == r),
                          "chunks_local": sum(1 for v in per_chunk.values() if v == {r}),
                          "chunks_shared": sum(1 for v in per_chunk.values() if r in v and len(v) > 1),
                          "sent_in_the_last_step": rows[r]} for r in range(world)],
            "transport": "RCCL (device records) + gloo side group (host records)" if pipe.comm.nccl and pipe.comm.host_group is not None
                         else ("RCCL" if pipe.comm.nccl else "gloo")}


def rank_check(args):
    """--rank-check-only: the rendezvous of the N ranks and nothing else, over gloo, without touching a GPU (CPU test of the
    launch path: `python bench.py --gpus 2 --rank-check-only` must report ranks_seen == 2)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(t)
        seen = int(t.item())
        rank = dist.get_rank()
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen, rank = 1, 0
    if rank == 0:
        print(json.dumps({"rank_check_only": True, "n_gpus": world, "ranks_seen": seen, "gpus_argument": args.gpus}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["60min", "8h", "clips"], default="60min",
                    help="60min: BASELINE configs[1], per GPU (weak scaling); 8h: configs[3], one 8 h stream over all ranks (strong "
                         "scaling); clips: configs[4], 1024 x 30 s clips over all ranks")
    ap.add_argument("--minutes", type=float, default=None, help="audio minutes (60min: per GPU, default 60; 8h: in total, default 480)")
    ap.add_argument("--clips", type=int, default=1024, help="--config clips: clips in total")
    ap.add_argument("--cpu-sample-seconds", type=float, default=3600.0)
    ap.add_argument("--refine-form", type=int, default=-1, help="A / B only: the form of K4s (awm_debug_set_refine_form): 3 = rounds 3 - 5, 4 = default, 5 = update term in float")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-detect-speed-config", action="store_true", help="skip the 48 kHz --detect-speed configuration (BASELINE configs[2])")
    ap.add_argument("--lanes", type=int, default=4, help="lanes `get` spreads the chunks of a stream over (1: kernels back to back, for profiling)")
    ap.add_argument("--one-call", action="store_true",
                    help="a step = awm_add_get_watermark_d (add, then get of its output, as one call) instead of the two entry points")
    ap.add_argument("--sharded", action="store_true",
                    help="debug: take the multi-GPU (ShardedStream / torch.distributed) code path even with one process")
    ap.add_argument("--viterbi-form", choices=["auto", "chain", "one-launch"], default="auto",
                    help="K8: auto = chosen per process from the probed cost of a dependent launch (the default of the library); the profile "
                         "passes force the form the untraced run uses (a tracer makes launches expensive and would flip the choice)")
    ap.add_argument("--same-device", action="store_true",
                    help="debug: the N ranks of --gpus N all use cuda:0 and talk over gloo (host-staged transfers) -- what the multi-GPU "
                         "protocol itself costs when no second GPU pays anything back; never a scaling number")
    ap.add_argument("--rank-check-only", action="store_true",
                    help="start the ranks (as --gpus N does), count them over gloo and stop: CPU test of the launch path")
    args = ap.parse_args()
    ensure_ranks(args, sys.argv[1:])
    if args.rank_check_only:
        return rank_check(args)

    import torch
    import audiowmark_amd as awm

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
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
        elif args.same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    ctx = awm.Context(local_rank)
    awm.lib.awm_debug_set_viterbi_persistent({"auto": -1, "chain": 0, "one-launch": 1}[args.viterbi_form])
    awm.lib.awm_ctx_set_chunk_lanes(ctx._h, args.lanes)
    if args.refine_form >= 0:
        awm.lib.awm_debug_set_refine_form(args.refine_form)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    strong = args.config == "8h"
    if args.config == "clips":
        # ---- configs[4]: independent 30 s clips, replicas: rank r takes clips r + 1, r + 1 + world, ...; no data-path collective.
        # clip k is `test-gen-noise` with --test-key k (16 bit), watermarked and decoded with --test-key k.
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor
        n_clip = 30 * RATE
        mine = list(range(rank + 1, args.clips + 1, world))
        keys = [awm.test_key(k) for k in mine]

        def make_clip(k):                                    # (the C call releases the GIL: the generator runs on all host cores)
            return quantise16(np, awm.binding.gen_noise(awm.test_key(k), 2 * n_clip)).reshape(n_clip, 2)
        with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 8)) as pool:
            clips = [torch.from_numpy(c).to(dev) for c in pool.map(make_clip, mine)]
        outs = [torch.empty_like(c) for c in clips]
        audio_seconds = args.clips * 30.0
        workload = (f"{args.clips} clips of 30 s stereo 44.1 kHz test-gen-noise (--test-key k, 16 bit) over {world} GPU(s) (replicas: {len(mine)} on rank 0), "
                    f"add + get per clip with the clip's own key k (awm_add_watermark_batch_keys_d / awm_get_watermark_batch_keys_d: groups of 64 clips, "
                    f"`add`: 256 clips per launch of every stage (block maxima, K2, limiter), their frame_mod tables from 256 keys per launch of the device's table kernel (K16) on its own stream while the previous 256 clips are watermarked; `get`: a group's sync / mix / bit order tables from the device as well (K16g, one group ahead on a stream beside the lane's); one launch per stage and group)")

        def step():
            ctx.add_watermark_batch_keys(keys, PAYLOAD, clips, outs)
            return ctx.get_watermark_batch_keys(keys, outs)
    else:
        minutes = args.minutes if args.minutes is not None else (480.0 if strong else 60.0)
        n_total = int(minutes * 60 * RATE) * (1 if strong else world)
        if sharded_path:
            per = (n_total // world) // 1024 * 1024          # spans of a sharded stream are whole frames (only the last one may be ragged)
            n = per if rank < world - 1 or not strong else n_total - per * (world - 1)
            if not strong:
                n = per
        else:
            n = n_total
        x = torch.rand((n, 2), generator=gen, device=dev, dtype=torch.float32) * 2 - 1   # test-gen-noise distribution
        out = torch.empty_like(x)
        audio_seconds = (n_total if strong or not sharded_path else n * world) / RATE
        workload = (f"{minutes:g} min stereo 44.1 kHz white noise " + ("in total" if strong else "per GPU") +
                    f" resident in HBM, add+get incl. payload decode, payload {PAYLOAD}, strength 10" +
                    (", a step = awm_add_get_watermark_d (add, then get of its output, one call)" if args.one_call and not sharded_path else ""))
        if sharded_path:
            from audiowmark_amd import sharded
            pipe = sharded.ShardedStream(ctx, dist, n_frames_local=n, n_channels=2)

            def step():
                pipe.add_watermark(None, PAYLOAD, x, out)
                return pipe.get_watermark(None, out)
        elif args.one_call:
            # add + get of the output as ONE call (awm_add_get_watermark_d): the same kernels and results; `get` may start a chunk behind
            # the limiter pass that covers it instead of behind the whole `add` (measured: 5.25 against 5.29 ms -- the step is bound by the
            # sum of its kernels' work, not by where `get` starts; profiles/r05/README.md)
            def step():
                return ctx.add_get_watermark(None, PAYLOAD, x, out)
        else:
            def step():
                ctx.add_watermark(None, PAYLOAD, x, out=out)
                return ctx.get_watermark(None, out)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # the very first step of the process on its own (first_step_ms): what a caller without warm-up sees -- lane streams (4 - 10 ms each in the
    # runtime), workspaces, key tables; the timed steps below are the steady state
    first_step_ms = None
    if args.warmup > 0:
        sync()
        t_first = time.perf_counter()
        step()
        sync()
        first_step_ms = round((time.perf_counter() - t_first) * 1e3, 3)
    for _ in range(max(0, args.warmup - 1)):
        step()
    awm.lib.awm_prof_reset(ctx._h)
    # per-kernel HIP events: two events per launch -- noise for the 60 min kernels, a sizeable cost for the ~40 small launches
    # of a 30 s clip, so the clips mode is timed without them
    awm.lib.awm_prof_enable(ctx._h, 0 if args.config == "clips" else 1)
    sync()
    t0 = time.perf_counter()
    pats = None
    for i in range(args.steps):
        if i == args.steps - 1 and sharded_path and args.config != "clips":
            pipe.comm.reset_stats()
        pats = step()
    sync()
    elapsed = time.perf_counter() - t0
    awm.lib.awm_prof_enable(ctx._h, 0)
    matches_local = 0
    if args.config == "clips":
        matches_local = sum(any(p["bits"] == PAYLOAD for p in clip) for clip in pats)
    ranks_seen = 1
    if dist is not None:
        t = torch.tensor([elapsed, float(matches_local), 1.0], device="cpu" if args.same_device else dev, dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)             # final gather of the per-rank results (clips mode: match counts)
        elapsed = float(tmax[0].item())
        matches_local = int(t[1].item())
        ranks_seen = int(t[2].item())

    # one stream over several ranks: what every rank held, worked on and sent in the LAST timed step, so that a scaling curve explains
    # itself (plan: awm_sharded_plan, pure host; bytes: counted by the transport)
    shard_plan = None
    try:
        shard_plan = _shard_plan(args, sharded_path, pipe if (sharded_path and args.config != "clips") else None, dist, world, RATE)
    except Exception as e:                                   # (an explanation, never a reason to lose the line)
        shard_plan = {"error": repr(e)}
    single_stream = args.config == "60min" and world == 1 and not args.sharded
    prof = read_prof(awm, ctx)
    two_calls_ms = None
    one_call_equal = None
    if world == 1 and not args.sharded and args.config != "clips" and not args.one_call:
        # the same step as ONE call (awm_add_get_watermark_d; untimed extra): what the add -> get hand-over per chunk is worth
        awm.lib.awm_prof_enable(ctx._h, 0)
        def two():
            return ctx.add_get_watermark(None, PAYLOAD, x, out)
        two()
        sync()
        t0 = time.perf_counter()
        for _ in range(max(3, args.steps // 2)):
            pats2 = two()
        sync()
        two_calls_ms = round((time.perf_counter() - t0) / max(3, args.steps // 2) * 1e3, 3)
        key2 = lambda p: (p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"])
        one_call_equal = [key2(p) for p in pats2] == [key2(p) for p in (pats or [])]
        if not one_call_equal:                               # (a bug to report -- in the line, which it must not cost)
            print("bench.py: awm_add_get_watermark_d and the two separate calls disagree", file=sys.stderr)
    serial, serial_steps = {}, 3
    if args.config != "clips":
        # the same kernels one after the other (a single lane), outside the timed region: the duration a kernel has when it
        # owns the GPU -- the number a roofline is about, and what a rocprofv3 --kernel-trace of this pass shows
        awm.lib.awm_ctx_set_chunk_lanes(ctx._h, 1)
        step()
        awm.lib.awm_prof_reset(ctx._h)
        awm.lib.awm_prof_enable(ctx._h, 1)
        for _ in range(serial_steps):
            step()
        sync()
        awm.lib.awm_prof_enable(ctx._h, 0)
        serial = {p[0]: p for p in read_prof(awm, ctx)}
        awm.lib.awm_ctx_set_chunk_lanes(ctx._h, args.lanes)

    if rank == 0:
        if args.config == "clips":
            # measured, untimed: profiled launch scopes per clip (one extra pass with the per-kernel events on), the host cost of one
            # key's tables on one core, and the same batch with ONE key for all clips (what the per-clip keys add)
            awm.lib.awm_prof_reset(ctx._h)
            awm.lib.awm_prof_enable(ctx._h, 1)
            step()
            sync()
            awm.lib.awm_prof_enable(ctx._h, 0)
            scopes = sum(p[2] for p in read_prof(awm, ctx))
            # stand-alone durations of the batch kernels: ONE group of 64 clips = one lane of `get` (the groups of a larger batch run
            # on several lanes side by side), per-kernel HIP events; `add` of the same 64 clips (its lanes run side by side: K2's
            # duration there is a concurrent one and says so)
            g = min(64, len(clips))
            calls = 3
            ctx.get_watermark_batch_keys(keys[:g], outs[:g])
            awm.lib.awm_prof_reset(ctx._h)
            awm.lib.awm_prof_enable(ctx._h, 1)
            for _ in range(calls):
                ctx.add_watermark_batch_keys(keys[:g], PAYLOAD, clips[:g], outs[:g])
                ctx.get_watermark_batch_keys(keys[:g], outs[:g])
            sync()
            awm.lib.awm_prof_enable(ctx._h, 0)
            one_group = scope_table(read_prof(awm, ctx), calls)
            for k in one_group:
                if k["scope"] in ("add_mix_kernel", "limiter_kernel"):
                    k["note"] = "add of the group: ONE launch per stage for its 64 clips (blockIdx.y = clip; block maxima, K2, limiter table, limiter)"
            t0 = time.perf_counter()
            for k in mine[:8]:
                awm.tab_frame_mod(awm.test_key(k), PAYLOAD); awm.tab_sync_bits(awm.test_key(k), True); awm.tab_mix_entries(awm.test_key(k))
            t_tab = (time.perf_counter() - t0) / max(1, len(mine[:8]))
            sync()
            t0 = time.perf_counter()
            for _ in range(2):
                ctx.add_watermark_batch(keys[0], PAYLOAD, clips, outs)
                ctx.get_watermark_batch(keys[0], outs)
            sync()
            one_key = (time.perf_counter() - t0) / 2
            per_clip = elapsed / args.steps * 1e3 / max(1, len(clips))
            cfg = {"workload": workload, "parallelism": f"{world} replica(s)", "clips_with_payload": matches_local,
                   "clip_batch_config": {"ms_per_clip_add_and_get": round(per_clip, 4),
                                         "ms_per_clip_with_one_key_for_all": round(one_key * 1e3 / max(1, len(clips)), 4),
                                         "key_tables_ms_per_clip_amortised": round(per_clip - one_key * 1e3 / max(1, len(clips)), 4),
                                         "key_tables_host_ms_per_key_on_one_core": round(t_tab * 1e3, 3), "host_cores": os.cpu_count(),
                                         "profiled_launch_scopes_per_clip": round(scopes / max(1, len(clips)), 3),
                                         "kernels_one_group_of_64_clips": one_group}}
        else:
            matches = sum(1 for p in (pats or []) if p["bits"] == PAYLOAD)
            cfg = {"workload": workload, "parallelism": (f"one stream sharded over {world} GPU(s)" if sharded_path else "1 GPU") +
                   (" -- DEBUG: all ranks on ONE device over gloo (--same-device), not a scaling measurement" if args.same_device else ""),
                   "patterns": len(pats or []), "payload_matches": matches}
        cfg["ranks_seen"] = ranks_seen
        if shard_plan is not None:
            cfg["shard_plan"] = shard_plan
        if dist is not None:
            try:
                cfg["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                pass
        roofline = None
        scopes = None
        if serial:
            total_alone = sum(v[1] for v in serial.values()) or 1.0
            singles = [v for v in serial.values() if v[0] in SINGLE_KERNEL_SCOPES]
            name, sms, sl, sb = max(singles, key=lambda v: v[1])
            achieved = sb / (sms * 1e-3) / 1e9               # algorithmic GB/s: bytes per launch / average launch duration
            roofline = {"kernel": KERNELS.get(name, (name, ""))[0], "scope": name, "bound": "hbm", "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": pmc_traffic(name, args.minutes if args.minutes is not None else 60.0) if single_stream else None,
                        "algorithmic_bytes_per_launch": int(sb / sl), "launches_per_step": round(sl / serial_steps, 2), "avg_ms": round(sms / sl, 4),
                        "share_of_gpu_time_alone": round(sms / total_alone, 3), "limited_by": KERNELS.get(name, ("", "?"))[1]}
            # scopes that bracket SEVERAL launches: reported, never the roofline kernel.  ms = the scope's events per step, one lane.
            scopes = {k: {"ms_per_step_alone": round(v[1] / serial_steps, 4), "scopes_per_step": round(v[2] / serial_steps, 2),
                          "launches_per_scope": SCOPE_LAUNCHES.get(k)}
                      for k, v in serial.items() if k not in SINGLE_KERNEL_SCOPES}
            if "sync_scan_kernel(approx)" in serial:
                # K5w works out of LDS: every candidate start gathers its 510 sync rows x 60 bands from the dB ring (four
                # candidates per ds_read_b128; 256 B/clk/CU = ~150 TB/s for the chip at 2.4 GHz, MI355X_MICROARCH.md).
                _, kms, kl, kb = serial["sync_scan_kernel(approx)"]
                frame_shifts = kb / 324.0
                candidates = max(frame_shifts - 4 * 2226 * kl, 0.0)
                lds_tbps = candidates * 510 * 60 * 4 / (kms * 1e-3) / 1e12
                roofline["lds_gather_of_the_scan"] = {"achieved": round(lds_tbps, 1), "peak": 150.0, "unit": "TB/s", "frac": round(lds_tbps / 150.0, 3),
                                                      "note": "peak at 2.4 GHz; effective clock under this kernel 2.3 GHz (GRBM_GUI_ACTIVE, profiles/r03/effective_clock.txt). "
                                                              "Per sync frame and wave 30 gathers feed 120 float additions in the reference's order: the VALU issues 42 % of the cycles (one "
                                                              "wave64 FP32 instruction per ~2.5 cycles is the SIMD's rate, tools/valu_rate.hip), the LDS is active 51 %; its 846 tiles are "
                                                              "3.3 rounds on 256 CUs: the stand-alone time includes a 17 % tail that the other lanes fill in the timed configuration"}
            # the single-launch kernels one by one, same definition, with their measured traffic: the dominant kernel above works out of
            # LDS (its HBM fraction says little), these are the ones the 8 TB/s are the yardstick for
            roofline["single_launch_kernels"] = [
                {"kernel": KERNELS.get(v[0], (v[0], ""))[0], "scope": v[0], "bound": "hbm", "achieved": round(v[3] / (v[1] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(v[3] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_ms": round(v[1] / v[2], 4),
                 "algorithmic_bytes_per_launch": int(v[3] / v[2]),
                 "traffic": pmc_traffic(v[0], args.minutes if args.minutes is not None else 60.0) if single_stream else None,
                 "share_of_gpu_time_alone": round(v[1] / total_alone, 3)}
                for v in sorted(singles, key=lambda v: -v[1])]
            # every kernel above 5 % of the stand-alone GPU time, same definition (algorithmic bytes / stand-alone duration / 8 TB/s)
            roofline["all_kernels_above_5_percent"] = [
                {"kernel": KERNELS.get(k, (k, ""))[0], "scope": k, "single_launch_scope": k in SINGLE_KERNEL_SCOPES,
                 "share_of_gpu_time_alone": round(v[1] / total_alone, 3), "avg_ms_per_scope": round(v[1] / v[2], 4),
                 "achieved_GBps": round(v[3] / (v[1] * 1e-3) / 1e9, 1), "frac": round(v[3] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "limited_by": KERNELS.get(k, ("", "?"))[1]}
                for k, v in sorted(serial.items(), key=lambda kv: -kv[1][1]) if v[1] / total_alone > 0.05]
        res = {
            "metric": "audio seconds watermarked+decoded per wall-second (xRT), 44.1 kHz stereo",
            "value": round(audio_seconds * args.steps / elapsed, 1),
            "unit": "xRT",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "first_step_ms": first_step_ms,
            "ms_per_step_as_one_call": two_calls_ms,
            "one_call_patterns_equal_two_calls": one_call_equal,
            "higher_is_better": True,
            "scaling": "strong" if strong or args.config == "clips" else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": cfg,
            "roofline": roofline,
            "scopes_ms_per_step": scopes,
            "viterbi_form": dict(viterbi_form(awm) or {}, forced=args.viterbi_form != "auto"),
            "traffic_source": traffic_provenance(),
        }
        if serial:
            am = serial.get("add_mix_kernel")
            # the batched STFT north_star puts the 40 % HBM target on: the fused add kernel (STFT + band edit + inverse +
            # overlap-add), 8192 algorithmic bytes per frame-channel, stand-alone duration
            res["stft_roofline"] = ({"kernel": "add_mix_kernel", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                     "achieved": round(am[3] / (am[1] * 1e-3) / 1e9, 1), "frac": round(am[3] / (am[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "avg_ms": round(am[1] / am[2], 4),
                                     "traffic": pmc_traffic("add_mix_kernel", args.minutes if args.minutes is not None else 60.0) if single_stream else None}
                                    if am else None)
            # concurrent lanes (timed region): durations overlap and are stretched by the kernels they share the GPU with
            res["kernels_ms_per_step"] = {p[0]: round(p[1] / args.steps, 3) for p in prof}
            # one lane, kernels back to back (untimed extra pass): true per-kernel cost
            res["kernels_ms_per_step_alone"] = {k: round(v[1] / serial_steps, 3) for k, v in serial.items()}
            # algorithmic GB/s (SURVEY.md 8d bytes / stand-alone time), same definition as roofline.achieved
            res["kernels_algorithmic_GBps"] = {k: round(v[3] / (v[1] * 1e-3) / 1e9, 1) for k, v in serial.items() if v[1] > 0}
        if single_stream:
            if not args.no_e2e:
                try:
                    res["e2e"] = e2e_leg(torch, awm, ctx, x, elapsed / args.steps * 1e3)
                except Exception as e:
                    res["e2e"] = {"error": str(e)}
            if not args.no_detect_speed_config:
                try:
                    res["detect_speed_config"] = detect_speed_config(torch, awm, ctx, None, PAYLOAD, args.minutes if args.minutes is not None else 60.0, args.lanes)
                except Exception as e:                       # reported, never fatal for the headline line
                    res["detect_speed_config"] = {"error": str(e)}
            if not args.no_cpu_baseline:
                try:
                    res["cpu_baseline"], res["parity"] = cpu_baseline_and_parity(torch, awm, ctx, args.cpu_sample_seconds)
                except Exception as e:
                    res["cpu_baseline"], res["parity"] = None, {"error": str(e)}
            else:
                res["cpu_baseline"] = None
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
