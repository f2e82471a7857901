"""BASELINE.json configs at FULL size against the compiled reference (oracle/_ref = the unmodified reference sources).

The reference's own functions run on the GPU box's host threads (`add` single threaded, `get` on hardware_concurrency
threads, exactly like `audiowmark add` / `audiowmark get`, reference wmadd.cc:448-618, wmget.cc:971-1013); the HIP path
runs through the C ABI on the same input.  Input = `audiowmark test-gen-noise` samples (reference audiowmark.cc:399-417)
quantised to 16 bit like the WAV file that command writes.

Bars (north_star): embedded PCM within 1e-5 RMS (enforced: 1e-6); decoded pattern list -- time, sync index, pattern type,
block type -- identical line by line, payload bits identical for every watermark (see compare_patterns for the reference's
n_best fallback lines, which are Viterbi decodes of noise); sync quality within 1e-5; decode error within 1e-4.

Scenarios follow tests/block-decoder-test.sh:8-18 (configs[1]), tests/detect-speed-test.sh:9-16 (configs[2]) and
tests/clip-decoder-test.sh with --test-key (configs[4]).  The measured differences are written to
gpurun_out/fullsize_parity.json."""
import json
import os
import time

import numpy as np
import pytest

import _ref

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _ref.available(), reason="oracle/_ref (compiled reference) not built")]

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PAY1 = "0123456789abcdef0011223344556677"
RMS_TOL = 1e-6
QUALITY_TOL = 1e-5
ERROR_TOL = 1e-4
REPORT = {}


def pkey(p):
    return (round(p["time"], 6), p["sync_index"], p["type"], p["block_type"], p["bits"])


def quantise16(x):
    """what reading back the 16 bit file of `test-gen-noise` gives (stdout WAV path: truncation towards zero, /32768)"""
    return (np.clip(np.trunc(x.astype(np.float64) * 32768.0), -32768, 32767) / 32768.0).astype(np.float32)


JUNK_ERROR = 0.6      # decode error of a real watermark: 0.10-0.14 (block), 0.33-0.36 (30 s clip); of noise: 0.75-0.79


def compare_patterns(got, want, what, speed_tol=None, max_ties=0):
    """STRICT by default (max_ties = 0): every pattern the reference reports is reported at the same sync index.  A caller that
    knows of a refinement tie in its fixture passes max_ties explicitly (none of the full-size configurations does: all measured 0).

    Every pattern the reference reports must be reported at the same position with the same types; the payload bits must be
    identical for every pattern that IS a watermark (reference decode error < 0.6).  The reference also prints its n_best
    fallback candidates: Viterbi decodes of noise (decode error ~0.77), whose 128 bits are decided by path metric differences
    at float rounding level -- there a different FFT rounding (the reference's FFTW vs. the oracle's double FFT vs. this one)
    may flip bits; such patterns must still agree in position and types, the number of differing ones is reported."""
    assert len(got) == len(want), f"{what}: {len(got)} patterns, the reference has {len(want)}"
    junk_diff = ties = 0
    tied = []
    for i, (g, w) in enumerate(zip(got, want)):
        if pkey(g)[:4] != pkey(w)[:4]:
            # A refinement TIE: the sync quality is flat to ~1e-7 over neighbouring fine offsets (8 samples) around a block start and
            # the FFTs of the two detectors differ at float rounding level, so strict `>` (syncfinder.cc:441) may keep different
            # neighbours (two CPU FFT backends under the unmodified reference do the same).  Same type, same
            # bits, quality within the tolerance, position 8 samples apart; counted, reported, bounded.
            assert (g["type"], g["block_type"], g["bits"]) == (w["type"], w["block_type"], w["bits"]) \
                and abs(int(g["sync_index"]) - int(w["sync_index"])) <= 8 and abs(g["time"] - w["time"]) < 1e-3, \
                f"{what}: pattern position / type differs: {pkey(g)} != {pkey(w)}"
            ties += 1
            tied.append(i)
            assert ties <= max_ties, f"{what}: sync position differs from the reference's (8 samples away, {ties} so far, {max_ties} allowed): {pkey(g)} != {pkey(w)}"
            continue
        if g["bits"] != w["bits"]:
            assert w["decode_error"] >= JUNK_ERROR, f"{what}: payload bits of a watermark differ: {pkey(g)} != {pkey(w)}"
            junk_diff += 1
    assert ties <= max_ties, f"{what}: {ties} patterns at neighbouring fine offsets, {max_ties} allowed"
    dq = max((abs(g["sync_quality"] - w["sync_quality"]) for g, w in zip(got, want)), default=0.0)
    # (a noise pattern decoded to other bits took another path through the trellis: its error value is another path's; a block read
    # 8 samples apart -- a tie, also inside an AB pair or the "all" pattern -- has another path metric)
    de = max((abs(g["decode_error"] - w["decode_error"]) for g, w in zip(got, want) if g["bits"] == w["bits"]), default=0.0)
    assert dq < QUALITY_TOL, f"{what}: sync quality differs by {dq}"
    assert de < (0.01 if tied else ERROR_TOL), f"{what}: decode error differs by {de}"
    out = {"patterns": len(want), "max_abs_sync_quality_diff": dq, "max_abs_decode_error_diff": de, "noise_patterns_with_other_bits": junk_diff,
           "refinement_ties": ties}
    if speed_tol is not None:
        ds = max((abs(g["speed"] - w["speed"]) for g, w in zip(got, want)), default=0.0)
        assert ds <= speed_tol, f"{what}: speed differs by {ds}"
        out["max_abs_speed_diff"] = ds
    return out


def rms_max(a, b, block=1 << 24):
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    assert a.size == b.size
    ss, mx = 0.0, 0.0
    for i in range(0, a.size, block):
        d = a[i:i + block].astype(np.float64) - b[i:i + block].astype(np.float64)
        ss += float(np.dot(d, d))
        mx = max(mx, float(np.abs(d).max())) if d.size else mx
    return (ss / max(a.size, 1)) ** 0.5, mx


@pytest.fixture(scope="module")
def gpu():
    import torch
    import audiowmark_amd as awm
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    ctx = awm.Context(0)

    class G:
        pass
    g = G()
    g.torch, g.awm, g.ctx = torch, awm, ctx
    g.dev = lambda a, ch=2: torch.from_numpy(np.ascontiguousarray(a).reshape(-1, ch)).cuda()
    yield g
    awm.set_params()
    awm.set_speed_params()
    _ref.set_params()
    _ref.set_speed_params(False, False, -1)
    ctx.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fullsize_parity.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


def test_noise_generator_is_the_references(gpu):
    """the product's `test-gen-noise` samples (C ABI awm_test_gen_noise) == the reference's, so the long inputs below may
    come from the fast generator"""
    a = gpu.awm.binding.gen_noise(None, 3_000_000)
    assert np.array_equal(a, _ref.gen_noise(None, 3_000_000))
    k = gpu.awm.test_key(5)
    assert np.array_equal(gpu.awm.binding.gen_noise(k, 100_000), _ref.gen_noise(k, 100_000))


def test_config1_60min_stereo_add_get_equal_reference(gpu):
    """BASELINE.json configs[1]: 60 min stereo 44.1 kHz, add + get, vs the reference on the host cores"""
    n = 60 * 60 * 44100
    x = quantise16(gpu.awm.binding.gen_noise(None, 2 * n))
    t0 = time.perf_counter()
    ref_w = _ref.add(None, x, 2, PAY1)
    t1 = time.perf_counter()
    ref_pats = _ref.get(None, ref_w, 2)
    t2 = time.perf_counter()
    w = gpu.ctx.add_watermark(None, PAY1, gpu.dev(x))
    r, m = rms_max(w.cpu().numpy(), ref_w)
    assert r < RMS_TOL and m < 4e-6, f"embedded PCM differs from the reference: rms {r}, max {m}"
    # decode the REFERENCE's output on the GPU: byte-identical input on both sides
    got = gpu.ctx.get_watermark(None, gpu.dev(ref_w))
    rep = compare_patterns(got, ref_pats, "configs[1] get")
    # and the GPU's own output (PCM differs in the 8th digit): same positions and payloads
    own = gpu.ctx.get_watermark(None, w)
    compare_patterns(own, ref_pats, "configs[1] get of the GPU's own output")
    matches = sum(p["bits"] == PAY1 for p in got)
    assert matches >= 100
    rep.update({"pcm_rms": r, "pcm_max_abs": m, "payload_matches": matches, "reference_add_s": round(t1 - t0, 2),
                "reference_get_s": round(t2 - t1, 2), "reference_threads": os.cpu_count()})
    REPORT["config1_60min_stereo"] = rep
    print("configs[1]:", rep)


def test_130min_stereo_five_chunks_equal_reference(gpu):
    """Longer than BASELINE configs[1]: 2 h 10 min stereo = FIVE reference chunks (wavchunkloader.cc:75-84), so the four chunk
    lanes are reused and overlap merging (wmget.cc:288-316) happens four times -- the complete pattern list against the reference.
    (Why an 8 h run reports the payload in all of its patterns while a 60 min run has four lines without it: the last chunk of 60
    min is 4.5 minutes = 5 blocks, fewer than n_best = 8, so sync_select_threshold_and_n_best keeps three or four noise peaks
    whose decodes are printed as well (syncfinder.cc:364-383); the last chunk of 8 h is 8.1 minutes = 9 blocks.  Here the last
    chunk is 19 minutes: no such lines either -- asserted below.)"""
    n = 130 * 60 * 44100
    x = quantise16(gpu.awm.binding.gen_noise(None, 2 * n))
    t0 = time.perf_counter()
    ref_w = _ref.add(None, x, 2, PAY1)
    t1 = time.perf_counter()
    del x
    ref_pats = _ref.get(None, ref_w, 2)
    t2 = time.perf_counter()
    assert len(gpu.awm.plan_chunks(n)) == 5
    got = gpu.ctx.get_watermark(None, gpu.dev(ref_w))
    rep = compare_patterns(got, ref_pats, "130 min get")
    matches = sum(p["bits"] == PAY1 for p in got)
    assert matches == len(got) and matches >= 220            # 150 blocks + AB pairs + 5 "all" patterns, no n_best filler
    rep.update({"payload_matches": matches, "chunks": 5, "reference_add_s": round(t1 - t0, 2), "reference_get_s": round(t2 - t1, 2),
                "reference_threads": os.cpu_count()})
    REPORT["stereo_130min_five_chunks"] = rep
    print("130 min:", rep)


# BASELINE configs[3] holds the ONE refinement tie of the full-size configurations: in chunk 9 (4 h 28 min into the stream) the block the
# reference build of oracle/_ref (double-precision FFT) puts at sync index 48 075 408 of its chunk is found at 48 075 400 here -- neighbouring
# fine offsets with qualities 1.8e-6 apart; `search_refine` keeps the first strictly better one (syncfinder.cc:441).  The reference does
# the same to ITSELF when only its FFT library changes: its double-FFT build and its MKL float-FFT build (oracle/Makefile: ref, ref_mkl)
# pick different fine offsets for 1 block in 1630 on byte-identical input, with qualities up to 3.9e-6 apart
# (tools/ref_backend_census.py -> profiles/r05/ref_backend_census.json; this detector against either build:
# profiles/r05/census_three_way.json).  The tie is pinned: that block, 8 samples, its AB pair -- anything else fails.
KNOWN_8H_TIE = {"ours": 48075400, "reference": 48075408, "max_pattern_lines": 3}


def test_config3_8h_stereo_equal_reference(gpu):
    """BASELINE.json configs[3] at FULL size on one GPU: 8 h stereo 44.1 kHz (18 reference chunks), watermarked by the REFERENCE, decoded
    by both detectors from the same samples: the complete pattern list (836 lines) -- payload bits everywhere, sync index and types
    everywhere except the one pinned tie; the HIP `add` of the same 8 h against the reference's PCM; the multi-GPU protocol with 2
    contexts on the one device equal to the single-GPU list.  (~3.5 minutes: the reference's `add` + `get` of 8 h on the host cores.)"""
    from audiowmark_amd import sharded
    n = 8 * 3600 * 44100
    x = quantise16(gpu.awm.binding.gen_noise(None, 2 * n))
    t0 = time.perf_counter()
    ref_w = _ref.add(None, x, 2, PAY1)
    t1 = time.perf_counter()
    ref_pats = _ref.get(None, ref_w, 2)
    t2 = time.perf_counter()
    assert len(gpu.awm.plan_chunks(n)) == 18
    wd = gpu.dev(ref_w)
    got = gpu.ctx.get_watermark(None, wd)
    rep = compare_patterns(got, ref_pats, "configs[3] 8 h get", max_ties=KNOWN_8H_TIE["max_pattern_lines"])
    moved = sorted({(g["sync_index"], w["sync_index"]) for g, w in zip(got, ref_pats) if g["sync_index"] != w["sync_index"]})
    assert moved in ([], [(KNOWN_8H_TIE["ours"], KNOWN_8H_TIE["reference"])]), f"sync positions beside the reference's other than the pinned tie: {moved}"
    matches = sum(p["bits"] == PAY1 for p in got)
    assert matches == len(got) == len(ref_pats) and matches >= 830            # 557 blocks + AB pairs + 18 "all" patterns, no n_best filler
    # the HIP add of the same 8 h
    w = gpu.ctx.add_watermark(None, PAY1, gpu.dev(x))
    del x
    r, m = rms_max(w.cpu().numpy(), ref_w)
    assert r < RMS_TOL and m < 4e-6, f"8 h embedded PCM differs from the reference: rms {r}, max {m}"
    del w, ref_w
    # the multi-GPU protocol: the stream in two spans on two contexts of this device == the single-GPU list, to the last bit
    ctx2 = gpu.awm.Context(0)
    try:
        per = (n // 2) // 1024 * 1024
        multi = sharded.multi_get([gpu.ctx, ctx2], None, [wd[:per], wd[per:]], max_out=8192)
    finally:
        ctx2.close()
    full = lambda p: pkey(p) + (p["sync_quality"], p["decode_error"])
    assert [full(p) for p in multi] == [full(p) for p in got]
    rep.update({"single_blocks": sum(1 for p in ref_pats if p["type"] == 0 and p["block_type"] < 2), "moved_blocks": moved, "pcm_rms": r, "pcm_max_abs": m,
                "payload_matches": matches, "chunks": 18, "awm_multi_get_d_2_contexts_equal_to_single": True,
                "reference_add_s": round(t1 - t0, 2), "reference_get_s": round(t2 - t1, 2), "reference_threads": os.cpu_count()})
    REPORT["config3_8h_stereo"] = rep
    print("configs[3]:", rep)


def test_config2_60min_48k_detect_speed_equal_reference(gpu):
    """BASELINE.json configs[2]: 60 min stereo 48 kHz, watermarked at 48 kHz, replayed 2 % fast, `get --detect-speed`.
    `add` at 48 kHz is compared with the reference's WatermarkResampler path (wmadd.cc:353-430); the replay (the attacker's
    part) is produced once and handed to both sides as a 48 kHz stream; the reference reads it through its own
    WavChunkLoader (one streaming 48 -> 44.1 kHz resampler for the whole file, wavchunkloader.cc:70-72,200-240), the HIP side
    through awm_resample_d.  zita-resampler itself is absent from the reference tree: the reference sources are compiled
    against the restated classes (oracle/zita_restated.h) -- parity with the real library is unpinned."""
    rate, speed = 48000, 1.02
    n = 60 * 60 * rate
    x = quantise16(gpu.awm.binding.gen_noise(None, 2 * n))
    t0 = time.perf_counter()
    ref_w = _ref.add(None, x, 2, PAY1, sample_rate=rate)
    t1 = time.perf_counter()
    w = gpu.ctx.add_watermark(None, PAY1, gpu.dev(x), sample_rate=rate)
    del x
    r, m = rms_max(w.cpu().numpy(), ref_w)
    assert r < RMS_TOL and m < 4e-6, f"48 kHz embedded PCM differs from the reference: rms {r}, max {m}"
    del w
    fast = gpu.ctx.resample_ratio(gpu.dev(ref_w), 1 / speed, rate=rate)       # test-change-speed
    del ref_w
    fast_host = fast.cpu().numpy()
    y = gpu.ctx.resample(fast, rate, 44100)                                   # what WavChunkLoader hands to the decoders
    del fast
    gpu.awm.set_speed_params(detect_speed=True)
    _ref.set_speed_params(True, False, -1)
    try:
        t2 = time.perf_counter()
        ref_pats = _ref.get(None, fast_host, 2, sample_rate=rate)
        t3 = time.perf_counter()
        got = gpu.ctx.get_watermark(None, y)
    finally:
        gpu.awm.set_speed_params()
        _ref.set_speed_params(False, False, -1)
    rep = compare_patterns(got, ref_pats, "configs[2] get --detect-speed", speed_tol=2e-6)
    hits = [p for p in got if p["bits"] == PAY1]
    assert len(hits) >= 60 and all(p["speed"] != 1 for p in hits)
    assert all(abs(p["speed"] - speed) / speed < 2e-4 for p in hits)
    rep.update({"pcm_rms": r, "pcm_max_abs": m, "payload_matches": len(hits), "reference_add_s": round(t1 - t0, 2),
                "reference_get_s": round(t3 - t2, 2), "reference_threads": os.cpu_count(),
                "detected_speeds": sorted({round(p["speed"], 6) for p in hits})})
    REPORT["config2_60min_48k_detect_speed"] = rep
    print("configs[2]:", rep)


def _reference_clip(k):
    """worker process: clip k of configs[4] through the compiled reference (noise and watermark key = --test-key k)"""
    key = int(k).to_bytes(8, "big") + bytes(8)
    x = quantise16(_ref.gen_noise(key, 2 * 30 * 44100))
    w = _ref.add(key, x, 2, PAY1)
    return k, x, w, _ref.get(key, w, 2)


def test_config4_sample_of_64_clips_equal_reference(gpu):
    """BASELINE.json configs[4], a 64 clip sample: 30 s stereo clips, clip k uses --test-key k for the noise and the
    watermark, add + get per clip vs the reference (ClipDecoder path, wmget.cc:764-884)."""
    n = 30 * 44100
    worst = {"pcm_rms": 0.0, "pcm_max_abs": 0.0, "max_abs_sync_quality_diff": 0.0, "max_abs_decode_error_diff": 0.0}
    junk_diff = n_patterns = 0
    found = 0
    clips = []
    # the reference side runs in 8 worker processes (its own thread pool does not scale to one 30 s clip: 1.2 s per clip on 256 threads)
    import concurrent.futures
    import multiprocessing
    t0 = time.perf_counter()
    with concurrent.futures.ProcessPoolExecutor(max_workers=8, mp_context=multiprocessing.get_context("spawn")) as pool:
        reference = list(pool.map(_reference_clip, range(1, 65)))
    t_ref = time.perf_counter() - t0
    for k, x, ref_w, ref_pats in reference:
        key = gpu.awm.test_key(k)
        assert np.array_equal(x, quantise16(gpu.awm.binding.gen_noise(key, 2 * n)))       # same input on both sides
        w = gpu.ctx.add_watermark(key, PAY1, gpu.dev(x))
        r, m = rms_max(w.cpu().numpy(), ref_w)
        assert r < RMS_TOL and m < 4e-6, f"clip {k}: embedded PCM differs: rms {r}, max {m}"
        d = gpu.dev(ref_w)
        rep = compare_patterns(gpu.ctx.get_watermark(key, d), ref_pats, f"clip {k}")
        found += any(p["bits"] == PAY1 for p in ref_pats)
        junk_diff += rep["noise_patterns_with_other_bits"]
        n_patterns += rep["patterns"]
        worst["pcm_rms"] = max(worst["pcm_rms"], r)
        worst["pcm_max_abs"] = max(worst["pcm_max_abs"], m)
        for f in ("max_abs_sync_quality_diff", "max_abs_decode_error_diff"):
            worst[f] = max(worst[f], rep[f])
        clips.append((d, ref_pats, gpu.dev(x), w, key))
    assert found >= 60                                                       # clip-decoder-test.sh expects the payload from a 30 s clip
    # the batch entry points with ONE KEY PER CLIP (awm_add_watermark_batch_keys_d / awm_get_watermark_batch_keys_d: the key tables of
    # a group travel in one copy and are indexed per clip inside the group's launches): all 64 clips in one call each -- the PCM of the
    # single calls bit for bit, the reference's pattern lists clip by clip
    keys = [c[4] for c in clips]
    outs = gpu.ctx.add_watermark_batch_keys(keys, PAY1, [c[2] for c in clips])
    for o, c in zip(outs, clips):
        assert gpu.torch.equal(o, c[3])
    batch_keys = gpu.ctx.get_watermark_batch_keys(keys, [c[0] for c in clips])
    batch_ties = 0
    for k, (b, c) in enumerate(zip(batch_keys, clips), start=1):
        assert [pkey(p) for p in b] == [pkey(p) for p in gpu.ctx.get_watermark(c[4], c[0])], f"batch clip {k} differs from the single call"
        batch_ties += compare_patterns(b, c[1], f"batch clip {k} (per-clip keys)")["refinement_ties"]
    assert sum(any(p["bits"] == PAY1 for p in b) for b in batch_keys) >= 60
    worst["batch_with_per_clip_keys"] = {"clips": len(clips), "equal_to_single_calls": True, "refinement_ties_vs_reference": batch_ties}
    clips = clips[:16]
    # the batch entry point (one key for the whole batch): the same 16 clips decoded with key 1 -- clip 1 carries it, the others do not
    key1 = gpu.awm.test_key(1)
    batch = gpu.ctx.get_watermark_batch(key1, [c[0] for c in clips])
    singles = [gpu.ctx.get_watermark(key1, c[0]) for c in clips]
    assert [[pkey(p) for p in b] for b in batch] == [[pkey(p) for p in s] for s in singles]
    compare_patterns(batch[0], clips[0][1], "batch clip 1")
    worst.update({"clips": 64, "patterns": n_patterns, "noise_patterns_with_other_bits": junk_diff, "clips_with_payload": found, "reference_seconds_for_64_clips_in_8_processes": round(t_ref, 2),
                  "reference_threads": os.cpu_count()})
    REPORT["config4_64_clips"] = worst
    print("configs[4] sample:", worst)
