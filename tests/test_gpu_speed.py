"""Speed detection (`get --detect-speed`, reference wmspeed.cc) and the variable-ratio resampler behind it, HIP path through
the C ABI against the oracle and the golden vectors of the compiled reference (tests/golden/speed_v1.json).

Tolerances: the detected speed is a float result of a search over float scores -- it has to agree within 2e-6 (two steps of
the final smoothing grid, wmspeed.cc:407); scores within 1e-4, magnitudes within 1e-2 dB-sum units (values ~2000); stretched
PCM within 1e-6; everything that is decoded from the stretched stream (payload bits, sync positions, pattern types) exactly."""
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KEY = bytes(range(16))
SPEED_TOL = 2e-6


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


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
    awm.set_speed_params()
    orc.set_speed_params(False, False, -1)
    ctx.close()


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "speed_v1.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def marked(golden):
    C = golden["channels"]
    x = orc.gen_noise(KEY, golden["seconds"] * 44100 * C)
    y = orc.add(KEY, x, C, golden["payload"])
    assert sha(y) == golden["marked_sha"]
    return y


@pytest.fixture(scope="module")
def replayed(golden, marked):
    """the reference's detect-speed-test.sh inputs: the marked noise replayed at 0.9764 / 1.0 / 1.01 (oracle's resampler)"""
    out = {}
    for speed, case in golden["cases"].items():
        z = orc.resample_ratio(marked, golden["channels"], 1 / float(speed))
        assert sha(z) == case["sha"]
        out[speed] = z
    return out


@pytest.mark.parametrize("ratio,ch", [(1 / 0.9764, 2), (1 / 1.01, 2), (0.49, 1), (0.8, 3), (1.25, 2), (2.5, 1)])
def test_resample_ratio_matches_restated_zita(gpu, ratio, ch):
    x = np.random.default_rng(int(ratio * 1000)).uniform(-1, 1, (200000 + ch, ch)).astype(np.float32)
    got = gpu.ctx.resample_ratio(gpu.dev(x, ch), ratio).cpu().numpy()
    want = orc.resample_ratio(x, ch, ratio).reshape(-1, ch)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-6
    # truncated input (prepare_mags uses it)
    got = gpu.ctx.resample_ratio(gpu.dev(x, ch), ratio, max_in_seconds=1.5).cpu().numpy()
    want = orc.resample_ratio(x, ch, ratio, max_in_seconds=1.5).reshape(-1, ch)
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-6


def test_clip_location_and_magnitudes(gpu, golden, replayed):
    for speed in ("0.9764", "1.01"):
        case = golden["cases"][speed]
        z = replayed[speed]
        zd = gpu.dev(z)
        assert gpu.ctx.speed_clip_location(KEY, zd, 25.0) == case["clip_location_25"]
        mags = gpu.ctx.speed_mags(KEY, zd, case["clip_location_25"], 0.98, 25.0)
        assert mags.shape[0] == case["mags_rows"]
        for r, col, u, d in case["mags_samples"]:
            assert abs(float(mags[r, col, 0]) - u) < 1e-2 and abs(float(mags[r, col, 1]) - d) < 1e-2
    want = orc.speed_mags(KEY, replayed["1.01"], 2, golden["cases"]["1.01"]["clip_location_25"], 0.98, 25.0)
    assert np.abs(mags - want).max() < 1e-2


def test_scan_pass(gpu, golden, replayed):
    case = golden["cases"]["0.9764"]
    s, q = gpu.ctx.speed_scan(KEY, gpu.dev(replayed["0.9764"]), case["clip_location_25"], 25.0, 1.0007, 5, 2, [0.98])
    assert s.tolist() == case["scan_speed"]
    assert np.abs(q - np.array(case["scan_quality"])).max() < 1e-4
    assert abs(s[q.argmax()] - 0.9764) < 1e-3


@pytest.mark.parametrize("patient", [False, True])
def test_detect_speed(gpu, golden, replayed, patient):
    for speed, case in golden["cases"].items():
        use, best, quality = gpu.ctx.detect_speed(KEY, gpu.dev(replayed[speed]), patient)
        want = case["detect_patient" if patient else "detect"]
        if want is None:
            assert use is None                           # replay speed 1: nothing to correct (|speed - 1| < 1e-4)
            assert abs(best - 1) < 1e-4 and quality > 1
        else:
            assert abs(use - want) <= SPEED_TOL
            assert abs(use - float(speed)) / float(speed) < 2e-4


def test_decode_with_detect_speed(gpu, golden, replayed):
    """decode() with --detect-speed (wmget.cc:886-939): the patterns of the stretched stream come first, then the normal ones"""
    for speed in ("0.9764", "1.01", "1"):
        want = golden["cases"][speed]["decode_detect_speed"]
        gpu.awm.set_speed_params(detect_speed=True)
        try:
            got = gpu.ctx.decode_chunk(KEY, gpu.dev(replayed[speed]), True)
            got_all = gpu.ctx.get_watermark(KEY, gpu.dev(replayed[speed]))
        finally:
            gpu.awm.set_speed_params()
        assert [(p["sync_index"], p["type"], p["block_type"], p["bits"]) for p in got] == \
               [(p["sync_index"], p["type"], p["block_type"], p["bits"]) for p in want]
        for g, w in zip(got, want):
            assert abs(g["speed"] - w["speed"]) <= SPEED_TOL and abs(g["time"] - w["time"]) < 1e-3
            assert abs(g["sync_quality"] - w["sync_quality"]) < 1e-4
        hits = [p for p in got_all if p["bits"] == golden["payload"]]
        assert hits and (speed == "1") == all(p["speed"] == 1 for p in hits)


def test_try_speed(gpu, golden, replayed):
    gpu.awm.set_speed_params(try_speed=0.9764)
    orc.set_speed_params(False, False, 0.9764)
    try:
        got = gpu.ctx.decode_chunk(KEY, gpu.dev(replayed["0.9764"]), True)
        want = orc.decode_chunk(KEY, replayed["0.9764"], 2, True)
    finally:
        gpu.awm.set_speed_params()
        orc.set_speed_params(False, False, -1)
    assert [(p["sync_index"], p["type"], p["block_type"], p["bits"], p["speed"]) for p in got] == \
           [(p["sync_index"], p["type"], p["block_type"], p["bits"], p["speed"]) for p in want]
    assert any(p["bits"] == golden["payload"] and p["speed"] == 0.9764 for p in got)


def test_mono_and_long_input_against_oracle(gpu):
    """mono, 100 s (the block decoder finds whole blocks on the stretched stream; the clip location matters)"""
    n = 100 * 44100
    x = orc.gen_noise(KEY, n)
    y = orc.add(KEY, x, 1, "f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0")
    z = orc.resample_ratio(y, 1, 1 / 1.07)
    zd = gpu.dev(z, 1)
    use, best, quality = gpu.ctx.detect_speed(KEY, zd)
    o_use, o_best, o_quality = orc.detect_speed(KEY, z, 1)
    assert abs(use - o_use) <= SPEED_TOL and abs(quality - o_quality) < 1e-4 and abs(use - 1.07) < 3e-4
    gpu.awm.set_speed_params(detect_speed=True)
    orc.set_speed_params(True, False, -1)
    try:
        got = gpu.ctx.get_watermark(KEY, zd)
        want = orc.get(KEY, z, 1)
    finally:
        gpu.awm.set_speed_params()
        orc.set_speed_params(False, False, -1)
    assert [(p["sync_index"], p["type"], p["block_type"], p["bits"]) for p in got] == \
           [(p["sync_index"], p["type"], p["block_type"], p["bits"]) for p in want]
    assert all(abs(g["speed"] - w["speed"]) <= SPEED_TOL for g, w in zip(got, want))
    assert sum(p["bits"] == "f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0" and p["speed"] != 1 for p in got) >= 2


def test_detect_speed_with_three_frames_per_bit(gpu):
    """--frames-per-bit 3 (blocks of 3084 frames): the speed search's tables follow the geometry (the sync frames are drawn over the whole
    block), the tables cached per key are keyed by it: 130 s stereo replayed 4 % slow, `get --detect-speed` against the oracle, then the
    default geometry again with the same key."""
    pay = "f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0"
    pk = lambda p: (p["sync_index"], p["type"], p["block_type"], p["bits"])

    def both(fpb):
        gpu.awm.set_params(frames_per_bit=fpb)
        orc.set_params(frames_per_bit=fpb)
        gpu.awm.set_speed_params(detect_speed=True)
        orc.set_speed_params(True, False, -1)
        try:
            y = orc.add(KEY, orc.gen_noise(KEY, 130 * 44100 * 2), 2, pay)
            z = orc.resample_ratio(y, 2, 1 / 0.96)
            return gpu.ctx.get_watermark(KEY, gpu.dev(z)), orc.get(KEY, z, 2)
        finally:
            gpu.awm.set_speed_params()
            orc.set_speed_params(False, False, -1)
            gpu.awm.set_params()
            orc.set_params()
    for fpb in (3, 2):
        got, want = both(fpb)
        assert [pk(p) for p in got] == [pk(p) for p in want], fpb
        assert all(abs(g["speed"] - w["speed"]) <= SPEED_TOL for g, w in zip(got, want))
        assert sum(p["bits"] == pay and abs(p["speed"] - 0.96) < 3e-4 for p in got) >= 1, fpb


def test_short_silent_and_one_silent_channel(gpu):
    assert gpu.ctx.detect_speed(KEY, gpu.dev(np.zeros((5000, 2), np.float32)))[0] is None        # < 0.25 s
    # digital silence: every score is 0, the second pass searches around "speed 0": the reference exits with
    # "failed to setup vresampler with ratio=0.000000" (resample.cc:110-114), the library reports the same error
    with pytest.raises(gpu.awm.AwmError, match="failed to setup vresampler with ratio=0.000000"):
        gpu.ctx.detect_speed(KEY, gpu.dev(np.zeros((3 * 44100, 2), np.float32)))
    # one digitally silent channel: its bands are -96 dB in the reference (exact zeros), not the other channel's rounding noise
    x = np.random.default_rng(77).uniform(-1, 1, (20 * 44100, 2)).astype(np.float32)
    x[:, 1] = 0
    loc = orc.speed_clip_location(KEY, x, 2, 25.0)
    got = gpu.ctx.speed_mags(KEY, gpu.dev(x), loc, 1.0, 25.0)
    want = orc.speed_mags(KEY, x.ravel(), 2, loc, 1.0, 25.0)
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-2


def test_config3_full_size_48k_detect_speed(gpu):
    """BASELINE.json configs[2] at full size (60 min stereo 48 kHz, `get --detect-speed`), through size independent properties:
    the stream is watermarked at 48 kHz, replayed 2 % fast, read back like the reference's loader does (48 -> 44.1 kHz) and
    decoded with speed detection: every 30-minute chunk finds the replay speed on its own and the payload is recovered from
    the stretched stream all along the file, at the original time positions."""
    t = gpu.torch
    rate, speed, payload = 48000, 1.02, "0123456789abcdef0011223344556677"
    n = 60 * 60 * rate
    g = t.Generator(device="cuda")
    g.manual_seed(9)
    x = t.rand((n, 2), generator=g, device="cuda", dtype=t.float32) * 2 - 1
    w = gpu.ctx.add_watermark(KEY, payload, x, sample_rate=rate)
    del x
    fast = gpu.ctx.resample_ratio(w, 1 / speed, rate=rate)                  # test-change-speed
    assert abs(fast.shape[0] - n / speed) < 1
    del w
    y = gpu.ctx.resample(fast, rate, 44100)                                 # WavChunkLoader
    del fast
    plain = gpu.ctx.get_watermark(KEY, y)
    assert not any(p["bits"] == payload for p in plain)                     # undecodable without the speed correction
    gpu.awm.set_speed_params(detect_speed=True)
    try:
        pats = gpu.ctx.get_watermark(KEY, y)
    finally:
        gpu.awm.set_speed_params()
    hits = [p for p in pats if p["bits"] == payload]
    speeds = sorted({p["speed"] for p in hits})
    assert hits and all(p["speed"] != 1 for p in hits)
    assert len(speeds) == len(gpu.awm.plan_chunks(y.shape[0])) == 3         # every chunk detects its own speed
    assert all(abs(s - speed) / speed < 2e-4 for s in speeds)
    # block positions: pattern times are seconds of the replayed file (index in the stretched stream / int (44100 * speed),
    # wmget.cc:543,916): consecutive A / B blocks are 2226 frames of the original apart, i.e. 2226 * 1024 / 44100 / speed seconds
    block_s = 2226 * 1024 / 44100 / speed
    starts = np.array(sorted(p["time"] for p in hits if p["type"] == 0 and p["block_type"] < 2))
    assert len(starts) >= 60
    gaps = np.diff(starts)
    steps = np.round(gaps / block_s)
    assert np.all(steps >= 1) and np.abs(gaps - steps * block_s).max() < 0.1 and (steps == 1).sum() >= 50
    assert len(plain) > 0 and len(pats) > len(plain)


@pytest.mark.parametrize("cuts", [[0.5], [0.27, 0.7]])
def test_detect_speed_over_several_contexts_equals_single(gpu, cuts):
    """`get --detect-speed` through the multi-GPU protocol (awm_multi_get_d with 2 / 3 contexts on the one device): the speed part is
    sharded by chunk (the rank that holds most of a chunk fetches the rest and runs speed search, stretch and the decoders of the
    stretched copy), the plain decoders by position as always; the merged list -- stretched and plain patterns, their speeds,
    qualities and error values -- equals the single-context get of the whole stream.  26 minutes at 10 minute chunks, replayed 3 %
    slow, the cuts inside chunks."""
    from audiowmark_amd import sharded
    t = gpu.torch
    payload, speed = "0123456789abcdef0011223344556677", 0.97
    g = t.Generator(device="cuda")
    g.manual_seed(31)
    x = t.rand((26 * 60 * 44100, 2), generator=g, device="cuda", dtype=t.float32) * 2 - 1
    y = gpu.ctx.resample_ratio(gpu.ctx.add_watermark(KEY, payload, x), 1 / speed)
    del x
    total = y.shape[0]
    gpu.awm.set_params(chunk_size_min=10.0)
    gpu.awm.set_speed_params(detect_speed=True)
    try:
        want = gpu.ctx.get_watermark(KEY, y)
        edges = [0] + [int(total * c) // 1024 * 1024 for c in cuts] + [total]
        spans = [y[a:b].contiguous() for a, b in zip(edges[:-1], edges[1:])]
        ctxs = [gpu.ctx] + [gpu.awm.Context(0) for _ in spans[1:]]
        got = sharded.multi_get(ctxs, KEY, spans)
        for c in ctxs[1:]:
            c.close()
    finally:
        gpu.awm.set_speed_params()
        gpu.awm.set_params()
    key = lambda p: (round(p["time"], 6), p["sync_index"], p["type"], p["block_type"], p["bits"], p["speed"], p["sync_quality"], p["decode_error"])
    assert [key(p) for p in got] == [key(p) for p in want]
    hits = [p for p in want if p["bits"] == payload]
    assert len(hits) >= 20 and all(p["speed"] != 1 for p in hits)


def test_batch_of_clips_with_detect_speed(gpu, golden, replayed):
    """awm_get_watermark_batch_d with --detect-speed set: the clips are decoded one after the other (the speed search and the
    stretched copy live in per-context buffers) and every clip's result equals awm_get_watermark_d on it."""
    clips = [gpu.dev(replayed["0.9764"]), gpu.dev(replayed["1.01"]), gpu.dev(replayed["1"])]
    gpu.awm.set_speed_params(detect_speed=True)
    try:
        batch = gpu.ctx.get_watermark_batch(KEY, clips)
        single = [gpu.ctx.get_watermark(KEY, c) for c in clips]
    finally:
        gpu.awm.set_speed_params()
    assert batch == single
    assert [any(p["bits"] == golden["payload"] and p["speed"] != 1 for p in b) for b in batch] == [True, True, False]
    assert all(any(p["bits"] == golden["payload"] for p in b) for b in batch)


def test_three_channels_and_short_material(gpu):
    """odd channel count (the transform of a time step carries channel pairs: the last one rides alone) and material that is
    shorter than the 25 s / 50 s the passes ask for (fewer rows in the magnitude matrices)"""
    n, ch = 14 * 44100, 3
    x = orc.gen_noise(KEY, n * ch)
    y = orc.add(KEY, x, ch, "0123456789abcdef0011223344556677")
    z = orc.resample_ratio(y, ch, 1 / 0.93)
    zd = gpu.dev(z, ch)
    loc = orc.speed_clip_location(KEY, z, ch, 25.0)
    assert gpu.ctx.speed_clip_location(KEY, zd, 25.0) == loc
    want = orc.speed_mags(KEY, z, ch, loc, 0.93, 25.0)
    got = gpu.ctx.speed_mags(KEY, zd, loc, 0.93, 25.0)
    assert got.shape == want.shape and got.shape[0] < 4303 and np.abs(got - want).max() < 1e-2
    use, best, quality = gpu.ctx.detect_speed(KEY, zd)
    o_use, o_best, o_quality = orc.detect_speed(KEY, z, ch)
    assert (use is None) == (o_use is None) and abs(best - o_best) <= SPEED_TOL and abs(quality - o_quality) < 1e-4
    assert abs(best - 0.93) < 1e-3
