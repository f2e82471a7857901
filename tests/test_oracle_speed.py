"""The oracle's restatement of the speed detection (wmspeed.cc) and of the VResampler call sequences (resample.cc:96-125)
against the golden vectors made from the compiled reference (tests/golden/make_speed_golden.py), and -- where
oracle/_ref is present -- against the compiled reference directly.  zita-resampler itself is absent from the reference
tree: both sides run the restated zita classes (parity with the real library unpinned, oracle/zita_restated.h)."""
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as orc
import _ref

HERE = os.path.dirname(os.path.abspath(__file__))
KEY = bytes(range(16))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


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


def test_resample_ratio_and_clip_location(golden, marked):
    C = golden["channels"]
    for speed, case in golden["cases"].items():
        z = orc.resample_ratio(marked, C, 1 / float(speed))
        assert len(z) // C == case["frames"] and sha(z) == case["sha"]
        assert orc.speed_clip_location(KEY, z, C, 25.0) == case["clip_location_25"]


def test_scan_pass_and_mags(golden, marked):
    C = golden["channels"]
    case = golden["cases"]["0.9764"]
    z = orc.resample_ratio(marked, C, 1 / 0.9764)
    mags = orc.speed_mags(KEY, z, C, case["clip_location_25"], 0.98, 25.0)
    assert mags.shape[0] == case["mags_rows"]
    for r, col, u, d in case["mags_samples"]:
        assert (float(mags[r, col, 0]), float(mags[r, col, 1])) == (u, d)
    s, q = orc.speed_scan(KEY, z, C, case["clip_location_25"], 25.0, 1.0007, 5, 2, [0.98])
    assert s.tolist() == case["scan_speed"] and q.tolist() == case["scan_quality"]


def test_detect_speed_and_decode(golden, marked):
    C = golden["channels"]
    for speed in ("0.9764", "1"):
        case = golden["cases"][speed]
        z = orc.resample_ratio(marked, C, 1 / float(speed))
        use, best, quality = orc.detect_speed(KEY, z, C, False)
        assert use == case["detect"]
        if speed != "1":
            assert abs(best - float(speed)) / float(speed) < 1e-3 and quality > 1
    case = golden["cases"]["0.9764"]
    z = orc.resample_ratio(marked, C, 1 / 0.9764)
    orc.set_speed_params(True, False, -1)
    try:
        pats = orc.decode_chunk(KEY, z, C, True)
    finally:
        orc.set_speed_params(False, False, -1)
    want = case["decode_detect_speed"]
    assert [(p["time"], p["sync_index"], p["type"], p["block_type"], p["bits"], p["speed"], p["sync_quality"]) for p in pats] == \
           [(p["time"], p["sync_index"], p["type"], p["block_type"], p["bits"], p["speed"], p["sync_quality"]) for p in want]
    assert any(p["bits"] == golden["payload"] and p["speed"] != 1 for p in pats)


def test_select_and_smooth_helpers():
    rng = np.random.default_rng(5)
    speed = np.sort(rng.uniform(0.8, 1.25, 40))
    quality = rng.uniform(0, 1, 40)
    quality[10] = quality[11] = 2.0                      # double peak: only the first of two equal values is kept
    s, q = orc.speed_select_n_best(speed, quality, 5)
    assert len(s) == 5 and q[0] == 2.0 and s[0] == speed[10] and list(q) == sorted(q, reverse=True)
    peaks = [i for i in range(40) if (quality[i - 1] if i else 0) <= quality[i] >= (quality[i + 1] if i < 39 else 0)]
    assert set(s) <= set(speed[peaks])
    if _ref.available():
        rs, rq = _ref.speed_select_n_best(speed, quality, 5)
        assert list(rs) == list(s) and list(rq) == list(q)
    sp = 0.97 + np.arange(81) * 0.00005
    qq = np.exp(-((sp - 0.9712) / 0.0004) ** 2) + rng.uniform(0, 0.05, 81)
    best = orc.speed_smooth_best(sp, qq, 1 - 1.00005, 20)
    assert abs(best - 0.9712) < 2e-4
    if _ref.available():
        assert _ref.speed_smooth_best(sp, qq, 1 - 1.00005, 20) == best


@pytest.mark.skipif(not _ref.available(), reason="oracle/_ref not built")
def test_patient_mode_matches_reference(golden, marked):
    C = golden["channels"]
    z = orc.resample_ratio(marked, C, 1 / 1.01)
    use, best, quality = orc.detect_speed(KEY, z, C, True)
    assert use == golden["cases"]["1.01"]["detect_patient"] == _ref.detect_speed(KEY, z, C, True)


def test_short_and_silent_inputs():
    assert orc.detect_speed(KEY, np.zeros(2 * 5000, np.float32), 2)[0] is None            # < 0.25 s: no search at all
    # digital silence: every score is 0, the second pass searches around "speed 0" and the reference exits with
    # "failed to setup vresampler with ratio=0.000000" (resample.cc:110-114)
    with pytest.raises(RuntimeError, match="vresampler"):
        orc.detect_speed(KEY, np.zeros(2 * 44100 * 3, np.float32), 2)
