"""Pins the CPU restatement against fixtures generated from the compiled reference
(tests/golden/make_golden.py) and -- where oracle/_ref exists -- against the compiled reference directly."""
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as orc
import _ref

HERE = os.path.dirname(os.path.abspath(__file__))
PAY1 = "0123456789abcdef0011223344556677"
PAY2 = "f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0"
KEYS = {"zero": bytes(16), "test42": (42).to_bytes(8, "big") + bytes(8), "ramp": bytes(range(16))}


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "golden_v1.json")) as f:
        j = json.load(f)
    return j, np.load(os.path.join(HERE, "golden", "golden_v1.npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def noise(seed, n, ch):
    return np.random.default_rng(seed).uniform(-1, 1, (n, ch)).astype(np.float32)


def pkey(p):
    return (round(p["time"], 9), p["sync_index"], p["type"], p["block_type"], p["bits"])


@pytest.mark.parametrize("name", list(KEYS))
def test_tables(golden, name):
    j, _ = golden
    k, key = j["keys"][name], KEYS[name]
    assert [int(v) for v in orc.random_u64(key, 0xf00f1234b00b5678, 5, 24)] == k["prng_stream5_seed_f00f"]
    assert sha(orc.bit_pos(key)) == k["bit_pos_sha"]
    assert sha(orc.mix_entries(key)) == k["mix_entries_sha"]
    assert sha(orc.bit_order(key, 858)) == k["bit_order_858_sha"]
    assert sha(orc.sync_bits(key, False)) == k["sync_bits_block_sha"]
    assert sha(orc.sync_bits(key, True)) == k["sync_bits_clip_sha"]
    for pay in (PAY1, PAY2):
        for ab in (0, 1):
            assert sha(orc.frame_mod(key, pay, ab)) == k[f"frame_mod_{pay[:4]}_{ab}_sha"]
    assert [a.tolist() for a in orc.up_down(key, 2, 0)] == k["up_down_sync_0"]
    assert [a.tolist() for a in orc.up_down(key, 1, 1715)] == k["up_down_data_1715"]


def test_conv(golden):
    j, _ = golden
    bits = np.random.default_rng(11).integers(0, 2, 128)
    for bt in (0, 1, 2):
        coded = orc.conv_encode(bt, bits)
        assert sha(coded) == j["conv"][str(bt)]["coded_sha"]
        soft = np.clip(coded + np.random.default_rng(12 + bt).normal(0, 0.5, coded.shape), -1, 2).astype(np.float32)
        dec, err = orc.conv_decode_soft(bt, soft)
        assert dec.tolist() == j["conv"][str(bt)]["decoded"]
        assert err == np.float32(j["conv"][str(bt)]["error"])


def test_stft_and_add(golden):
    j, z = golden
    assert sha(orc.window(1024)) == j["window1024_sha"]
    got = orc.fft_range(noise(21, 8000, 2), 2, 100, 3)
    np.testing.assert_allclose(got, z["fft_range_s21"], rtol=0, atol=2e-7)      # identical FFT algorithm: expect 0
    orc.set_params(test_no_limiter=False)
    a = orc.add(None, noise(31, 2 * 44100 + 77, 1), 1, PAY1)
    orc.set_params(test_no_limiter=True)
    b = orc.add(KEYS["test42"], noise(32, 44100 + 500, 2), 2, PAY2)
    orc.set_params()
    assert np.sqrt(np.mean((a - z["add_mono_s31"]) ** 2)) < 1e-7
    assert np.sqrt(np.mean((b - z["add_stereo_s32_nolimiter"]) ** 2)) < 1e-7
    assert np.array_equal(a, z["add_mono_s31"]) and np.array_equal(b, z["add_stereo_s32_nolimiter"])


@pytest.fixture(scope="module")
def stream70():
    n = 70 * 44100
    return orc.add(None, noise(41, n, 2), 2, PAY1).reshape(n, 2)


def test_sync_search(golden, stream70):
    j, z = golden
    assert sha(stream70) == j["stream70_sha"]
    idx, raw, mean = orc.search_approx(None, stream70, 2)
    assert len(idx) == j["approx70"]["n"]
    top = np.argsort(-np.abs(raw - mean))[:24]
    assert idx[top].tolist() == j["approx70"]["top_index"]
    np.testing.assert_allclose(raw[top], j["approx70"]["top_raw"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(mean[top], j["approx70"]["top_mean"], rtol=0, atol=1e-12)
    si, sq, sb = orc.sync_search(None, stream70, 2)
    assert si.tolist() == j["sync70"]["index"] and sb.tolist() == j["sync70"]["block_type"]
    np.testing.assert_allclose(sq, j["sync70"]["quality"], rtol=0, atol=1e-12)
    got = orc.mix_decode(None, stream70, 2, j["mix_decode70_index"])
    np.testing.assert_array_equal(got, z["mix_decode70"])


def test_decode_and_clip(golden, stream70):
    j, _ = golden
    got = sorted(orc.decode_chunk(None, stream70, 2, True), key=lambda p: (p["time"], p["type"], p["block_type"], p["bits"]))
    want = j["decode_chunk70"]
    assert [pkey(p) for p in got] == [pkey(p) for p in want]
    for g, w in zip(got, want):
        assert abs(g["sync_quality"] - w["sync_quality"]) < 1e-12 and abs(g["decode_error"] - w["decode_error"]) < 1e-7
    assert any(p["bits"] == PAY1 for p in got)
    clip = stream70[20 * 44100: 44 * 44100]
    got = orc.get(None, clip, 2)
    assert [pkey(p) for p in got] == [pkey(p) for p in j["get_clip24"]]
    assert got[0]["bits"] == PAY1 and got[0]["type"] == 1


@pytest.mark.skipif(not _ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_against_compiled_reference():
    rng = np.random.default_rng(99)
    key = bytes(rng.integers(0, 256, 16, dtype=np.uint8))
    assert np.array_equal(orc.mix_entries(key), _ref.mix_entries(key))
    assert np.array_equal(orc.frame_mod(key, PAY2, 1), _ref.frame_mod(key, PAY2, 1))
    x = rng.uniform(-1, 1, (56 * 44100 + 3, 1)).astype(np.float32)
    a, b = orc.add(key, x, 1, PAY2), _ref.add(key, x, 1, PAY2)
    assert np.array_equal(a, b)
    w = a.reshape(-1, 1)
    for u, v in zip(orc.sync_search(key, w, 1), _ref.sync_search(key, w, 1)):
        assert np.array_equal(u, v)
    assert [pkey(p) for p in orc.get(key, w, 1)] == [pkey(p) for p in _ref.get(key, w, 1)]
