#!/usr/bin/env python3
"""Regenerates tests/golden/golden_v1.{json,npz} from the COMPILED REFERENCE (oracle/_ref/libawm_ref.so,
i.e. the unmodified sources under /root/reference built by oracle/Makefile).  Run in the build container:

    make -C oracle ref && python tests/golden/make_golden.py

The fixtures pin the CPU restatement (oracle/awm_oracle.cc) on machines where /root/reference does not exist.
Inputs are regenerated from numpy seeds by the tests (PCG64 + uniform are stable across numpy versions),
so only outputs are stored."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _ref  # noqa: E402

PAY1 = "0123456789abcdef0011223344556677"
PAY2 = "f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def noise(seed, n, ch):
    return np.random.default_rng(seed).uniform(-1, 1, (n, ch)).astype(np.float32)


def main():
    assert _ref.available(), "build oracle/_ref first"
    j = {"version": 1, "keys": {}}
    arrays = {}
    keys = {"zero": bytes(16), "test42": (42).to_bytes(8, "big") + bytes(8), "ramp": bytes(range(16))}
    for name, key in keys.items():
        k = {}
        k["prng_stream5_seed_f00f"] = [int(v) for v in _ref.random_u64(key, 0xf00f1234b00b5678, 5, 24)]
        k["bit_pos_sha"] = sha(_ref.bit_pos(key))
        k["bit_pos_head"] = _ref.bit_pos(key)[:12].tolist()
        k["mix_entries_sha"] = sha(_ref.mix_entries(key))
        k["bit_order_858_sha"] = sha(_ref.bit_order(key, 858))
        k["sync_bits_block_sha"] = sha(_ref.sync_bits(key, False))
        k["sync_bits_clip_sha"] = sha(_ref.sync_bits(key, True))
        for pay in (PAY1, PAY2):
            for ab in (0, 1):
                k[f"frame_mod_{pay[:4]}_{ab}_sha"] = sha(_ref.frame_mod(key, pay, ab))
        k["up_down_sync_0"] = [a.tolist() for a in _ref.up_down(key, 2, 0)]
        k["up_down_data_1715"] = [a.tolist() for a in _ref.up_down(key, 1, 1715)]
        j["keys"][name] = k
    j["window1024_sha"] = sha(_ref.window(1024))
    j["noise_zero_key_first8_int16"] = [int(v * 32768) for v in _ref.gen_noise(None, 8)]

    # convolutional code
    bits = np.random.default_rng(11).integers(0, 2, 128)
    j["conv"] = {}
    for bt in (0, 1, 2):
        coded = _ref.conv_encode(bt, bits)
        soft = np.clip(coded + np.random.default_rng(12 + bt).normal(0, 0.5, coded.shape), -1, 2).astype(np.float32)
        dec, err = _ref.conv_decode_soft(bt, soft)
        j["conv"][str(bt)] = {"coded_sha": sha(coded), "decoded": dec.tolist(), "error": float(err)}

    # STFT
    x = noise(21, 8000, 2)
    arrays["fft_range_s21"] = _ref.fft_range(x, 2, 100, 3)

    # add: mono with limiter, stereo without, odd lengths
    _ref.set_params(test_no_limiter=False)
    arrays["add_mono_s31"] = _ref.add(None, noise(31, 2 * 44100 + 77, 1), 1, PAY1)
    _ref.set_params(test_no_limiter=True)
    arrays["add_stereo_s32_nolimiter"] = _ref.add(keys["test42"], noise(32, 44100 + 500, 2), 2, PAY2)
    _ref.set_params()

    # sync + decode on a 70 s stereo stream and a 24 s clip cut out of it
    n = 70 * 44100
    w = _ref.add(None, noise(41, n, 2), 2, PAY1).reshape(n, 2)
    j["stream70_sha"] = sha(w)
    idx, raw, mean = _ref.search_approx(None, w, 2)
    top = np.argsort(-np.abs(raw - mean))[:24]
    j["approx70"] = {"n": int(len(idx)), "top_index": idx[top].tolist(), "top_raw": raw[top].tolist(), "top_mean": mean[top].tolist()}
    si, sq, sb = _ref.sync_search(None, w, 2)
    j["sync70"] = {"index": si.tolist(), "quality": sq.tolist(), "block_type": sb.tolist()}
    arrays["mix_decode70"] = _ref.mix_decode(None, w, 2, int(si[np.argmax(sq)]))
    j["mix_decode70_index"] = int(si[np.argmax(sq)])
    j["decode_chunk70"] = sorted(_ref.decode_chunk(None, w, 2, True), key=lambda p: (p["time"], p["type"], p["block_type"], p["bits"]))
    clip = w[20 * 44100: 44 * 44100]
    j["get_clip24"] = _ref.get(None, clip, 2)
    with open(os.path.join(HERE, "golden_v1.json"), "w") as f:
        json.dump(j, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **arrays)
    print("golden written:", {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    main()
