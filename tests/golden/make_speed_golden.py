#!/usr/bin/env python3
"""Regenerates tests/golden/speed_v1.json from the COMPILED REFERENCE (oracle/_ref/libawm_ref.so: the unmodified
src/wmspeed.cc, resample.cc, wmget.cc; zita-resampler replaced by oracle/zita_restated.h).  Run in the build container:

    make -C oracle ref && python tests/golden/make_speed_golden.py

The scenario is the reference's tests/detect-speed-test.sh: 30 s of key-generated noise, watermarked, replayed at another
speed (test-change-speed = resample_ratio (1 / speed)), then `get --detect-speed [--patient]`.  The tests rebuild the
input with the oracle and check its checksum against the one stored here, so that only outputs are kept."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _ref  # noqa: E402

KEY = bytes(range(16))
PAYLOAD = "0123456789abcdef0011223344556677"
SECONDS, CHANNELS = 30, 2


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert _ref.available(), "build oracle/_ref first"
    x = _ref.gen_noise(KEY, SECONDS * 44100 * CHANNELS)
    y = _ref.add(KEY, x, CHANNELS, PAYLOAD)
    out = {"version": 1, "seconds": SECONDS, "channels": CHANNELS, "payload": PAYLOAD, "marked_sha": sha(y), "cases": {}}
    for speed in (0.9764, 1.0, 1.01):
        z = _ref.resample_ratio(y, CHANNELS, 1 / speed)
        c = {"frames": len(z) // CHANNELS, "sha": sha(z)}
        c["clip_location_25"] = _ref.speed_clip_location(KEY, z, CHANNELS, 25.0)
        mags = _ref.speed_mags(KEY, z, CHANNELS, c["clip_location_25"], 0.98, 25.0)
        c["mags_rows"] = int(mags.shape[0])
        c["mags_samples"] = [[r, col, float(mags[r, col, 0]), float(mags[r, col, 1])]
                             for r, col in ((0, 0), (1, 17), (100, 509), (2000, 255), (mags.shape[0] - 1, 300))]
        s, q = _ref.speed_scan(KEY, z, CHANNELS, c["clip_location_25"], 25.0, 1.0007, 5, 2, [0.98])
        c["scan_speed"] = s.tolist()
        c["scan_quality"] = q.tolist()
        c["detect"] = _ref.detect_speed(KEY, z, CHANNELS, False)
        c["detect_patient"] = _ref.detect_speed(KEY, z, CHANNELS, True)
        _ref.set_speed_params(True, False, -1)
        pats = _ref.decode_chunk(KEY, z, CHANNELS, True)
        _ref.set_speed_params(False, False, -1)
        c["decode_detect_speed"] = [dict(time=p["time"], sync_index=p["sync_index"], type=p["type"], block_type=p["block_type"],
                                         bits=p["bits"], speed=p["speed"], sync_quality=p["sync_quality"]) for p in pats]
        out["cases"]["%g" % speed] = c
        print(speed, c["frames"], c["detect"], c["detect_patient"], len(pats))
    with open(os.path.join(HERE, "speed_v1.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
