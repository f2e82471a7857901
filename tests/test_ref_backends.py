"""The two builds of the UNMODIFIED reference (oracle/Makefile: `ref` = double-precision FFT behind fftw3.h, `ref_mkl` = MKL's float
FFTW3 wrapper) against each other: what changes in the reference's own results when only its FFT library changes -- the calibration
behind the parity bars (tools/ref_backend_census.py).  CPU only; skipped where the MKL build
is not there."""
import os

import numpy as np
import pytest

import _ref

pytestmark = pytest.mark.skipif(not (_ref.available() and os.path.exists(_ref.PATH_MKL)), reason="oracle/_ref builds (ref + ref_mkl) not there")

PAY = "f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0"


def test_the_fft_library_moves_the_references_results_at_rounding_level_only():
    os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
    n = 200 * 44100
    x = _ref.gen_noise(None, 2 * n)
    x = (np.trunc(x.astype(np.float64) * 32768.0) / 32768.0).astype(np.float32)
    try:
        _ref.use_backend("double")
        w_d = _ref.add(None, x, 2, PAY)
        s_d = _ref.sync_search(None, w_d, 2)
        _ref.use_backend("mkl")
        w_m = _ref.add(None, x, 2, PAY)
        s_m = _ref.sync_search(None, w_d, 2)                    # (the same samples for both detectors)
    finally:
        _ref.use_backend("double")
    d = w_d.astype(np.float64) - w_m.astype(np.float64)
    assert 0 < np.sqrt(np.mean(d * d)) < 1e-7 and np.abs(d).max() < 1e-6          # embedded PCM: ~1e-8 RMS apart, not identical
    assert len(s_d[0]) == len(s_m[0]) == 8                                       # 8 sync scores in this fixture
    for i_d, q_d, b_d, i_m, q_m, b_m in zip(*s_d, *s_m):
        assert b_d == b_m and abs(int(i_d) - int(i_m)) <= 8 and abs(q_d - q_m) < 1e-5
    assert max(abs(a - b) for a, b in zip(s_d[1], s_m[1])) > 0                   # qualities differ in the 6th digit
    # the block positions of this fixture
    assert [int(i) for i in s_d[0]][:2] == [256000, 2164480]
