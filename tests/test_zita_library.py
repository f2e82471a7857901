"""zita-resampler: the restatement (oracle/zita_restated.h) against the LIBRARY, the day there is one.

SURVEY.md section 8 (f2): the reference resamples through zita-resampler (resample.cc:30-126), which is neither in the reference tree nor
in this image -- K10 / K12 are pinned to a restatement of its published algorithm, "parity unpinned".  This test arms itself: if a
libzita-resampler and its headers are installed, `oracle/zita_probe.cc` -- the reference's use of the two classes, written once against
the library's public interface -- is compiled against the library and against the restatement, and the two outputs must be equal bit
for bit (fixed ratios: 48 / 96 / 22.05 / 88.2 kHz <-> 44.1 kHz; variable ratios around 1 and far from it; 1 - 3 channels).  Without
the library the probe is still built against the restatement and held against the oracle's own driver (orc_resample /
orc_resample_ratio), so that the code that will do the pinning is exercised, and the library half skips saying why.
(CPU only; nothing of the product is involved.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import _oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(ROOT, "oracle", "zita_probe.cc")
OUT = os.path.join(ROOT, "oracle", "_ref")

FIXED = [(48000, 44100), (44100, 48000), (96000, 44100), (22050, 44100), (88200, 44100), (44100, 96000)]
RATIOS = [1.0 / 1.02, 1.0 / 0.98, 1.0 / 1.25, 1.0 / 0.8, 0.918749, 1.0883, 0.442992, 44100.0 / 33333.0]


def _build(name, extra):
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, name)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", so] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    return (so if r.returncode == 0 else None), r.stderr


def _load(so):
    lib = C.CDLL(so)
    lib.zita_probe_fixed.restype = C.c_size_t
    lib.zita_probe_fixed.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_size_t]
    lib.zita_probe_var.restype = C.c_size_t
    lib.zita_probe_var.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_uint, C.c_void_p, C.c_size_t]
    return lib


def _fixed(lib, x, ch, fs_in, fs_out):
    n = x.size // ch
    cap = int(n * fs_out / fs_in) + 64
    out = np.zeros(cap * ch, np.float32)
    got = lib.zita_probe_fixed(x.ctypes.data, n, ch, fs_in, fs_out, 16, out.ctypes.data, cap)
    assert got <= cap
    return out[:got * ch]


def _var(lib, x, ch, ratio):
    n = x.size // ch
    cap = int(n * ratio) + 64
    out = np.zeros(cap * ch, np.float32)
    got = lib.zita_probe_var(x.ctypes.data, n, ch, ratio, 16, out.ctypes.data, cap)
    assert got <= cap
    return out[:got * ch]


def _noise(seed, n, ch):
    return np.ascontiguousarray(np.random.default_rng(seed).uniform(-1, 1, n * ch).astype(np.float32))


@pytest.fixture(scope="module")
def restated():
    so, err = _build("libzita_probe_restated.so", ["-I", os.path.join(ROOT, "oracle", "ref_shim", "include")])
    assert so, err
    return _load(so)


def test_probe_driver_equals_the_oracles_use_of_the_restated_classes(restated):
    """the probe (the driver that will face the library) and the oracle (what K10 / K12 are pinned to) drive the restated classes the same
    way: identical streams for fixed and variable ratios"""
    for i, (fs_in, fs_out) in enumerate(FIXED):
        for ch in (1, 2):
            x = _noise(10 + i, 30000 + 517 * i, ch)
            assert np.array_equal(_fixed(restated, x, ch, fs_in, fs_out), orc.resample(x, ch, fs_in, fs_out)), (fs_in, fs_out, ch)
    # a pair of rates the fixed-ratio class refuses goes to the variable one (resample.cc:233-270)
    x = _noise(99, 41000, 2)
    assert _fixed(restated, x, 2, 33333, 44100).size == 0
    assert np.array_equal(_var(restated, x, 2, 44100.0 / 33333.0), orc.resample(x, 2, 33333, 44100))
    for i, ratio in enumerate(RATIOS[:6]):
        x = _noise(200 + i, 25000 + 311 * i, 2)
        got, want = _var(restated, x, 2, ratio), orc.resample_ratio(x, 2, ratio)
        n = min(got.size, want.size)                          # (the oracle's truncating form may stop a few frames earlier)
        assert n > 0 and abs(got.size - want.size) <= 2 * 40 and np.array_equal(got[:n], want[:n]), ratio


def test_restatement_equals_the_library_bit_for_bit(restated):
    """ARMS ITSELF: needs libzita-resampler + headers.  Fixed and variable ratios, 1 - 3 channels, ragged lengths: every output sample of
    the restated classes equals the library's."""
    so, err = _build("libzita_probe_library.so", ["-lzita-resampler"])
    if not so:
        pytest.skip("zita-resampler (library + headers) is not installed here: the restatement stays unpinned -- "
                    + (err.strip().splitlines()[0] if err.strip() else "g++ failed"))
    library = _load(so)
    for i, (fs_in, fs_out) in enumerate(FIXED):
        for ch in (1, 2, 3):
            x = _noise(1000 + 7 * i + ch, 50000 + 977 * i + ch, ch)
            a, b = _fixed(library, x, ch, fs_in, fs_out), _fixed(restated, x, ch, fs_in, fs_out)
            assert a.size == b.size and a.size > 0 and np.array_equal(a, b), ("fixed", fs_in, fs_out, ch)
    for i, ratio in enumerate(RATIOS):
        for ch in (1, 2, 3):
            x = _noise(2000 + 7 * i + ch, 40000 + 613 * i + ch, ch)
            a, b = _var(library, x, ch, ratio), _var(restated, x, ch, ratio)
            assert a.size == b.size and a.size > 0 and np.array_equal(a, b), ("variable", ratio, ch)
