"""ctypes view of oracle/_ref/libawm_ref.so -- the UNMODIFIED reference sources compiled by
oracle/Makefile.  TEST INFRASTRUCTURE ONLY.  `available()` is False when the library has not been
built (e.g. /root/reference absent and no prebuilt copy travelled with the repo)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libawm_ref.so")
PATH_MKL = os.path.join(ROOT, "oracle", "_ref", "libawm_ref_mkl.so")   # same sources, fftwf_* from MKL's float FFTW wrapper (`make -C oracle ref_mkl`)
BIN = os.path.join(ROOT, "oracle", "_ref", "audiowmark_ref")

_lib = None
_backend = "double"
_libs = {}


def use_backend(name):
    """switch every wrapper below between the two builds of the reference: "double" (oracle/ref_shim/fftw_shim.cc, the default and the
    only one the tests use) and "mkl" (MKL's single-precision FFTW3 wrapper; calibration only: tools/ref_backend_census.py)"""
    global _lib, _backend
    assert name in ("double", "mkl")
    _libs[_backend] = _lib
    _backend = name
    _lib = _libs.get(name)


def available():
    return os.path.exists(PATH)


class Pattern(C.Structure):
    _fields_ = [("time", C.c_double), ("sync_index", C.c_uint64), ("sync_quality", C.c_double),
                ("block_type", C.c_int), ("type", C.c_int), ("decode_error", C.c_float), ("speed", C.c_double),
                ("bits", C.c_int * 128), ("n_bits", C.c_int)]

    def hex(self):
        b = list(self.bits[:self.n_bits])
        return "".join("%x" % (b[i] * 8 + b[i + 1] * 4 + b[i + 2] * 2 + b[i + 3]) for i in range(0, len(b) - 3, 4))

    def as_dict(self):
        return dict(time=self.time, sync_index=int(self.sync_index), sync_quality=self.sync_quality,
                    block_type=self.block_type, type=self.type, decode_error=self.decode_error, speed=self.speed,
                    bits=self.hex())


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(PATH if _backend == "double" else PATH_MKL)
        _lib.ref_sync_decode.restype = C.c_double
        _lib.ref_mix_entries.restype = C.c_size_t
        _lib.ref_conv_encode.restype = C.c_size_t
        _lib.ref_conv_decode_soft.restype = C.c_size_t
        _lib.ref_search_approx.restype = C.c_size_t
        _lib.ref_set_params.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double]
        _lib.ref_set_quiet(1)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _key(key):
    return bytes(16) if key is None else bytes(key)


def set_params(water_delta=0.01, mix=True, frames_per_bit=2, test_no_limiter=False, sync_threshold2=0.35, n_best=8,
               chunk_size_min=30.0):
    lib().ref_set_params(water_delta, int(mix), frames_per_bit, int(test_no_limiter), sync_threshold2, n_best, chunk_size_min)


def random_u64(key, seed, stream, n):
    out = np.zeros(n, np.uint64)
    lib().ref_random_u64(_key(key), C.c_uint64(seed), stream, C.c_size_t(n), _p(out))
    return out


def random_double(key, seed, stream, n):
    out = np.zeros(n, np.float64)
    lib().ref_random_double(_key(key), C.c_uint64(seed), stream, C.c_size_t(n), _p(out))
    return out


def gen_noise(key, n_values):
    out = np.zeros(n_values, np.float32)
    lib().ref_gen_noise(_key(key), C.c_size_t(n_values), _p(out))
    return out


def up_down(key, stream, f):
    up = np.zeros(30, np.int32)
    down = np.zeros(30, np.int32)
    lib().ref_up_down(_key(key), stream, f, _p(up), _p(down))
    return up, down


def bit_pos(key):
    out = np.zeros(2226, np.int32)
    lib().ref_bit_pos(_key(key), _p(out))
    return out


def mix_entries(key):
    out = np.zeros((51480, 3), np.int32)
    n = lib().ref_mix_entries(_key(key), _p(out))
    return out[:n]


def window(n):
    out = np.zeros(n, np.float32)
    lib().ref_window(C.c_size_t(n), _p(out))
    return out


def bit_order(key, n):
    out = np.zeros(n, np.uint32)
    lib().ref_bit_order(_key(key), C.c_size_t(n), _p(out))
    return out


def conv_encode(block_type, bits):
    bits = np.ascontiguousarray(bits, np.int32)
    out = np.zeros((len(bits) + 15) * 12, np.int32)
    n = lib().ref_conv_encode(block_type, _p(bits), C.c_size_t(len(bits)), _p(out))
    return out[:n]


def conv_decode_soft(block_type, coded):
    coded = np.ascontiguousarray(coded, np.float32)
    out = np.zeros(len(coded), np.int32)
    err = C.c_float()
    n = lib().ref_conv_decode_soft(block_type, _p(coded), C.c_size_t(len(coded)), _p(out), C.byref(err))
    return out[:n], err.value


def frame_mod(key, payload_hex, ab):
    out = np.zeros((2226, 101), np.uint8)
    n = lib().ref_frame_mod(_key(key), payload_hex.encode(), ab, _p(out))
    assert n == 2226
    return out


def sync_bits(key, clip_mode=False):
    rows = 170 if clip_mode else 85
    out = np.zeros((6, rows, 61), np.int32)
    r = lib().ref_sync_bits(_key(key), int(clip_mode), _p(out))
    assert r == rows
    return out


def fft_range(samples, n_channels, start_index, frame_count):
    samples = np.ascontiguousarray(samples, np.float32).ravel()
    out = np.zeros((frame_count, n_channels, 513, 2), np.float32)
    r = lib().ref_fft_range(_p(samples), C.c_size_t(samples.size), n_channels, C.c_size_t(start_index),
                            C.c_size_t(frame_count), _p(out))
    return out if r else None


def ifft(spect):
    spect = np.ascontiguousarray(spect, np.float32)
    n = (spect.shape[0] - 1) * 2
    out = np.zeros(n, np.float32)
    lib().ref_ifft(C.c_size_t(n), _p(spect), _p(out))
    return out


def add(key, samples, n_channels, payload_hex, sample_rate=44100):
    samples = np.ascontiguousarray(samples, np.float32).ravel()
    n_frames = samples.size // n_channels
    out = np.zeros(samples.size + 4096 * n_channels, np.float32)
    of = C.c_size_t()
    rc = lib().ref_add(_key(key), _p(samples), C.c_size_t(n_frames), n_channels, sample_rate, payload_hex.encode(), _p(out),
                       C.byref(of), None)
    assert rc == 0
    return out[:of.value * n_channels]


def add_at(key, samples, n_channels, payload_hex, zero_frames, sample_rate=44100):
    """add_stream_watermark (..., zero_frames) of the reference (wmadd.cc:448): the stream starts zero_frames samples in"""
    samples = np.ascontiguousarray(samples, np.float32).ravel()
    n_frames = samples.size // n_channels
    out = np.zeros(samples.size + 4096 * n_channels, np.float32)
    of = C.c_size_t()
    rc = lib().ref_add_at(_key(key), _p(samples), C.c_size_t(n_frames), n_channels, sample_rate, payload_hex.encode(),
                          C.c_size_t(zero_frames), _p(out), C.byref(of))
    assert rc == 0
    return out[:of.value * n_channels]


def sync_fft(samples, n_channels, index, frame_count, want_frames=None, first=0, last=None):
    samples = np.ascontiguousarray(samples, np.float32).ravel()
    if last is None:
        last = samples.size
    db = np.zeros((frame_count, 81), np.float32)
    have = np.zeros(frame_count, np.int8)
    want = None if want_frames is None else np.ascontiguousarray(want_frames, np.int8)
    n = lib().ref_sync_fft(_p(samples), C.c_size_t(samples.size), n_channels, C.c_size_t(index), C.c_size_t(frame_count),
                           _p(want) if want is not None else None, C.c_size_t(first), C.c_size_t(last), _p(db), _p(have))
    return (db, have) if n else (None, None)


def sync_decode(key, clip_mode, start_frame, db, have):
    db = np.ascontiguousarray(db, np.float32).ravel()
    have = np.ascontiguousarray(have, np.int8)
    return lib().ref_sync_decode(_key(key), int(clip_mode), C.c_size_t(start_frame), _p(db), C.c_size_t(db.size), _p(have),
                                 C.c_size_t(have.size))


def sync_search(key, samples, n_channels, clip_mode=False, max_out=4096):
    samples = np.ascontiguousarray(samples, np.float32).ravel()
    idx = np.zeros(max_out, np.uint64)
    q = np.zeros(max_out, np.float64)
    bt = np.zeros(max_out, np.int32)
    n = lib().ref_sync_search(_key(key), _p(samples), C.c_size_t(samples.size), n_channels, int(clip_mode), C.c_size_t(max_out),
                              _p(idx), _p(q), _p(bt))
    return idx[:n], q[:n], bt[:n]


def search_approx(key, samples, n_channels, clip_mode=False):
    samples = np.ascontiguousarray(samples, np.float32).ravel()
    max_out = 4 * (samples.size // n_channels // 1024 + 1)
    idx = np.zeros(max_out, np.uint64)
    raw = np.zeros(max_out, np.float64)
    mean = np.zeros(max_out, np.float64)
    n = lib().ref_search_approx(_key(key), _p(samples), C.c_size_t(samples.size), n_channels, int(clip_mode),
                                C.c_size_t(max_out), _p(idx), _p(raw), _p(mean))
    return idx[:n], raw[:n], mean[:n]


def mix_decode(key, samples, n_channels, index):
    samples = np.ascontiguousarray(samples, np.float32).ravel()
    out = np.zeros(858, np.float32)
    n = lib().ref_mix_decode(_key(key), _p(samples), C.c_size_t(samples.size), n_channels, C.c_size_t(index), _p(out))
    return out if n else None


def _patterns(fn, *args, max_out=4096):
    buf = (Pattern * max_out)()
    n = fn(*args, C.c_size_t(max_out), C.cast(buf, C.c_void_p))
    assert n >= 0
    return [buf[i].as_dict() for i in range(min(n, max_out))]


def decode_chunk(key, samples, n_channels, first_chunk=True):
    samples = np.ascontiguousarray(samples, np.float32).ravel()
    return _patterns(lib().ref_decode_chunk, _key(key), _p(samples), C.c_size_t(samples.size), n_channels, int(first_chunk))


def get(key, samples, n_channels, sample_rate=44100):
    samples = np.ascontiguousarray(samples, np.float32).ravel()
    return _patterns(lib().ref_get_rate, _key(key), _p(samples), C.c_size_t(samples.size), n_channels, sample_rate)


# ---- speed detection (wmspeed.cc) / VResampler paths; zita-resampler is the oracle's restatement ---------------------
def set_speed_params(detect_speed=False, patient=False, try_speed=-1.0):
    lib().ref_set_speed_params.argtypes = [C.c_int, C.c_int, C.c_double]
    lib().ref_set_speed_params(int(detect_speed), int(patient), float(try_speed))


def resample_ratio(samples, n_channels, ratio, rate=44100, new_rate=44100, max_in_seconds=-1.0):
    s = np.ascontiguousarray(samples, dtype=np.float32)
    n = s.size // n_channels
    cap = int(n * ratio) + 16
    out = np.zeros(cap * n_channels, dtype=np.float32)
    f = lib().ref_resample_ratio
    f.restype = C.c_size_t
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_size_t, C.c_void_p]
    got = f(_p(s), n, n_channels, rate, ratio, new_rate, max_in_seconds, cap, _p(out))
    assert got <= cap
    return out[:got * n_channels]


def speed_clip_location(key, samples, n_channels, seconds, candidates=5, rate=44100):
    s = np.ascontiguousarray(samples, dtype=np.float32)
    f = lib().ref_speed_clip_location
    f.restype = C.c_double
    f.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_int]
    return f(_key(key), _p(s), s.size, n_channels, rate, seconds, candidates)


def speed_mags(key, samples, n_channels, clip_location, center, seconds, rate=44100):
    s = np.ascontiguousarray(samples, dtype=np.float32)
    max_rows = int(seconds * 22050 / 128) + 8
    out = np.zeros((max_rows, 510, 2), dtype=np.float32)
    f = lib().ref_speed_mags
    f.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_size_t, C.c_void_p]
    rows = f(_key(key), _p(s), s.size, n_channels, rate, clip_location, center, seconds, max_rows, _p(out))
    assert rows <= max_rows
    return out[:rows]


def speed_scan(key, samples, n_channels, clip_location, seconds, step, n_steps, n_center_steps, speeds, rate=44100):
    s = np.ascontiguousarray(samples, dtype=np.float32)
    sp = np.ascontiguousarray(speeds, dtype=np.float64)
    cap = len(sp) * (2 * n_center_steps + 1) * (2 * n_steps + 1)
    o_s = np.zeros(cap)
    o_q = np.zeros(cap)
    f = lib().ref_speed_scan
    f.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                  C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
    n = f(_key(key), _p(s), s.size, n_channels, rate, clip_location, seconds, step, n_steps, n_center_steps,
          _p(sp), len(sp), cap, _p(o_s), _p(o_q))
    assert n == cap
    return o_s, o_q


def speed_select_n_best(speed, quality, n):
    sp = np.array(speed, dtype=np.float64)
    q = np.array(quality, dtype=np.float64)
    f = lib().ref_speed_select_n_best
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    k = f(_p(sp), _p(q), len(sp), n)
    return sp[:k], q[:k]


def speed_smooth_best(speed, quality, step, distance):
    sp = np.ascontiguousarray(speed, dtype=np.float64)
    q = np.ascontiguousarray(quality, dtype=np.float64)
    f = lib().ref_speed_smooth_best
    f.restype = C.c_double
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double]
    return f(_p(sp), _p(q), len(sp), step, distance)


def detect_speed(key, samples, n_channels, patient=False, rate=44100):
    s = np.ascontiguousarray(samples, dtype=np.float32)
    out = C.c_double(0)
    f = lib().ref_detect_speed
    f.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
    n = f(_key(key), _p(s), s.size, n_channels, rate, int(patient), C.byref(out))
    return out.value if n else None


def raw_convert(values_or_bytes, bit_depth, encoding, big_endian, to_raw):
    """the reference's RawConverter (rawconverter.cc): floats -> bytes (to_raw) or bytes -> floats"""
    if to_raw:
        src = np.ascontiguousarray(values_or_bytes, np.float32)
        n = src.size
        out = np.zeros(n * (bit_depth // 8), np.uint8)
    else:
        src = np.ascontiguousarray(values_or_bytes, np.uint8)
        n = src.size // (bit_depth // 8)
        out = np.zeros(n, np.float32)
    rc = lib().ref_raw_convert(bit_depth, encoding, int(big_endian), int(to_raw), _p(src), _p(out), C.c_size_t(n))
    assert rc == 0
    return out
