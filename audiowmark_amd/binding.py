"""ctypes binding of include/awm_hip.h.  Names follow the reference's operators:
add_watermark / get_watermark (wmcommon.hh:226-228), SyncFinder.search, fft_range, ..."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
N_BANDS = 81
BLOCK_FRAMES = 2226
SOFT_BITS = 858


class AwmError(RuntimeError):
    pass


def library_path():
    return os.path.join(_HERE, "libawm_hip.so")


# (Runtime settings are the application's business, not this module's: a process that wants every lane of `get` on a hardware
# queue of its own exports GPU_MAX_HW_QUEUES=16 before its first HIP call -- bench.py and tests/conftest.py do.)


def _load_hip_runtime():
    """libawm_hip.so has no DT_NEEDED on a HIP runtime: bind it to the runtime of this process.  PyTorch
    ships its own libamdhip64.so (no SONAME) next to libtorch_hip.so; a second runtime in the same process
    cannot open the KFD device, so torch's copy is preferred and made global before our library is loaded."""
    candidates = []
    try:
        import torch
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    candidates += [os.path.join(rocm, "lib", "libamdhip64.so.7"), os.path.join(rocm, "lib", "libamdhip64.so")]
    for c in candidates:
        if os.path.exists(c):
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
    raise AwmError("no HIP runtime (libamdhip64) found for libawm_hip.so")


def _load():
    path = library_path()
    if not os.path.exists(path):
        raise AwmError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C audiowmark_amd/csrc` (there is no CPU fallback)")
    _load_hip_runtime()
    return C.CDLL(path)


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


_HEX = np.frombuffer(b"0123456789abcdef", np.uint8)


def patterns_to_dicts(buf, count, first=0):
    """The first `count` entries of a ctypes Pattern array as dicts (Pattern.as_dict for each, vectorised: the payload
    hex strings are built with numpy instead of 128 Python operations per pattern)."""
    if count <= 0:
        return []
    if isinstance(buf, np.ndarray):
        a = buf[first:first + count]
    else:
        a = np.frombuffer(buf, dtype=np.dtype(Pattern), count=count, offset=first * C.sizeof(Pattern))
    bits = a["bits"].astype(np.uint8)
    nib = (bits[:, 0::4] << 3) | (bits[:, 1::4] << 2) | (bits[:, 2::4] << 1) | bits[:, 3::4]
    text = _HEX[nib]                                   # [count, 32] ASCII
    n_hex = (a["n_bits"] // 4).tolist()
    rows = text.tobytes().decode()
    w = text.shape[1]
    cols = [a[f].tolist() for f in ("time", "sync_index", "sync_quality", "block_type", "type", "decode_error", "speed")]
    return [dict(time=t, sync_index=si, sync_quality=sq, block_type=bt, type=ty, decode_error=de, speed=sp,
                 bits=rows[i * w:i * w + n_hex[i]])
            for i, (t, si, sq, bt, ty, de, sp) in enumerate(zip(*cols))]


lib = _load()
_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p
lib.awm_last_error.restype = C.c_char_p
lib.awm_version.restype = C.c_char_p
lib.awm_ctx_stream.restype = C.c_void_p
lib.awm_search_approx_d.restype = C.c_long
lib.awm_set_params.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double]
lib.awm_set_params.restype = None
lib.awm_ctx_create.argtypes = [C.c_int, C.POINTER(_vp)]
lib.awm_ctx_destroy.argtypes = [_vp]
lib.awm_ctx_destroy.restype = None
lib.awm_ctx_synchronize.argtypes = [_vp]
lib.awm_ctx_stream.argtypes = [_vp]
lib.awm_ctx_set_stream.argtypes = [_vp, _vp]
lib.awm_stft_d.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, _vp]
lib.awm_add_init_block_max_d.argtypes = [_vp, _vp, C.c_size_t]
lib.awm_add_mix_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp, C.c_double, C.c_size_t, _vp, _vp, _vp,
                              C.c_size_t, C.c_size_t]
lib.awm_add_limit_d.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_size_t, _vp, C.c_size_t, C.c_size_t]
lib.awm_add_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp, C.c_double, C.c_int]
lib.awm_sync_fft_d.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t, _vp, _vp]
lib.awm_sync_search_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp]
lib.awm_search_approx_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp]
lib.awm_block_soft_bits_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, _vp, _vp]
lib.awm_viterbi_decode.argtypes = [_vp, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp, _vp]
lib.awm_add_watermark_d.argtypes = [_vp, _vp, C.c_char_p, _vp, _vp, C.c_size_t, C.c_int, C.c_int]
lib.awm_get_watermark_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_size_t, _vp]
lib.awm_add_get_watermark_d.argtypes = [_vp, _vp, C.c_char_p, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_size_t, _vp]
lib.awm_resample_frames.argtypes = [_vp, C.c_size_t, C.c_int, C.c_int]
lib.awm_resample_frames.restype = C.c_size_t
lib.awm_resample_d.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t]
lib.awm_add_watermark_batch_d.argtypes = [_vp, _vp, C.c_char_p, C.c_size_t, _vp, _vp, _vp, C.c_int]
lib.awm_get_watermark_batch_d.argtypes = [_vp, _vp, C.c_size_t, _vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp]
lib.awm_add_watermark_batch_keys_d.argtypes = [_vp, _vp, C.c_char_p, C.c_size_t, _vp, _vp, _vp, C.c_int]
lib.awm_get_watermark_batch_keys_d.argtypes = [_vp, _vp, C.c_size_t, _vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp]
lib.awm_decode_chunk_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_size_t, _vp]
lib.awm_tab_up_down.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp]
lib.awm_tab_bit_pos.argtypes = [_vp, _vp]
lib.awm_tab_mix_entries.argtypes = [_vp, _vp]
lib.awm_tab_bit_order.argtypes = [_vp, C.c_size_t, _vp]
lib.awm_tab_frame_mod.argtypes = [_vp, C.c_char_p, _vp]
lib.awm_tab_sync_bits.argtypes = [_vp, C.c_int, _vp]
lib.awm_tab_window.argtypes = [C.c_size_t, _vp]
lib.awm_tab_synth_window.argtypes = [_vp]
lib.awm_conv_encode.argtypes = [C.c_int, _vp, C.c_size_t, _vp]
lib.awm_test_gen_noise.argtypes = [_vp, C.c_size_t, _vp]


class RawFormat(C.Structure):
    """awm_raw_format: headerless PCM as with --format raw --raw-rate / --raw-channels / --raw-bits / --raw-encoding / --raw-endian"""
    _fields_ = [("n_channels", C.c_int), ("sample_rate", C.c_int), ("bit_depth", C.c_int), ("encoding", C.c_int), ("big_endian", C.c_int)]


lib.awm_add_watermark_file.argtypes = [_vp, _vp, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(RawFormat), C.POINTER(RawFormat)]
lib.awm_add_stream_watermark_file.argtypes = [_vp, _vp, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(RawFormat), C.POINTER(RawFormat), C.c_size_t]
lib.awm_add_stream_create_at.argtypes = [_vp, _vp, C.c_char_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(_vp)]
lib.awm_debug_sync_db_sliding_d.argtypes = [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, C.c_int, C.c_int, _vp]
lib.awm_get_watermark_file.argtypes = [_vp, _vp, C.c_char_p, C.POINTER(RawFormat), C.c_size_t, _vp]
lib.awm_add_get_watermark_file.argtypes = [_vp, _vp, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(RawFormat), C.POINTER(RawFormat), C.c_size_t, _vp]
lib.awm_add_stream_create.argtypes = [_vp, _vp, C.c_char_p, C.c_int, C.c_size_t, C.POINTER(_vp)]
lib.awm_add_stream_destroy.argtypes = [_vp]
lib.awm_add_stream_destroy.restype = None
lib.awm_add_stream_input.argtypes = [_vp]
lib.awm_add_stream_input.restype = C.c_void_p
lib.awm_add_stream_push.argtypes = [_vp, C.c_size_t, C.c_int, _vp, _vp]


lib.awm_decode_chunks_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_size_t, _vp, _vp]
lib.awm_pcm_decode_d.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, _vp]
lib.awm_pcm_encode_d.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, _vp]
lib.awm_plan_chunks.argtypes = [C.c_size_t, C.c_size_t, _vp, _vp, _vp]
lib.awm_merge_patterns.argtypes = [_vp, _vp, _vp, C.c_int, C.c_size_t, _vp]
lib.awm_prof_name.restype = C.c_char_p
lib.awm_set_speed_params.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
lib.awm_set_speed_params.restype = None
lib.awm_resample_ratio_frames.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double]
lib.awm_resample_ratio_frames.restype = C.c_size_t
lib.awm_resample_ratio_d.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, _vp, C.c_size_t]
lib.awm_detect_speed_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, _vp, _vp]
lib.awm_speed_clip_location_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_int, _vp]
lib.awm_speed_mags_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_size_t, _vp]
lib.awm_speed_scan_d.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                 _vp, C.c_int, C.c_size_t, _vp, _vp]


class Params(C.Structure):
    """awm_params (include/awm_hip.h): the reference's Params, process-wide or per context"""
    _fields_ = [("struct_size", C.c_size_t), ("water_delta", C.c_double), ("mix", C.c_int), ("hard", C.c_int), ("strict", C.c_int),
                ("snr", C.c_int), ("payload_size", C.c_int), ("frames_per_bit", C.c_int), ("sync_threshold2", C.c_double),
                ("get_n_best", C.c_int), ("get_chunk_size", C.c_double), ("detect_speed", C.c_int), ("detect_speed_patient", C.c_int),
                ("try_speed", C.c_double), ("test_speed", C.c_double), ("test_cut", C.c_int), ("test_no_sync", C.c_int),
                ("test_no_limiter", C.c_int), ("test_truncate", C.c_int)]

    def __init__(self, **kw):
        super().__init__()
        lib.awm_params_init(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError(f"awm_params has no field {k}")
            setattr(self, k, v)


lib.awm_params_init.argtypes = [C.POINTER(Params)]
lib.awm_params_init.restype = None
lib.awm_set_global_params.argtypes = [C.POINTER(Params)]
lib.awm_ctx_set_params.argtypes = [_vp, C.POINTER(Params)]
lib.awm_ctx_get_params.argtypes = [_vp, C.POINTER(Params)]
lib.awm_ctx_snr_begin.argtypes = [_vp]
lib.awm_ctx_snr_end.argtypes = [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
lib.awm_get_watermark_keys_d.argtypes = [_vp, _vp, C.c_int, _vp, C.c_size_t, C.c_int, C.c_size_t, _vp, _vp]
lib.awm_get_watermark_keys_file.argtypes = [_vp, _vp, C.c_int, C.c_char_p, C.POINTER(RawFormat), C.c_size_t, _vp, _vp]


def set_global_params(**kw):
    """the process-wide parameter set from the reference's defaults + the given fields (awm_set_global_params)"""
    p = Params(**kw)
    _check(lib.awm_set_global_params(C.byref(p)), "awm_set_global_params")


def _check(rc, what):
    if rc < 0:
        raise AwmError(f"{what} failed (rc={rc}): {lib.awm_last_error().decode(errors='replace')}")
    return rc


def key_bytes(key=None):
    """16-byte AES key: None -> the all-zero default key, bytes/hex -> as given."""
    if key is None:
        return bytes(16)
    if isinstance(key, str):
        key = bytes.fromhex(key)
    key = bytes(key)
    if len(key) != 16:
        raise ValueError("key must be 16 bytes")
    return key


def test_key(n):
    """Key::set_test_key (reference random.cc:204-209): big-endian u64 in the first 8 bytes."""
    return int(n).to_bytes(8, "big") + bytes(8)


def _np(a):
    return a.ctypes.data_as(C.c_void_p)


def set_params(water_delta=0.01, mix=True, frames_per_bit=2, test_no_limiter=False, sync_threshold2=0.35, n_best=8,
               chunk_size_min=30.0):
    lib.awm_set_params(water_delta, int(mix), frames_per_bit, int(test_no_limiter), sync_threshold2, n_best,
                       chunk_size_min)


def set_speed_params(detect_speed=False, patient=False, try_speed=-1.0, test_speed=-1.0):
    """--detect-speed / --detect-speed-patient / --try-speed / --test-speed of `get` (reference wmcommon.hh:49-52)."""
    lib.awm_set_speed_params(int(detect_speed), int(patient), float(try_speed), float(test_speed))


def plan_chunks(n_frames):
    """WavChunkLoader chunking (host only): list of (first_frame, n_frames, time_offset_seconds)."""
    mx = 4096
    a = np.zeros(mx, np.uint64)
    b = np.zeros(mx, np.uint64)
    t = np.zeros(mx, np.float64)
    n = lib.awm_plan_chunks(n_frames, mx, _np(a), _np(b), _np(t))
    return [(int(a[i]), int(b[i]), float(t[i])) for i in range(n)]


def _pattern_from_dict(d):
    p = Pattern()
    p.time = d["time"]
    p.sync_index = d["sync_index"]
    p.sync_quality = d["sync_quality"]
    p.block_type = d["block_type"]
    p.type = d["type"]
    p.decode_error = d["decode_error"]
    p.speed = d["speed"]
    bits = []
    for ch in d["bits"]:
        v = int(ch, 16)
        bits += [(v >> 3) & 1, (v >> 2) & 1, (v >> 1) & 1, v & 1]
    p.n_bits = len(bits)
    for i, b in enumerate(bits):
        p.bits[i] = b
    return p


PATTERN_DTYPE = np.dtype(Pattern)


def patterns_from_dicts(dicts):
    """pattern dicts -> structured array (PATTERN_DTYPE)"""
    out = np.zeros(len(dicts), PATTERN_DTYPE)
    for i, d in enumerate(dicts):
        p = _pattern_from_dict(d)
        out[i] = np.frombuffer(bytes(p), PATTERN_DTYPE)[0]
    return out


def merge_patterns_raw(key, per_chunk_arrays):
    """ResultSet.merge + sort over per-chunk structured arrays (dtype PATTERN_DTYPE, times already offset); returns dicts.
    No per-pattern Python work: this runs inside the timed region of the multi-GPU path."""
    counts = np.array([len(a) for a in per_chunk_arrays], np.int32)
    total = int(counts.sum())
    if total == 0:
        return []
    arr = np.ascontiguousarray(np.concatenate([a for a in per_chunk_arrays if len(a)]))
    out = np.zeros(total, PATTERN_DTYPE)
    n = lib.awm_merge_patterns(key_bytes(key), arr.ctypes.data_as(C.c_void_p), _np(counts), len(per_chunk_arrays), total,
                               out.ctypes.data_as(C.c_void_p))
    return patterns_to_dicts(out, min(n, total))


def merge_patterns(key, per_chunk):
    """ResultSet.merge + sort over per-chunk pattern lists (times already offset), host only."""
    flat = [_pattern_from_dict(d) for chunk in per_chunk for d in chunk]
    counts = np.array([len(c) for c in per_chunk], np.int32)
    arr = (Pattern * max(1, len(flat)))(*flat)
    mx = max(1, len(flat))
    out = (Pattern * mx)()
    n = lib.awm_merge_patterns(key_bytes(key), C.cast(arr, C.c_void_p), _np(counts), len(per_chunk), mx, C.cast(out, C.c_void_p))
    return [out[i].as_dict() for i in range(min(n, mx))]


# ---- key-derived tables (host only) ---------------------------------------------------------
def tab_up_down(key, stream, frame):
    up = np.zeros(30, np.int32)
    down = np.zeros(30, np.int32)
    lib.awm_tab_up_down(key_bytes(key), stream, frame, _np(up), _np(down))
    return up, down


def tab_bit_pos(key):
    pos = np.zeros(BLOCK_FRAMES, np.int32)
    lib.awm_tab_bit_pos(key_bytes(key), _np(pos))
    return pos


def tab_mix_entries(key):
    out = np.zeros((51480, 3), np.int32)
    n = lib.awm_tab_mix_entries(key_bytes(key), _np(out))
    return out[:n]


def tab_bit_order(key, n):
    out = np.zeros(n, np.uint32)
    lib.awm_tab_bit_order(key_bytes(key), n, _np(out))
    return out


def tab_frame_mod(key, payload_hex):
    out = np.zeros((2, BLOCK_FRAMES, N_BANDS), np.int8)
    _check(lib.awm_tab_frame_mod(key_bytes(key), payload_hex.encode(), _np(out)), "awm_tab_frame_mod")
    return out


def tab_sync_bits(key, clip_mode=False):
    rows = 170 if clip_mode else 85
    out = np.zeros((6, rows, 61), np.int32)
    r = lib.awm_tab_sync_bits(key_bytes(key), int(clip_mode), _np(out))
    assert r == rows
    return out


def tab_window(n=1024):
    out = np.zeros(n, np.float32)
    lib.awm_tab_window(n, _np(out))
    return out


def tab_synth_window():
    out = np.zeros(3072, np.float32)
    lib.awm_tab_synth_window(_np(out))
    return out


def gen_noise(key, n_values):
    """`audiowmark test-gen-noise` samples (reference audiowmark.cc:399-417): interleaved float32 in [-1, 1)."""
    out = np.zeros(n_values, np.float32)
    lib.awm_test_gen_noise(key_bytes(key), C.c_size_t(n_values), _np(out))
    return out


def conv_encode(block_type, bits):
    bits = np.ascontiguousarray(bits, np.int32)
    out = np.zeros((len(bits) + 15) * 12, np.int32)
    n = lib.awm_conv_encode(block_type, _np(bits), len(bits), _np(out))
    return out[:n]


def _hip_memcpy_dtod(ctx, dst, src, nbytes):
    """device-to-device copy on the context's stream (through torch: the library exports no raw copy)"""
    if nbytes <= 0:
        return
    _as_tensor(dst, nbytes).copy_(_as_tensor(src, nbytes))       # views over raw device pointers; torch's current stream == ctx stream


def _as_tensor(ptr, nbytes, device=None):
    """uint8 VIEW over raw device memory (no copy).  `device`: the torch device the memory lives on (default: the current one) -- a view
    created under another current device would silently become a copy, and bytes received into it would never reach the buffer."""
    import torch

    class _Iface:
        pass
    o = _Iface()
    o.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    t = torch.as_tensor(o, device=device if device is not None else "cuda")
    if nbytes and t.data_ptr() != int(ptr):
        raise AwmError("_as_tensor: torch copied the buffer instead of viewing it (wrong device?)")
    return t


# ---- device side ---------------------------------------------------------------------------
def _dev_ptr(t):
    import torch
    assert isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _pcm_shape(pcm):
    """(n_frames, n_channels) of an interleaved float32 CUDA tensor shaped [frames, channels] or [frames]."""
    import torch
    assert pcm.dtype == torch.float32 and pcm.is_cuda and pcm.is_contiguous()
    if pcm.dim() == 1:
        return pcm.shape[0], 1
    return pcm.shape[0], pcm.shape[1]


class Context:
    """One awm_ctx: a GPU, its stream and workspaces.  Work is enqueued on torch's current stream
    of that device so that torch.cuda events/synchronisation bracket it."""

    def __init__(self, device=0, use_torch_stream=True):
        import torch
        self.device = device
        h = C.c_void_p()
        _check(lib.awm_ctx_create(device, C.byref(h)), "awm_ctx_create")
        self._h = h
        if use_torch_stream:
            s = torch.cuda.current_stream(device)
            _check(lib.awm_ctx_set_stream(self._h, C.c_void_p(s.cuda_stream)), "awm_ctx_set_stream")

    def close(self):
        if self._h:
            lib.awm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib.awm_ctx_synchronize(self._h), "awm_ctx_synchronize")

    # RawConverter::from_raw / to_raw on the device
    def pcm_decode(self, raw_bytes, bit_depth, encoding=0, big_endian=False):
        import torch
        assert raw_bytes.dtype == torch.uint8 and raw_bytes.is_cuda and raw_bytes.is_contiguous()
        n = raw_bytes.numel() // (bit_depth // 8)
        out = torch.empty(n, dtype=torch.float32, device=raw_bytes.device)
        _check(lib.awm_pcm_decode_d(self._h, _dev_ptr(raw_bytes), n, bit_depth, encoding, int(big_endian), _dev_ptr(out)), "awm_pcm_decode_d")
        return out

    def pcm_encode(self, samples, bit_depth, encoding=0, big_endian=False, direct16=True):
        import torch
        assert samples.dtype == torch.float32 and samples.is_cuda and samples.is_contiguous()
        n = samples.numel()
        out = torch.empty(n * (bit_depth // 8), dtype=torch.uint8, device=samples.device)
        _check(lib.awm_pcm_encode_d(self._h, _dev_ptr(samples), n, bit_depth, encoding, int(big_endian), int(direct16), _dev_ptr(out)),
               "awm_pcm_encode_d")
        return out

    # FFTAnalyzer::fft_range
    def fft_range(self, pcm, start_index, frame_count, hop=1024):
        import torch
        n, ch = _pcm_shape(pcm)
        out = torch.empty((frame_count, ch, 513, 2), dtype=torch.float32, device=pcm.device)
        _check(lib.awm_stft_d(self._h, _dev_ptr(pcm), n, ch, start_index, hop, frame_count, _dev_ptr(out)), "awm_stft_d")
        return out

    # add_watermark on resident PCM
    def add_watermark(self, key, payload_hex, pcm, out=None, sample_rate=44100):
        import torch
        n, ch = _pcm_shape(pcm)
        if out is None:
            out = torch.empty_like(pcm)
        _check(lib.awm_add_watermark_d(self._h, key_bytes(key), payload_hex.encode(), _dev_ptr(pcm), _dev_ptr(out), n, ch,
                                       sample_rate), "awm_add_watermark_d")
        return out

    def add_watermark_batch(self, key, payload_hex, clips, outs=None):
        """add_watermark of many independent resident clips (44.1 kHz, same channel count) with one key and payload, dealt to
        the context's work lanes; returns the list of outputs."""
        import torch
        if not clips:
            return []
        shapes = [_pcm_shape(c) for c in clips]
        ch = shapes[0][1]
        assert all(s[1] == ch for s in shapes)
        if outs is None:
            outs = [torch.empty_like(c) for c in clips]
        src = (C.c_void_p * len(clips))(*[_dev_ptr(c) for c in clips])
        dst = (C.c_void_p * len(clips))(*[_dev_ptr(o) for o in outs])
        frames = (C.c_size_t * len(clips))(*[s[0] for s in shapes])
        _check(lib.awm_add_watermark_batch_d(self._h, key_bytes(key), payload_hex.encode(), len(clips), src, dst, frames, ch),
               "awm_add_watermark_batch_d")
        return outs

    def add_watermark_batch_keys(self, keys, payload_hex, clips, outs=None):
        """the same with ONE KEY PER CLIP (awm_add_watermark_batch_keys_d; BASELINE configs[4]: `--test-key k` for clip k)"""
        import torch
        if not clips:
            return []
        assert len(keys) == len(clips)
        shapes = [_pcm_shape(c) for c in clips]
        ch = shapes[0][1]
        assert all(s[1] == ch for s in shapes)
        if outs is None:
            outs = [torch.empty_like(c) for c in clips]
        src = (C.c_void_p * len(clips))(*[_dev_ptr(c) for c in clips])
        dst = (C.c_void_p * len(clips))(*[_dev_ptr(o) for o in outs])
        frames = (C.c_size_t * len(clips))(*[s[0] for s in shapes])
        flat = b"".join(key_bytes(k) for k in keys)
        _check(lib.awm_add_watermark_batch_keys_d(self._h, flat, payload_hex.encode(), len(clips), src, dst, frames, ch),
               "awm_add_watermark_batch_keys_d")
        return outs

    def get_watermark_batch_keys(self, keys, clips, n_threads=0, max_out_per_clip=64):
        """get_watermark of many independent resident clips, clip i with keys[i] alone (awm_get_watermark_batch_keys_d)"""
        if not clips:
            return []
        assert len(keys) == len(clips)
        shapes = [_pcm_shape(c) for c in clips]
        ch = shapes[0][1]
        assert all(s[1] == ch for s in shapes)
        ptrs = (C.c_void_p * len(clips))(*[_dev_ptr(c) for c in clips])
        frames = (C.c_size_t * len(clips))(*[s[0] for s in shapes])
        n_out = (C.c_int * len(clips))()
        flat = b"".join(key_bytes(k) for k in keys)
        while True:
            buf = self._pattern_buffer(len(clips) * max_out_per_clip)
            _check(lib.awm_get_watermark_batch_keys_d(self._h, flat, len(clips), ptrs, frames, ch, n_threads, max_out_per_clip,
                                                      C.cast(buf, C.c_void_p), n_out), "awm_get_watermark_batch_keys_d")
            most = max(n_out)
            if most <= max_out_per_clip:
                return [patterns_to_dicts(buf, n_out[i], i * max_out_per_clip) for i in range(len(clips))]
            max_out_per_clip = most

    # ---- file level: the reference's add_watermark / get_watermark (wmcommon.hh:226-228) ----
    def add_watermark_file(self, key, payload_hex, in_path, out_path, raw_in=None, raw_out=None, zero_frames=None):
        """infile -> outfile; raw_* = RawFormat for headerless PCM, None for WAV; zero_frames: add_stream_watermark's start offset"""
        if zero_frames is not None:
            _check(lib.awm_add_stream_watermark_file(self._h, key_bytes(key), payload_hex.encode(), os.fsencode(in_path), os.fsencode(out_path),
                                                     C.byref(raw_in) if raw_in is not None else None,
                                                     C.byref(raw_out) if raw_out is not None else None, zero_frames), "awm_add_stream_watermark_file")
            return
        _check(lib.awm_add_watermark_file(self._h, key_bytes(key), payload_hex.encode(), os.fsencode(in_path), os.fsencode(out_path),
                                          C.byref(raw_in) if raw_in is not None else None,
                                          C.byref(raw_out) if raw_out is not None else None), "awm_add_watermark_file")

    def add_get_watermark_file(self, key, payload_hex, in_path, out_path, raw_in=None, raw_out=None):
        """awm_add_get_watermark_file: infile -> outfile, and the pattern list of `get` on what was written (never read back)"""
        return self._patterns(lib.awm_add_get_watermark_file, "awm_add_get_watermark_file", self._h, key_bytes(key), payload_hex.encode(),
                              os.fsencode(in_path), os.fsencode(out_path), C.byref(raw_in) if raw_in is not None else None,
                              C.byref(raw_out) if raw_out is not None else None)

    def get_watermark_file(self, key, in_path, raw_in=None):
        return self._patterns(lib.awm_get_watermark_file, "awm_get_watermark_file", self._h, key_bytes(key), os.fsencode(in_path),
                              C.byref(raw_in) if raw_in is not None else None)

    def sync_db_sliding(self, pcm, bases, count, ld=72):
        """K4s alone (awm_debug_sync_db_sliding_d): dB [stream][81][ld] of `count` windows advancing by 8 samples from every base"""
        import torch
        n, ch = _pcm_shape(pcm)
        b = torch.as_tensor(bases, dtype=torch.int64, device=pcm.device).contiguous()
        out = torch.zeros((b.numel(), 81, ld), dtype=torch.float32, device=pcm.device)
        _check(lib.awm_debug_sync_db_sliding_d(self._h, _dev_ptr(pcm), n, ch, C.c_void_p(b.data_ptr()), b.numel(), count, ld, _dev_ptr(out)),
               "awm_debug_sync_db_sliding_d")
        self.synchronize()
        return out

    def add_watermark_tiles(self, key, payload_hex, pcm, tile_frames1024=128, zero_frames=0):
        """awm_add_stream: `add` as a tile loop over resident PCM (the bounded-memory form the file path uses); returns the
        concatenated output -- bit-identical to add_watermark on the whole stream.  zero_frames: the stream starts that many
        samples into the frame / block grid (add_stream_watermark's zero_frames, reference wmadd.cc:501-526)."""
        import torch
        n, ch = _pcm_shape(pcm)
        h = C.c_void_p()
        _check(lib.awm_add_stream_create_at(self._h, key_bytes(key), payload_hex.encode(), ch, tile_frames1024, zero_frames, C.byref(h)),
               "awm_add_stream_create_at")
        out = torch.empty_like(pcm)
        tile = tile_frames1024 * 1024
        done_p = (C.c_void_p * 3)()
        done_n = (C.c_size_t * 3)()
        pos = written = 0
        esz = pcm.element_size() * ch
        try:
            while True:
                got = min(tile, n - pos)
                last = pos + got >= n
                slot = lib.awm_add_stream_input(h)
                if got:
                    _hip_memcpy_dtod(self, slot, pcm.data_ptr() + pos * esz, got * esz)
                k = _check(lib.awm_add_stream_push(h, got, int(last), done_p, done_n), "awm_add_stream_push")
                for i in range(k):
                    _hip_memcpy_dtod(self, out.data_ptr() + written * esz, done_p[i], done_n[i] * esz)
                    written += done_n[i]
                pos += got
                if last:
                    break
            self.synchronize()
        finally:
            lib.awm_add_stream_destroy(h)
        assert written == n
        return out

    def resample(self, pcm, rate_in, rate_out):
        """The stream the reference's loader hands to the decoder for a file at rate_in (zita-resampler restated)."""
        import torch
        n, ch = _pcm_shape(pcm)
        m = lib.awm_resample_frames(self._h, n, rate_in, rate_out)
        if n and not m:
            raise AwmError("resampling %d -> %d Hz is not supported: %s" % (rate_in, rate_out, lib.awm_last_error().decode()))
        out = torch.empty((m, ch), dtype=torch.float32, device=pcm.device)
        _check(lib.awm_resample_d(self._h, _dev_ptr(pcm), n, ch, rate_in, rate_out, _dev_ptr(out), m), "awm_resample_d")
        return out

    def resample_ratio(self, pcm, ratio, rate=44100, max_in_seconds=-1.0):
        """resample_ratio_truncate (reference resample.cc:96-119): zita's VResampler, restated."""
        import torch
        n, ch = _pcm_shape(pcm)
        m = lib.awm_resample_ratio_frames(n, ch, rate, ratio, max_in_seconds)
        out = torch.empty((m, ch), dtype=torch.float32, device=pcm.device)
        _check(lib.awm_resample_ratio_d(self._h, _dev_ptr(pcm), n, ch, rate, ratio, max_in_seconds, _dev_ptr(out), m),
               "awm_resample_ratio_d")
        return out

    def detect_speed(self, key, pcm, patient=False, rate=44100):
        """detect_speed for one key (reference wmspeed.cc:622-781): (speed to try or None, best speed, best quality)."""
        n, ch = _pcm_shape(pcm)
        speed, quality = C.c_double(0), C.c_double(0)
        rc = lib.awm_detect_speed_d(self._h, key_bytes(key), _dev_ptr(pcm), n, ch, rate, int(patient), C.byref(speed), C.byref(quality))
        if rc < 0:
            _check(rc, "awm_detect_speed_d")
        return (speed.value if rc else None), speed.value, quality.value

    def speed_clip_location(self, key, pcm, seconds, candidates=5, rate=44100):
        n, ch = _pcm_shape(pcm)
        loc = C.c_double(0)
        _check(lib.awm_speed_clip_location_d(self._h, key_bytes(key), _dev_ptr(pcm), n, ch, rate, seconds, candidates, C.byref(loc)),
               "awm_speed_clip_location_d")
        return loc.value

    def speed_mags(self, key, pcm, clip_location, center, seconds, rate=44100):
        n, ch = _pcm_shape(pcm)
        max_rows = int(seconds * 22050 / 128) + 8
        out = np.zeros((max_rows, 510, 2), np.float32)
        rows = lib.awm_speed_mags_d(self._h, key_bytes(key), _dev_ptr(pcm), n, ch, rate, clip_location, center, seconds, max_rows, _np(out))
        if rows < 0:
            _check(rows, "awm_speed_mags_d")
        return out[:rows]

    def speed_scan(self, key, pcm, clip_location, seconds, step, n_steps, n_center_steps, speeds, rate=44100):
        n, ch = _pcm_shape(pcm)
        sp = np.ascontiguousarray(speeds, np.float64)
        cap = len(sp) * (2 * n_center_steps + 1) * (2 * n_steps + 1)
        o_s, o_q = np.zeros(cap), np.zeros(cap)
        cnt = lib.awm_speed_scan_d(self._h, key_bytes(key), _dev_ptr(pcm), n, ch, rate, clip_location, seconds, step, n_steps,
                                   n_center_steps, _np(sp), len(sp), cap, _np(o_s), _np(o_q))
        if cnt < 0:
            _check(cnt, "awm_speed_scan_d")
        return o_s[:cnt], o_q[:cnt]

    def add_d(self, pcm, frame_mod, water_delta=0.01, use_limiter=True, out=None):
        import torch
        n, ch = _pcm_shape(pcm)
        if out is None:
            out = torch.empty_like(pcm)
        fm = np.ascontiguousarray(frame_mod, np.int8)
        _check(lib.awm_add_d(self._h, _dev_ptr(pcm), _dev_ptr(out), n, ch, _np(fm), water_delta, int(use_limiter)), "awm_add_d")
        return out

    def add_mix(self, pcm, out, frame_mod, water_delta, first_frame, halo_before, halo_after, block_max, first_block=0):
        n, ch = _pcm_shape(pcm)
        fm = np.ascontiguousarray(frame_mod, np.int8)
        _check(lib.awm_add_mix_d(self._h, _dev_ptr(pcm), _dev_ptr(out), n, ch, _np(fm), water_delta, first_frame,
                                 _dev_ptr(halo_before) if halo_before is not None else None,
                                 _dev_ptr(halo_after) if halo_after is not None else None,
                                 _dev_ptr(block_max) if block_max is not None else None, first_block,
                                 block_max.numel() if block_max is not None else 0), "awm_add_mix_d")

    def add_init_block_max(self, block_max):
        _check(lib.awm_add_init_block_max_d(self._h, _dev_ptr(block_max), block_max.numel()), "awm_add_init_block_max_d")

    def add_limit(self, out, first_sample, block_max, first_block=0):
        n, ch = _pcm_shape(out)
        _check(lib.awm_add_limit_d(self._h, _dev_ptr(out), n, ch, first_sample, _dev_ptr(block_max), first_block,
                                   block_max.numel()), "awm_add_limit_d")

    # SyncFinder::sync_fft
    def sync_fft(self, pcm, index, frame_count, want_frames=None, first=0, last=None):
        import torch
        n, ch = _pcm_shape(pcm)
        if last is None:
            last = n * ch
        db = torch.zeros((frame_count, N_BANDS), dtype=torch.float32, device=pcm.device)
        have = torch.zeros(frame_count, dtype=torch.int8, device=pcm.device)
        want = None
        if want_frames is not None:
            want = np.ascontiguousarray(want_frames, np.int8)
        _check(lib.awm_sync_fft_d(self._h, _dev_ptr(pcm), n, ch, index, frame_count, _np(want) if want is not None else None,
                                  first, last, _dev_ptr(db), _dev_ptr(have)), "awm_sync_fft_d")
        return db, have

    # SyncFinder::search
    def sync_search(self, key, pcm, clip_mode=False, max_out=4096):
        n, ch = _pcm_shape(pcm)
        while True:
            idx = np.zeros(max_out, np.uint64)
            q = np.zeros(max_out, np.float64)
            bt = np.zeros(max_out, np.int32)
            cnt = _check(lib.awm_sync_search_d(self._h, key_bytes(key), _dev_ptr(pcm), n, ch, int(clip_mode), max_out, _np(idx),
                                               _np(q), _np(bt)), "awm_sync_search_d")
            if cnt <= max_out:
                return idx[:cnt], q[:cnt], bt[:cnt]
            max_out = cnt                                    # more scores than the buffer holds: repeat, never truncate

    def search_approx(self, key, pcm, clip_mode=False):
        n, ch = _pcm_shape(pcm)
        max_out = 4 * (n // 1024 + 1)
        idx = np.zeros(max_out, np.uint64)
        raw = np.zeros(max_out, np.float64)
        mean = np.zeros(max_out, np.float64)
        cnt = _check(lib.awm_search_approx_d(self._h, key_bytes(key), _dev_ptr(pcm), n, ch, int(clip_mode), max_out, _np(idx),
                                             _np(raw), _np(mean)), "awm_search_approx_d")
        return idx[:cnt], raw[:cnt], mean[:cnt]

    # fft_range + mix_decode
    def block_soft_bits(self, key, pcm, indices):
        n, ch = _pcm_shape(pcm)
        idx = np.ascontiguousarray(indices, np.uint64)
        out = np.zeros((len(idx), SOFT_BITS), np.float32)
        ok = np.zeros(len(idx), np.int32)
        _check(lib.awm_block_soft_bits_d(self._h, key_bytes(key), _dev_ptr(pcm), n, ch, _np(idx), len(idx), _np(out), _np(ok)),
               "awm_block_soft_bits_d")
        return out, ok

    # conv_decode_soft
    def viterbi_decode(self, block_type, soft):
        soft = np.ascontiguousarray(soft, np.float32)
        if soft.ndim == 1:
            soft = soft[None, :]
        n, coded_len = soft.shape
        rate = 12 if block_type == 2 else 6
        bits = np.zeros((n, coded_len // rate - 15), np.int32)
        err = np.zeros(n, np.float32)
        _check(lib.awm_viterbi_decode(self._h, block_type, _np(soft), coded_len, n, _np(bits), _np(err)), "awm_viterbi_decode")
        return bits, err

    def _pattern_buffer(self, max_out):
        # one page-sized ctypes array per context, reused: allocating and zeroing 4096 patterns costs ~0.4 ms per call
        buf = getattr(self, "_pat_buf", None)
        if buf is None or len(buf) < max_out:
            buf = self._pat_buf = (Pattern * max_out)()
        return buf

    def _patterns(self, fn, what, *args, max_out=4096):
        # the C ABI returns the number of patterns FOUND; when that exceeds the buffer the call is repeated with a buffer
        # that holds them all (never a silently truncated list)
        while True:
            buf = self._pattern_buffer(max_out)
            cnt = _check(fn(*args, max_out, C.cast(buf, C.c_void_p)), what)
            if cnt <= max_out:
                return patterns_to_dicts(buf, cnt)
            max_out = cnt

    # get_watermark on resident PCM (chunk loop, BlockDecoder, ClipDecoder, merge, sort)
    def get_watermark(self, key, pcm):
        n, ch = _pcm_shape(pcm)
        return self._patterns(lib.awm_get_watermark_d, "awm_get_watermark_d", self._h, key_bytes(key), _dev_ptr(pcm), n, ch)

    def add_get_watermark(self, key, payload_hex, pcm, out, sample_rate=44100):
        """awm_add_get_watermark_d: add_watermark into `out`, then get_watermark of `out`, as one call (`get` starts a chunk as soon as
        the limiter has passed it); returns the pattern list -- the results of the two separate calls."""
        n, ch = _pcm_shape(pcm)
        assert _pcm_shape(out) == (n, ch) and out.data_ptr() != pcm.data_ptr()
        return self._patterns(lib.awm_add_get_watermark_d, "awm_add_get_watermark_d", self._h, key_bytes(key), payload_hex.encode(),
                              _dev_ptr(pcm), _dev_ptr(out), n, ch, sample_rate)

    def set_params(self, params=None, **kw):
        """Give this context its own parameter set (awm_ctx_set_params): a Params object, or the fields that differ from what is
        in force now as keywords; set_params() without anything returns the context to the process-wide set."""
        if params is None and kw:
            params = self.get_params()
            for k, v in kw.items():
                if k not in dict(Params._fields_):
                    raise TypeError(f"awm_params has no field {k}")
                setattr(params, k, v)
        _check(lib.awm_ctx_set_params(self._h, C.byref(params) if params is not None else None), "awm_ctx_set_params")

    def get_params(self):
        p = Params()
        _check(lib.awm_ctx_get_params(self._h, C.byref(p)), "awm_ctx_get_params")
        return p

    def snr_begin(self):
        """`add --snr`: every add of this context accumulates input and watermark power (before the limiter) until snr_end"""
        _check(lib.awm_ctx_snr_begin(self._h), "awm_ctx_snr_begin")

    def snr_end(self):
        """-> SNR in dB = 10 log10 (power of the input / power of the watermark signal), reference wmadd.cc:553-563, 591-592"""
        import math
        sig, delta = C.c_double(), C.c_double()
        _check(lib.awm_ctx_snr_end(self._h, C.byref(sig), C.byref(delta)), "awm_ctx_snr_end")
        return 10 * math.log10(sig.value / delta.value) if delta.value > 0 else float("inf")

    def _patterns_keys(self, fn, what, keys, *args, max_out=4096):
        flat = b"".join(key_bytes(k) for k in keys)
        while True:
            buf = self._pattern_buffer(max_out)
            which = (C.c_int * max_out)()
            cnt = _check(fn(self._h, flat, len(keys), *args, max_out, C.cast(buf, C.c_void_p), which), what)
            if cnt <= max_out:
                pats = patterns_to_dicts(buf, cnt)
                for p, k in zip(pats, which):
                    p["key_index"] = int(k)
                return pats
            max_out = cnt

    def get_watermark_keys(self, keys, pcm):
        """get_watermark with the reference's key LIST (`--key a --key b`): one pass over the material, every pattern tells the
        position of its key in the list ("key_index")."""
        n, ch = _pcm_shape(pcm)
        return self._patterns_keys(lib.awm_get_watermark_keys_d, "awm_get_watermark_keys_d", keys, _dev_ptr(pcm), n, ch)

    def get_watermark_keys_file(self, keys, in_path, raw_in=None):
        return self._patterns_keys(lib.awm_get_watermark_keys_file, "awm_get_watermark_keys_file", keys, os.fsencode(in_path),
                                   C.byref(raw_in) if raw_in is not None else None)

    def get_watermark_batch(self, key, clips, n_threads=0, max_out_per_clip=64):
        """get_watermark of many independent resident clips (same channel count), spread over the context's work lanes;
        returns one pattern list per clip."""
        if not clips:
            return []
        shapes = [_pcm_shape(c) for c in clips]
        ch = shapes[0][1]
        assert all(s[1] == ch for s in shapes)
        ptrs = (C.c_void_p * len(clips))(*[_dev_ptr(c) for c in clips])
        frames = (C.c_size_t * len(clips))(*[s[0] for s in shapes])
        n_out = (C.c_int * len(clips))()
        while True:
            buf = self._pattern_buffer(len(clips) * max_out_per_clip)
            _check(lib.awm_get_watermark_batch_d(self._h, key_bytes(key), len(clips), ptrs, frames, ch, n_threads, max_out_per_clip,
                                                 C.cast(buf, C.c_void_p), n_out), "awm_get_watermark_batch_d")
            most = max(n_out)
            if most <= max_out_per_clip:
                return [patterns_to_dicts(buf, n_out[i], i * max_out_per_clip) for i in range(len(clips))]
            max_out_per_clip = most           # a clip found more patterns than a slot holds: repeat with slots that fit

    def decode_chunks(self, key, pcm, chunks, first_is_stream_start, max_out=8192):
        """decode() of several chunks [(first_frame, n_frames), ...] of one resident buffer; returns one pattern list per
        chunk (times relative to the chunk)."""
        n, ch = _pcm_shape(pcm)
        first = np.array([c[0] for c in chunks], np.uint64)
        count = np.array([c[1] for c in chunks], np.uint64)
        while True:
            buf = self._pattern_buffer(max_out)
            which = np.zeros(max_out, np.int32)
            cnt = _check(lib.awm_decode_chunks_d(self._h, key_bytes(key), _dev_ptr(pcm), n, ch, len(chunks), _np(first), _np(count),
                                                 int(first_is_stream_start), max_out, C.cast(buf, C.c_void_p), _np(which)),
                         "awm_decode_chunks_d")
            if cnt <= max_out:
                break
            max_out = cnt
        out = [[] for _ in chunks]
        for i, d in enumerate(patterns_to_dicts(buf, cnt)):
            out[which[i]].append(d)
        return out

    def decode_chunks_raw(self, key, pcm, chunks, first_is_stream_start, max_out=8192):
        """decode_chunks without building Python objects: (patterns, which) -- a structured array (PATTERN_DTYPE, times relative
        to the chunk) and the index into `chunks` each pattern belongs to."""
        n, ch = _pcm_shape(pcm)
        first = np.array([c[0] for c in chunks], np.uint64)
        count = np.array([c[1] for c in chunks], np.uint64)
        while True:
            out = np.zeros(max_out, PATTERN_DTYPE)
            which = np.zeros(max_out, np.int32)
            cnt = _check(lib.awm_decode_chunks_d(self._h, key_bytes(key), _dev_ptr(pcm), n, ch, len(chunks), _np(first), _np(count),
                                                 int(first_is_stream_start), max_out, out.ctypes.data_as(C.c_void_p), _np(which)),
                         "awm_decode_chunks_d")
            if cnt <= max_out:
                return out[:cnt], which[:cnt]
            max_out = cnt

    def decode_chunk(self, key, pcm, first_chunk=True):
        n, ch = _pcm_shape(pcm)
        return self._patterns(lib.awm_decode_chunk_d, "awm_decode_chunk_d", self._h, key_bytes(key), _dev_ptr(pcm), n, ch,
                              int(first_chunk))
