"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/awm_hip.h
declares, the pure-host table builders agree bit for bit with the oracle, compute entry points fail
loudly without a GPU (no CPU fallback), chunk planning / pattern merging match the reference semantics."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _oracle as orc
import audiowmark_amd as awm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAY = "0123456789abcdef0011223344556677"
KEYS = [None, awm.test_key(42), bytes(range(16))]


def test_every_declared_symbol_is_exported():
    with open(os.path.join(ROOT, "include", "awm_hip.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = set(re.findall(r"\b(awm_[a-z0-9_]+)\s*\(", text))
    assert len(names) >= 30
    missing = [n for n in sorted(names) if not hasattr(awm.lib, n)]
    assert not missing, missing


@pytest.mark.parametrize("key", KEYS)
def test_host_tables_match_oracle(key):
    assert np.array_equal(awm.tab_bit_pos(key), orc.bit_pos(key))
    assert np.array_equal(awm.tab_mix_entries(key), orc.mix_entries(key))
    assert np.array_equal(awm.tab_bit_order(key, 858), orc.bit_order(key, 858))
    for stream in (1, 2):
        for f in (0, 7, 509, 1715):
            for a, b in zip(awm.tab_up_down(key, stream, f), orc.up_down(key, stream, f)):
                assert np.array_equal(a, b)
    for clip in (False, True):
        assert np.array_equal(awm.tab_sync_bits(key, clip), orc.sync_bits(key, clip))
    for pay in (PAY, "f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0", "a"):
        fm = awm.tab_frame_mod(key, pay)
        for ab in (0, 1):
            want = orc.frame_mod(key, pay, ab)
            assert not want[:, :20].any()
            assert np.array_equal(fm[ab], want[:, 20:101])
            assert ((fm[ab] != 0).sum(axis=1) == 60).all()            # 30 up + 30 down bands in every frame


def test_linear_mode_tables_match_oracle():
    """--linear (Params::mix = false): per-frame band lists instead of the shuffled mix entries (reference wmadd.cc:115-126)"""
    key = KEYS[-1]
    awm.set_params(mix=False)
    orc.set_params(mix=False)
    try:
        fm = awm.tab_frame_mod(key, PAY)
        for ab in (0, 1):
            want = orc.frame_mod(key, PAY, ab)
            assert np.array_equal(fm[ab], want[:, 20:101])
            assert ((fm[ab] != 0).sum(axis=1) == 60).all()
    finally:
        awm.set_params()
        orc.set_params()
    assert not np.array_equal(awm.tab_frame_mod(key, PAY)[0], fm[0])          # the default (mix) tables are different ones


def test_windows_and_conv_encode_match_oracle():
    assert np.array_equal(awm.tab_window(1024), orc.window(1024))
    assert np.array_equal(awm.tab_synth_window(), orc.synth_window())
    bits = np.random.default_rng(3).integers(0, 2, 128)
    for bt in (0, 1, 2):
        assert np.array_equal(awm.conv_encode(bt, bits), orc.conv_encode(bt, bits))


def test_bad_payload_is_an_error():
    with pytest.raises(awm.AwmError):
        awm.tab_frame_mod(None, "xyz")


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = awm.lib.awm_ctx_create(0, C.byref(h))
    assert rc == -2                                                        # AWM_ERR_NO_DEVICE
    assert b"no CPU fallback" in awm.lib.awm_last_error()
    with pytest.raises(awm.AwmError):
        awm.Context(0)


def test_group_key_tables_build_on_host_threads():
    """The host side of the per-clip-key tables of `get` (what the device's K16g is checked against, and the path of --linear): a group of
    keys builds on several threads without a GPU, and the debug entry points that need one say so instead of computing on the CPU."""
    import ctypes as C
    f = awm.lib.awm_debug_time_group_key_tables
    f.restype = C.c_double
    assert f(8, 4) > 0
    assert f(3, 0) > 0
    bad = (C.c_longlong * 9)()
    assert awm.lib.awm_debug_clip_key_tables_check_d(None, bytes(16), C.c_size_t(1), bad) != 0


def test_plan_chunks_reference_semantics():
    # wavchunkloader.cc:75-84: 30 min chunks, overlap lrint(2 * 51.69 s * 1.3 * 44100) samples
    L, O = 79380000, 5926502
    assert awm.plan_chunks(0) == []
    assert awm.plan_chunks(100) == [(0, 100, 0.0)]
    c = awm.plan_chunks(60 * 60 * 44100)
    assert [(a, b) for a, b, _ in c] == [(0, L), (L - O, L), (2 * (L - O), 60 * 60 * 44100 - 2 * (L - O))]
    assert abs(c[1][2] - (L - O) / 44100) < 1e-9
    # a stream that ends exactly on a chunk boundary yields one more chunk made of the overlap only
    assert awm.plan_chunks(L) == [(0, L, 0.0), (L - O, O, (L - O) / 44100)]


def test_merge_patterns_dedup_and_sort():
    def pat(time, bits, q, bt=0, typ=0):
        return dict(time=time, sync_index=int(time * 44100), sync_quality=q, block_type=bt, type=typ, decode_error=0.1,
                    speed=1.0, bits=bits)
    good, junk = PAY, "ffffffffffffffffffffffffffffffff"
    chunk0 = [pat(5.8, good, 1.3), pat(57.4, good, 1.4, 1), pat(57.4, good, 1.35, 2), pat(20.0, junk, 0.2)]
    chunk1 = [pat(57.4 + 0.001, good, 1.39, 1), pat(109.1, good, 1.2), pat(0.0, good, 1.3, 0, 2)]
    out = awm.merge_patterns(None, [chunk0, chunk1])
    # the B block seen by both chunks is kept once; best-rated payload first, ALL pattern last within it
    assert len(out) == 6
    assert [p["bits"] for p in out[:5]] == [good] * 5 and out[5]["bits"] == junk
    assert out[4]["type"] == 2 and [round(p["time"], 1) for p in out[:4]] == [5.8, 57.4, 57.4, 109.1]
    # the array path (no per-pattern Python objects; used inside the timed region of the multi-GPU path) gives the same list
    def arr(chunk):
        a = np.zeros(len(chunk), awm.PATTERN_DTYPE)
        for i, d in enumerate(chunk):
            p = awm.binding._pattern_from_dict(d)
            a[i] = np.frombuffer(bytes(p), awm.PATTERN_DTYPE)[0]
        return a
    assert awm.merge_patterns_raw(None, [arr(chunk0), arr(chunk1)]) == out
    assert awm.merge_patterns_raw(None, [arr([]), arr([])]) == []


def test_speed_host_pieces_match_oracle():
    """the host side of the speed search (no GPU): peak selection, smoothing, and the clip location generator -- the positions
    that get hashed and the candidate locations seeded from their SHA-1 (AES / SHA units when the CPU has them) -- against
    the oracle's restatement"""
    import ctypes as C
    import _oracle as orc
    lib = awm.lib
    rng = np.random.default_rng(11)
    speed = np.sort(rng.uniform(0.8, 1.25, 627))
    quality = rng.uniform(0, 1, 627)
    quality[100] = quality[101] = 1.5
    s1, q1 = speed.copy(), quality.copy()
    lib.awm_speed_select_n_best.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    n = lib.awm_speed_select_n_best(s1.ctypes.data, q1.ctypes.data, len(s1), 5)
    so, qo = orc.speed_select_n_best(speed, quality, 5)
    assert n == 5 and s1[:5].tolist() == so.tolist() and q1[:5].tolist() == qo.tolist()
    sp = 0.97 + np.arange(81) * 0.00005
    qq = np.exp(-((sp - 0.9712) / 0.0004) ** 2) + rng.uniform(0, 0.05, 81)
    lib.awm_speed_smooth_best.restype = C.c_double
    lib.awm_speed_smooth_best.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double]
    assert lib.awm_speed_smooth_best(sp.ctypes.data, qq.ctypes.data, 81, 1 - 1.00005, 20.0) == orc.speed_smooth_best(sp, qq, 1 - 1.00005, 20)
    # clip location: same samples hashed, same candidates, same winner as the oracle's get_best_clip_location
    key = bytes(range(16))
    x = orc.gen_noise(key, 2 * 44100 * 40)
    x[: 2 * 44100 * 10] *= 0.1                      # a quiet start: the loudest candidate clip is not arbitrary
    lib.awm_speed_clip_positions.restype = C.c_size_t
    lib.awm_speed_clip_positions.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p]
    cap = x.size // 300
    pos = np.zeros(cap, np.uint64)
    cnt = lib.awm_speed_clip_positions(key, x.size, cap, pos.ctypes.data)
    assert 0 < cnt <= cap and pos[0] == 0 and np.all(np.diff(pos[:cnt].astype(np.int64)) < 1000)
    hashed = np.ascontiguousarray(x[pos[:cnt].astype(np.int64)])
    loc = np.zeros(5)
    lib.awm_speed_clip_candidates.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    assert lib.awm_speed_clip_candidates(key, hashed.ctypes.data, cnt, 5, loc.ctypes.data) == 0
    assert np.all((0 <= loc) & (loc < 1))

    def energy(l, seconds=25.0):
        frames = x.size // 2
        start = int(max(l * (frames / 44100 - seconds), 0) * 44100)
        end = min(int(start + seconds * 44100), frames)
        c = x[2 * start:2 * end].astype(np.float64)
        return float((c * c).sum())
    best = max(loc, key=energy)
    assert orc.speed_clip_location(key, x, 2, 25.0) == best


def test_params_struct_round_trip_and_validation():
    """awm_params (reference Params, wmcommon.hh:33-89): defaults, the process-wide set, refusal of a foreign struct size"""
    p = awm.binding.Params()
    assert (p.water_delta, p.mix, p.hard, p.strict, p.snr, p.payload_size, p.frames_per_bit) == (0.01, 1, 0, 0, 0, 128, 2)
    assert (p.sync_threshold2, p.get_n_best, p.get_chunk_size, p.try_speed, p.test_speed) == (0.35, 8, 30.0, -1.0, -1.0)
    try:
        awm.binding.set_global_params(hard=1, water_delta=0.02, get_n_best=5)
        q = awm.binding.Params()
        assert awm.lib.awm_ctx_get_params(None, C.byref(q)) == 0
        assert (q.hard, q.water_delta, q.get_n_best, q.mix) == (1, 0.02, 5, 1)
        # the legacy setters address the same process-wide set
        awm.set_params(water_delta=0.015)
        awm.lib.awm_ctx_get_params(None, C.byref(q))
        assert q.water_delta == 0.015 and q.hard == 1
        bad = awm.binding.Params()
        bad.struct_size = 8
        assert awm.lib.awm_set_global_params(C.byref(bad)) == -3 and b"struct_size" in awm.lib.awm_last_error()
        bad = awm.binding.Params(get_n_best=0)
        assert awm.lib.awm_set_global_params(C.byref(bad)) == -3
        assert awm.lib.awm_ctx_set_params(None, C.byref(p)) == -3               # no context
    finally:
        awm.binding.set_global_params()
    awm.lib.awm_ctx_get_params(None, C.byref(q))
    assert (q.hard, q.water_delta, q.get_n_best) == (0, 0.01, 8)


def test_library_leaves_the_environment_alone():
    """awm_ctx_create used to export GPU_MAX_HW_QUEUES into its host's environment"""
    import subprocess, sys
    code = ("import os, ctypes as C, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None);"
            "import audiowmark_amd as awm; h = C.c_void_p(); awm.lib.awm_ctx_create(0, C.byref(h));"
            "print('GPU_MAX_HW_QUEUES' in os.environ, C.CDLL(None).getenv(b'GPU_MAX_HW_QUEUES'))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, timeout=300).stdout.split()
    assert out == ["False", "0"], out
