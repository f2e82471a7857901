"""Bounded-memory paths: `add` as a tile loop (awm_add_stream), file level add / get with chunked staging
(awm_add_watermark_file / awm_get_watermark_file = the reference's add_watermark / get_watermark, wmcommon.hh:226-228).

Bar: the tile loop's output is BIT-IDENTICAL to the whole-buffer path (same kernels on spans with the carried halo frame and
limiter maxima; the reference's streaming add keeps the same state, wmadd.cc:173,220-222, limiter.cc:51-64), for every length
class: shorter than a frame, shorter than a tile, exact tiles, one sample more."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PAY = "0123456789abcdef0011223344556677"
TILE = 128                      # frames of 1024 samples: the smallest tile the API takes


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
    yield g
    awm.set_params()
    ctx.close()


def noise(g, n, ch, seed):
    gen = g.torch.Generator(device="cuda")
    gen.manual_seed(seed)
    return g.torch.rand((n, ch), generator=gen, device="cuda", dtype=g.torch.float32) * 2 - 1


@pytest.mark.parametrize("n,ch", [(1, 2), (700, 1), (1024, 2), (TILE * 1024 - 1, 2), (TILE * 1024, 2), (TILE * 1024 + 1, 2),
                                  (2 * TILE * 1024, 1), (2 * TILE * 1024 + 300, 2), (5 * TILE * 1024 + 44100 * 3 + 17, 2),
                                  (3 * TILE * 1024 + 1023, 3)])
def test_add_tiles_equal_whole(gpu, n, ch):
    x = noise(gpu, n, ch, n % 1000 + ch)
    whole = gpu.ctx.add_watermark(None, PAY, x)
    tiles = gpu.ctx.add_watermark_tiles(None, PAY, x, TILE)
    assert gpu.torch.equal(whole, tiles)


def test_add_tiles_without_limiter_and_with_key(gpu):
    x = noise(gpu, 3 * TILE * 1024 + 5000, 2, 99) * 0.5
    key = gpu.awm.test_key(3)
    gpu.awm.set_params(test_no_limiter=True)
    try:
        whole = gpu.ctx.add_watermark(key, PAY, x)
        tiles = gpu.ctx.add_watermark_tiles(key, PAY, x, TILE)
    finally:
        gpu.awm.set_params()
    assert gpu.torch.equal(whole, tiles)


ZERO_FRAMES = [0, 1, 1023, 1024, 44100, 3 * 1024 + 17, 44100 * 61 + 5, TILE * 1024 * 2 + 1023]


@pytest.mark.parametrize("zero_frames", ZERO_FRAMES)
def test_add_stream_zero_frames_equals_the_reference(gpu, zero_frames):
    """add_stream_watermark (key, in, out, bits, zero_frames) (reference wmcommon.hh:226; wmadd.cc:501-526, 574-580; limiter.cc:69-88):
    the stream starts zero_frames samples into the frame / block / limiter grid.  Against the compiled reference's own
    add_stream_watermark with that argument (PCM RMS < 1e-6, max < 2e-6), and BIT-IDENTICAL to this library's whole-buffer add of
    "zero_frames zeros, then the input" with the zeros cut again -- which is what the reference's skip logic amounts to."""
    import _ref
    t = gpu.torch
    n = 3 * TILE * 1024 + 44100 + 311                          # three tiles and a bit: the carried prefix crosses every tile edge
    x = noise(gpu, n, 2, 31 + zero_frames % 97) * 0.98          # (loud: the limiter is at work in every block)
    got = gpu.ctx.add_watermark_tiles(None, PAY, x, TILE, zero_frames=zero_frames)
    assert got.shape == x.shape
    whole = gpu.ctx.add_watermark(None, PAY, t.cat([t.zeros((zero_frames, 2), device="cuda"), x]))[zero_frames:]
    assert t.equal(got, whole)
    if not _ref.available():
        pytest.skip("oracle/_ref is not built")
    ref = _ref.add_at(None, x.cpu().numpy(), 2, PAY, zero_frames).reshape(-1, 2)
    assert ref.shape == (n, 2)
    d = got.cpu().numpy().astype(np.float64) - ref
    assert np.sqrt((d ** 2).mean()) < 1e-6 and np.abs(d).max() < 2e-6
    # and the watermark it carries is the one a detector finds at that offset: not the one of zero_frames = 0
    if zero_frames % (1024 * 2226 * 2) > 4096:
        plain = gpu.ctx.add_watermark_tiles(None, PAY, x, TILE)
        assert not t.equal(got, plain)


@pytest.mark.parametrize("n,ch,zero_frames", [(0, 2, 5000), (1, 1, 1023), (500, 2, 700), (1024, 1, 1), (TILE * 1024, 2, 1023),
                                              (TILE * 1024 - 1023, 2, 1023), (TILE * 1024 + 1, 3, 2048 + 9)])
def test_add_stream_zero_frames_edge_lengths(gpu, n, ch, zero_frames):
    """lengths around the tile size with a prefix that hangs over the last tile; empty input; mono and 3 channels"""
    t = gpu.torch
    x = noise(gpu, n, ch, 5 + n % 50)
    got = gpu.ctx.add_watermark_tiles(None, PAY, x, TILE, zero_frames=zero_frames)
    whole = gpu.ctx.add_watermark(None, PAY, t.cat([t.zeros((zero_frames, ch), device="cuda"), x]))[zero_frames:]
    assert got.shape == x.shape and t.equal(got, whole)


@pytest.mark.parametrize("fmt,rate,zero_frames", [("s16", 44100, 3 * 1024 + 17), ("s16", 44100, 44100 * 200), ("f32", 48000, 48000 * 3 + 100)])
def test_file_add_with_zero_frames(gpu, tmp_path, fmt, rate, zero_frames):
    """awm_add_stream_watermark_file: the file level form; at 48 kHz the zeros are materialised in front of the resamplers"""
    import _ref
    t, awm = gpu.torch, gpu.awm
    bits, enc, big = {"s16": (16, 0, False), "f32": (32, 2, False)}[fmt]
    n = int(130.3 * rate)
    x = noise(gpu, n, 2, 41) * 0.9
    raw_in = gpu.ctx.pcm_encode(x.reshape(-1), bits, enc, big, True).cpu().numpy()
    src, dst = tmp_path / "in.raw", tmp_path / "out.raw"
    raw_in.tofile(src)
    rf = awm.binding.RawFormat(2, rate, bits, enc, int(big))
    gpu.ctx.add_watermark_file(None, PAY, src, dst, rf, rf, zero_frames=zero_frames)
    got = gpu.ctx.pcm_decode(t.from_numpy(np.fromfile(dst, np.uint8)).cuda(), bits, enc, big).reshape(-1, 2)
    assert got.shape == (n, 2)
    x_file = gpu.ctx.pcm_decode(t.from_numpy(raw_in).cuda(), bits, enc, big).reshape(n, 2)
    if not _ref.available():
        pytest.skip("oracle/_ref is not built")
    ref = _ref.add_at(None, x_file.cpu().numpy(), 2, PAY, zero_frames, sample_rate=rate).reshape(-1, 2)
    ref_q = gpu.ctx.pcm_decode(gpu.ctx.pcm_encode(t.from_numpy(ref).cuda().reshape(-1), bits, enc, big, True), bits, enc, big).reshape(-1, 2)
    d = (got - ref_q).cpu().numpy().astype(np.float64)
    # (16 bit output: the two pipelines may land on different sides of a quantisation step for a handful of samples)
    tol = 1.01 / 32768 if bits == 16 else 4e-6
    assert np.abs(d).max() <= tol and np.sqrt((d ** 2).mean()) < (2e-6 if bits == 16 else 1e-6)


def test_add_tiles_rejects_bad_use(gpu):
    import ctypes as C
    lib = gpu.awm.lib
    h = C.c_void_p()
    assert lib.awm_add_stream_create(gpu.ctx._h, bytes(16), PAY.encode(), 2, 16, C.byref(h)) < 0          # tile below one limiter look-ahead
    assert lib.awm_add_stream_create(gpu.ctx._h, bytes(16), b"xyz", 2, TILE, C.byref(h)) < 0               # payload does not parse
    assert lib.awm_add_stream_create(gpu.ctx._h, bytes(16), PAY.encode(), 2, TILE, C.byref(h)) == 0
    p = (C.c_void_p * 3)()
    k = (C.c_size_t * 3)()
    assert lib.awm_add_stream_push(h, 1000, 0, p, k) < 0                                                   # a middle tile must be full
    assert lib.awm_add_stream_push(h, 1000, 1, p, k) == 1 and k[0] == 1000
    assert lib.awm_add_stream_push(h, 0, 1, p, k) < 0                                                      # nothing after the last tile
    lib.awm_add_stream_destroy(h)


@pytest.mark.parametrize("seconds,fmt", [(200.5, "s16"), (96.0, "s24be"), (30.0, "f32")])
def test_file_add_equals_whole_buffer_path_and_get_finds_it(gpu, tmp_path, seconds, fmt):
    """raw PCM file -> file through the tile loop (95 s tiles: 200 s = 3 tiles) == decode, whole-buffer add, encode"""
    t, awm = gpu.torch, gpu.awm
    bits, enc, big = {"s16": (16, 0, False), "s24be": (24, 0, True), "f32": (32, 2, False)}[fmt]
    n = int(seconds * 44100)
    x = noise(gpu, n, 2, 7) * 0.9
    raw_in = gpu.ctx.pcm_encode(x.reshape(-1), bits, enc, big, True).cpu().numpy()
    src, dst = tmp_path / "in.raw", tmp_path / "out.raw"
    raw_in.tofile(src)
    rf = awm.binding.RawFormat(2, 44100, bits, enc, int(big))
    gpu.ctx.add_watermark_file(None, PAY, src, dst, rf, rf)
    got = np.fromfile(dst, np.uint8)
    x_file = gpu.ctx.pcm_decode(t.from_numpy(raw_in).cuda(), bits, enc, big).reshape(n, 2)
    want = gpu.ctx.pcm_encode(gpu.ctx.add_watermark(None, PAY, x_file).reshape(-1), bits, enc, big, True).cpu().numpy()
    assert got.size == want.size and np.array_equal(got, want)
    # get from the file == get from the resident samples
    from_file = gpu.ctx.get_watermark_file(None, dst, rf)
    resident = gpu.ctx.get_watermark(None, gpu.ctx.pcm_decode(t.from_numpy(got).cuda(), bits, enc, big).reshape(n, 2))
    key = lambda p: (round(p["time"], 6), p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"])
    assert [key(p) for p in from_file] == [key(p) for p in resident]
    assert any(p["bits"] == PAY for p in from_file)


def test_file_wav_in_wav_out(gpu, tmp_path):
    """WAV header handling on both sides of the streamed path (RIFF, 16 bit), incl. a length that is not a multiple of a frame"""
    import struct
    t = gpu.torch
    n = 44100 * 100 + 333
    x = noise(gpu, n, 2, 11) * 0.8
    pcm = gpu.ctx.pcm_encode(x.reshape(-1), 16, 0, False, True).cpu().numpy()
    hdr = b"RIFF" + struct.pack("<I", 36 + pcm.size) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 44100, 44100 * 4, 4, 16) \
        + b"data" + struct.pack("<I", pcm.size)
    src, dst = tmp_path / "in.wav", tmp_path / "out.wav"
    with open(src, "wb") as f:
        f.write(hdr)
        f.write(pcm.tobytes())
    gpu.ctx.add_watermark_file(None, PAY, src, dst)
    out = np.fromfile(dst, np.uint8)
    assert out.size == 44 + pcm.size and bytes(out[:4]) == b"RIFF" and struct.unpack("<I", bytes(out[40:44]))[0] == pcm.size
    x_file = gpu.ctx.pcm_decode(t.from_numpy(pcm).cuda(), 16, 0, False).reshape(n, 2)
    # a NAMED WAV file goes through libsndfile in the reference: samples are clipped at 32 bit and the top 16 bits kept
    # (sfoutputstream.cc:148-155), not the truncate-at-16-bit rule of the stdout / raw path
    want = gpu.ctx.pcm_encode(gpu.ctx.add_watermark(None, PAY, x_file).reshape(-1), 16, 0, False, False).cpu().numpy()
    assert np.array_equal(out[44:], want)
    assert any(p["bits"] == PAY for p in gpu.ctx.get_watermark_file(None, dst))


def _wav(pcm_bytes, announce=None, channels=2, rate=44100, bits=16):
    import struct
    n = len(pcm_bytes) if announce is None else announce
    return b"RIFF" + struct.pack("<I", 36 + n) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits) \
        + b"data" + struct.pack("<I", n) + pcm_bytes


def test_file_io_modes_write_the_same_bytes(gpu, tmp_path):
    """host side of the file level calls (host/wmfile.cc): input by the I/O workers through AudioInputStream::raw_region (pread of tile
    parts, read ahead of the GPU) or by one reader thread; output by one writer thread or by the workers (one shared mapping of the
    output / pwrite) -- every combination, with 3 and 16 workers, writes the file of the default mode byte for byte: a 16 bit WAV (a
    44 byte header: no tile starts on a page boundary of the file) of 500 s = 6 tiles; `get` of it finds the same patterns in every mode"""
    awm, t = gpu.awm, gpu.torch
    n = 500 * 44100 + 77
    x = noise(gpu, n, 2, 21) * 0.9
    pcm = gpu.ctx.pcm_encode(x.reshape(-1), 16, 0, False, True).cpu().numpy().tobytes()
    src = tmp_path / "in.wav"
    src.write_bytes(_wav(pcm))
    ref_bytes = ref_pats = None
    key = lambda p: (round(p["time"], 6), p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"])
    try:
        for threads in (0, 3, 16):
            for flags in (1, 0, 9, 11, 15):
                awm.lib.awm_set_io_threads(threads)
                awm.lib.awm_debug_set_io_flags(flags)
                dst = tmp_path / f"out_{threads}_{flags}.wav"
                gpu.ctx.add_watermark_file(None, PAY, src, dst)
                data = dst.read_bytes()
                pats = [key(p) for p in gpu.ctx.get_watermark_file(None, dst)]
                if ref_bytes is None:
                    ref_bytes, ref_pats = data, pats
                    assert len(data) == 44 + len(pcm) and any(p[4] == PAY for p in pats)
                assert data == ref_bytes, (threads, flags)
                assert pats == ref_pats, (threads, flags)
                dst.unlink()
    finally:
        awm.lib.awm_set_io_threads(0)
        awm.lib.awm_debug_set_io_flags(1)


def test_file_shorter_than_its_header_says_and_input_from_a_pipe(gpu, tmp_path):
    """a WAV file cut off in the middle (the header announces 300 s, the file holds 211.3 s): `add` writes what it got -- the samples of the
    whole-buffer path for those frames -- in every I/O mode; the same stream through a FIFO (length unknown to the reader: the ordered
    reader / writer threads) gives the same sample bytes"""
    import threading
    awm, t = gpu.awm, gpu.torch
    n_full, n_have = 300 * 44100, int(211.3 * 44100)
    x = noise(gpu, n_have, 2, 23) * 0.9
    pcm = gpu.ctx.pcm_encode(x.reshape(-1), 16, 0, False, True).cpu().numpy().tobytes()
    src = tmp_path / "cut.wav"
    src.write_bytes(_wav(pcm, announce=n_full * 4))
    x_file = gpu.ctx.pcm_decode(t.from_numpy(np.frombuffer(pcm, np.uint8).copy()).cuda(), 16, 0, False).reshape(n_have, 2)
    want = gpu.ctx.pcm_encode(gpu.ctx.add_watermark(None, PAY, x_file).reshape(-1), 16, 0, False, False).cpu().numpy().tobytes()
    try:
        for flags in (1, 0, 15):
            awm.lib.awm_debug_set_io_flags(flags)
            dst = tmp_path / f"cut_out_{flags}.wav"
            gpu.ctx.add_watermark_file(None, PAY, src, dst)
            out = dst.read_bytes()
            assert out[44:] == want, flags
    finally:
        awm.lib.awm_debug_set_io_flags(1)
    # a FIFO as input (raw samples): the reader thread, output of unknown length
    fifo = tmp_path / "in.fifo"
    os.mkfifo(fifo)
    feeder = threading.Thread(target=lambda: open(fifo, "wb").write(pcm))
    feeder.start()
    rf = awm.binding.RawFormat(2, 44100, 16, 0, 0)
    dst = tmp_path / "fifo_out.raw"
    gpu.ctx.add_watermark_file(None, PAY, fifo, dst, rf, rf)
    feeder.join()
    want_raw = gpu.ctx.pcm_encode(gpu.ctx.add_watermark(None, PAY, x_file).reshape(-1), 16, 0, False, True).cpu().numpy().tobytes()
    assert dst.read_bytes() == want_raw


def test_get_of_a_long_file_that_ends_before_its_header_says(gpu, tmp_path):
    """the file level `get` starts its chunks while the stream is still being loaded, with the chunk plan of the ANNOUNCED length (two chunks
    or more: > 30 min).  A WAV whose header announces 62 min and that holds 33.5: what the chunks saw beyond the end was not the stream --
    the result is thrown away and the plain order takes over: the pattern list is that of the same samples under a truthful header, with
    the chunks started during the load and with the whole stream loaded first."""
    awm, t = gpu.awm, gpu.torch
    n_have, n_said = int(33.5 * 60 * 44100), 62 * 60 * 44100
    x = gpu.ctx.add_watermark(None, PAY, noise(gpu, n_have, 2, 29) * 0.9)
    pcm = gpu.ctx.pcm_encode(x.reshape(-1), 16, 0, False, True).cpu().numpy().tobytes()
    del x
    cut, whole = tmp_path / "cut.wav", tmp_path / "whole.wav"
    cut.write_bytes(_wav(pcm, announce=n_said * 4))
    whole.write_bytes(_wav(pcm))
    key = lambda p: (round(p["time"], 6), p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"])
    want = [key(p) for p in gpu.ctx.get_watermark_file(None, whole)]
    assert sum(p[4] == PAY for p in want) > 30
    assert [key(p) for p in gpu.ctx.get_watermark_file(None, cut)] == want
    # headerless samples with three stray bytes behind the last whole frame (the length comes from the file's size)
    ragged = tmp_path / "ragged.raw"
    ragged.write_bytes(pcm + b"\x01\x02\x03")
    rf = awm.binding.RawFormat(2, 44100, 16, 0, 0)
    assert [key(p) for p in gpu.ctx.get_watermark_file(None, ragged, rf)] == want
    awm.lib.awm_debug_set_get_overlap(0)
    try:
        assert [key(p) for p in gpu.ctx.get_watermark_file(None, cut)] == want
        assert [key(p) for p in gpu.ctx.get_watermark_file(None, whole)] == want
        assert [key(p) for p in gpu.ctx.get_watermark_file(None, ragged, rf)] == want
    finally:
        awm.lib.awm_debug_set_get_overlap(1)


def test_ctx_warm_up_and_trim(gpu, tmp_path):
    """awm_ctx_warm_up creates the streams a first call would create, awm_ctx_trim gives the kept workspaces back (lanes' scratch buffers, the
    file level rings, the stream's PCM buffer): results before, between and after are the same, a trimmed context allocates again by itself"""
    import ctypes as C
    awm, t = gpu.awm, gpu.torch
    lib = awm.lib
    lib.awm_ctx_warm_up.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.awm_ctx_trim.argtypes = [C.c_void_p]
    ctx = awm.Context(0)
    try:
        assert lib.awm_ctx_warm_up(ctx._h, 1, 1) == 0
        n = 200 * 44100
        x = noise(gpu, n, 2, 61) * 0.9
        raw = ctx.pcm_encode(x.reshape(-1), 16, 0, False, True).cpu().numpy()
        src, dst = tmp_path / "in.raw", tmp_path / "out.raw"
        raw.tofile(src)
        rf = awm.binding.RawFormat(2, 44100, 16, 0, 0)
        key = lambda p: (round(p["time"], 6), p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"])
        ctx.add_watermark_file(None, PAY, src, dst, rf, rf)
        first_bytes = dst.read_bytes()
        first = [key(p) for p in ctx.get_watermark_file(None, dst, rf)]
        assert any(p[4] == PAY for p in first)
        free0 = t.cuda.mem_get_info()[0]
        assert lib.awm_ctx_trim(ctx._h) == 0
        assert t.cuda.mem_get_info()[0] > free0                       # something came back
        ctx.add_watermark_file(None, PAY, src, dst, rf, rf)
        assert dst.read_bytes() == first_bytes
        assert [key(p) for p in ctx.get_watermark_file(None, dst, rf)] == first
        assert lib.awm_ctx_trim(ctx._h) == 0 and lib.awm_ctx_trim(ctx._h) == 0
        w = ctx.add_watermark(None, PAY, x)
        assert [key(p) for p in ctx.get_watermark(None, w)] == [key(p) for p in gpu.ctx.get_watermark(None, w)]
    finally:
        ctx.close()


@pytest.mark.parametrize("seconds,fmt,rate", [(200.5, "s16", 44100), (3700.0, "s16", 44100), (96.0, "s24be", 44100), (40.0, "f32", 44100), (120.0, "s16", 48000)])
def test_file_add_then_get_as_one_call(gpu, tmp_path, seconds, fmt, rate):
    """awm_add_get_watermark_file ("watermark, then verify"): the file of awm_add_watermark_file byte for byte and the pattern list of
    awm_get_watermark_file on it -- without reading the file back: the output stage keeps the samples as the file holds them (after the
    sample format's quantisation).  Tile loop (44.1 kHz: 3 tiles; 62 minutes: 40 tiles, four chunks), 24 bit big endian, float samples,
    and the resampled path (48 kHz)."""
    t, awm = gpu.torch, gpu.awm
    bits, enc, big = {"s16": (16, 0, False), "s24be": (24, 0, True), "f32": (32, 2, False)}[fmt]
    n = int(seconds * rate)
    x = noise(gpu, n, 2, 7 + int(seconds)) * 0.9
    raw_in = gpu.ctx.pcm_encode(x.reshape(-1), bits, enc, big, True).cpu().numpy()
    del x
    src, dst, dst2 = tmp_path / "in.raw", tmp_path / "out.raw", tmp_path / "out2.raw"
    raw_in.tofile(src)
    rf = awm.binding.RawFormat(2, rate, bits, enc, int(big))
    key = lambda p: (round(p["time"], 6), p["sync_index"], p["type"], p["block_type"], p["bits"], p["sync_quality"], p["decode_error"])
    gpu.ctx.add_watermark_file(None, PAY, src, dst, rf, rf)
    want = [key(p) for p in gpu.ctx.get_watermark_file(None, dst, rf)]
    got = [key(p) for p in gpu.ctx.add_get_watermark_file(None, PAY, src, dst2, rf, rf)]
    assert dst2.read_bytes() == dst.read_bytes()
    assert got == want and any(p[4] == PAY for p in got)
    # the same list with the whole stream loaded before the first chunk starts (default: the chunks start during the load)
    awm.lib.awm_debug_set_get_overlap(0)
    try:
        assert [key(p) for p in gpu.ctx.get_watermark_file(None, dst, rf)] == want
    finally:
        awm.lib.awm_debug_set_get_overlap(1)
    # ... and with ONE reader thread instead of the I/O workers (headerless PCM then announces no length itself: the chunks during the load
    # take it from the file -- round 6: "input stream is longer than announced" for streams of two chunks or more)
    awm.lib.awm_debug_set_io_flags(0)
    try:
        assert [key(p) for p in gpu.ctx.get_watermark_file(None, dst, rf)] == want
    finally:
        awm.lib.awm_debug_set_io_flags(1)
    # and the context is back to normal: a plain `add` afterwards keeps nothing
    gpu.ctx.add_watermark_file(None, PAY, src, dst2, rf, rf)
    assert dst2.read_bytes() == dst.read_bytes()
    assert [key(p) for p in gpu.ctx.get_watermark_file(None, dst2, rf)] == want
