"""Parity of the HIP path (through the C ABI) against the oracle on seeded inputs, against the golden
fixtures, and -- at BASELINE.json's full size -- through size independent properties.

Tolerances (north_star): decoded payload bits and sync positions bit exact; embedded PCM within
1e-5 RMS of the reference's float output (the tests enforce 1e-6, SURVEY.md Appendix C calibration);
sync qualities within 1e-5.  Integer / bit results (Viterbi output, tables) are compared exactly."""
import json
import os

import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PAY1 = "0123456789abcdef0011223344556677"
PAY2 = "f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0"
RMS_TOL = 1e-6
QUALITY_TOL = 1e-5


def noise(seed, n, ch):
    return np.random.default_rng(seed).uniform(-1, 1, (n, ch)).astype(np.float32)


def rms(a, b):
    d = np.asarray(a, np.float64).ravel() - np.asarray(b, np.float64).ravel()
    return float(np.sqrt(np.mean(d * d))) if d.size else 0.0


def pkey(p):
    return (round(p["time"], 9), p["sync_index"], p["type"], p["block_type"], p["bits"])


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
    g.dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    yield g
    awm.set_params()
    orc.set_params()
    ctx.close()


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "golden_v1.json")) as f:
        j = json.load(f)
    return j, np.load(os.path.join(HERE, "golden", "golden_v1.npz"))


@pytest.fixture(scope="module")
def stream70():
    n = 70 * 44100
    return orc.add(None, noise(41, n, 2), 2, PAY1).reshape(n, 2)


# ---- K1: FFTAnalyzer::fft_range ----------------------------------------------------------------
@pytest.mark.parametrize("ch,start", [(1, 0), (1, 77), (2, 100), (2, 333), (3, 5)])
def test_fft_range(gpu, ch, start):
    x = noise(21 + ch, 12000, ch)
    got = gpu.ctx.fft_range(gpu.dev(x), start, 8).cpu().numpy()
    want = orc.fft_range(x, ch, start, 8)
    assert np.abs(got - want).max() < 1e-6 * np.abs(want).max()
    with pytest.raises(gpu.awm.AwmError):                      # reading past the end is refused
        gpu.ctx.fft_range(gpu.dev(x), 12000 - 1000, 1)


def test_fft_range_golden(gpu, golden):
    _, z = golden
    got = gpu.ctx.fft_range(gpu.dev(noise(21, 8000, 2)), 100, 3).cpu().numpy()
    assert np.abs(got - z["fft_range_s21"]).max() < 1e-6 * np.abs(z["fft_range_s21"]).max()


# ---- K2/K3: add ---------------------------------------------------------------------------------
@pytest.mark.parametrize("ch,n,limiter,key", [
    (2, 70 * 44100 + 13, True, None), (2, 30 * 44100, False, "t42"), (1, 20 * 44100 + 1, True, None),
    (2, 1024, True, None), (2, 700, True, None), (1, 1025, False, None), (3, 4 * 44100 + 5, True, None),
    (2, 1, True, None)])
def test_add_parity(gpu, ch, n, limiter, key):
    key = gpu.awm.test_key(42) if key else None
    x = noise(100 + ch + n % 97, n, ch)
    gpu.awm.set_params(test_no_limiter=not limiter)
    orc.set_params(test_no_limiter=not limiter)
    want = orc.add(key, x, ch, PAY1).reshape(n, ch)
    got = gpu.ctx.add_watermark(key, PAY1, gpu.dev(x)).cpu().numpy()
    assert got.shape == x.shape                                # output length == input length (wmadd.cc:570-572)
    assert rms(got, want) < RMS_TOL
    assert np.abs(got - want).max() < 2e-6
    if n >= 44100:
        assert 0.005 < rms(want, x) < 0.05                     # a watermark was actually embedded
    gpu.awm.set_params()
    orc.set_params()


def test_add_empty(gpu):
    x = gpu.torch.zeros((0, 2), dtype=gpu.torch.float32, device="cuda")
    assert gpu.ctx.add_watermark(None, PAY1, x).shape == (0, 2)


def test_add_golden(gpu, golden):
    _, z = golden
    got = gpu.ctx.add_watermark(None, PAY1, gpu.dev(noise(31, 2 * 44100 + 77, 1))).cpu().numpy().ravel()
    assert rms(got, z["add_mono_s31"]) < RMS_TOL
    gpu.awm.set_params(test_no_limiter=True)
    got = gpu.ctx.add_watermark(gpu.awm.test_key(42), PAY2, gpu.dev(noise(32, 44100 + 500, 2))).cpu().numpy().ravel()
    gpu.awm.set_params()
    assert rms(got, z["add_stereo_s32_nolimiter"]) < RMS_TOL


def test_add_snr_bound(gpu):
    # tests/block-decoder-test.sh:18 of the reference: SNR >= 32.4 dB on white noise without limiter
    x = noise(7, 200 * 44100, 2)
    gpu.awm.set_params(test_no_limiter=True)
    w = gpu.ctx.add_watermark(None, PAY2, gpu.dev(x)).cpu().numpy()
    gpu.awm.set_params()
    d = w.astype(np.float64) - x
    snr = 10 * np.log10((x.astype(np.float64) ** 2).sum() / (d ** 2).sum())
    assert snr >= 32.4
    # the device-side meter behind `add --snr` (awm_ctx_snr_begin / _end) measures the same thing, also with the limiter at work
    for no_limiter in (True, False):
        gpu.awm.set_params(test_no_limiter=no_limiter)
        gpu.ctx.snr_begin()
        gpu.ctx.add_watermark(None, PAY2, gpu.dev(x))
        metered = gpu.ctx.snr_end()
        gpu.awm.set_params()
        assert abs(metered - snr) < 1e-6, (no_limiter, metered, snr)


def test_add_sharded_spans_equal_whole(gpu):
    """Two frame spans with halo frames + a max-combined limiter table reproduce the single launch bit for bit
    (this is the multi-GPU decomposition of `add`, run here on one device)."""
    t = gpu.torch
    n, split = 12 * 44100 + 321, 200 * 1024
    x = gpu.dev(noise(55, n, 2))
    fm = gpu.awm.tab_frame_mod(None, PAY1)
    whole = gpu.ctx.add_d(x, fm, 0.01, True)
    n_blocks = n // 44100 + 2
    bm = t.empty(n_blocks, dtype=t.float32, device="cuda")
    gpu.ctx.add_init_block_max(bm)
    out = t.empty_like(x)
    a, b = x[:split], x[split:]
    halo_after = x[split:split + 1024].contiguous()
    halo_before = x[split - 1024:split].contiguous()
    gpu.ctx.add_mix(a, out[:split], fm, 0.01, 0, None, halo_after, bm)
    gpu.ctx.add_mix(b, out[split:], fm, 0.01, split // 1024, halo_before, None, bm)
    gpu.ctx.add_limit(out[:split], 0, bm)
    gpu.ctx.add_limit(out[split:], split, bm)
    assert t.equal(out, whole)


# ---- K4: SyncFinder::sync_fft --------------------------------------------------------------------
def test_sync_fft(gpu, stream70):
    w = stream70
    got_db, got_have = gpu.ctx.sync_fft(gpu.dev(w), 264, 40)
    want_db, want_have = orc.sync_fft(w, 2, 264, 40)
    assert np.array_equal(got_have.cpu().numpy(), want_have)
    assert np.abs(got_db.cpu().numpy() - want_db).max() < 2e-4             # sums of dB values ~ -60
    want = np.zeros(40, np.int8)
    want[[1, 2, 3, 17, 39]] = 1
    first, last = 2 * (264 + 10 * 1024) + 1, 2 * (264 + 30 * 1024)
    got_db, got_have = gpu.ctx.sync_fft(gpu.dev(w), 264, 40, want, first, last)
    want_db, want_have = orc.sync_fft(w, 2, 264, 40, want, first, last)
    assert np.array_equal(got_have.cpu().numpy(), want_have) and want_have.sum() == 1
    assert np.abs(got_db.cpu().numpy() - want_db).max() < 2e-4


# ---- K5: search_approx / search --------------------------------------------------------------------
def test_search_approx(gpu, golden, stream70):
    j, _ = golden
    gi, graw, gmean = gpu.ctx.search_approx(None, gpu.dev(stream70))
    oi, oraw, omean = orc.search_approx(None, stream70, 2)
    assert np.array_equal(gi, oi) and len(gi) == j["approx70"]["n"]
    assert np.abs(graw - oraw).max() < 5e-5 and np.abs(gmean - omean).max() < 1e-5
    top = np.argsort(-np.abs(graw - gmean))[:8]
    assert sorted(gi[top].tolist()) == sorted(j["approx70"]["top_index"][:8])


@pytest.mark.parametrize("ch", [1, 2])
def test_sync_search_block(gpu, ch):
    n = 115 * 44100
    w = orc.add(None, noise(60 + ch, n, ch), ch, PAY2).reshape(n, ch)
    gi, gq, gb = gpu.ctx.sync_search(None, gpu.dev(w))
    oi, oq, ob = orc.sync_search(None, w, ch)
    assert gi.tolist() == oi.tolist() and gb.tolist() == ob.tolist()       # positions and block types: exact
    assert np.abs(gq - oq).max() < QUALITY_TOL
    assert any(abs(int(i) - 250 * 1024) < 512 for i in gi)                 # first A block after the 250 frame pad


def test_sync_search_golden(gpu, golden, stream70):
    j, _ = golden
    gi, gq, gb = gpu.ctx.sync_search(None, gpu.dev(stream70))
    assert gi.tolist() == j["sync70"]["index"] and gb.tolist() == j["sync70"]["block_type"]
    assert np.abs(gq - np.array(j["sync70"]["quality"])).max() < QUALITY_TOL


def test_sync_search_cut_start(gpu, stream70):
    # reference tests/sync-test.sh: cutting an arbitrary (odd) number of samples only moves the sync positions
    cut = 12345
    w = np.ascontiguousarray(stream70[cut:])
    gi, gq, gb = gpu.ctx.sync_search(None, gpu.dev(w))
    oi, oq, ob = orc.sync_search(None, w, 2)
    assert gi.tolist() == oi.tolist() and gb.tolist() == ob.tolist()
    assert np.abs(gq - oq).max() < QUALITY_TOL


# ---- K7: soft bits, K8: Viterbi -------------------------------------------------------------------
def test_block_soft_bits(gpu, golden, stream70):
    j, z = golden
    idx = j["mix_decode70_index"]
    got, ok = gpu.ctx.block_soft_bits(None, gpu.dev(stream70), [idx, idx + 8, len(stream70) - 1000])
    assert ok.tolist() == [1, 1, 0]                                        # fft_range refuses to read past the end
    assert np.abs(got[0] - z["mix_decode70"]).max() < 5e-3 and np.abs(z["mix_decode70"]).mean() > 50
    want = orc.mix_decode(None, stream70, 2, idx + 8)
    assert np.abs(got[1] - want).max() < 5e-3


@pytest.mark.parametrize("bt", [0, 1, 2])
def test_viterbi_bit_exact(gpu, bt):
    rng = np.random.default_rng(200 + bt)
    soft, want_bits, want_err = [], [], []
    for sigma in (0.0, 0.3, 0.5, 0.7, 1.5):
        bits = rng.integers(0, 2, 128)
        coded = orc.conv_encode(bt, bits).astype(np.float32)
        s = (coded + rng.normal(0, sigma, coded.shape)).astype(np.float32)
        b, e = orc.conv_decode_soft(bt, s)
        soft.append(s)
        want_bits.append(b)
        want_err.append(e)
    got_bits, got_err = gpu.ctx.viterbi_decode(bt, np.stack(soft))
    assert np.array_equal(got_bits, np.stack(want_bits))
    assert np.array_equal(got_err, np.array(want_err, np.float32))         # path metric: same float operations, same order


def test_viterbi_one_launch_equals_the_launch_chain(gpu):
    """K8 as ONE launch (8 resident workgroups per decode meeting at a per-decode counter, tickets instead of blockIdx) against the
    chain of 16 launches: bits and error values identical, also for a batch that oversubscribes the chip several times (600 decodes
    = 4800 workgroups; the ticket order is what keeps that free of deadlock), for the three code types mixed in one call, for blocks
    whose soft bits are NaN (the checked path), and repeated (the counters must be back at zero after every launch)."""
    rng = np.random.default_rng(77)
    lib = gpu.awm.lib
    for n, sigma in ((3, 0.5), (40, 0.7), (600, 1.0)):
        for bt in (0, 1, 2):
            bits = rng.integers(0, 2, (n, 128))
            coded = np.stack([orc.conv_encode(bt, b) for b in bits]).astype(np.float32)
            soft = (coded + rng.normal(0, sigma, coded.shape)).astype(np.float32)
            if n == 40:
                soft[5] = np.nan
            try:
                lib.awm_debug_set_viterbi_persistent(0)
                want_bits, want_err = gpu.ctx.viterbi_decode(bt, soft)
            finally:
                lib.awm_debug_set_viterbi_persistent(1)
            for rep in range(3):
                got_bits, got_err = gpu.ctx.viterbi_decode(bt, soft)
                assert np.array_equal(got_bits, want_bits), (n, bt, rep)
                assert np.array_equal(got_err, want_err, equal_nan=True), (n, bt, rep)
            if n == 3:
                for i in range(n):
                    b, e = orc.conv_decode_soft(bt, soft[i])
                    assert np.array_equal(got_bits[i], b) and got_err[i] == np.float32(e)
    # a whole `get` (three chunks on concurrent lanes, each with its own batch of A, B and AB decodes) with either form forced
    try:
        x = gpu.dev(noise(4242, 25 * 60 * 44100, 2))
        gpu.awm.set_params(chunk_size_min=10.0)
        w = gpu.ctx.add_watermark(None, PAY1, x)
        lists = []
        for form in (0, 1, 0, 1):
            lib.awm_debug_set_viterbi_persistent(form)
            lists.append([(pkey(p), p["sync_quality"], p["decode_error"]) for p in gpu.ctx.get_watermark(None, w)])
        assert lists[0] == lists[1] == lists[2] == lists[3] and len(lists[0]) > 40
    finally:
        gpu.awm.set_params()
        lib.awm_debug_set_viterbi_persistent(-1)                 # back to the choice by the measured launch cost


# ---- whole decode -----------------------------------------------------------------------------------
def test_decode_chunk_golden(gpu, golden, stream70):
    j, _ = golden
    got = sorted(gpu.ctx.decode_chunk(None, gpu.dev(stream70), True), key=lambda p: (p["time"], p["type"], p["block_type"], p["bits"]))
    want = j["decode_chunk70"]
    assert [pkey(p) for p in got] == [pkey(p) for p in want]
    for g, w in zip(got, want):
        assert abs(g["sync_quality"] - w["sync_quality"]) < QUALITY_TOL and abs(g["decode_error"] - w["decode_error"]) < 1e-5


def test_get_clip(gpu, golden, stream70):
    j, _ = golden
    clip = np.ascontiguousarray(stream70[20 * 44100: 44 * 44100])
    got = gpu.ctx.get_watermark(None, gpu.dev(clip))
    assert [pkey(p) for p in got] == [pkey(p) for p in j["get_clip24"]]
    assert got[0]["bits"] == PAY1 and got[0]["type"] == 1                  # CLIP pattern first
    mono = np.ascontiguousarray(clip[:, :1])
    want = orc.get(None, mono, 1)
    assert [pkey(p) for p in gpu.ctx.get_watermark(None, gpu.dev(mono))] == [pkey(p) for p in want]


def test_wrong_key_finds_nothing(gpu, stream70):
    # reference tests/key-test.sh
    pats = gpu.ctx.get_watermark(gpu.awm.test_key(2), gpu.dev(stream70))
    assert all(p["bits"] != PAY1 for p in pats)


def test_add_then_get_end_to_end_vs_oracle(gpu):
    n = 118 * 44100 + 999
    x = noise(77, n, 2)
    key = gpu.awm.test_key(7)
    w = gpu.ctx.add_watermark(key, PAY2, gpu.dev(x))
    got = gpu.ctx.get_watermark(key, w)
    want = orc.get(key, orc.add(key, x, 2, PAY2).reshape(n, 2), 2)
    assert [pkey(p) for p in got] == [pkey(p) for p in want]
    assert sum(p["bits"] == PAY2 for p in got) >= 4                        # A, B, AB, all (block-decoder-test.sh)


# ---- full size (BASELINE.json configs[1]): size independent properties ------------------------------
def test_full_size_60min_roundtrip(gpu):
    t = gpu.torch
    n = 60 * 60 * 44100
    g = t.Generator(device="cuda")
    g.manual_seed(5)
    x = t.rand((n, 2), generator=g, device="cuda", dtype=t.float32) * 2 - 1
    w = gpu.ctx.add_watermark(None, PAY1, x)
    assert w.shape == x.shape
    assert float(w.abs().max()) <= 0.99 * (1 + 1e-6)                       # limiter ceiling
    delta = (w - x)
    assert 0.01 < float(delta.double().pow(2).mean().sqrt()) < 0.06
    # idempotence of the deterministic pipeline
    assert t.equal(gpu.ctx.add_watermark(None, PAY1, x), w)
    pats = gpu.ctx.get_watermark(None, w)
    matches = [p for p in pats if p["bits"] == PAY1]
    assert len(matches) >= 100                                             # 69 blocks + AB pairs + per-chunk "all"
    block = 2226 * 1024
    found = sorted({round(p["time"] * 44100) for p in matches if p["type"] == 0 and p["block_type"] < 2})
    expected = [250 * 1024 + i * block for i in range(69)]
    hit = sum(any(abs(f - e) < 512 for f in found) for e in expected)
    assert hit >= 66                                                       # every full block located within half a frame
    # each 30 min reference chunk decodes on its own to the same patterns that the merged result holds
    chunks = gpu.awm.plan_chunks(n)
    assert len(chunks) == 3
    first, count, off = chunks[1]
    sub = gpu.ctx.decode_chunk(None, w[first:first + count], False)
    merged = {(round(p["time"], 3), p["block_type"], p["bits"]) for p in pats if p["type"] == 0}
    for p in sub:
        if p["type"] == 0 and p["bits"] == PAY1:
            assert (round(p["time"] + off, 3), p["block_type"], p["bits"]) in merged


def test_sharded_stream_world1(gpu):
    """The torch.distributed (RCCL) code path with a single rank gives the same PCM / patterns as the plain calls."""
    import torch.distributed as dist
    from audiowmark_amd import sharded
    t = gpu.torch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=t.device("cuda", 0))
        created = True
    try:
        n = 75 * 44100 + 5
        x = gpu.dev(noise(91, n, 2))
        pipe = sharded.ShardedStream(gpu.ctx, dist, n, 2)
        out = t.empty_like(x)
        pipe.add_watermark(None, PAY1, x, out)
        assert t.equal(out, gpu.ctx.add_watermark(None, PAY1, x))
        got = pipe.get_watermark(None, out)
        want = gpu.ctx.get_watermark(None, out)
        assert [pkey(p) for p in got] == [pkey(p) for p in want]
    finally:
        if created:
            dist.destroy_process_group()


def test_three_channels_end_to_end(gpu):
    """Channel counts other than 1 / 2 take the generic kernels (strided fetch, FFT based refinement)."""
    n = 58 * 44100
    x = noise(93, n, 3)
    w = gpu.ctx.add_watermark(None, PAY2, gpu.dev(x))
    want_w = orc.add(None, x, 3, PAY2).reshape(n, 3)
    assert rms(w.cpu().numpy(), want_w) < RMS_TOL
    got = gpu.ctx.get_watermark(None, gpu.dev(want_w))
    want = orc.get(None, want_w, 3)
    assert [pkey(p) for p in got] == [pkey(p) for p in want]
    assert any(p["bits"] == PAY2 for p in got)


def test_unmarked_audio_uses_n_best_fallback(gpu):
    """No watermark -> fewer than n_best peaks above the threshold: the n_best largest maxima are still refined and
    decoded (sync_select_threshold_and_n_best, reference syncfinder.cc:364-383) and agree with the oracle."""
    n = 75 * 44100
    x = noise(95, n, 2)
    gi, gq, gb = gpu.ctx.sync_search(None, gpu.dev(x))
    oi, oq, ob = orc.sync_search(None, x, 2)
    assert gi.tolist() == oi.tolist() and gb.tolist() == ob.tolist() and len(gi) == 8
    assert np.abs(gq - oq).max() < QUALITY_TOL
    assert all(p["bits"] != PAY1 for p in gpu.ctx.get_watermark(None, gpu.dev(x)))


def test_unsupported_parameters_are_refused(gpu):
    for fpb in (0, 9, -2):                                    # --frames-per-bit: 1 .. 8 are block geometries the kernels take
        gpu.awm.set_params(frames_per_bit=fpb)
        try:
            with pytest.raises(gpu.awm.AwmError):
                gpu.ctx.add_watermark(None, PAY1, gpu.dev(noise(1, 5000, 2)))
        finally:
            gpu.awm.set_params()


@pytest.mark.parametrize("fpb", [1, 3, 4])
def test_frames_per_bit_other_than_two(gpu, fpb):
    """--frames-per-bit (reference audiowmark.cc:675, wmcommon.cc:36-48: a block is 510 sync + 858 x frames_per_bit data frames: 1368 / 3084 /
    3942 instead of 2226).  Every kernel takes the geometry as an argument; the tables that are built on the device for clip batches with
    a key per clip hand over to the host builders.  `add` against the oracle and the compiled reference (PCM), `get` of the oracle's
    output in BLOCK mode (200 s) and in CLIP mode (40 s): positions, types, payloads equal, qualities within the tolerance; the tile loop
    equals the whole buffer; a batch of clips with a key per clip equals the single calls; and the tables cached per key follow the
    parameter (the same key with two geometries in one process)."""
    import _ref as ref
    t = gpu.torch
    base = gpu.ctx.get_watermark(None, gpu.dev(orc.add(None, noise(440, 60 * 44100, 2), 2, PAY1).reshape(-1, 2)))     # default geometry first
    gpu.awm.set_params(frames_per_bit=fpb)
    orc.set_params(frames_per_bit=fpb)
    ref.set_params(frames_per_bit=fpb)
    try:
        for seconds in (200, 40):
            x = noise(441 + fpb + seconds, seconds * 44100 + 321, 2)
            want = orc.add(None, x, 2, PAY1).reshape(-1, 2)
            got = gpu.ctx.add_watermark(None, PAY1, gpu.dev(x)).cpu().numpy()
            assert rms(got, want) < RMS_TOL and np.abs(got - want).max() < 2e-6, (fpb, seconds)
            if seconds == 40:
                assert rms(got, np.asarray(ref.add(None, x, 2, PAY1)).reshape(-1, 2)) < RMS_TOL
            ours, theirs = gpu.ctx.get_watermark(None, gpu.dev(want)), orc.get(None, want, 2)
            assert [pkey(p) for p in ours] == [pkey(p) for p in theirs] and len(ours) > 0, (fpb, seconds)
            assert max(abs(a["sync_quality"] - b["sync_quality"]) for a, b in zip(ours, theirs)) < QUALITY_TOL
            assert any(p["bits"] == PAY1 for p in ours), (fpb, seconds)
            if seconds == 200:
                assert t.equal(gpu.ctx.add_watermark_tiles(None, PAY1, gpu.dev(x), tile_frames1024=128), gpu.ctx.add_watermark(None, PAY1, gpu.dev(x)))
        if fpb == 3:
            # a stream that starts inside the frame / block grid (zero_frames), and one stream over two contexts
            from audiowmark_amd import sharded
            x = noise(470, 150 * 44100 + 5, 2)
            for zf in (1, 44100 + 17):
                want = np.asarray(ref.add_at(None, x, 2, PAY1, zf)).reshape(-1, 2)
                assert rms(gpu.ctx.add_watermark_tiles(None, PAY1, gpu.dev(x), tile_frames1024=128, zero_frames=zf).cpu().numpy(), want) < RMS_TOL, zf
            other = gpu.awm.Context(0)
            try:
                xd = gpu.dev(x)
                cut = 71 * 1024
                outs = [t.empty_like(xd[:cut]), t.empty_like(xd[cut:])]
                sharded.multi_add([gpu.ctx, other], None, PAY1, [xd[:cut].contiguous(), xd[cut:].contiguous()], outs)
                whole = gpu.ctx.add_watermark(None, PAY1, xd)
                assert t.equal(t.cat(outs), whole)
                assert [pkey(p) for p in sharded.multi_get([gpu.ctx, other], None, outs)] == [pkey(p) for p in gpu.ctx.get_watermark(None, whole)]
            finally:
                other.close()
        # clips with a key per clip: the batch entry points against the single calls
        keys = [gpu.awm.test_key(k) for k in range(1, 6)]
        clips = [gpu.dev(noise(460 + k, (25 + 3 * k) * 44100, 2)) for k in range(5)]
        marked = gpu.ctx.add_watermark_batch_keys(keys, PAY2, clips)
        for k in range(5):
            assert t.equal(marked[k], gpu.ctx.add_watermark(keys[k], PAY2, clips[k])), k
        batch = gpu.ctx.get_watermark_batch_keys(keys, marked)
        for k in range(5):
            assert [pkey(p) for p in batch[k]] == [pkey(p) for p in gpu.ctx.get_watermark(keys[k], marked[k])], k
        assert sum(any(p["bits"] == PAY2 for p in c) for c in batch) >= 4
    finally:
        gpu.awm.set_params()
        orc.set_params()
        ref.set_params()
    again = gpu.ctx.get_watermark(None, gpu.dev(orc.add(None, noise(440, 60 * 44100, 2), 2, PAY1).reshape(-1, 2)))
    assert [pkey(p) + (p["sync_quality"],) for p in again] == [pkey(p) + (p["sync_quality"],) for p in base] and len(base) > 0


def _sharded_worker(rank, world, port, lengths, q):
    import torch
    import torch.distributed as dist
    import audiowmark_amd as awm
    from audiowmark_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # both ranks share the one GPU of the test box
    try:
        awm.set_params(chunk_size_min=10.0)
        total = sum(lengths)
        whole = noise(97, total, 2)
        s = sum(lengths[:rank])
        local = torch.from_numpy(whole[s:s + lengths[rank]].copy()).cuda()
        ctx = awm.Context(0)
        pipe = sharded.ShardedStream(ctx, dist, lengths[rank], 2)
        out = torch.empty_like(local)
        pipe.add_watermark(None, PAY1, local, out)
        pats = pipe.get_watermark(None, out)
        q.put((rank, "ok", out.cpu().numpy(), pats, pipe.part.work()))
    except Exception:
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


def test_sharded_two_ranks_equal_single_process(gpu):
    """Two processes (gloo transport, one shared GPU) run the sharded add + get on a 21 minute stream cut into two
    spans: the PCM must equal the single-call result bit for bit and the merged patterns must be identical.
    Same code path as the multi-GPU bench except for the transport (RCCL there)."""
    import socket
    import torch.multiprocessing as mp
    lengths = [14 * 60 * 44100 // 1024 * 1024, 7 * 60 * 44100 + 333]
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_sharded_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[2]
    gpu.awm.set_params(chunk_size_min=10.0)
    try:
        whole = gpu.dev(noise(97, sum(lengths), 2))
        want = gpu.ctx.add_watermark(None, PAY1, whole)
        got = np.concatenate([results[0][2], results[1][2]])
        assert np.array_equal(got, want.cpu().numpy())
        want_pats = gpu.ctx.get_watermark(None, want)
        assert [pkey(p) for p in results[0][3]] == [pkey(p) for p in want_pats]
        assert results[1][3] is None and all(w > 0 for w in results[0][4])     # both ranks worked on start frames
        assert sum(p["bits"] == PAY1 for p in want_pats) >= 20
    finally:
        gpu.awm.set_params()


@pytest.mark.parametrize("minutes,cuts", [(21, [0.5]), (21, [0.31, 0.34, 0.8]), (21, [0.0, 0.55]), (45, [0.2, 0.45, 0.7])])
def test_multi_context_get_equals_single(gpu, minutes, cuts):
    """awm_multi_add_d / awm_multi_get_d (the protocol of host/wmshard.cc with one host thread per context and device copies as
    transport; here all contexts on the one GPU of the box): spans cut anywhere in frames -- also a span much shorter than a block
    and an empty one -- give the PCM of the whole-stream add bit for bit and exactly its pattern list.  10 minute chunks: the
    cuts fall inside chunks, inside chunk overlaps and near chunk ends."""
    from audiowmark_amd import sharded
    t = gpu.torch
    gpu.awm.set_params(chunk_size_min=10.0)
    try:
        total = minutes * 60 * 44100 + 777
        whole = gpu.dev(noise(131 + minutes, total, 2))
        edges = [0] + [int(total * c) // 1024 * 1024 for c in cuts] + [total]
        spans = [whole[a:b].contiguous() for a, b in zip(edges[:-1], edges[1:])]
        ctxs = [gpu.ctx] + [gpu.awm.Context(0) for _ in spans[1:]]
        outs = [t.empty_like(s) for s in spans]
        sharded.multi_add(ctxs, None, PAY1, spans, outs)
        want = gpu.ctx.add_watermark(None, PAY1, whole)
        assert t.equal(t.cat(outs), want)
        got = sharded.multi_get(ctxs, None, outs)
        want_pats = gpu.ctx.get_watermark(None, want)
        assert [pkey(p) for p in got] == [pkey(p) for p in want_pats]
        assert [(p["sync_quality"], p["decode_error"]) for p in got] == [(p["sync_quality"], p["decode_error"]) for p in want_pats]
        assert sum(p["bits"] == PAY1 for p in want_pats) >= 20
    finally:
        gpu.awm.set_params()


def test_key_tables_on_the_device(gpu):
    """K16 (hip/keytab.hip): the frame_mod tables of `add` built on the device -- AES-128-CTR draws, the per-frame band shuffles, the three
    key-wide Fisher-Yates shuffles (targets in parallel, swaps by one lane), the table fill -- equal the host's awm_tab_frame_mod byte for
    byte, for 1024 keys (test keys 1 .. 1000, the zero key, random 128 bit keys) and two payloads; and a batch `add` with one key per
    clip gives the same PCM with the tables from either side."""
    import ctypes as C
    rng = np.random.default_rng(99)
    keys = [gpu.awm.test_key(k) for k in range(1, 1001)] + [bytes(16)] + [bytes(rng.integers(0, 256, 16, dtype=np.uint8)) for _ in range(23)]
    flat = b"".join(gpu.awm.key_bytes(k) for k in keys)
    for payload in (PAY1, "ffffffffffffffffffffffffffffffff"):
        out = np.zeros((len(keys), 2 * 2226 * 81), np.int8)
        rc = gpu.awm.lib.awm_debug_frame_mod_tables_d(gpu.ctx._h, flat, C.c_size_t(len(keys)), payload.encode(), out.ctypes.data_as(C.c_void_p))
        assert rc == 0, gpu.awm.lib.awm_last_error()
        check = range(len(keys)) if payload == PAY1 else range(0, len(keys), 37)
        for i in check:
            want = np.asarray(gpu.awm.tab_frame_mod(keys[i], payload), np.int8).ravel()
            assert np.array_equal(out[i], want), (payload, i)
    clips = [gpu.dev(noise(500 + i, 20 * 44100 + 13 * i, 2)) for i in range(24)]
    try:
        gpu.awm.lib.awm_debug_set_key_tables_on_device(0)
        host_side = gpu.ctx.add_watermark_batch_keys(keys[:24], PAY1, clips)
    finally:
        gpu.awm.lib.awm_debug_set_key_tables_on_device(1)
    device_side = gpu.ctx.add_watermark_batch_keys(keys[:24], PAY1, clips)
    assert all(gpu.torch.equal(a, b) for a, b in zip(host_side, device_side))


def test_get_key_tables_on_the_device(gpu):
    """K16g (hip/keytab.hip): the tables `get` needs per key of a clip batch -- K5w's sync chains and row frames, the want list, K4s's
    gathered layout (perm, pos), the mix entries in shuffled order, the inverse bit order -- built on the device, group by group as the
    batch path builds them, equal the host's build of the same tables element for element, for 1024 keys (test keys, the zero key, random
    128 bit keys); and a batch `get` with one key per clip finds the same patterns with the tables from either side."""
    import ctypes as C
    rng = np.random.default_rng(98)
    keys = [gpu.awm.test_key(k) for k in range(1, 1001)] + [bytes(16)] + [bytes(rng.integers(0, 256, 16, dtype=np.uint8)) for _ in range(23)]
    flat = b"".join(gpu.awm.key_bytes(k) for k in keys)
    bad = (C.c_longlong * 9)()
    rc = gpu.awm.lib.awm_debug_clip_key_tables_check_d(gpu.ctx._h, flat, C.c_size_t(len(keys)), bad)
    assert rc == 0, gpu.awm.lib.awm_last_error()
    assert list(bad) == [0] * 9, dict(zip("chains row_frames want perm pos mix_frame mix_up mix_down inv_order".split(), bad))
    # 150 clips = three groups on two lanes (one of them with two groups: both table areas in turn), lengths ragged, one clip silent
    n = 150
    clips = [noise(700 + i, (12 + i % 19) * 44100 + 11 * i, 2) for i in range(n)]
    clips[40][:] = 0
    marked = gpu.ctx.add_watermark_batch_keys(keys[:n], PAY2, [gpu.dev(c) for c in clips])
    try:
        gpu.awm.lib.awm_debug_set_key_tables_on_device(0)
        host_side = gpu.ctx.get_watermark_batch_keys(keys[:n], marked)
        gpu.awm.lib.awm_debug_set_key_tables_on_device(2)     # the tables of all keys first
        all_first = gpu.ctx.get_watermark_batch_keys(keys[:n], marked)
    finally:
        gpu.awm.lib.awm_debug_set_key_tables_on_device(1)     # (default: a group's tables one group ahead of its lane)
    device_side = gpu.ctx.get_watermark_batch_keys(keys[:n], marked)
    assert device_side == host_side
    assert all_first == host_side
    assert device_side == gpu.ctx.get_watermark_batch_keys(keys[:n], marked)          # (and again: the areas are reused)
    assert sum(any(p["bits"] == PAY2 for p in c) for c in device_side) >= 100


def test_multi_context_clip_batches_equal_single_context(gpu):
    """awm_multi_add_watermark_batch_d / awm_multi_get_watermark_batch_d: 40 clips of 20 - 30 s dealt unevenly to three contexts (all on
    the one GPU of the box), every clip with its own key and, in a second pass, one key for all: PCM and pattern lists equal the
    single-context batch calls, in clip order; an empty share is fine."""
    from audiowmark_amd import sharded
    t = gpu.torch
    rng = np.random.default_rng(5)
    n = 40
    clips = [gpu.dev(noise(900 + i, int(rng.integers(20, 31)) * 44100 + int(rng.integers(0, 999)), 2)) for i in range(n)]
    keys = [gpu.awm.test_key(i + 1) for i in range(n)]
    owner = [0 if i % 5 == 0 else (1 if i % 2 else 2) for i in range(n)]
    ctxs = [gpu.ctx, gpu.awm.Context(0), gpu.awm.Context(0), gpu.awm.Context(0)]          # (the fourth gets nothing)
    for key_arg in (keys, keys[3]):
        per_clip = isinstance(key_arg, list)
        want_pcm = gpu.ctx.add_watermark_batch_keys(keys, PAY1, clips) if per_clip else gpu.ctx.add_watermark_batch(key_arg, PAY1, clips)
        got_pcm = sharded.multi_add_batch(ctxs, key_arg, PAY1, clips, owner)
        assert all(t.equal(a, b) for a, b in zip(got_pcm, want_pcm))
        want = gpu.ctx.get_watermark_batch_keys(keys, want_pcm) if per_clip else gpu.ctx.get_watermark_batch(key_arg, want_pcm)
        got = sharded.multi_get_batch(ctxs, key_arg, got_pcm, owner)
        assert got == want
        assert sum(any(p["bits"] == PAY1 for p in g) for g in got) >= 35
    with pytest.raises(gpu.awm.AwmError):
        sharded.multi_get_batch(ctxs, keys, clips, [9] * n)
    for c in ctxs[1:]:
        c.close()


def test_get_right_after_add_is_ordered_behind_it(gpu):
    """add -> get of the same buffer on one context without a wait in between: the chunks of the `get` run on other streams than the
    `add` and must still see all of its output (event ordering behind the context's stream).  Same patterns as a `get` after a
    host-side wait, also for a `get` of a PART of the buffer, repeatedly (a missing wait shows as a changed pattern); and the whole-stream
    `add` equals the tile loop of the file path (whose limiter runs tile by tile) sample by sample on a stream of three chunks."""
    torch = gpu.torch
    n = 61 * 60 * 44100 + 777                           # three chunks
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    x = torch.rand((n, 2), generator=g, device="cuda") * 2 - 1
    out = torch.empty_like(x)
    gpu.ctx.add_watermark(None, PAY1, x, out=out)
    torch.cuda.synchronize()
    plain = [pkey(p) for p in gpu.ctx.get_watermark(None, out)]
    assert len(plain) > 100
    reference = out.clone()
    for _ in range(4):
        out.zero_()
        gpu.ctx.add_watermark(None, PAY1, x, out=out)
        armed = [pkey(p) for p in gpu.ctx.get_watermark(None, out)]
        assert armed == plain
        assert torch.equal(out, reference)
    # a part of the buffer (offset inside the first chunk's segments, whole frames or not)
    for off in (1024 * 100, 12345):
        part = out[off:]
        torch.cuda.synchronize()
        want = [pkey(p) for p in gpu.ctx.get_watermark(None, part)]
        gpu.ctx.add_watermark(None, PAY1, x, out=out)
        assert [pkey(p) for p in gpu.ctx.get_watermark(None, part)] == want
    # the tile loop of the file path limits tile by tile: the same samples
    tiles = gpu.ctx.add_watermark_tiles(None, PAY1, x, tile_frames1024=4096)
    assert torch.equal(tiles, reference)


def test_multi_context_short_stream_and_errors(gpu):
    """a stream in the ClipDecoder's range is decoded by rank 0 alone (same patterns); a failing rank does not hang the others"""
    from audiowmark_amd import sharded
    t = gpu.torch
    n = 40 * 44100
    whole = gpu.ctx.add_watermark(None, PAY2, gpu.dev(noise(77, n, 2)))
    cut = n // 2 // 1024 * 1024
    ctxs = [gpu.ctx, gpu.awm.Context(0)]
    got = sharded.multi_get(ctxs, None, [whole[:cut].contiguous(), whole[cut:].contiguous()])
    want = gpu.ctx.get_watermark(None, whole)
    assert [pkey(p) for p in got] == [pkey(p) for p in want] and any(p["bits"] == PAY2 for p in got)
    # a span that is not a whole number of frames in the middle of the stream is refused by every rank
    with pytest.raises(gpu.awm.AwmError):
        sharded.multi_add(ctxs, None, PAY1, [whole[:cut + 5].contiguous(), whole[cut + 5:].contiguous()],
                          [t.empty_like(whole[:cut + 5]), t.empty_like(whole[cut + 5:])])
    # the MAIN context's settings are in force on every rank (helpers with other settings would build other plans and wait for messages
    # that never come): a helper's own parameter set is ignored ...
    ctxs[1].set_params(frames_per_bit=9)
    again = sharded.multi_get(ctxs, None, [whole[:cut].contiguous(), whole[cut:].contiguous()])
    assert [pkey(p) for p in again] == [pkey(p) for p in want]
    ctxs[1].set_params()
    # ... and the main context's unsupported ones fail the call
    ctxs[0].set_params(frames_per_bit=9)
    try:
        with pytest.raises(gpu.awm.AwmError):
            sharded.multi_get(ctxs, None, [whole[:cut].contiguous(), whole[cut:].contiguous()])
    finally:
        ctxs[0].set_params()
    # ONE failing rank (its span has a length but no samples) does not hang the other: the call returns its error
    import ctypes as C
    long_stream = gpu.dev(noise(78, 6 * 60 * 44100, 2))
    half = long_stream.shape[0] // 2 // 1024 * 1024
    h = (C.c_void_p * 2)(*[c._h for c in ctxs])
    first_span = long_stream[:half].contiguous()
    ptr = (C.c_void_p * 2)(gpu.awm.binding._dev_ptr(first_span), None)
    lens = np.asarray([half, long_stream.shape[0] - half], np.uint64)
    buf = ctxs[0]._pattern_buffer(64)
    rc = gpu.awm.lib.awm_multi_get_d(h, 2, gpu.awm.key_bytes(None), ptr, 2, lens.ctypes.data, 64, C.cast(buf, C.c_void_p))
    assert rc < 0


# ---- sample rates other than 44100 Hz (zita-resampler restated on both sides: parity with zita itself is unpinned) ----
@pytest.mark.parametrize("rate_in,rate_out,ch,n", [(48000, 44100, 2, 100000), (44100, 48000, 2, 77777), (96000, 44100, 1, 50001),
                                                   (22050, 44100, 2, 30000), (48000, 44100, 1, 5), (32000, 44100, 2, 1)])
def test_resample_matches_restated_zita(gpu, rate_in, rate_out, ch, n):
    x = noise(700 + n % 89, n, ch)
    want = orc.resample(x, ch, rate_in, rate_out).reshape(-1, ch)
    got = gpu.ctx.resample(gpu.dev(x), rate_in, rate_out).cpu().numpy()
    assert got.shape == want.shape and got.shape[0] == -(-n * rate_out // rate_in)      # ceil (n * rate_out / rate_in)
    assert np.array_equal(got, want)                                                     # same products, same order


@pytest.mark.parametrize("rate_in,rate_out,ch,n", [(33333, 44100, 2, 100000), (44100, 33333, 1, 70001), (44101, 44100, 2, 50000),
                                                   (11111, 44100, 1, 20000), (33333, 44100, 2, 3)])
def test_resample_vresampler_fallback(gpu, rate_in, rate_out, ch, n):
    """rates zita's fixed-ratio Resampler refuses: ResamplerImpl::create falls back to VResampler (resample.cc:233-270)"""
    x = noise(900 + n % 97, n, ch)
    want = orc.resample(x, ch, rate_in, rate_out).reshape(-1, ch)
    got = gpu.ctx.resample(gpu.dev(x), rate_in, rate_out).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-6          # phase from the exact product vs zita's accumulated double


def test_resample_unsupported_ratio(gpu):
    with pytest.raises(gpu.awm.AwmError):
        gpu.ctx.resample(gpu.dev(noise(1, 1000, 1)), 1000000, 44100)                     # ratio < 1 / 16: neither zita class takes it


@pytest.mark.parametrize("rate,ch,limiter", [(48000, 2, True), (96000, 1, True), (32000, 2, False), (33333, 2, True)])
def test_add_at_other_rates(gpu, rate, ch, limiter):
    n = 60 * rate + 123
    x = noise(800 + ch, n, ch)
    gpu.awm.set_params(test_no_limiter=not limiter)
    orc.set_params(test_no_limiter=not limiter)
    try:
        want = orc.add(None, x, ch, PAY1, sample_rate=rate).reshape(n, ch)
        got = gpu.ctx.add_watermark(None, PAY1, gpu.dev(x), sample_rate=rate).cpu().numpy()
    finally:
        gpu.awm.set_params()
        orc.set_params()
    assert rms(got, want) < RMS_TOL and np.abs(got - want).max() < 2e-6
    assert 0.003 < rms(want, x) < 0.05


def test_roundtrip_48k(gpu):
    """add at 48 kHz, then what `get` does with a 48 kHz file: resample to 44.1 kHz and decode -- against the oracle's same steps."""
    rate, n = 48000, 115 * 48000
    x = noise(811, n, 2)
    w = gpu.ctx.add_watermark(None, PAY2, gpu.dev(x), sample_rate=rate)
    y = gpu.ctx.resample(w, rate, 44100)
    got = gpu.ctx.get_watermark(None, y)
    want = orc.get(None, orc.resample(w.cpu().numpy(), 2, rate, 44100), 2)
    assert [pkey(p) for p in got] == [pkey(p) for p in want]
    assert sum(p["bits"] == PAY2 for p in got) >= 3


def test_many_chunks_on_lanes(gpu):
    """A stream cut into more chunks than there are work lanes (3 minute chunks of a 16 minute stream): the chunk groups, the
    concurrent lanes and the single-lane order must all give the oracle's pattern list."""
    import os
    x = noise(123, 16 * 60 * 44100 + 777, 2)
    gpu.awm.set_params(chunk_size_min=3.0)
    orc.set_params(chunk_size_min=3.0)
    try:
        assert len(gpu.awm.plan_chunks(len(x))) > 4
        w = gpu.ctx.add_watermark(None, PAY1, gpu.dev(x))
        got = gpu.ctx.get_watermark(None, w)
        gpu.awm.lib.awm_ctx_set_chunk_lanes(gpu.ctx._h, 1)
        try:
            one = gpu.ctx.get_watermark(None, w)
        finally:
            gpu.awm.lib.awm_ctx_set_chunk_lanes(gpu.ctx._h, 4)
        want = orc.get(None, w.cpu().numpy(), 2)
        assert [pkey(p) for p in got] == [pkey(p) for p in one] == [pkey(p) for p in want]
        assert sum(p["bits"] == PAY1 for p in got) >= 10
    finally:
        gpu.awm.set_params()
        orc.set_params()


def test_clip_batch_on_lanes(gpu):
    """BASELINE config 5 in miniature: a batch of independent 30 s clips (different noise, keys as in SURVEY 8d are not
    needed for parity) decoded concurrently on the context's lanes gives, clip by clip, what the single call and the oracle give."""
    n = 30 * 44100
    clips = [noise(900 + i, n + 17 * i, 2) for i in range(6)]
    marked = [gpu.ctx.add_watermark(None, PAY1 if i % 2 else PAY2, gpu.dev(c)) for i, c in enumerate(clips)]
    one_by_one = [gpu.ctx.get_watermark(None, m) for m in marked]
    for threads in (1, 3, 0):
        batch = gpu.ctx.get_watermark_batch(None, marked, n_threads=threads)
        assert [[pkey(p) for p in b] for b in batch] == [[pkey(p) for p in o] for o in one_by_one]
    for i in (0, 5):
        want = orc.get(None, marked[i].cpu().numpy(), 2)
        assert [pkey(p) for p in one_by_one[i]] == [pkey(p) for p in want]
        assert any(p["bits"] == (PAY1 if i % 2 else PAY2) for p in one_by_one[i])
    assert gpu.ctx.get_watermark_batch(None, []) == []
    # awm_add_watermark_batch_d: the clips dealt to lanes -- bit-identical to one call per clip
    import torch
    same_payload = [gpu.ctx.add_watermark(None, PAY1, gpu.dev(c)) for c in clips]
    batch_add = gpu.ctx.add_watermark_batch(None, PAY1, [gpu.dev(c) for c in clips])
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(same_payload, batch_add))
    assert gpu.ctx.add_watermark_batch(None, PAY1, []) == []


def test_add_batch_in_one_launch_per_stage_equals_the_per_clip_launches(gpu):
    """awm_add_watermark_batch_d / _batch_keys_d watermark a batch of stereo clips with ONE launch per stage (block maxima, K2, limiter
    table, limiter; blockIdx.y = the clip, spans sized for the batch): bit-identical to four launches per clip (awm_debug_set_add_batched 0)
    and to one call per clip -- ragged lengths, odd frame counts (the limiter's values behind the last whole float4), a clip shorter than
    a frame, one of 70 s, digital silence, clips that reach the limiter's ceiling; more clips than one group of keys (256)."""
    t = gpu.torch
    lengths = [30 * 44100, 7 * 44100 + 1, 500, 70 * 44100 + 3, 1024, 1025, 12 * 44100 + 777] + [(3 + i % 5) * 44100 + 17 * i for i in range(270)]
    clips = []
    for i, n in enumerate(lengths):
        x = noise(2000 + i, n, 2)
        if i % 11 == 3:
            x[:] = 0
        if i % 7 == 2:
            x *= 1.6                                      # beyond the ceiling: the limiter has work to do
        clips.append(gpu.dev(x))
    keys = [gpu.awm.test_key(1 + i) for i in range(len(clips))]
    one_by_one = [gpu.ctx.add_watermark(None, PAY1, c) for c in clips[:12]]
    try:
        gpu.awm.lib.awm_debug_set_add_batched(0)
        per_clip = gpu.ctx.add_watermark_batch(None, PAY1, clips)
        per_clip_keys = gpu.ctx.add_watermark_batch_keys(keys, PAY2, clips)
        gpu.awm.lib.awm_debug_set_add_batched(1)              # a group's tables while the previous group is watermarked
        overlapped_keys = gpu.ctx.add_watermark_batch_keys(keys, PAY2, clips)
    finally:
        gpu.awm.lib.awm_debug_set_add_batched(2)              # (default: the tables of all keys first)
    batched = gpu.ctx.add_watermark_batch(None, PAY1, clips)
    batched_keys = gpu.ctx.add_watermark_batch_keys(keys, PAY2, clips)
    t.cuda.synchronize()
    assert all(t.equal(a, b) for a, b in zip(per_clip, batched))
    assert all(t.equal(a, b) for a, b in zip(per_clip_keys, batched_keys))
    assert all(t.equal(a, b) for a, b in zip(per_clip_keys, overlapped_keys))
    assert all(t.equal(a, b) for a, b in zip(one_by_one, batched[:12]))
    assert not t.equal(batched[0], clips[0]) and float(batched[2 + 7].abs().max()) <= 1.0
    # and again into the same outputs (the staging of the arguments is reused)
    again = gpu.ctx.add_watermark_batch_keys(keys, PAY2, clips, outs=[t.empty_like(c) for c in clips])
    t.cuda.synchronize()
    assert all(t.equal(a, b) for a, b in zip(again, batched_keys))


def test_clip_batch_groups(gpu):
    """The group path of awm_get_watermark_batch_d (padded clips side by side, one launch per stage and group): more clips than one
    group holds, lengths from 3 s to 50 s, mono and stereo mixed (groups are per channel count), digital silence (no candidate at
    all), very short material (the selection falls back to the single-clip search) -- clip by clip what one call per clip gives.
    The padded copy of a clip carries 2048 frames of zeros on either side and leaves the rest of the slice alone: with the slices
    poisoned (NaNs) before the copy the results must stay the same -- nobody reads past those zeros -- also when every clip is made to
    take the sequential selection, which does read whole slices and zeroes them first."""
    import torch
    rng = np.random.default_rng(77)
    clips = []
    for i in range(70):
        seconds = [30, 30, 30, 12, 50, 3, 30, 21][i % 8]
        ch = 1 if i % 9 == 4 else 2
        x = noise(1500 + i, seconds * 44100 + 13 * i, ch)
        if i % 16 == 7:
            x[:] = 0
        if i % 10 == 3:
            x[len(x) // 3: len(x) // 2] = 0          # a gap of digital silence inside
        if i % 10 == 6:
            x[:len(x) // 4] = 0                      # ... at the start
        if i % 10 == 8:
            x[-(len(x) // 5):] = 0                   # ... at the end
        clips.append(x)
    marked = []
    for i, c in enumerate(clips):
        marked.append(gpu.ctx.add_watermark(None, PAY1 if i % 2 else PAY2, gpu.dev(c)) if len(c) >= 20 * 44100 and c.any() else gpu.dev(c))
    # (mixed channel counts: one batch call per channel count, as the binding asserts equal shapes)
    for ch in (1, 2):
        sel = [m for m, c in zip(marked, clips) if c.shape[1] == ch]
        one_by_one = [gpu.ctx.get_watermark(None, m) for m in sel]
        batch = gpu.ctx.get_watermark_batch(None, sel)
        assert len(batch) == len(sel)
        assert batch == one_by_one
        if ch == 2:
            assert sum(any(p["bits"] in (PAY1, PAY2) for p in b) for b in batch) >= 20
        try:
            gpu.awm.lib.awm_debug_set_clip_poison(1)
            assert gpu.ctx.get_watermark_batch(None, sel) == one_by_one
            gpu.awm.lib.awm_debug_set_group_fallback(1)
            assert gpu.ctx.get_watermark_batch(None, sel) == one_by_one
        finally:
            gpu.awm.lib.awm_debug_set_clip_poison(0)
            gpu.awm.lib.awm_debug_set_group_fallback(0)


@pytest.mark.parametrize("n", [0, 1, 1000, 1024, 2049, 44100])
def test_get_on_tiny_inputs(gpu, n):
    """Inputs far too short to carry a block: the reference pads and searches anyway (ClipDecoder); results must agree."""
    x = noise(300 + n, n, 2)
    if n == 0:
        t = gpu.torch.zeros((0, 2), dtype=gpu.torch.float32, device="cuda")
        assert gpu.ctx.get_watermark(None, t) == []
        return
    got = gpu.ctx.get_watermark(None, gpu.dev(x))
    want = orc.get(None, x, 2)
    if n == 1:
        # a single sample: every soft bit is an exact tie except a handful that carry nothing but FFT rounding noise,
        # so the decoded junk is not comparable -- only the (all-tie) sync decisions are, see test_degenerate_sync_ties
        assert len(got) >= 1 and all(p["type"] == 1 for p in got)
        return
    assert [pkey(p) for p in got] == [pkey(p) for p in want]


def _clip_padded(x):
    ch = x.shape[1]
    n = (2226 + 5) * 1024 * ch
    vals = x.ravel()
    last = min(n, vals.size)
    pad_start = n + (n - last if last < n else 0)
    return np.concatenate([np.zeros(pad_start, np.float32), vals[:last], np.zeros(n, np.float32)]).reshape(-1, ch)


@pytest.mark.parametrize("kind", ["one_sample", "all_zero"])
def test_degenerate_sync_ties(gpu, kind):
    """All scores are exactly 0: thousands of tied local maxima.  The device-side selection (tie runs, -96 dB for exact
    zeros, frames skipped as silence) must hand the same sequence to the same std::sort as the reference does."""
    x = noise(301, 1, 2) if kind == "one_sample" else np.zeros((50 * 44100, 2), np.float32)
    p = _clip_padded(x)
    gi, graw, gmean = gpu.ctx.search_approx(None, gpu.dev(p), clip_mode=True)
    oi, oraw, omean = orc.search_approx(None, p, 2, True)
    assert np.array_equal(gi, oi) and np.array_equal(graw, oraw) and np.array_equal(gmean, omean)
    g = gpu.ctx.sync_search(None, gpu.dev(p), clip_mode=True)
    o = orc.sync_search(None, p, 2, True)
    assert g[0].tolist() == o[0].tolist() and g[2].tolist() == o[2].tolist() and np.array_equal(g[1], o[1])
    w = gpu.ctx.add_watermark(None, PAY1, gpu.dev(x)).cpu().numpy()
    assert rms(w, orc.add(None, x, 2, PAY1).reshape(x.shape)) < RMS_TOL
    assert all(q["bits"] != PAY1 for q in gpu.ctx.get_watermark(None, gpu.dev(x)))           # and no crash on NaN soft bits


def _np_encode(x, bits, encoding, big, direct16):
    """The reference's RawConverter::to_raw rules (rawconverter.cc:155-215, rawconverter.hh:34-50) in numpy float32."""
    x = x.astype(np.float32)
    width = bits // 8
    if encoding == 2:
        c = np.clip(x, -1, 1)
        a = c.astype(">f4" if big else "<f4") if bits == 32 else c.astype(np.float64).astype(">f8" if big else "<f8")
        return a.view(np.uint8)
    if direct16:
        s = (x * np.float32(32768)).astype(np.float32)
        i = np.where(s >= 32767, 32767, np.where(s <= -32768, -32768, np.trunc(s))).astype(np.int64) << 16
    else:
        s = (x * np.float32(2147483648)).astype(np.float32)
        i = np.where(s >= np.float32(2147483648), 2147483647, np.where(s <= np.float32(-2147483648), -2147483648, np.trunc(s.astype(np.float64)))).astype(np.int64)
    u = (i & 0xFFFFFFFF).astype(np.uint64)
    if encoding == 1:
        u ^= 0x80000000
    out = np.zeros((len(x), width), np.uint8)
    for b in range(width):
        sig = width - 1 - b if big else b
        out[:, b] = (u >> (8 * (4 - width + sig))) & 0xFF
    return out.ravel()


def _np_decode(raw, bits, encoding, big):
    width = bits // 8
    b = raw.reshape(-1, width)
    if encoding == 2:
        a = b.copy().view((">f4" if big else "<f4") if bits == 32 else (">f8" if big else "<f8")).ravel()
        return a.astype(np.float32)
    u = np.zeros(len(b), np.uint64)
    for k in range(width):
        sig = width - 1 - k if big else k
        u |= b[:, k].astype(np.uint64) << (8 * (4 - width + sig))
    if encoding == 1:
        u ^= 0x80000000
    i = u.astype(np.uint32).view(np.int32)
    return (i.astype(np.float32) * np.float32(1.0 / 2147483648.0)).astype(np.float32)


@pytest.mark.parametrize("bits,encoding,big,direct16", [
    (16, 0, False, True), (16, 0, False, False), (16, 0, True, True), (24, 0, False, True), (24, 0, True, True),
    (32, 0, False, True), (8, 1, False, True), (16, 1, False, True), (32, 2, False, True), (64, 2, True, True)])
def test_pcm_staging_bit_exact(gpu, bits, encoding, big, direct16):
    """Device-side RawConverter: every sample format byte for byte (encode) / bit for bit (decode)."""
    rng = np.random.default_rng(bits * 10 + encoding)
    x = np.concatenate([rng.uniform(-1.2, 1.2, 100003), [0, 1, -1, 0.99999994, -0.5 / 32768, 0.5 / 32768, 32767 / 32768, 1e-9, -1e-9]]).astype(np.float32)
    got = gpu.ctx.pcm_encode(gpu.dev(x), bits, encoding, big, direct16).cpu().numpy()
    d16 = direct16 and bits == 16 and encoding == 0 and not big
    want = _np_encode(x, bits, encoding, big, d16)
    assert np.array_equal(got, want)
    back = gpu.ctx.pcm_decode(gpu.dev(want), bits, encoding, big).cpu().numpy()
    assert np.array_equal(back.view(np.uint32), _np_decode(want, bits, encoding, big).view(np.uint32))


RAW_FORMATS = [  # the 18 formats of the reference's tests/raw-format-test.sh:36-69: (name, bits, encoding, big endian)
    ("s8", 8, 0, False), ("u8", 8, 1, False),
    ("s16le", 16, 0, False), ("s24le", 24, 0, False), ("s32le", 32, 0, False),
    ("u16le", 16, 1, False), ("u24le", 24, 1, False), ("u32le", 32, 1, False), ("f32le", 32, 2, False), ("f64le", 64, 2, False),
    ("s16be", 16, 0, True), ("s24be", 24, 0, True), ("s32be", 32, 0, True),
    ("u16be", 16, 1, True), ("u24be", 24, 1, True), ("u32be", 32, 1, True), ("f32be", 32, 2, True), ("f64be", 64, 2, True)]


@pytest.mark.parametrize("name,bits,encoding,big", RAW_FORMATS, ids=[f[0] for f in RAW_FORMATS])
def test_pcm_staging_equals_reference_rawconverter(gpu, name, bits, encoding, big):
    """Device-side conversion == the reference's RawConverter (rawconverter.cc:155-286, compiled unmodified in oracle/_ref):
    to_raw byte for byte, from_raw bit for bit, for every format of tests/raw-format-test.sh."""
    import _ref
    if not _ref.available():
        pytest.skip("oracle/_ref (compiled reference) not built")
    rng = np.random.default_rng(len(name) * 1000 + bits + encoding)
    x = np.concatenate([rng.uniform(-1.2, 1.2, 200003), rng.uniform(-3e-5, 3e-5, 5000),
                        [0, 1, -1, 0.99999994, -0.99999994, -0.5 / 32768, 0.5 / 32768, 32767 / 32768, -32768.5 / 32768, 1e-9, -1e-9,
                         1 - 2.0 ** -24, 2.0 ** -31, -2.0 ** -31]]).astype(np.float32)
    want = _ref.raw_convert(x, bits, encoding, big, True)
    got = gpu.ctx.pcm_encode(gpu.dev(x), bits, encoding, big, True).cpu().numpy()
    assert np.array_equal(got, want)
    raw = rng.integers(0, 256, want.size, dtype=np.uint8) if encoding != 2 else want      # any byte pattern for the integer formats
    back = gpu.ctx.pcm_decode(gpu.dev(raw), bits, encoding, big).cpu().numpy()
    assert np.array_equal(back.view(np.uint32), _ref.raw_convert(raw, bits, encoding, big, False).view(np.uint32))


def test_linear_mode(gpu):
    """--linear (Params::mix = false; reference wmadd.cc:115-126, wmget.cc:110-152): per-frame up / down bands instead of the
    shuffled mix entries -- same kernels with another table."""
    x = noise(4242, 130 * 44100, 2)
    gpu.awm.set_params(mix=False)
    orc.set_params(mix=False)
    try:
        want = orc.add(None, x, 2, PAY2).reshape(-1, 2)
        w = gpu.ctx.add_watermark(None, PAY2, gpu.dev(x))
        got = w.cpu().numpy()
        assert rms(got, want) < RMS_TOL and np.abs(got - want).max() < 2e-6
        pats = gpu.ctx.get_watermark(None, w)
        ref = orc.get(None, got, 2)
        assert [pkey(p) for p in pats] == [pkey(p) for p in ref]
        assert sum(p["bits"] == PAY2 for p in pats) >= 3
    finally:
        gpu.awm.set_params()
        orc.set_params()
    # the two modes do not read each other's watermarks, and the tables are cached per mode
    assert not any(p["bits"] == PAY2 for p in gpu.ctx.get_watermark(None, w))
    w_mix = gpu.ctx.add_watermark(None, PAY2, gpu.dev(x))
    assert any(p["bits"] == PAY2 for p in gpu.ctx.get_watermark(None, w_mix))


def test_digital_silence_inside_the_stream(gpu):
    """Gaps of digital silence inside the material (found by the randomised cross-check tools/gpu_fuzz.py).  A window of zeros
    has an exactly zero spectrum in the reference (-96 dB per band), and so has a window whose only non-zero sample sits at
    position 0 (Hann weight 0): the sliding DFT of the refinement resets its bins / forces -96 dB there.  Windows that slide
    INTO a gap keep loud samples only at their low-weight edge: the first transform of a row is done in double, else its
    rounding (1e-7 of the unwindowed content) shows there (2.4e-4 in a sync quality on this material with a float transform)."""
    rng = np.random.default_rng(77)
    for ch, marked in ((2, False), (1, True), (2, True)):
        x = noise(500 + ch, 66 * 44100, ch)
        if marked:
            x = orc.add(None, x, ch, PAY1).reshape(-1, ch)
        for _ in range(60):                                    # gaps in all channels ...
            a = int(rng.integers(0, len(x) - 50000))
            x[a:a + int(rng.integers(1100, 45000))] = 0
        for _ in range(20):                                    # ... and in one channel only
            a = int(rng.integers(0, len(x) - 50000))
            x[a:a + int(rng.integers(1100, 45000)), ch - 1] = 0
        got = gpu.ctx.get_watermark(None, gpu.dev(x))
        want = orc.get(None, x, ch)
        assert [pkey(p) for p in got] == [pkey(p) for p in want]
        assert max([abs(g["sync_quality"] - w["sync_quality"]) for g, w in zip(got, want)] + [0]) < QUALITY_TOL


def test_one_clean_gap_is_exact(gpu):
    """one long gap: full windows of silence, windows entering and leaving it"""
    x = noise(503, 66 * 44100, 2)
    x[20 * 44100:24 * 44100] = 0
    got = gpu.ctx.get_watermark(None, gpu.dev(x))
    want = orc.get(None, x, 2)
    assert [pkey(p) for p in got] == [pkey(p) for p in want]
    assert max([abs(g["sync_quality"] - w["sync_quality"]) for g, w in zip(got, want)] + [0]) < QUALITY_TOL


@pytest.mark.parametrize("minutes,limiter", [(61, True), (61, False), (95, True), (0.9, True)])
def test_add_get_as_one_call_equals_the_two_calls(gpu, minutes, limiter):
    """awm_add_get_watermark_d (add, then get of its output, one call: `get` starts a chunk behind the limiter pass that covers it, on
    other streams than the add): the PCM of awm_add_watermark_d bit for bit and the pattern list of awm_get_watermark_d on it, incl.
    quality and error values -- three chunks, four chunks (more chunks than a first round of lanes), a clip (one chunk: no hand-over),
    with and without the limiter; repeated, with the output buffer cleared in between (a chunk that started too early would read
    zeros or unlimited samples and change the list)."""
    torch = gpu.torch
    n = int(minutes * 60 * 44100) + 333
    g = torch.Generator(device="cuda"); g.manual_seed(17)
    x = torch.rand((n, 2), generator=g, device="cuda") * 2 - 1
    full = lambda p: pkey(p) + (p["sync_quality"], p["decode_error"])
    gpu.awm.set_params(test_no_limiter=not limiter)
    try:
        want_pcm = gpu.ctx.add_watermark(None, PAY1, x)
        want = [full(p) for p in gpu.ctx.get_watermark(None, want_pcm)]
        assert any(p[4] == PAY1 for p in want)
        out = torch.empty_like(x)
        for _ in range(4):
            out.zero_()
            got = [full(p) for p in gpu.ctx.add_get_watermark(None, PAY1, x, out)]
            assert got == want
            assert torch.equal(out, want_pcm)
        # the separate calls afterwards are unaffected (the marks are disarmed when the call returns)
        out.zero_()
        gpu.ctx.add_watermark(None, PAY1, x, out=out)
        assert [full(p) for p in gpu.ctx.get_watermark(None, out)] == want
    finally:
        gpu.awm.set_params()


def test_refinement_kernel_forms(gpu):
    """K4s (the refinement's sliding DFT, SyncFinder::search_refine -> sync_fft, reference syncfinder.cc:393-458, 560-605) exists in four
    forms (awm_debug_set_refine_form): 0 and 3 are the kernels of rounds 2 - 5, 4 (the default) the restructured step of round 6 with the
    SAME arithmetic -- its output must equal theirs to the last bit, on noise, on material with gaps of digital silence (the scalar
    zero-window / reset rules), at 65 and at fewer offsets, and through the whole `get` (every quality a double equal).
    Form 5 keeps the recurrence's state in double but accumulates the update term in float.  It is NOT the default and this test says why:
    on stationary noise its dB values are within 1e-2 of form 4's (mean 1e-6) and the detector's positions and qualities agree, but its
    error is relative to the UNWINDOWED content that slid through the window -- where a window slides into a gap of digital silence
    (loud samples leave, little is left) the dB values are off by whole dB (measured: 4.1), where form 4 and the reference are exact."""
    t, lib = gpu.torch, gpu.awm.lib
    rng = np.random.default_rng(4242)
    x = noise(901, 40 * 44100, 2)
    plain = gpu.dev(x)
    x[5 * 44100:5 * 44100 + 30000] = 0                          # a gap in both channels, one in the right channel only, a lone sample
    x[11 * 44100:11 * 44100 + 9000, 1] = 0
    x[17 * 44100:17 * 44100 + 5000] = 0
    x[17 * 44100 + 2500, 0] = 0.25
    x *= np.linspace(1.0, 1e-3, len(x), dtype=np.float32)[:, None]     # 60 dB of level across the stream
    xd = gpu.dev(x)
    bases = np.concatenate([rng.integers(0, len(x) - 1024 - 8 * 65, 300), 5 * 44100 + np.arange(-1600, 31000, 997),
                            11 * 44100 + np.arange(-1200, 9500, 511), 17 * 44100 + np.arange(-1100, 5200, 333)]).astype(np.int64)
    try:
        outs = {}
        for count in (65, 64, 17, 1):
            for form in (3, 0, 4, 5):
                lib.awm_debug_set_refine_form(form)
                outs[form] = gpu.ctx.sync_db_sliding(xd, bases, count)
            assert t.equal(outs[3], outs[0]) and t.equal(outs[3], outs[4]), count
            assert (outs[4][:, :, :count] <= 0).any() and t.isfinite(outs[4]).all() and t.isfinite(outs[5]).all()
        gap_error = (outs[5] - outs[4]).abs().max().item()
        for form in (4, 5):
            lib.awm_debug_set_refine_form(form)
            outs[form] = gpu.ctx.sync_db_sliding(plain, bases, 65)
        d = (outs[5] - outs[4]).abs()
        assert d.max().item() < 1e-2 and d.mean().item() < 1e-5, (d.max().item(), d.mean().item())
        print("form 5 against form 4: max |d dB| on stationary noise %.3g (mean %.3g), next to gaps of digital silence %.3g"
              % (d.max().item(), d.mean().item(), gap_error))
        # mono streams take the generic kernel whatever the form
        lib.awm_debug_set_refine_form(4)
        m4 = gpu.ctx.sync_db_sliding(gpu.dev(x[:, :1].copy()), bases, 65)
        lib.awm_debug_set_refine_form(3)
        assert t.equal(m4, gpu.ctx.sync_db_sliding(gpu.dev(x[:, :1].copy()), bases, 65))
        # the whole detector: BLOCK mode (three minutes) and CLIP mode (30 s, padded, rows inside the padding are skipped), with a gap
        full = lambda p: pkey(p) + (p["sync_quality"], p["decode_error"])
        for seconds in (185, 30):
            w = orc.add(None, noise(902 + seconds, seconds * 44100, 2), 2, PAY1).reshape(-1, 2)
            for gap in (False, True):
                if gap:
                    w[len(w) // 2:len(w) // 2 + 20000] = 0
                wd = gpu.dev(w)
                res = {}
                for form in (3, 4, 5):
                    lib.awm_debug_set_refine_form(form)
                    res[form] = gpu.ctx.get_watermark(None, wd)
                assert [full(p) for p in res[4]] == [full(p) for p in res[3]] and len(res[4]) > 0
                if not gap:
                    assert [pkey(p) for p in res[5]] == [pkey(p) for p in res[4]]
                    assert max(abs(a["sync_quality"] - b["sync_quality"]) for a, b in zip(res[5], res[4])) < QUALITY_TOL
    finally:
        lib.awm_debug_set_refine_form(4)


@pytest.mark.parametrize("kind", ["harmonic", "bursts", "clipped", "dc_offset"])
def test_add_and_get_on_material_that_is_not_noise(gpu, kind):
    """The material of tools/ref_backend_census.py's round 6 kinds, three minutes each, 16 bit: a sparse spectrum (harmonic stacks on slow
    chirps), speech-like bursts with DIGITAL SILENCE between them, hard clipped full scale noise (the limiter at work in every block), a DC
    offset with noise at -60 dB -- where `mag > 1e-7` (wmadd.cc:64-84; here abs2 > 1e-14f with v_log / v_exp for powf (hypotf)), exactly zero
    power (wmcommon.hh:207-214) and `umag == 0 || dmag == 0` (syncfinder.cc:101) fire in bulk.  `add` against the reference: PCM RMS < 1e-6,
    max < 4e-6; `get` of the reference's output: same positions, types and payloads, qualities within the tolerance the two builds of
    the reference keep between themselves on this material (profiles/r06/ref_backend_census_other.json: 1.2e-4 on the harmonic stacks,
    where the float rounding of the reference's own window shows against the little power between the partials)."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    import ref_backend_census as rbc
    n = 180 * 44100
    x = rbc.quantise16(rbc.material(kind, 0, n))
    ref = orc.add(None, x, 2, PAY1).reshape(-1, 2)
    got = gpu.ctx.add_watermark(None, PAY1, gpu.dev(x)).cpu().numpy()
    d = got.astype(np.float64) - ref
    assert np.sqrt((d ** 2).mean()) < RMS_TOL and np.abs(d).max() < 4e-6, (np.sqrt((d ** 2).mean()), np.abs(d).max())
    w = rbc.quantise16(ref)
    want = orc.get(None, w, 2)
    have = gpu.ctx.get_watermark(None, gpu.dev(w))
    assert [pkey(p) for p in have] == [pkey(p) for p in want] and len(want) > 0
    tol = 3e-4 if kind in ("harmonic", "dc_offset") else QUALITY_TOL
    assert max(abs(a["sync_quality"] - b["sync_quality"]) for a, b in zip(have, want)) < tol
