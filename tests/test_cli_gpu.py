"""End-to-end checks of the `audiowmark` command line front end (add / get / cmp on raw and WAV data) -- the
reference's own `make check` scenarios that fit the 44.1 kHz raw/WAV scope (tests/block-decoder-test.sh,
sync-test.sh, clip-decoder-test.sh, pipe-test.sh, key-test.sh, raw-format-test.sh), plus cross checks against
the compiled reference binary where oracle/_ref is available."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import _ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AWM = os.path.join(ROOT, "audiowmark_amd", "audiowmark")
PAY = "f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0"


def run(cmd, stdin=None, check=True):
    r = subprocess.run(cmd, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if check:
        assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-2000:])
    return r


# The ONE known refinement tie of this suite: in the 200 s fixture watermarked by THIS library's `add`, the
# block that the reference's detector puts at sync index 2535416 is found at 2535408 here -- neighbouring fine offsets (8 samples)
# whose qualities are 1.338650118 / 1.338649997, a difference below the two float pipelines' rounding noise; `search_refine` keeps
# the first strictly better offset (syncfinder.cc:441).  The block is then read 8 samples apart and its path metric differs in the
# third digit.  One tie shows in up to three report lines (the block, its AB pair, the "all" pattern).  Only the tests that read
# that fixture pass it; every other comparison is strict.
KNOWN_TIE = {"ours": 2535408, "reference": 2535416, "max_lines": 3}


def same_report(ours, theirs, what="", max_tie_lines=0):
    """Two `cmp` / `get` reports agree LINE BY LINE (strict by default).  With max_tie_lines > 0 (only for the fixture of KNOWN_TIE)
    the decode error column of that many pattern lines may differ by < 0.01 while time, payload, printed sync quality and type agree."""
    skip = ("key", "expect_matches")
    a = [l for l in ours if not l.startswith(skip)]
    b = [l for l in theirs if not l.startswith(skip)]
    assert len(a) == len(b), (what, a, b)
    ties = 0
    for x, y in zip(a, b):
        if x == y:
            continue
        assert max_tie_lines > 0, (what, x, y)
        fx, fy = x.split(), y.split()                   # pattern <time | all> <bits> <quality> <error> [<type>]
        assert fx[0] == fy[0] == "pattern" and len(fx) == len(fy) and len(fx) in (5, 6), (what, x, y)
        assert fx[:4] == fy[:4] and fx[5:] == fy[5:] and abs(float(fx[4]) - float(fy[4])) < 0.01, (what, x, y)
        ties += 1
    assert ties <= max_tie_lines, (what, ties, a, b)
    return ties


def wav_samples(path):
    with open(path, "rb") as f:
        data = f.read()
    pos = data.index(b"data") + 8
    return np.frombuffer(data[pos:pos + (len(data) - pos) // 2 * 2], dtype="<i2")


@pytest.fixture(scope="module")
def work(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    noise = d / "noise200.wav"
    r = run([AWM, "test-gen-noise", "-", "200", "44100"])
    assert hashlib.md5(r.stdout).hexdigest() == "a7079f22b4ddd9c1bd338b564b726aad"
    noise.write_bytes(r.stdout)
    marked = d / "marked200.wav"
    r = run([AWM, "add", "--format", "wav-pipe", str(noise), "-", PAY])
    marked.write_bytes(r.stdout)
    return d, noise, marked


def test_block_decoder_scenario(work):
    d, noise, marked = work
    assert os.path.getsize(noise) == os.path.getsize(marked)                             # check_length
    r = run([AWM, "cmp", "--input-format", "wav-pipe", str(marked), PAY, "--expect-matches", "5"])
    out = r.stdout.decode()
    assert "match_count 5 10" in out and "sync_match 3 8" in out
    lines = [l for l in out.splitlines() if l.startswith("pattern")]
    assert lines[0].split()[1:] == ["0:05", PAY, "1.383", "0.139", "A"]
    # SNR without limiter >= 32.4 dB (tests/block-decoder-test.sh:18)
    nolim = d / "nolim.wav"
    nolim.write_bytes(run([AWM, "add", "--format", "wav-pipe", "--test-no-limiter", str(noise), "-", PAY]).stdout)
    snr = float(run([AWM, "test-snr", str(noise), str(nolim)]).stdout.split()[1])
    assert snr >= 32.4


def test_json_and_get(work):
    d, _, marked = work
    js = d / "out.json"
    r = run([AWM, "get", "--input-format", "wav-pipe", "--json", str(js), str(marked)])
    import json
    j = json.loads(js.read_text())
    assert j["length"] == "3:20" and len(j["matches"]) == 10
    assert j["matches"][0]["bits"] == PAY and j["matches"][0]["type"] == "A"


def test_pipe_raw_and_wrong_key(work):
    d, noise, marked = work
    # stdin -> stdout (tests/pipe-test.sh) with raw s16le on both sides (tests/raw-format-test.sh)
    pcm = wav_samples(str(noise)).tobytes()
    raw = ["--format", "raw", "--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16"]
    out = run([AWM, "add", "--test-key", "3"] + raw + ["-", "-", PAY], stdin=pcm).stdout
    assert len(out) == len(pcm)
    r = run([AWM, "cmp", "--test-key", "3", "--input-format", "raw", "--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16",
             "-", PAY, "--expect-matches", "5"], stdin=out)
    assert b"match_count 5" in r.stdout
    # wrong key -> no match, exit code 1 (tests/key-test.sh)
    r = run([AWM, "cmp", "--test-key", "4", "--input-format", "raw", "--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16",
             "-", PAY], stdin=out, check=False)
    assert r.returncode == 1 and b"match_count 0" in r.stdout


def test_sync_and_clip_scenarios(work):
    d, _, marked = work
    s = wav_samples(str(marked)).reshape(-1, 2)
    raw = ["--input-format", "raw", "--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16"]
    # tests/sync-test.sh: cut 441150 frames from the start -> 3 matches
    r = run([AWM, "cmp"] + raw + ["-", PAY, "--expect-matches", "3", "--test-cut", "441150"], stdin=s[441150:].tobytes())
    assert b"match_count 3" in r.stdout
    # tests/clip-decoder-test.sh: a 30 s clip -> 1 match, also after cutting a few more samples
    clip = s[:30 * 44100]
    r = run([AWM, "cmp"] + raw + ["-", PAY, "--expect-matches", "1"], stdin=clip.tobytes())
    assert b"CLIP" in r.stdout
    run([AWM, "cmp"] + raw + ["-", PAY, "--expect-matches", "1"], stdin=clip[22150:].tobytes())


@pytest.mark.skipif(not os.path.exists(_ref.BIN), reason="oracle/_ref/audiowmark_ref not built")
def test_against_reference_binary(work):
    d, noise, marked = work
    ref_marked = d / "ref_marked.wav"
    ref_marked.write_bytes(run([_ref.BIN, "add", "--format", "wav-pipe", str(noise), "-", PAY]).stdout)
    a, b = wav_samples(str(marked)).astype(np.int32), wav_samples(str(ref_marked)).astype(np.int32)
    assert a.shape == b.shape
    diff = np.abs(a - b)
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3          # quantisation-boundary flips only
    # both detectors agree on both files, line by line
    for f, tie_lines in ((marked, KNOWN_TIE["max_lines"]), (ref_marked, 0)):
        ours = run([AWM, "cmp", "--input-format", "wav-pipe", str(f), PAY]).stdout.decode().splitlines()
        theirs = run([_ref.BIN, "cmp", "--x-in-wav-pipe", str(f), PAY]).stdout.decode().splitlines()
        same_report(ours, theirs, str(f), max_tie_lines=tie_lines)


@pytest.mark.skipif(not os.path.exists(_ref.BIN), reason="oracle/_ref/audiowmark_ref not built")
def test_frames_per_bit_option_against_reference_binary(tmp_path):
    """--frames-per-bit (reference audiowmark.cc:675: a block of 510 + 858 x N frames) through both command lines: 90 s of test-gen-noise
    watermarked with N = 3 by either binary differ by quantisation-boundary flips only, both detectors print the same report for both files,
    and neither finds the payload without the option (the blocks do not line up)."""
    noise = tmp_path / "noise.wav"
    noise.write_bytes(run([AWM, "test-gen-noise", "-", "90", "44100"]).stdout)
    ours_marked, ref_marked = tmp_path / "ours.wav", tmp_path / "ref.wav"
    ours_marked.write_bytes(run([AWM, "add", "--frames-per-bit", "3", "--format", "wav-pipe", str(noise), "-", PAY]).stdout)
    ref_marked.write_bytes(run([_ref.BIN, "add", "--frames-per-bit", "3", "--format", "wav-pipe", str(noise), "-", PAY]).stdout)
    a, b = wav_samples(str(ours_marked)).astype(np.int32), wav_samples(str(ref_marked)).astype(np.int32)
    assert a.shape == b.shape
    diff = np.abs(a - b)
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
    for f in (ours_marked, ref_marked):
        ours = run([AWM, "get", "--frames-per-bit", "3", "--input-format", "wav-pipe", str(f)]).stdout.decode().splitlines()
        theirs = run([_ref.BIN, "get", "--frames-per-bit", "3", "--x-in-wav-pipe", str(f)]).stdout.decode().splitlines()
        same_report(ours, theirs, str(f))
        assert any(PAY in l for l in ours)
    plain = run([AWM, "get", "--input-format", "wav-pipe", str(ours_marked)]).stdout.decode()
    assert PAY not in plain


@pytest.mark.skipif(not _ref.available(), reason="oracle/_ref (compiled reference) not built")
def test_the_known_refinement_tie_is_exactly_that_one(work):
    """The fixture's tie, pinned by its sync indices: on the file watermarked by this library every pattern of the compiled
    reference's detector is found at the SAME sync index here, except the block(s) of KNOWN_TIE -- and if a later change of the
    arithmetic removes the tie, this test says so (then KNOWN_TIE and the three max_tie_lines arguments go)."""
    import torch
    import audiowmark_amd as awm
    d, _, marked = work
    x = (wav_samples(str(marked)).astype(np.float32) / 32768.0).reshape(-1, 2)
    want = _ref.get(None, x, 2)
    got = awm.Context(0).get_watermark(None, torch.from_numpy(x).cuda())
    assert len(got) == len(want)
    moved = [(g["sync_index"], w["sync_index"]) for g, w in zip(got, want) if g["sync_index"] != w["sync_index"]]
    assert 0 < len(moved) <= KNOWN_TIE["max_lines"] and set(moved) == {(KNOWN_TIE["ours"], KNOWN_TIE["reference"])}, moved
    for g, w in zip(got, want):
        assert (g["type"], g["block_type"], g["bits"]) == (w["type"], w["block_type"], w["bits"])
        assert abs(g["sync_quality"] - w["sync_quality"]) < 1e-5


def test_snr_report_equals_reference_with_the_limiter_active(work, tmp_path):
    """`add --snr` (reference wmadd.cc:553-563, 591-592): the watermark is measured BEFORE the limiter.  test-gen-noise is full-scale
    noise, so the limiter is at work on every block; the reported figure equals the reference binary's and the one of a run without
    the limiter (a measurement of output - input would come out several dB lower).  Also at 48 kHz (WatermarkResampler path)."""
    d, noise, marked = work
    def snr_of(cmd):
        r = run(cmd)
        lines = [l for l in (r.stderr + b"\n" + r.stdout[:4096]).decode(errors="replace").splitlines() if l.startswith("SNR:")]
        assert len(lines) == 1, (cmd, r.stdout[-500:], r.stderr[-500:])
        return float(lines[0].split()[1])
    out = str(tmp_path / "o.wav")
    ours = snr_of([AWM, "add", "--snr", str(noise), out, PAY])
    nolim = snr_of([AWM, "add", "--snr", "--test-no-limiter", str(noise), out, PAY])
    assert abs(ours - nolim) < 1e-6 and ours >= 32.3         # (the float mix; test-snr on the 16 bit files: 32.43, test_block_decoder_scenario)
    # (the limiter does change the output: that difference is what the old report measured)
    a, b = wav_samples(str(marked)).astype(np.float64), wav_samples(str(noise)).astype(np.float64)
    assert 10 * np.log10((b ** 2).sum() / ((a - b) ** 2).sum()) < ours - 1
    if os.path.exists(_ref.BIN):
        theirs = snr_of([_ref.BIN, "add", "--snr", "--format", "wav-pipe", str(noise), "-", PAY])      # (the oracle build has no libsndfile)
        assert abs(ours - theirs) < 2e-3, (ours, theirs)
    noise48 = tmp_path / "noise48.wav"
    noise48.write_bytes(run([AWM, "test-gen-noise", "-", "60", "48000"]).stdout)
    s48 = snr_of([AWM, "add", "--snr", str(noise48), out, PAY])
    s48n = snr_of([AWM, "add", "--snr", "--test-no-limiter", str(noise48), out, PAY])
    assert abs(s48 - s48n) < 1e-6 and s48 > 30


def test_sample_rate_scenario(tmp_path):
    """tests/sample-rate-test.sh of the reference in miniature: 48 kHz and 96 kHz material through `add` and `cmp`
    (resampling to the 44.1 kHz watermark rate and back happens inside; zita-resampler restated, parity unpinned)."""
    for rate in (48000, 96000, 33333):                 # 33333 Hz: zita's fixed-ratio Resampler refuses, VResampler takes over
        noise = tmp_path / f"noise{rate}.wav"
        noise.write_bytes(run([AWM, "test-gen-noise", "-", "140", str(rate)]).stdout)
        marked = tmp_path / f"marked{rate}.wav"
        marked.write_bytes(run([AWM, "add", "--format", "wav-pipe", str(noise), "-", PAY]).stdout)
        assert os.path.getsize(noise) == os.path.getsize(marked)
        out = run([AWM, "cmp", "--input-format", "wav-pipe", str(marked), PAY]).stdout.decode()
        assert "match_count" in out and int(out.split("match_count")[1].split()[0]) >= 2
    # a ratio neither zita class takes (< 1 / 16) is refused with the reference's message (resample.cc:262)
    odd = tmp_path / "noise1M.wav"
    odd.write_bytes(run([AWM, "test-gen-noise", "-", "1", "1000000"]).stdout)
    r = run([AWM, "add", "--format", "wav-pipe", str(odd), "-", PAY], check=False)
    assert r.returncode != 0 and b"not implemented" in r.stderr


def test_detect_speed_scenario(work, tmp_path):
    """reference tests/detect-speed-test.sh: 30 s of noise, watermarked, replayed at another speed, found with --detect-speed"""
    noise = tmp_path / "n30.wav"
    marked = tmp_path / "m30.wav"
    run([AWM, "test-gen-noise", str(noise), "30", "44100"])
    run([AWM, "add", "--test-key", "1", str(noise), str(marked), PAY])
    for speed in ("0.9764", "1.0", "1.01"):
        spd = tmp_path / ("s%s.wav" % speed)
        run([AWM, "test-change-speed", str(marked), str(spd), speed])
        for opt in ("--detect-speed", "--detect-speed-patient"):
            r = run([AWM, "cmp", "--test-key", "1", str(spd), PAY, opt, "--test-speed", speed])
            out = r.stdout.decode()
            line = [l for l in out.splitlines() if l.startswith("detect_speed ")]
            assert len(line) == 1, out
            found, quality, delta = (float(v) for v in line[0].split()[1:])
            assert delta < 0.02 and quality > 1, line                       # percent
            assert "match_count" in out and not out.split("match_count")[1].strip().startswith("0 "), out
            if speed != "1.0":
                assert any(l.startswith("speed %.6f" % found) for l in out.splitlines()), out   # ResultSet::print (wmget.cc:397-404)
    # without speed detection the replayed file is not decodable
    r = run([AWM, "cmp", "--test-key", "1", str(tmp_path / "s0.9764.wav"), PAY], check=False)
    assert r.returncode != 0
    # --try-speed with the known speed finds it as well; only one speed option at a time
    run([AWM, "cmp", "--test-key", "1", str(tmp_path / "s0.9764.wav"), PAY, "--try-speed", "0.9764"])
    r = run([AWM, "get", "--test-key", "1", str(spd), "--detect-speed", "--try-speed", "1.01"], check=False)
    assert r.returncode != 0 and b"can only use one option" in r.stderr


def test_key_scenarios_and_two_keys_in_one_get(tmp_path):
    """tests/key-test.sh:13-37 of the reference: key files, wrong key, no key, double watermark with two keys -- and both keys
    in ONE `get` (syncfinder.cc:171-256 searches all keys over the same dB matrices), compared with the reference binary"""
    raw = ["--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16"]
    msg2 = "0123456789abcdef0123456789abcdef"
    noise = wav_samples_from_stdout(run([AWM, "test-gen-noise", "-", "30", "44100"]).stdout)
    k1, k2 = tmp_path / "key-test-1.key", tmp_path / "key-test-2.key"
    run([AWM, "gen-key", str(k1)])
    run([AWM, "gen-key", str(k2), "--name", "second"])
    assert k1.read_text().startswith("# watermarking key for audiowmark") and "name second" in k2.read_text()
    add = lambda keyopt, pcm, msg: run([AWM, "add", "-q", "--format", "raw"] + raw + keyopt + ["-", "-", msg], stdin=pcm).stdout
    cmp_ = lambda keyopt, pcm, msg, n: run([AWM, "cmp", "--input-format", "raw"] + raw + keyopt + ["-", msg, "--expect-matches", str(n)], stdin=pcm)
    out1 = add(["--key", str(k1)], noise, PAY)
    out2 = add(["--key", str(k2)], noise, msg2)
    cmp_(["--key", str(k1)], out1, PAY, 1)
    cmp_(["--key", str(k2)], out1, PAY, 0)                                   # shouldn't be able to detect without the correct key
    cmp_([], out1, PAY, 0)
    cmp_(["--key", str(k2)], out2, msg2, 1)
    cmp_(["--key", str(k1)], out2, msg2, 0)
    # double watermark with two different keys
    both = add(["--test-key", "42"], add([], noise, PAY), msg2)
    cmp_([], both, PAY, 1)
    cmp_(["--test-key", "42"], both, msg2, 1)
    # both keys in one run: a "key" line per named key, each with its own patterns
    r = run([AWM, "get", "--input-format", "raw"] + raw + ["--test-key", "42", "--test-key", "7", "-"], stdin=both)
    text = r.stdout.decode()
    assert "key test-key-42" in text and msg2 in text.split("key test-key-42")[1].split("key test-key-7")[0]
    assert "key test-key-7" in text and PAY not in text                      # the default key was not asked for
    if os.path.exists(_ref.BIN):
        theirs = run([_ref.BIN, "get", "--x-in-raw", "--test-key", "42", "--test-key", "7", "-"], stdin=both, check=False)
        if theirs.returncode == 0:
            assert text.splitlines() == theirs.stdout.decode().splitlines()


def wav_samples_from_stdout(data):
    pos = data.index(b"data") + 8
    return data[pos:]


def test_hard_decision_option(work):
    """--hard (wmget.cc:46-50): soft bits are replaced by 0 / 1 before the Viterbi decoder; same lines as the reference binary"""
    d, _, marked = work
    ours = run([AWM, "cmp", "--hard", "--input-format", "wav-pipe", str(marked), PAY, "--expect-matches", "5"]).stdout.decode().splitlines()
    assert any(l.startswith("pattern") and PAY in l for l in ours)
    if os.path.exists(_ref.BIN):
        theirs = run([_ref.BIN, "cmp", "--hard", "--x-in-wav-pipe", str(marked), PAY]).stdout.decode().splitlines()
        same_report(ours, theirs, "--hard on the fixture of KNOWN_TIE", max_tie_lines=KNOWN_TIE["max_lines"])


def test_rf64_output_and_riff_size_limit(work, tmp_path):
    """--output-format rf64 writes the ds64 chunk with 64 bit sizes (what libsndfile's SF_FORMAT_RF64 produces in the
    reference); a plain RIFF header that cannot hold the size is refused instead of being written with wrapped sizes"""
    import struct
    d, noise, _ = work
    out = tmp_path / "out.rf64"
    run([AWM, "add", "-q", "--output-format", "rf64", str(noise), str(out), PAY])
    data = out.read_bytes()
    pcm = os.path.getsize(noise) - 44
    assert data[:4] == b"RF64" and data[4:8] == b"\xff\xff\xff\xff" and data[8:16] == b"WAVEds64"
    riff_size, data_size, frames = struct.unpack("<QQQ", data[20:44])
    assert data_size == pcm and frames == pcm // 4 and riff_size == len(data) - 8
    assert data[72:76] == b"data" and data[76:80] == b"\xff\xff\xff\xff" and len(data) == 80 + pcm
    r = run([AWM, "cmp", str(out), PAY, "--expect-matches", "5"])            # and it reads its own RF64 back
    assert b"match_count 5" in r.stdout
    # a header that announces 5 GiB of samples: the RIFF output header cannot hold that
    big = tmp_path / "big.wav"
    n = 5 << 30
    big.write_bytes(b"RF64" + b"\xff" * 4 + b"WAVEds64" + struct.pack("<IQQQI", 28, n + 72, n, n // 4, 0) + b"fmt "
                    + struct.pack("<IHHIIHH", 16, 1, 2, 44100, 44100 * 4, 4, 16) + b"data" + b"\xff" * 4 + b"\0" * 4096)
    # (through a pipe: the length a REGULAR file announces is clamped to what the file holds, like libsndfile does)
    r = run([AWM, "add", "-q", "-", str(tmp_path / "o.wav"), PAY], stdin=big.read_bytes(), check=False)
    assert r.returncode == 1 and b"does not fit a RIFF header" in r.stderr
    # the same header on a regular file of 4 KiB: 1024 frames are what can be read, and what gets written
    run([AWM, "add", "-q", str(big), str(tmp_path / "o.wav"), PAY])
    assert os.path.getsize(tmp_path / "o.wav") == 44 + 4096


def test_malformed_wav_headers_are_rejected(tmp_path):
    import struct
    for ch, rate, align, what in [(0, 44100, 4, b"invalid fmt chunk"), (2, 0, 4, b"invalid fmt chunk"), (2, 44100, 3, b"inconsistent fmt chunk")]:
        f = tmp_path / "bad.wav"
        f.write_bytes(b"RIFF" + struct.pack("<I", 436) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * align, align, 16)
                      + b"data" + struct.pack("<I", 400) + b"\0" * 400)
        r = run([AWM, "test-snr", str(f), str(f)], check=False)
        assert r.returncode == 1 and what in r.stderr


def _ref_lines(args, stdin=None):
    r = run([_ref.BIN] + args, stdin=stdin)
    return [l.split() for l in r.stdout.decode().splitlines() if l.startswith("pattern")]


def test_hard_option_through_the_c_abi(work):
    """--hard without the command line: a context with its own parameter set (awm_ctx_set_params) next to one that follows the
    process-wide defaults, in one process; the patterns of both equal the reference binary's with / without --hard"""
    import ctypes as C
    import audiowmark_amd as awm
    d, _, marked = work
    soft_ctx, hard_ctx = awm.Context(0), awm.Context(0)
    hard_ctx.set_params(hard=1)
    assert hard_ctx.get_params().hard == 1 and soft_ctx.get_params().hard == 0
    hard = hard_ctx.get_watermark_file(None, str(marked))
    soft = soft_ctx.get_watermark_file(None, str(marked))
    again = hard_ctx.get_watermark_file(None, str(marked))                     # the other context's call changed nothing
    assert [p["bits"] for p in again] == [p["bits"] for p in hard]
    assert [p["decode_error"] for p in hard] != [p["decode_error"] for p in soft]
    assert sum(p["bits"] == PAY for p in hard) >= 5
    if os.path.exists(_ref.BIN):
        for ours, opt in ((hard, ["--hard"]), (soft, [])):
            theirs = _ref_lines(["get"] + opt + ["--x-in-wav-pipe", str(marked)])
            assert len(theirs) == len(ours)
            ties = 0
            for a, b in zip(ours, theirs):                                      # pattern time bits quality error type
                assert a["bits"] == b[2] and "%.3f" % a["sync_quality"] == b[3]
                if "%.3f" % a["decode_error"] != b[4]:                          # KNOWN_TIE: this is that fixture
                    assert abs(a["decode_error"] - float(b[4])) < 0.01
                    ties += 1
            assert ties <= KNOWN_TIE["max_lines"]
    hard_ctx.set_params()                                                       # back to the process-wide set
    assert [p["decode_error"] for p in hard_ctx.get_watermark_file(None, str(marked))] == [p["decode_error"] for p in soft]
    # a parameter the kernels are not built for is refused at the entry point, not silently ignored
    soft_ctx.set_params(frames_per_bit=9)
    with pytest.raises(awm.AwmError):
        soft_ctx.get_watermark_file(None, str(marked))


def test_two_keys_in_one_get_through_the_c_abi(tmp_path):
    """tests/key-test.sh:13-37 (double watermark with two different keys) through awm_get_watermark_keys_d / _file only: one pass
    over the material, every pattern names its key; equal to one call per key and to the reference binary with two --test-key"""
    import torch
    import audiowmark_amd as awm
    msg2 = "0123456789abcdef0123456789abcdef"
    ctx = awm.Context(0)
    k_default, k42, k7 = None, awm.test_key(42), awm.test_key(7)
    n = 30 * 44100
    x = torch.from_numpy((np.clip(np.trunc(awm.binding.gen_noise(None, 2 * n).astype(np.float64) * 32768), -32768, 32767) / 32768)
                         .astype(np.float32).reshape(n, 2)).cuda()
    both = ctx.add_watermark(k42, msg2, ctx.add_watermark(k_default, PAY, x))
    pats = ctx.get_watermark_keys([k42, k7, k_default], both)
    by_key = lambda i: [p for p in pats if p["key_index"] == i]
    assert any(p["bits"] == msg2 for p in by_key(0)) and not any(p["bits"] == PAY for p in by_key(0))
    assert any(p["bits"] == PAY for p in by_key(2)) and not any(p["bits"] == msg2 for p in by_key(2))
    assert not any(p["bits"] in (PAY, msg2) for p in by_key(1))                # key 7 marks nothing
    strip = lambda ps: [{k: v for k, v in p.items() if k != "key_index"} for p in ps]
    for i, k in enumerate([k42, k7, k_default]):
        assert strip(by_key(i)) == ctx.get_watermark(k, both)                  # same patterns as one call per key
    # the file level entry point on the same material (s16 raw) and the reference binary with the same key list
    raw = tmp_path / "both.raw"
    ctx.pcm_encode(both.reshape(-1), 16, 0, False, True).cpu().numpy().tofile(raw)
    rf = awm.binding.RawFormat(2, 44100, 16, 0, 0)
    from_file = ctx.get_watermark_keys_file([k42, k_default], str(raw), rf)
    assert any(p["bits"] == msg2 and p["key_index"] == 0 for p in from_file)
    assert any(p["bits"] == PAY and p["key_index"] == 1 for p in from_file)
    if os.path.exists(_ref.BIN):
        r = run([_ref.BIN, "get", "--x-in-raw", "--test-key", "42", "--test-key", "0", "-"], stdin=raw.read_bytes(), check=False)
        if r.returncode == 0:
            theirs = [l.split() for l in r.stdout.decode().splitlines() if l.startswith(("pattern", "key"))]
            ours, last = [], None
            for p in from_file:
                if p["key_index"] != last:
                    ours.append(["key", "test-key-%d" % (42, 0)[p["key_index"]]])
                    last = p["key_index"]
                ours.append(p)
            assert len(ours) == len(theirs)
            for a, b in zip(ours, theirs):
                if b[0] == "key":
                    assert a == b
                else:
                    assert a["bits"] == b[2] and "%.3f" % a["sync_quality"] == b[3]


def test_get_over_several_devices_equals_one(tmp_path):
    """AWM_DEVICES=a,b: the command line reads the file through the first GPU and spreads a long stream over all of them
    (awm_ctx_set_helpers -> awm_multi_get_d: the multi-GPU protocol with device-to-device copies).  On this one-GPU box the list names
    device 0 twice and three times -- same code path, same report as the plain `get`, line by line."""
    raw = ["--input-format", "raw", "--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16"]
    noise = wav_samples_from_stdout(run([AWM, "test-gen-noise", "-", "780", "44100"]).stdout)          # 13 min: > 4 blocks per GPU
    marked = tmp_path / "m.raw"
    marked.write_bytes(run([AWM, "add", "-q", "--format", "raw", "--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16",
                            "-", "-", PAY], stdin=noise).stdout)
    one = run([AWM, "cmp"] + raw + [str(marked), PAY]).stdout.decode().splitlines()
    assert sum(l.startswith("pattern") and PAY in l for l in one) >= 20
    for devs in ("0,0", "0,0,0"):
        env = dict(os.environ, AWM_DEVICES=devs)
        r = subprocess.run([AWM, "cmp"] + raw + [str(marked), PAY], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr
        assert r.stdout.decode().splitlines() == one, devs
    r = subprocess.run([AWM, "get"] + raw + [str(marked)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, AWM_DEVICES="0,x"))
    assert r.returncode != 0 and b"AWM_DEVICES" in r.stderr
