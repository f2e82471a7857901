"""Reference-vs-reference calibration of the refinement ties (CPU only, no GPU, no product code).

`search_refine` (reference syncfinder.cc:393-458) keeps the first fine offset that is STRICTLY better (`>` at :441); around a block
start the refined quality is flat to ~1e-6 over neighbouring 8-sample offsets.  How often do two builds of the UNMODIFIED reference
that differ only in the FFT library behind fftw3.h pick different offsets on byte-identical input?

  backend "double": oracle/_ref/libawm_ref.so      (oracle/ref_shim/fftw_shim.cc: double-precision FFT rounded to float once)
  backend "mkl":    oracle/_ref/libawm_ref_mkl.so  (MKL's single-precision FFTW3 wrapper; `make -C oracle ref_mkl`)

Material, every piece 30 minutes of stereo 44.1 kHz, watermarked by the reference's own `add` (double build) and quantised to 16 bit
(truncation towards zero, what a 16 bit file holds):
   * "testgen_8h": the 8 h `test-gen-noise` stream of BASELINE configs[3] (zero key), watermarked in one piece, searched in 16 pieces,
   * white noise at full scale, pink noise (1 / f power), noise low-passed at 3 kHz, white noise at -40 dB: independent pieces from
     numpy's PCG64 with a seed per piece (so tools/gpu_census_three_way.py regenerates the same bytes on the GPU box).
Both backends run `SyncFinder::search` (BLOCK mode) on every piece; every sync score pair is compared.

  python tools/ref_backend_census.py [pieces per synthetic kind = 8] [workers = 6] [skip_8h = 0] [first piece = 0] [kinds = noise] [dir = r05]
      -> profiles/<dir>/ref_backend_census.json  (summary + every differing position + the complete score lists of both backends)

Round 6 adds material that is NOT stationary noise (kinds = other): where `mag > 1e-7` (wmadd.cc:64-84), `umag == 0 || dmag == 0`
(syncfinder.cc:101) and exactly zero power (wmcommon.hh:207-214) fire in bulk and the limiter works in every block:
   * "harmonic": stacks of 30 harmonics (1 / h) on slow chirps between 110 and 440 Hz, another chirp per channel: a sparse spectrum,
   * "bursts": speech-like bursts of low-passed noise (0.2 - 1.5 s, raised-cosine edges, a level per burst) with DIGITAL SILENCE between them,
   * "clipped": white noise three times over full scale, hard clipped: two thirds of the samples sit on the rails, the limiter never rests,
   * "dc_offset": a constant 0.25 with white noise at -60 dB on it.
      -> profiles/r06/ref_backend_census_other.json, ref_backend_census_scores_other.json
"""
import concurrent.futures
import hashlib
import json
import multiprocessing
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

PAY = "0123456789abcdef0011223344556677"
RATE = 44100
PIECE = 30 * 60 * RATE                       # frames per piece
KINDS = ("white", "pink", "lowpass_3k", "white_minus_40dB")
KINDS_OTHER = ("harmonic", "bursts", "clipped", "dc_offset")
ALL_KINDS = KINDS + KINDS_OTHER
SHM = "/dev/shm/awm_census_8h.i16"


def quantise16(x):
    return (np.clip(np.trunc(x.astype(np.float64) * 32768.0), -32768, 32767) / 32768.0).astype(np.float32)


def shaped(x, amp_of_f):
    """noise with the amplitude response amp_of_f (f in Hz), in blocks of 2^22 frames in the frequency domain"""
    B = 1 << 22
    out = np.empty_like(x)
    for a in range(0, x.shape[0], B):
        blk = x[a:a + B]
        spec = np.fft.rfft(blk.astype(np.float64), axis=0)
        f = np.fft.rfftfreq(blk.shape[0], 1.0 / RATE)
        out[a:a + B] = np.fft.irfft(spec * amp_of_f(f)[:, None], n=blk.shape[0], axis=0).astype(np.float32)
    return out * np.float32(0.5 / float(np.abs(out).max()))


def harmonic(rng, n):
    """two channels of harmonic stacks on slow chirps (float64 phase carried from block to block; sin (h theta) by the Chebyshev recurrence)"""
    out = np.empty((n, 2), np.float32)
    B = 1 << 22
    for c in range(2):
        period = 97.0 + 16.0 * c + float(rng.random()) * 5.0
        phi = float(rng.random()) * 6.28
        phase0 = 0.0
        for a in range(0, n, B):
            t = np.arange(a, min(n, a + B), dtype=np.float64) / RATE
            f0 = 110.0 * 2.0 ** (1.0 + np.sin(2 * np.pi * t / period + phi))                          # 110 .. 440 Hz
            theta = phase0 + 2 * np.pi * np.cumsum(f0) / RATE
            phase0 = float(theta[-1]) % (2 * np.pi)
            c1, s_prev, s_cur = 2.0 * np.cos(theta), np.zeros(len(t)), np.sin(theta)
            acc = s_cur.copy()
            for h in range(2, 31):
                s_prev, s_cur = s_cur, c1 * s_cur - s_prev
                acc += s_cur * (np.where(h * f0 < 20000.0, 1.0, 0.0) / h)
            out[a:a + B, c] = (acc * 0.28).astype(np.float32)                                         # (sum of 1 / h sines stays below 1.8)
    return out


def bursts(rng, n):
    """bursts of low-passed noise with digital silence between them, the same envelope in both channels"""
    x = rng.random((n, 2), dtype=np.float32) * np.float32(2) - np.float32(1)
    x = shaped(x, lambda f: 1.0 / (1.0 + (f / 1500.0) ** 2))
    env = np.zeros(n, np.float32)
    pos = int(0.3 * RATE)
    edge = int(0.010 * RATE)
    ramp = (0.5 - 0.5 * np.cos(np.pi * np.arange(edge) / edge)).astype(np.float32)
    while pos < n:
        length = int((0.2 + 1.3 * float(rng.random())) * RATE)
        level = np.float32(0.1 + 0.8 * float(rng.random()))
        a, b = pos, min(n, pos + length)
        seg = np.full(b - a, level, np.float32)
        k = min(edge, (b - a) // 2)
        seg[:k] *= ramp[:k]
        seg[len(seg) - k:] *= ramp[:k][::-1]
        env[a:b] = seg
        pos = b + int((0.1 + 0.7 * float(rng.random())) * RATE)
    return (x * (np.float32(2.0) * env)[:, None]).astype(np.float32)


def material(kind, piece, n=PIECE):
    """deterministic: numpy PCG64 seeded by (kind, piece); the same bytes wherever numpy 2.x runs"""
    rng = np.random.Generator(np.random.PCG64(1000 * (ALL_KINDS.index(kind) + 1) + piece))
    if kind == "harmonic":
        return harmonic(rng, n)
    if kind == "bursts":
        return bursts(rng, n)
    x = rng.random((n, 2), dtype=np.float32) * np.float32(2) - np.float32(1)
    if kind == "clipped":
        return np.clip(x * np.float32(3), np.float32(-1), np.float32(1))
    if kind == "dc_offset":
        return np.float32(0.25) + x * np.float32(0.001)
    if kind == "white":
        return x
    if kind == "white_minus_40dB":
        return x * np.float32(0.01)
    if kind == "pink":
        return shaped(x, lambda f: 1.0 / np.sqrt(np.maximum(f, 20.0)))
    return shaped(x, lambda f: 1.0 / (1.0 + (f / 3000.0) ** 8))


def watermarked_piece(kind, piece):
    """the 16 bit samples both detectors read, as float32 [frames][2]"""
    import _ref
    if kind == "testgen_8h":
        a = np.memmap(SHM, np.int16, "r")
        return (a[piece * PIECE * 2:(piece + 1) * PIECE * 2].astype(np.float32) / np.float32(32768.0)).reshape(-1, 2)
    x = quantise16(material(kind, piece))
    _ref.use_backend("double")
    return quantise16(_ref.add(None, x.ravel(), 2, PAY)).reshape(-1, 2)


def work(item):
    kind, piece = item
    import _ref
    t0 = time.perf_counter()
    w = watermarked_piece(kind, piece).ravel()
    t_gen = time.perf_counter() - t0
    # md5 of the 16 bit samples: tools/gpu_census_three_way.py regenerates the piece on the GPU box and compares the HIP detector with
    # BOTH lists below only if it holds the same bytes
    import hashlib
    md5 = hashlib.md5(np.round(w.astype(np.float64) * 32768.0).astype(np.int16).tobytes()).hexdigest()
    res = {"kind": kind, "piece": piece, "md5_int16": md5}
    for backend in ("double", "mkl"):
        _ref.use_backend(backend)
        t0 = time.perf_counter()
        idx, q, bt = _ref.sync_search(None, w, 2)
        res[backend] = {"index": [int(i) for i in idx], "quality": [float(v) for v in q], "block_type": [int(b) for b in bt],
                        "search_s": round(time.perf_counter() - t0, 2)}
    res["prepare_s"] = round(t_gen, 2)
    return res


def compare(res, rec):
    a, b = res["double"], res["mkl"]
    rec["pieces"] += 1
    if len(a["index"]) != len(b["index"]):
        rec["other_differences"].append({"piece": res["piece"], "what": "score count", "double": len(a["index"]), "mkl": len(b["index"])})
        return
    for ia, qa, ta, ib, qb, tb in zip(a["index"], a["quality"], a["block_type"], b["index"], b["quality"], b["block_type"]):
        rec["scores"] += 1
        real = min(qa, qb) >= 0.5                      # a watermark block (quality 1.2 - 1.5 on this material); fillers of n_best sit at 0.1 - 0.3
        rec["blocks"] += bool(real)
        dq = abs(qa - qb)
        rec["max_abs_quality_diff"] = max(rec["max_abs_quality_diff"], dq)
        if ia == ib and ta == tb:
            rec["same_position"] += 1
            continue
        d = {"piece": res["piece"], "double": ia, "mkl": ib, "quality_double": qa, "quality_mkl": qb, "quality_gap": dq, "block_type": [ta, tb],
             "watermark_block": bool(real)}
        if ta == tb and abs(ia - ib) <= 16 and dq < 1e-5:
            rec["ties"].append(d)
            rec["ties_on_blocks"] += bool(real)
        else:
            rec["other_differences"].append(d)


def main():
    pieces = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    skip_8h = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    first_piece = int(sys.argv[4]) if len(sys.argv) > 4 else 0           # a second batch: other seeds, results into *_from<N>.json
    which = sys.argv[5] if len(sys.argv) > 5 else "noise"                # "noise": round 5's four kinds | "other": round 6's
    out_dir = sys.argv[6] if len(sys.argv) > 6 else "r05"
    kinds = KINDS if which == "noise" else KINDS_OTHER
    os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
    os.environ["AWM_REF_THREADS"] = "1"
    import _ref
    assert os.path.exists(_ref.PATH) and os.path.exists(_ref.PATH_MKL), "make -C oracle ref ref_mkl"
    items = []
    timing = {}
    if not skip_8h:
        t0 = time.perf_counter()
        n = 8 * 3600 * RATE
        x = quantise16(_ref.gen_noise(None, 2 * n))
        w = _ref.add(None, x, 2, PAY)
        del x
        np.clip(np.trunc(w.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16).tofile(SHM)
        del w
        timing["testgen_8h_generate_and_reference_add_s"] = round(time.perf_counter() - t0, 1)
        print("8 h stream ready", timing, flush=True)
        items += [("testgen_8h", p) for p in range(16)]
    items += [(k, p) for p in range(first_piece, first_piece + pieces) for k in kinds]
    suffix = ("_from%d" % first_piece if first_piece else "") + ("" if which == "noise" else "_" + which)
    out = {}
    lists = []
    t0 = time.perf_counter()
    with concurrent.futures.ProcessPoolExecutor(max_workers=workers, mp_context=multiprocessing.get_context("spawn")) as pool:
        for res in pool.map(work, items):
            rec = out.setdefault(res["kind"], {"pieces": 0, "scores": 0, "blocks": 0, "same_position": 0, "ties": [], "ties_on_blocks": 0,
                                               "other_differences": [], "max_abs_quality_diff": 0.0})
            compare(res, rec)
            lists.append(res)
            print(res["kind"], res["piece"], "scores", len(res["double"]["index"]), "ties so far", len(rec["ties"]),
                  res["double"]["search_s"], res["mkl"]["search_s"], flush=True)
    timing["wall_s"] = round(time.perf_counter() - t0, 1)
    timing["search_cpu_s_double"] = round(sum(r["double"]["search_s"] for r in lists), 1)
    timing["search_cpu_s_mkl"] = round(sum(r["mkl"]["search_s"] for r in lists), 1)
    if os.path.exists(SHM):
        os.remove(SHM)
    blocks = sum(v["blocks"] for v in out.values())
    scores = sum(v["scores"] for v in out.values())
    ties = sum(len(v["ties"]) for v in out.values())
    ties_b = sum(v["ties_on_blocks"] for v in out.values())
    summary = {"sync_scores_compared": scores, "watermark_blocks_compared": blocks, "scores_at_another_fine_offset": ties,
               "watermark_blocks_at_another_fine_offset": ties_b,
               "ties_per_1000_watermark_blocks": round(1000.0 * ties_b / max(1, blocks), 2),
               "other_differences": sum(len(v["other_differences"]) for v in out.values()),
               "max_abs_quality_diff": max(v["max_abs_quality_diff"] for v in out.values()),
               "note": "two builds of the UNMODIFIED reference on byte-identical 16 bit input; they differ only in the FFT behind fftw3.h "
                       "(double-precision FFT rounded once vs MKL's float FFTW wrapper); a tie = same block type, sync index <= 16 samples "
                       "apart, qualities < 1e-5 apart"}
    os.makedirs(os.path.join(ROOT, "profiles", out_dir), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", out_dir, "ref_backend_census%s.json" % suffix), "w") as f:
        json.dump({"summary": summary, "materials": out, "timing": timing, "piece_minutes": 30, "payload": PAY}, f, indent=1)
    with open(os.path.join(ROOT, "profiles", out_dir, "ref_backend_census_scores%s.json" % suffix), "w") as f:
        json.dump(lists, f, separators=(",", ":"))
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
