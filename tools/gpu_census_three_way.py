import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
"""Three-way refinement-tie census on IDENTICAL bytes: this library's detector against BOTH builds of the unmodified reference whose sync
score lists tools/ref_backend_census.py recorded (profiles/r05/ref_backend_census_scores.json: double-precision FFT build and MKL float
FFT build, SyncFinder::search per 30-minute piece).

Every synthetic piece is regenerated here from its seed (numpy PCG64), watermarked by the reference's `add` (oracle/_ref, double build) and
quantised to 16 bit exactly as the census did; the md5 of the 16 bit samples must equal the recorded one, otherwise the piece is skipped
(and counted).  The 8 h `test-gen-noise` pieces are rebuilt the same way when asked for (one single-threaded reference `add` of 8 h:
~35 s of one host core + 20 GB of host memory).

  python tools/gpu_census_three_way.py [with_8h = 1] [workers = 12]    ->  gpurun_out/census_three_way.json  (copy to profiles/rNN/)
"""
import concurrent.futures
import hashlib
import json
import multiprocessing
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

import ref_backend_census as rbc


def prepare(item):
    """worker process (CPU): the piece's 16 bit samples as int16 + md5"""
    kind, piece = item
    os.environ["AWM_REF_THREADS"] = "1"
    w = rbc.watermarked_piece(kind, piece)
    i16 = np.round(w.astype(np.float64) * 32768.0).astype(np.int16)
    return kind, piece, hashlib.md5(i16.tobytes()).hexdigest(), i16


def compare(ours, theirs, rec, piece):
    """ours / theirs: (index list, quality list, type list)"""
    if len(ours[0]) != len(theirs["index"]):
        rec["other_differences"].append({"piece": piece, "what": "score count", "ours": len(ours[0]), "reference": len(theirs["index"])})
        return
    for io, qo, to, ir, qr, tr in zip(ours[0], ours[1], ours[2], theirs["index"], theirs["quality"], theirs["block_type"]):
        real = min(qo, qr) >= 0.5
        rec["scores"] += 1
        rec["blocks"] += bool(real)
        dq = abs(qo - qr)
        rec["max_abs_quality_diff"] = max(rec["max_abs_quality_diff"], dq)
        if io == ir and to == tr:
            continue
        d = {"piece": piece, "ours": int(io), "reference": int(ir), "quality_ours": float(qo), "quality_reference": float(qr), "quality_gap": dq,
             "block_type": [int(to), int(tr)], "watermark_block": bool(real)}
        if to == tr and abs(int(io) - int(ir)) <= 16 and dq < 1e-5:
            rec["ties"].append(d)
            rec["ties_on_blocks"] += bool(real)
        else:
            rec["other_differences"].append(d)


def main():
    with_8h = (int(sys.argv[1]) if len(sys.argv) > 1 else 1) != 0
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    import torch
    import audiowmark_amd as awm
    import _ref
    assert _ref.available(), "oracle/_ref is not built"
    import glob
    recorded = {}
    # (every batch of the census: round 5's noise kinds, round 6's harmonic / bursts / clipped / dc_offset; AWM_CENSUS_GLOB narrows it)
    pattern = os.environ.get("AWM_CENSUS_GLOB", "r0[56]/ref_backend_census_scores*.json")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        with open(path) as f:
            recorded.update({(r["kind"], r["piece"]): r for r in json.load(f)})
    items = [k for k in recorded if k[0] != "testgen_8h"]
    t_all = time.perf_counter()
    if with_8h and any(k[0] == "testgen_8h" for k in recorded):
        t0 = time.perf_counter()
        n = 8 * 3600 * rbc.RATE
        x = rbc.quantise16(awm.binding.gen_noise(None, 2 * n))      # (== the reference's test-gen-noise, tests/test_host_abi.py; its AES runs on all cores)
        w = _ref.add(None, x, 2, rbc.PAY)
        del x
        np.clip(np.trunc(w.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16).tofile(rbc.SHM)
        del w
        print("8 h stream rebuilt in %.0f s" % (time.perf_counter() - t0), flush=True)
        items = [k for k in recorded if k[0] == "testgen_8h"] + items
    ctx = awm.Context(0)
    form = int(os.environ.get("AWM_REFINE_FORM", "-1"))                 # (which form of K4s: default = the library's; 5 = update term in float)
    if form >= 0:
        awm.lib.awm_debug_set_refine_form(form)
    form = awm.lib.awm_debug_refine_form()
    pairs = {"hip_vs_double": {}, "hip_vs_mkl": {}, "double_vs_mkl": {}}
    new = lambda: {"scores": 0, "blocks": 0, "ties": [], "ties_on_blocks": 0, "other_differences": [], "max_abs_quality_diff": 0.0}
    skipped = []
    done = 0
    with concurrent.futures.ProcessPoolExecutor(max_workers=workers, mp_context=multiprocessing.get_context("spawn")) as pool:
        for kind, piece, md5, i16 in pool.map(prepare, sorted(items)):
            rec = recorded[(kind, piece)]
            if rec.get("md5_int16") != md5:
                skipped.append({"kind": kind, "piece": piece, "md5_here": md5, "md5_recorded": rec.get("md5_int16")})
                continue
            dev = (torch.from_numpy(i16).cuda().float() / 32768.0).reshape(-1, 2)
            gi, gq, gb = ctx.sync_search(None, dev)
            ours = ([int(v) for v in gi], [float(v) for v in gq], [int(v) for v in gb])
            compare(ours, rec["double"], pairs["hip_vs_double"].setdefault(kind, new()), piece)
            compare(ours, rec["mkl"], pairs["hip_vs_mkl"].setdefault(kind, new()), piece)
            compare((rec["double"]["index"], rec["double"]["quality"], rec["double"]["block_type"]), rec["mkl"],
                    pairs["double_vs_mkl"].setdefault(kind, new()), piece)
            done += 1
    if os.path.exists(rbc.SHM):
        os.remove(rbc.SHM)
    summary = {}
    for name, kinds in pairs.items():
        blocks = sum(v["blocks"] for v in kinds.values())
        ties_b = sum(v["ties_on_blocks"] for v in kinds.values())
        summary[name] = {"watermark_blocks_compared": blocks, "watermark_blocks_at_another_fine_offset": ties_b,
                         "ties_per_1000_watermark_blocks": round(1000.0 * ties_b / max(1, blocks), 2),
                         "scores_compared": sum(v["scores"] for v in kinds.values()), "scores_at_another_fine_offset": sum(len(v["ties"]) for v in kinds.values()),
                         "other_differences": sum(len(v["other_differences"]) for v in kinds.values()),
                         "max_abs_quality_diff": max((v["max_abs_quality_diff"] for v in kinds.values()), default=0.0)}
    res = {"summary": summary, "refine_form": form, "pieces_compared": done, "pieces_skipped_md5_mismatch": skipped, "pairs": pairs,
           "wall_s": round(time.perf_counter() - t_all, 1),
           "note": "identical 16 bit input for all three detectors (md5 checked per piece); a tie = same block type, sync index <= 16 samples apart, qualities "
                   "< 1e-5 apart; 'double' / 'mkl' = the two builds of the unmodified reference (oracle/Makefile: ref, ref_mkl)"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", os.environ.get("AWM_CENSUS_OUT", "census_three_way.json")), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({"refine_form": form, "summary": summary, "pieces_compared": done, "skipped": len(skipped)}))


if __name__ == "__main__":
    main()
