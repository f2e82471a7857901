#!/usr/bin/env python3
"""Static instruction counts per kernel of a .hip file (device ISA of gfx950), for before / after comparisons of a kernel change.
  python tools/isa_stats.py audiowmark_amd/csrc/hip/kernels.hip [name-filter] [--git REV]      (--git: the file as of that revision)"""
import collections, os, re, subprocess, sys, tempfile

def isa(path, src_dir):
    out = tempfile.mktemp(suffix=".s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-fno-slp-vectorize",
           "--cuda-device-only", "-S", "-I", src_dir, "-x", "hip", path, "-o", out]
    subprocess.run(cmd, check=True)
    text = open(out).read()
    os.unlink(out)
    return text

def stats(text, flt):
    res = {}
    name = None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            name = m.group(1)
            continue
        if name is None:
            continue
        t = line.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        d = res.setdefault(name, collections.Counter())
        d["total"] += 1
        if op.startswith("v_"): d["valu"] += 1
        if op.startswith("s_"): d["salu"] += 1
        if op.startswith("ds_"): d["lds"] += 1
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")): d["mem"] += 1
        if op.startswith("v_mov") or op.startswith("v_accvgpr"): d["mov"] += 1
        if "_f64" in op: d["f64"] += 1
        if op.startswith("v_pk_"): d["pk"] += 1
        if op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32") or op.startswith(("v_exp", "v_log")): d["trans"] += 1
        if op == "s_endpgm":
            name = None
    for k, d in sorted(res.items()):
        dem = subprocess.run(["c++filt", k], stdout=subprocess.PIPE, text=True).stdout.strip()
        if flt and flt not in dem:
            continue
        print(dem.split("(")[0], dict(d))

if __name__ == "__main__":
    args = sys.argv[1:]
    rev = None
    if "--git" in args:
        i = args.index("--git"); rev = args[i + 1]; del args[i:i + 2]
    path = os.path.abspath(args[0])
    flt = args[1] if len(args) > 1 else ""
    src_dir = os.path.dirname(path)
    if rev:
        rel = os.path.relpath(path, subprocess.run(["git", "rev-parse", "--show-toplevel"], stdout=subprocess.PIPE, text=True).stdout.strip())
        tmp = os.path.join(src_dir, "_isa_stats_tmp.hip")
        open(tmp, "w").write(subprocess.run(["git", "show", f"{rev}:{rel}"], stdout=subprocess.PIPE, text=True, check=True).stdout)
        path = tmp
    try:
        text = isa(path, src_dir)
    finally:
        if rev:
            os.unlink(path)
    stats(text, flt)
