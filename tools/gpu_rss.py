import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the library leaves the environment alone)
"""Peak resident set size of the command line binary for `add` and `get` on s16 raw files of several lengths
(bounded-memory check of the streamed paths).  usage: python tools/gpu_rss.py [minutes ...]"""
import os
import resource
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "audiowmark_amd", "audiowmark")
PAY = "0123456789abcdef0011223344556677"
FMT = ["--format", "raw", "--raw-rate", "44100", "--raw-channels", "2", "--raw-bits", "16"]


def run(cmd):
    """returns (seconds, peak RSS in MB of this child, returncode, stdout)"""
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out, err = p.communicate()
    _, status, ru = os.wait4(p.pid, os.WNOHANG) if False else (0, 0, None)
    return time.perf_counter() - t0, p.returncode, out, err


def child_rss(cmd):
    pid = os.fork()
    if pid == 0:
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        os.dup2(devnull, 2)
        os.execv(cmd[0], cmd)
    t0 = time.perf_counter()
    _, status, ru = os.wait4(pid, 0)
    return time.perf_counter() - t0, ru.ru_maxrss / 1024.0, os.WEXITSTATUS(status)


def main():
    minutes = [float(a) for a in sys.argv[1:]] or [1, 60]
    d = "/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp"
    rng = np.random.default_rng(1)
    print("baseline (audiowmark --version): %.3f s, %.0f MB" % child_rss([CLI, "--version"])[:2])
    for m in minutes:
        src, dst = os.path.join(d, "awm_rss_in.raw"), os.path.join(d, "awm_rss_out.raw")
        n = int(m * 60 * 44100) * 2
        with open(src, "wb") as f:
            left = n
            while left:
                k = min(left, 1 << 24)
                f.write(rng.integers(-30000, 30000, k, dtype=np.int16).tobytes())
                left -= k
        ta, ra, rca = child_rss([CLI, "add", "-q"] + FMT + [src, dst, PAY])
        tg, rg, rcg = child_rss([CLI, "get"] + FMT + [dst])
        print("%6.1f min (%7.1f MB file): add %.3f s, peak RSS %.0f MB (rc %d); get %.3f s, peak RSS %.0f MB (rc %d); %.0f xRT incl. process start"
              % (m, n * 2 / 1e6, ta, ra, rca, tg, rg, rcg, m * 60 / (ta + tg)))
        os.unlink(src)
        os.unlink(dst)


if __name__ == "__main__":
    main()
