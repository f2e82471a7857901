import os, sys, ctypes as C, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import audiowmark_amd as awm
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "-")
f = awm.lib.awm_debug_time_group_key_tables; f.restype = C.c_double
for n, t in [(1, 1), (8, 8), (64, 8), (64, 16), (64, 32), (64, 64), (256, 64), (256, 128), (256, 256)]:
    f(n, t)
    print("keys %3d threads %3d: %.2f ms" % (n, t, min(f(n, t) for _ in range(4))))
