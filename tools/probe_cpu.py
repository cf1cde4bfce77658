"""Host probe: oracle sweep time vs OpenMP thread count on the GPU box's host (for cpu_baseline)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_c
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
M, N = 4096, 65536
rng = np.random.default_rng(0)
A = np.asfortranarray(rng.standard_normal((M, 8192), dtype=np.float32))
A = np.asfortranarray(np.tile(A, (1, N // 8192)))
r = rng.standard_normal(M)
for nt in (8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1):
        continue
    oracle_c.sweep_abs(A, r, nthreads=nt)
    t0 = time.perf_counter()
    for _ in range(3):
        oracle_c.sweep_abs(A, r, nthreads=nt)
    dt = (time.perf_counter() - t0) / 3
    print(f"threads={nt:4d}  sweep {dt*1e3:8.2f} ms  {M*N*4/dt/1e9:7.1f} GB/s")
