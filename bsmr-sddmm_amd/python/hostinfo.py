"""CPUs this process may actually use: the affinity mask capped by the control group's CPU
quota (a container that sees 256 CPUs but has a quota of 16 runs 256 OpenMP threads several
times slower than 16)."""
import math
import os


def effective_cpus() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = period = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]      # cgroup v2
        if q != "max":
            quota, period = int(q), int(p)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota and period and quota > 0:
        n = min(n, max(1, math.ceil(quota / period)))
    return max(1, n)


def limit_openmp_threads() -> int:
    """Sets OMP_NUM_THREADS (unless the caller already did) before any OpenMP runtime starts."""
    n = effective_cpus()
    os.environ.setdefault("OMP_NUM_THREADS", str(n))
    return int(os.environ["OMP_NUM_THREADS"])
