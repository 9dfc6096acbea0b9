"""Test infrastructure (as everything under oracle/): the number of CPUs this process may actually use.

A container often sees every CPU of its host (os.cpu_count(), sched_getaffinity) while a cgroup CPU quota limits it to
a few CPUs' worth of time (the GPU boxes of this project: 256 logical CPUs visible, cpu.max = 16 CPUs).  Thread pools
sized by the visible count -- torch's default is the physical core count -- then spin in barriers while the quota
throttles the whole group: LAPACK calls and the oracle run many times slower, occasionally by minutes.  tests/conftest.py
and bench.py's cpu_baseline size their thread pools with usable_cpus()."""
import os


def _quota_cpus():
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:  # cgroup v1
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            return max(1, quota // period)
    except (OSError, ValueError):
        pass
    return None


def usable_cpus():
    """min(CPUs in the affinity mask, cgroup quota in whole CPUs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = _quota_cpus()
    return max(1, min(n, q) if q else n)
