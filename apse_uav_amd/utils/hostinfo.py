"""Host CPU share of this process (containers: the cgroup quota, not os.cpu_count())."""
import math
import os


def usable_cpus():
    """CPUs this process may actually keep busy: min(affinity mask, cgroup v2/v1 CPU quota).  Running more compute threads
    than this gets the whole process throttled by the CFS quota (tens of ms stalls), so thread pools are sized by it."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(math.floor(float(quota) / float(period)))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    return max(1, n)
