// gf_host_cpus.hpp — how many hardware threads this process can really use: the affinity mask AND the container's CPU bandwidth quota.
// A GPU slice of a shared node typically shows every hardware thread of the host (256 here) behind a cgroup quota of a few cores (16 on the boxes of round 6:
// /sys/fs/cgroup/cpu.max = "1600000 100000"): threads beyond the quota do not run in parallel, they are throttled for the rest of every 100 ms period -- a worker pool
// sized from hardware_concurrency() (or spinning on a gate) then makes the whole process slower, by multiples (profiles/r06_e2e_pools.txt).
#pragma once
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <thread>

namespace gf {

inline int cgroup_cpu_quota() {   // CPUs' worth of bandwidth the cgroup grants (rounded up), or 0 when unlimited / unknown
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {               // cgroup v2: "<quota> <period>" or "max <period>"
        char q[64] = {0}; long long period = 0;
        const int n = fscanf(f, "%63s %lld", q, &period);
        fclose(f);
        if (n == 2 && q[0] != 'm' && period > 0) { const long long quota = atoll(q); if (quota > 0) return (int)((quota + period - 1) / period); }
        if (n >= 1) return 0;
    }
    long long quota = -1, period = 0;                                     // cgroup v1
    if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%lld", &quota) != 1) quota = -1; fclose(f); }
    if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%lld", &period) != 1) period = 0; fclose(f); }
    return (quota > 0 && period > 0) ? (int)((quota + period - 1) / period) : 0;
}

// hardware threads of the box, and the number this process can use in parallel (affinity mask, then quota); GF_HOST_CPUS overrides the latter
inline void host_cpus(int& box, int& usable) {
    box = (int)std::thread::hardware_concurrency();
    usable = box;
    { cpu_set_t cs; CPU_ZERO(&cs); if (sched_getaffinity(0, sizeof cs, &cs) == 0 && CPU_COUNT(&cs) > 0) usable = std::min(usable > 0 ? usable : CPU_COUNT(&cs), CPU_COUNT(&cs)); }
    static const int quota = cgroup_cpu_quota();
    if (quota > 0) usable = std::min(std::max(usable, 1), quota);
    if (const char* e = getenv("GF_HOST_CPUS")) if (atoi(e) > 0) usable = atoi(e);
    usable = std::max(usable, 1);
}

}  // namespace gf
