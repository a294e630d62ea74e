// host_threads.hpp -- how many host threads the native helpers (staging.cpp, segmenter.cpp) may use by default.
// std::thread::hardware_concurrency() reports the machine (256 on the MI355X hosts); a container usually owns far fewer
// CPUs through its cgroup quota (16 there), and running more busy threads than the quota gets the whole process throttled
// for the rest of the scheduling period -- measured: staging 4 GB on 32 threads under a 16-CPU quota took 458 ms instead
// of 35.  So: min(affinity mask, cgroup CPU quota, cap).
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace vad {

inline int cgroup_cpu_quota() {            // CPUs' worth of quota, 0 = unlimited / unknown
    long quota = -1, period = -1;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {               // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        if (std::fscanf(f, "%31s %ld", q, &period) == 2 && q[0] != 'm') quota = std::atol(q);
        std::fclose(f);
    } else {                                                                 // cgroup v1
        if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (std::fscanf(fq, "%ld", &quota) != 1) quota = -1;
            std::fclose(fq);
        }
        if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(fp, "%ld", &period) != 1) period = -1;
            std::fclose(fp);
        }
    }
    if (quota <= 0 || period <= 0) return 0;
    return (int)std::max(1L, quota / period);
}

inline int default_host_threads(int cap) {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    const int q = cgroup_cpu_quota();
    if (q > 0) n = std::min(n, q);
    return std::max(1, std::min(n, cap));
}

}  // namespace vad
