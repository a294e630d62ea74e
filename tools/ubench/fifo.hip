// Micro-benchmark (bring-up evidence, not product): is the per-CU vector-memory return path in-order
// ACROSS waves on gfx950?  Workgroup = 2 waves on one CU.  Wave 1 ("victim") times L2-hit loads from a
// small hot buffer in a loop.  Wave 0 ("aggressor") either idles (mode 0) or keeps issuing loads that
// miss to HBM from a 4 GiB buffer (mode 1), or issues the same misses through the scalar cache
// (s_load, mode 2).  If victim latency rises from L2-hit (~500-700 cycles) to HBM-miss level in mode 1,
// hits wait behind misses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(128) k(int mode, const float *hot, const float *big, size_t big_elems,
                                          long long *lat, float *sink, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    if (wave == 1) {
        for (int i = 0; i < iters; ++i) {
            const float *p = hot + ((size_t)(i * 64 + lane) * 32) % (1 << 18);      // 1 MiB hot set, L2 resident
            const long long t0 = __builtin_readcyclecounter();
            float v = __builtin_nontemporal_load(p);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t1 = __builtin_readcyclecounter();
            acc += v;
            if (lane == 0) lat[blockIdx.x * iters + i] = t1 - t0;
            __builtin_amdgcn_s_sleep(2);
        }
    } else {
        // sequential stream through this workgroup's own 16 MiB window: every lane a new 128-B line
        // (HBM miss, no reuse) but only 8 distinct 2-MiB pages per workgroup (TLB friendly)
        const size_t base = (size_t)blockIdx.x * (big_elems / gridDim.x);
        for (int i = 0; i < iters * 4; ++i) {
            const size_t e = base + (size_t)i * 64 * 32 + (size_t)lane * 32;
            if (mode == 1) {
                acc += big[e];                      // one distinct line per lane: 64 HBM misses per instruction
            } else if (mode == 2) {
                const float *q = big + (__builtin_amdgcn_readfirstlane((unsigned)(e >> 5)) << 5);
                float s;
                asm volatile("s_load_dword %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(s) : "s"(q) : "memory");
                acc += s;
            } else {
                __builtin_amdgcn_s_sleep(20);
            }
        }
    }
    sink[blockIdx.x * 128 + threadIdx.x] = acc;
}

int main() {
    const size_t big_elems = (size_t)1 << 30;     // 4 GiB
    float *hot, *big, *sink; long long *lat;
    const int grid = 256, iters = 400;
    (void)hipMalloc(&hot, 1 << 20); (void)hipMalloc(&big, big_elems * 4);
    (void)hipMalloc(&sink, grid * 128 * 4); (void)hipMalloc(&lat, (size_t)grid * iters * 8);
    (void)hipMemset(hot, 0, 1 << 20); (void)hipMemset(big, 0, big_elems * 4);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(grid), dim3(128), 0, 0, mode, hot, big, big_elems, lat, sink, iters);
            (void)hipDeviceSynchronize();
        }
        std::vector<long long> h((size_t)grid * iters);
        (void)hipMemcpy(h.data(), lat, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("mode %d (%s): victim L2-hit load latency [clock ticks]: p10 %lld p50 %lld p90 %lld p99 %lld\n", mode,
               mode == 0 ? "aggressor idle" : mode == 1 ? "aggressor vector HBM misses" : "aggressor scalar HBM misses",
               h[h.size() / 10], h[h.size() / 2], h[h.size() * 9 / 10], h[h.size() * 99 / 100]);
    }
    return 0;
}
