// occ -- how busy does the fp32 matrix pipe get with 2, 3, 4 waves per SIMD when every wave runs the throughput frontend's
// inner loop: [read the next step's two A fragments from LDS (ds_read_b128 x 2)] [8 x v_mfma_f32_16x16x4_f32 on two accumulators],
// one step ahead, sched_barrier'ed exactly like gemm_r (kernel_front_f43.hip)?  The product kernel's timing-only ablation "MFMAs
// only" (no FFT, no loads, no ring DMA, no barriers) reaches 86.5 % of the pipe at two waves per SIMD; is that the LDS latency two
// waves cannot cover, i.e. would a third wave recover it?
//   build + run ON the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/occ.hip -o /tmp/occ && /tmp/occ
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using lds_f32x4 = __attribute__((address_space(3))) const f32x4;
__device__ __forceinline__ f32x4 lds4(unsigned a) { return *reinterpret_cast<lds_f32x4 *>(a); }

template <int WPS, int VALU>
__global__ void __launch_bounds__(256, WPS) loop_kernel(int n, unsigned long long *cyc, float *sink) {
    __shared__ __attribute__((aligned(16))) float lds[12288];          // 48 KiB: the ring's three 16 KiB units
    for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) float *)lds) + (threadIdx.x & 63) * 16;
    f32x4 acc0 = {}, acc1 = {};
    float bv[4] = {1.f + threadIdx.x * 1e-3f, 0.5f, 0.25f, 2.f};
    f32x4 c0 = lds4(base), c1 = lds4(base + 1024);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int st = 0; st < 24; ++st) {                                // three units of 8 steps
            const f32x4 n0 = lds4(base + ((2 * (st + 1)) % 48) * 1024), n1 = lds4(base + ((2 * (st + 1) + 1) % 48) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            if (VALU) {                                                    // the B-operand transform of encoder 0: ~2 VALU per k-step
#pragma unroll
                for (int k = 0; k < 4; ++k) bv[k] = fmaf(bv[k], 0.999f, acc0[k] * 1e-30f);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[k], bv[k], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[k], bv[k], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = n0;
            c1 = n1;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc0[0] + acc1[1] == 12345.678f) sink[0] = acc0[0];
}

template <int WPS, int VALU>
static void run(int n) {
    unsigned long long *d;
    float *sink;
    const int grid = 256 * WPS;
    hipMalloc(&d, grid * 4 * 8);
    hipMalloc(&sink, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<WPS, VALU>), dim3(grid), dim3(256), 0, 0, n, d, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), d, grid * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    const double per_step = s / h.size() / n / 24;                      // wave cycles per step (8 MFMAs = 256 pipe cycles)
    printf("%d waves per SIMD%s: %.1f wave cycles per step -> matrix pipe busy %.3f  (kernel %.3f ms)\n", WPS,
           VALU ? " + 4 VALU per step" : "", per_step, 256.0 * WPS / per_step, ms);
    hipFree(d);
    hipFree(sink);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2000;
    run<1, 0>(n); run<2, 0>(n); run<3, 0>(n); run<4, 0>(n);
    run<1, 1>(n); run<2, 1>(n); run<3, 1>(n); run<4, 1>(n);
    return 0;
}
