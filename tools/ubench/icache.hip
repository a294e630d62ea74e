// Micro-benchmark (bring-up evidence, not product): is straight-line VALU code that does not fit the instruction cache
// (64 KB per two CUs) fetched at the rate the VALU consumes it?  Same instruction count both ways:
//   big:   NBIG v_fma_f32 in a row, fully unrolled (8 bytes each: ~100 KB of code), executed once per wave
//   small: a 64-instruction loop body executed NBIG / 64 times (fits the cache)
// 512 workgroups x 4 waves, 2 workgroups per CU, like the frontend kernels.  On the boxes where the 16 kHz frontend's
// FFT phases are slow, `big` is what to look at.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NBIG = 12288;

__global__ void __launch_bounds__(256, 2) big(float *out, float c) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < NBIG; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 15]) : "v"(c), "v"(c));
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 123.456f) out[0] = s;
}
__global__ void __launch_bounds__(256, 2) small(float *out, float c) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int rep = 0; rep < NBIG / 64; ++rep) {
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 15]) : "v"(c), "v"(c));
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 123.456f) out[0] = s;
}

int main() {
    float *out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int k = 0; k < 2; ++k) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) {
                if (k == 0) hipLaunchKernelGGL(big, dim3(4096), dim3(256), 0, 0, out, 0.999f);
                else hipLaunchKernelGGL(small, dim3(4096), dim3(256), 0, 0, out, 0.999f);
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("icache %s : %.3f ms per launch (4096 workgroups x %d v_fma)\n", k == 0 ? "big  " : "small", ms / 20, NBIG);
        }
    return 0;
}
