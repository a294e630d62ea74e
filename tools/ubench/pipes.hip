// Micro-benchmark (bring-up evidence, not product): which gfx950 issue pipes run side by side?
//   fp32 "matrix" work  v_mfma_f32_16x16x4_f32   (shares the VALU's pipe: tools/ubench/overlap.hip)
//   bf16 matrix work    v_mfma_f32_16x16x32_bf16 (the XDL matrix cores)
//   fp32 VALU work      v_fma_f32
// Modes (every workgroup = 4 waves, 2 workgroups per CU => 2 waves per SIMD, told apart by the HW wave slot parity):
//   0 all waves: F fp32 MFMAs                     1 all waves: B bf16 MFMAs                2 all waves: V VALU fmas
//   3 even slots fp32 MFMA, odd slots bf16 MFMA   4 even slots VALU, odd slots bf16 MFMA   5 even fp32 MFMA, odd VALU
//   6 one wave: fp32 MFMAs then bf16 MFMAs        7 one wave: interleaved 8 fp32 + 16 bf16 (independent accumulators)
//   8 one wave: interleaved 16 bf16 MFMA + 32 VALU     9 all waves: 32 v_pk_fma_f32 (packed fp32) per trip
// If two pipes are independent, mode 3 (4, 7, 8) takes max(...) of the single-pipe times instead of their sum.
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = float __attribute__((ext_vector_type(4)));
using bf8 = __bf16 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void f32_mfma(f32x4 (&acc)[4], float a, float b, int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
    }
}
__device__ __forceinline__ void bf_mfma(f32x4 (&acc)[4], bf8 a, bf8 b, int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
    }
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void valu_pk(f2 (&v)[16], f2 c, int n) {          // 32 v_pk_fma_f32 per trip
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = __builtin_elementwise_fma(v[k], c, f2{0.25f, 0.5f});
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = __builtin_elementwise_fma(v[k], c, f2{-0.25f, -0.5f});
    }
}
__device__ __forceinline__ void valu(float (&v)[16], float c, int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = fmaf(v[k], c, 0.25f);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = fmaf(v[k], c, -0.25f);
    }
}

// n: loop trips.  Per trip: 8 fp32 MFMAs (8 x 32 = 256 pipe cycles) | 16 bf16 MFMAs (16 x 16 = 256 if 4 passes) | 32 VALU (128 cycles)
__global__ void __launch_bounds__(256, 2) k(int mode, int n, float *out) {
    __shared__ float pad[12 * 1024];
    f32x4 accF[4] = {}, accB[4] = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1u;     // HW_ID[3:0] = wave slot
    const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    bf8 xa, xb;
    for (int i = 0; i < 8; ++i) { xa[i] = (__bf16)(a + i); xb[i] = (__bf16)(b - i); }
    switch (mode) {
        case 0: f32_mfma(accF, a, b, n); break;
        case 1: bf_mfma(accB, xa, xb, n); break;
        case 2: valu(v, 0.999f, n); break;
        case 3: if (slot) bf_mfma(accB, xa, xb, n); else f32_mfma(accF, a, b, n); break;
        case 4: if (slot) bf_mfma(accB, xa, xb, n); else valu(v, 0.999f, n); break;
        case 5: if (slot) valu(v, 0.999f, n); else f32_mfma(accF, a, b, n); break;
        case 6: f32_mfma(accF, a, b, n); bf_mfma(accB, xa, xb, n); break;
        case 7:
            for (int i = 0; i < n; ++i) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) accF[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, accF[m], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) accB[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, accB[m], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) accB[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, xa, accB[m], 0, 0, 0);
                }
            }
            break;
        case 9: {
            f2 w[16];
            for (int i = 0; i < 16; ++i) w[i] = f2{v[i], v[i] + 1.0f};
            valu_pk(w, f2{0.999f, 1.001f}, n);
            for (int i = 0; i < 16; ++i) v[i] = w[i].x + w[i].y;
            break;
        }
        default:
            for (int i = 0; i < n; ++i) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) accB[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, accB[m], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[(4 * kk + q) & 15] = fmaf(v[(4 * kk + q) & 15], 0.999f, 0.25f);
                }
            }
    }
    float s = 0.f;
    for (int m = 0; m < 4; ++m) s += accF[m][0] + accF[m][3] + accB[m][0] + accB[m][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    pad[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = pad[7];
}

int main() {
    const int grid = 512;                     // 2 workgroups per CU on 256 CUs
    float *out;
    hipMalloc(&out, grid * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 10; ++mode) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, mode, n, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d : %.3f ms  (%.1f cycles per trip at 2.4 GHz)\n", mode, ms, ms * 1e-3 * 2.4e9 / n);
        }
    return 0;
}
