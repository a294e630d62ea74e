// Micro-benchmark (bring-up evidence, not product): do VALU work of one wave and MFMA work of ANOTHER
// wave on the same SIMD overlap on gfx950, and how are two co-resident workgroups placed?
//   mode 0: every wave runs the MFMA loop        mode 1: every wave runs the VALU loop
//   mode 2: waves in odd HW wave slots run VALU, even slots run MFMA (same per-wave work as 0/1)
//   mode 3: every wave runs MFMA loop then VALU loop (serial, same wave)
//   mode 4: every wave runs an interleaved MFMA+VALU loop (same wave, independent streams)
// Prints kernel time per mode and a histogram of HW_ID wave slots / CU placement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>

using f32x4 = float __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_loop(f32x4 (&acc)[4], float a, float b, int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void valu_loop(float (&v)[16], float c, int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = fmaf(v[k], c, 0.25f);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = fmaf(v[k], c, -0.25f);
    }
}

__global__ void __launch_bounds__(256, 2) k(int mode, int n_mfma, int n_valu, float *out, unsigned *ids, long long *clk) {
    __shared__ float pad[12 * 1024];          // 48 KiB: at most 3 workgroups per CU by LDS, 2 by launch bounds
    f32x4 acc[4] = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID, all 32 bits
    const unsigned slot = hw & 15u;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    const long long t0 = wall_clock64();
    if (mode == 0) mfma_loop(acc, a, b, n_mfma);
    else if (mode == 1) valu_loop(v, 0.999f, n_valu);
    else if (mode == 2) { if (slot & 1) valu_loop(v, 0.999f, n_valu); else mfma_loop(acc, a, b, n_mfma); }
    else if (mode == 3) { mfma_loop(acc, a, b, n_mfma); valu_loop(v, 0.999f, n_valu); }
    else {
        // interleaved: per 32 MFMAs (1024 cycles of matrix pipe) issue n_valu/n_mfma * 32 VALU ops
        const int per = n_valu / n_mfma;
        for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            }
            for (int j = 0; j < per; ++j) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) v[kk] = fmaf(v[kk], 0.999f, 0.25f);
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) v[kk] = fmaf(v[kk], 0.999f, -0.25f);
            }
        }
    }
    const long long t1 = wall_clock64();
    float s = 0.f;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    pad[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = pad[7];
        ids[blockIdx.x] = hw;
        clk[2 * blockIdx.x] = t0;
        clk[2 * blockIdx.x + 1] = t1;
    }
}

int main() {
    const int grid = 512;                     // 2 workgroups per CU on 256 CUs
    float *out; unsigned *ids; long long *clk;
    hipMalloc(&out, grid * 4); hipMalloc(&ids, grid * 4); hipMalloc(&clk, grid * 16);
    const int n_mfma = 2000, n_valu = 2000;   // 64000 MFMAs (2.05M pipe cycles) ; 64000 VALU (256k cycles)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 5; ++mode) {
        for (int nv : {2000, 8000}) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, mode, n_mfma, nv, out, ids, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d n_mfma %d n_valu %d : %.3f ms\n", mode, n_mfma, nv, ms);
        }
    }
    std::vector<unsigned> h(grid); std::vector<long long> c(2 * grid);
    hipMemcpy(h.data(), ids, grid * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> slots; std::map<unsigned, std::vector<int>> percu;
    for (int i = 0; i < grid; ++i) {
        slots[h[i] & 15]++;
        // XCC id is not in HW_ID; key = SE(15:13) SH(12) CU(11:8) -> collisions across XCDs are expected (x8)
        percu[(h[i] >> 8) & 0xff].push_back(h[i] & 15);
    }
    printf("wave-slot histogram (wave 0 of each WG):");
    for (auto &kv : slots) printf(" slot%u:%d", kv.first, kv.second);
    printf("\nfirst 24 WGs: ");
    for (int i = 0; i < 24; ++i) printf("[b%d hw=%08x slot=%u simd=%u cu=%u se=%u] ", i, h[i], h[i] & 15, (h[i] >> 4) & 3, (h[i] >> 8) & 15, (h[i] >> 13) & 7);
    printf("\n");
    return 0;
}
