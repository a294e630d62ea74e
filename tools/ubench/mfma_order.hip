// mfma_order -- in which order does v_mfma_f32_16x16x4_f32 add its four products to C?  Compares one MFMA on random operands with
// candidate fp32 evaluations of c + a0 b0 + a1 b1 + a2 b2 + a3 b3 per output element (k = lane group g of the A / B operands):
//   chain_up:   fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))      chain_down: the same from k = 3 to 0
//   exact:      the sum of the four exact products and c, rounded once (evaluated in double: 5 terms of 48 bits -- exact enough)
//   build + run ON the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_order.hip -o /tmp/mo && /tmp/mo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));

__global__ void k(const float *A, const float *B, const float *C, float *D, int n) {
    // trial t: A[t][16][4], B[t][4][16], C[t][16][16]
    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        const float a = A[(t * 16 + j) * 4 + g];            // A operand: lane (g, i) supplies A[i][g]
        const float b = B[(t * 4 + g) * 16 + j];            // B operand: lane (g, j) supplies B[g][j]
        f32x4 c;
        for (int r = 0; r < 4; ++r) c[r] = C[(t * 16 + 4 * g + r) * 16 + j];
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[(t * 16 + 4 * g + r) * 16 + j] = c[r];
    }
}

int main() {
    const int n = 4096;
    std::vector<float> A(n * 64), B(n * 64), C(n * 256), D(n * 256);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    for (auto &v : A) v = rnd() * (1.0f + 1000.0f * (rnd() > 0.9f));
    for (auto &v : B) v = rnd();
    for (auto &v : C) v = rnd() * 3.0f;
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, n);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    long up = 0, down = 0, exact = 0, pair = 0, total = 0;
    for (int t = 0; t < n; ++t)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const float *a = &A[(t * 16 + i) * 4];
                float bb[4];
                for (int kk = 0; kk < 4; ++kk) bb[kk] = B[(t * 4 + kk) * 16 + j];
                const float c = C[(t * 16 + i) * 16 + j], d = D[(t * 16 + i) * 16 + j];
                float u = c, dn = c;
                for (int kk = 0; kk < 4; ++kk) u = fmaf(a[kk], bb[kk], u);
                for (int kk = 3; kk >= 0; --kk) dn = fmaf(a[kk], bb[kk], dn);
                double e = c;
                for (int kk = 0; kk < 4; ++kk) e += (double)a[kk] * (double)bb[kk];
                const float p = fmaf(a[3], bb[3], fmaf(a[2], bb[2], 0.f)) + fmaf(a[1], bb[1], fmaf(a[0], bb[0], c));
                up += (u == d); down += (dn == d); exact += ((float)e == d); pair += (p == d);
                ++total;
            }
    printf("v_mfma_f32_16x16x4_f32 against candidates over %ld outputs: chain k=0..3 %ld  chain k=3..0 %ld  exact-sum-rounded-once %ld  pairwise %ld\n",
           total, up, down, exact, pair);
    return 0;
}
