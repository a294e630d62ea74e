// pipes3 -- which VALU instructions run beside the bf16 matrix pipe on one gfx950 SIMD?  (follow-up of pipes2.hip)
// pipes2 showed v_fma_f32 of one wave executing beside v_mfma_f32_16x16x32_bf16 of the other wave of the SIMD.  The bf16 x 9
// frontend's VALU work is packed fp32 arithmetic (v_pk_fma_f32, v_pk_add_f32), conversions (v_cvt_pk_bf16_f32) and bit operations
// (v_and_b32, v_lshlrev_b32); its counters show almost no co-execution.  This pairs a bf16-MFMA wave with a wave running ONE kind
// of VALU instruction (two waves per SIMD, roles by HW wave slot parity, per-wave s_memtime spans), and also times both in ONE
// wave ([1 MFMA][K VALU] repeated).
//   build + run ON the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/pipes3.hip -o /tmp/pipes3 && /tmp/pipes3
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x2 = float __attribute__((ext_vector_type(2)));
using bf8 = __bf16 __attribute__((ext_vector_type(8)));

enum Kind { K_FMA = 0, K_PKFMA, K_PKADD, K_CVT, K_AND, K_LSHL, K_ADD, K_MOV, K_PKMUL, K_DOT2, NKIND };
static const char *kname[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "v_and_b32", "v_lshlrev_b32", "v_add_f32",
                              "v_mov_b32", "v_pk_mul_f32", "v_dot2c_f32_bf16"};

template <int KIND>
__device__ __forceinline__ void valu1(float &x, f32x2 &p, float c0, float c1, f32x2 pc) {
    if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
    else if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(pc));
    else if (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(pc));
    else if (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(pc));
    else if (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c0));
    else if (KIND == K_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(c0));
    else if (KIND == K_LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
    else if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c0));
    else if (KIND == K_DOT2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
    else asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(c0));
}

struct Rec {
    unsigned odd;
    unsigned long long cycles;
};

// role by wave-slot parity: even slots 16 bf16 MFMAs per trip (8 accumulators), odd slots V x 8 VALU instructions per trip;
// mode 0: both, 1: MFMA waves only (odd idle), 2: VALU waves only (even idle)
template <int KIND>
__global__ void __launch_bounds__(256, 2) pair_kernel(int mode, int n, int V, Rec *rec, float *sink) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const bool odd = hwid & 1;
    f32x4 acc[8] = {};
    float v[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 1e-3f + i; p[i] = f32x2{v[i], v[i] + 1}; }
    const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f, c0 = 0.999f, c1 = 0.25f;
    const f32x2 pc{0.999f, 1.001f};
    bf8 xa, xb;
    for (int i = 0; i < 8; ++i) { xa[i] = (__bf16)(a + i); xb[i] = (__bf16)(b - i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (!odd && mode != 2) {
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, acc[m], 0, 0, 0);
    } else if (odd && mode != 1) {
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < V; ++k) {
#pragma unroll
                for (int j = 0; j < 8; ++j) valu1<KIND>(v[j], p[j], c0, c1, pc);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3] + v[m] + p[m][0] + p[m][1];
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
        rec[wv].odd = odd;
        rec[wv].cycles = t1 - t0;
    }
    if (s == 12345.678f) sink[0] = s;
}


// pairing 2: even slots run 16 bf16 MFMAs per trip with the accumulators forced into arch VGPRs (ACC = 0) or AGPRs (ACC = 1);
// odd slots run a DENSE stream of one VALU kind (64 instructions per loop branch).  Does the co-execution depend on where the
// accumulators live (register-file ports)?
template <int KIND, int ACC>
__global__ void __launch_bounds__(256, 2) pair2_kernel(int mode, int n, int V, Rec *rec, float *sink) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const bool odd = hwid & 1;
    f32x4 acc[8] = {};
    float v[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 1e-3f + i; p[i] = f32x2{v[i], v[i] + 1}; }
    const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f, c0 = 0.999f, c1 = 0.25f;
    const f32x2 pc{0.999f, 1.001f};
    bf8 xa, xb;
    for (int i = 0; i < 8; ++i) { xa[i] = (__bf16)(a + i); xb[i] = (__bf16)(b - i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (!odd && mode != 2) {
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (ACC == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(xa), "v"(xb));
                    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(xa), "v"(xb));
                }
    } else if (odd && mode != 1) {
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < V; k += 8) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) valu1<KIND>(v[j], p[j], c0, c1, pc);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3] + v[m] + p[m][0] + p[m][1];
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
        rec[wv].odd = odd;
        rec[wv].cycles = t1 - t0;
    }
    if (s == 12345.678f) sink[0] = s;
}

// one wave per SIMD: [1 bf16 MFMA][K VALU of one kind] x 8 per trip
template <int KIND, int K>
__global__ void __launch_bounds__(256, 1) shadow_kernel(int n, Rec *rec, float *sink) {
    f32x4 acc[8] = {};
    float v[12];
    f32x2 p[12];
    for (int i = 0; i < 12; ++i) { v[i] = threadIdx.x * 1e-3f + i; p[i] = f32x2{v[i], v[i] + 1}; }
    const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f, c0 = 0.999f, c1 = 0.25f;
    const f32x2 pc{0.999f, 1.001f};
    bf8 xa, xb;
    for (int i = 0; i < 8; ++i) { xa[i] = (__bf16)(a + i); xb[i] = (__bf16)(b - i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) valu1<KIND>(v[k], p[k], c0, c1, pc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3];
    for (int i = 0; i < 12; ++i) s += v[i] + p[i][0] + p[i][1];
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
        rec[wv].odd = 0;
        rec[wv].cycles = t1 - t0;
    }
    if (s == 12345.678f) sink[0] = s;
}

static Rec *d_rec;
static float *d_sink;
static std::vector<Rec> h_rec;

template <int KIND>
static void pair(int mode, int n, int V, double &cycM, double &cycV) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(pair_kernel<KIND>, dim3(512), dim3(256), 0, 0, mode, n, V, d_rec, d_sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(h_rec.data(), d_rec, 2048 * sizeof(Rec), hipMemcpyDeviceToHost);
    double sm = 0, sv = 0;
    int nm = 0, nv = 0;
    for (int i = 0; i < 2048; ++i) {
        if (h_rec[i].odd) { sv += (double)h_rec[i].cycles; nv++; }
        else { sm += (double)h_rec[i].cycles; nm++; }
    }
    cycM = nm ? sm / nm / n : 0;
    cycV = nv ? sv / nv / n : 0;
}

template <int KIND, int K>
static double shadow(int n) {
    double best = 1e30;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((shadow_kernel<KIND, K>), dim3(256), dim3(256), 0, 0, n, d_rec, d_sink);
        hipDeviceSynchronize();
        hipMemcpy(h_rec.data(), d_rec, 1024 * sizeof(Rec), hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < 1024; ++i) s += (double)h_rec[i].cycles;
        best = s / 1024 / n < best ? s / 1024 / n : best;
    }
    return best;
}

template <int KIND>
static void row(int n) {
    double m0, v0, m1, v1, m2, v2;
    pair<KIND>(2, n, 8, m0, v0);                         // VALU alone, 64 instructions per trip
    const int V = (int)(264.0 / (v0 / 8) + 0.5);         // trips sized to the MFMA trip (16 x 16.5 cycles)
    pair<KIND>(1, n, V, m1, v1);
    pair<KIND>(2, n, V, m2, v2);
    double mp, vp;
    pair<KIND>(0, n, V, mp, vp);
    printf("%-20s | cyc/instr alone %5.2f | V=%3d x8 | MFMA alone %6.1f  VALU alone %6.1f | paired: MFMA %6.1f  VALU %6.1f | in one wave, K = 0 1 2 3 4 6: "
           "%6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n",
           kname[KIND], v0 / 64, V, m1, v2, mp, vp, shadow<KIND, 0>(n), shadow<KIND, 1>(n), shadow<KIND, 2>(n), shadow<KIND, 3>(n),
           shadow<KIND, 4>(n), shadow<KIND, 6>(n));
}


template <int KIND, int ACC>
static void pair2(int mode, int n, int V, double &cycM, double &cycV) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((pair2_kernel<KIND, ACC>), dim3(512), dim3(256), 0, 0, mode, n, V, d_rec, d_sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(h_rec.data(), d_rec, 2048 * sizeof(Rec), hipMemcpyDeviceToHost);
    double sm = 0, sv = 0;
    int nm = 0, nv = 0;
    for (int i = 0; i < 2048; ++i) {
        if (h_rec[i].odd) { sv += (double)h_rec[i].cycles; nv++; }
        else { sm += (double)h_rec[i].cycles; nm++; }
    }
    cycM = nm ? sm / nm / n : 0;
    cycV = nv ? sv / nv / n : 0;
}
template <int KIND, int ACC>
static void row2(int n) {
    double m0, v0, m1, v1, m2, v2, mp, vp;
    pair2<KIND, ACC>(2, n, 8, m0, v0);                   // 64 VALU instructions per trip, alone
    const int V = 8 * (int)(264.0 / v0 + 0.5);           // dense trips sized to the MFMA trip
    pair2<KIND, ACC>(1, n, V, m1, v1);
    pair2<KIND, ACC>(2, n, V, m2, v2);
    pair2<KIND, ACC>(0, n, V, mp, vp);
    printf("%-20s acc in %s | dense cyc/instr alone %5.2f | V=%3d x8 | MFMA alone %6.1f  VALU alone %6.1f | paired: MFMA %6.1f  VALU %6.1f\n",
           kname[KIND], ACC ? "AGPR" : "VGPR", v0 / 64, V, m1, v2, mp, vp);
}


int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4000;
    hipMalloc(&d_rec, 4096 * sizeof(Rec));
    hipMalloc(&d_sink, 64);
    h_rec.resize(4096);
    printf("two waves per SIMD (even slot: 16 x v_mfma_f32_16x16x32_bf16 per trip, odd slot: V x 8 VALU per trip); ticks per trip.  "
           "In one wave: ticks per 8 x [1 MFMA][K VALU]\n");
    row2<K_DOT2, 0>(n);
    row2<K_FMA, 0>(n);
    row2<K_FMA, 1>(n);
    row2<K_AND, 0>(n);
    row2<K_AND, 1>(n);
    row2<K_CVT, 0>(n);
    row2<K_PKFMA, 0>(n);
    row2<K_PKFMA, 1>(n);
    row<K_FMA>(n);
    row<K_ADD>(n);
    row<K_MOV>(n);
    row<K_AND>(n);
    row<K_LSHL>(n);
    row<K_CVT>(n);
    row<K_PKFMA>(n);
    row<K_PKADD>(n);
    row<K_PKMUL>(n);
    return 0;
}
