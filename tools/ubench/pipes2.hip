// Micro-benchmark (bring-up evidence, not product): do the fp32 matrix pipe (v_mfma_f32_16x16x4_f32), the bf16 matrix
// pipe (v_mfma_f32_16x16x32_bf16) and the plain VALU (v_fma_f32, v_exp_f32) of ONE SIMD run side by side on gfx950?
//
// tools/ubench/pipes.hip (round 2) said "no: everything serialises on one issue pipe"; MI355X_MICROARCH.md ("Wave
// scheduling") says MFMA and VALU pipes are separate.  The round-2 experiment let the two waves of a SIMD run the same
// number of loop trips of very different length and timed the kernel, i.e. the slower wave -- which cannot tell "the pipes
// are shared" from "the VALU wave won every issue arbitration until it was done".  This version
//   (a) BALANCES the two roles: the VALU wave's trip is sized to take as long as the MFMA wave's trip when each runs
//       alone (V fmas per trip, calibrated by the A and B rows), so that co-execution shows as ~1x and serialisation as
//       ~2x of the solo time whatever the arbitration does;
//   (b) records every wave's own s_memtime span and its HW_ID, so that the pairing (one wave of each role per SIMD) is
//       verified instead of assumed;
//   (c) repeats the pairings with s_setprio 3 on the MFMA wave / on the VALU wave;
//   (d) sweeps, inside ONE wave per SIMD, the number of independent VALU instructions placed between consecutive MFMAs
//       (0..12 v_fma_f32 or v_exp_f32): if the VALU executes beside the matrix pipe the trip time stays at the MFMA-only
//       time until the issue slots in the shadow are used up; if it shares the pipe the time grows from the first one;
//   (e) runs the pairings with 4 waves per SIMD (2 + 2) as well.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/pipes2.hip -o build/pipes2 ; run on the GPU box: build/pipes2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using bf8 = __bf16 __attribute__((ext_vector_type(8)));

enum Role { R_F32MFMA = 0, R_BF16MFMA = 1, R_VALU = 2, R_EXP = 3, R_IDLE = 4 };

struct Rec {
    unsigned hwid;
    unsigned role;
    unsigned long long cycles;
};

#define FMA8(v)                                                                                              \
    asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t"      \
                 "v_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"      \
                 "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"                                      \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
                 : "v"(c0), "v"(c1))
#define EXP8(v)                                                                                              \
    asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"          \
                 "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"              \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]))

// One trip of each role:  F32: 8 x v_mfma_f32_16x16x4_f32 on 8 independent accumulators (8 x 32 = 256 pipe cycles)
//                         BF16: 16 x v_mfma_f32_16x16x32_bf16 on 8 independent accumulators (16 x 16 = 256)
//                         VALU: V8 x 8 v_fma_f32 on 8 independent registers;  EXP: V8 x 8 v_exp_f32
// roleA for even HW wave slots, roleB for odd ones; prio applied to the wave before the loop.
template <int WPS>
__global__ void __launch_bounds__(256, WPS) pair_kernel(int roleA, int roleB, int prioA, int prioB, int n, int V8, Rec *rec,
                                                        float *sink) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // HW_REG_XCC_ID[3:0]
    const unsigned slot = hwid & 15u;
    const int role = (slot & 1) ? roleB : roleA;
    const int prio = (slot & 1) ? prioB : prioA;
    f32x4 acc[8] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f, c0 = 0.999f, c1 = 0.25f;
    bf8 xa, xb;
    for (int i = 0; i < 8; ++i) { xa[i] = (__bf16)(a + i); xb[i] = (__bf16)(b - i); }
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (role == R_F32MFMA) {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
        }
    } else if (role == R_BF16MFMA) {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, acc[m], 0, 0, 0);
        }
    } else if (role == R_VALU) {
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < V8; ++k) FMA8(v);
    } else if (role == R_EXP) {
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < V8; ++k) EXP8(v);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3] + v[m];
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
        rec[wv].hwid = (hwid & 0xFFFFu) | (xcc << 16);
        rec[wv].role = role;
        rec[wv].cycles = t1 - t0;
    }
    if (s == 12345.678f) sink[0] = s;
}

// One wave per SIMD: per trip 8 MFMAs, each followed by K independent VALU instructions (kind 0 v_fma_f32, 1 v_exp_f32).
template <int K, int KIND, int MF>
__global__ void __launch_bounds__(256, 1) shadow_kernel(int n, Rec *rec, float *sink) {
    f32x4 acc[8] = {};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f, c0 = 0.999f, c1 = 0.25f;
    bf8 xa, xb;
    for (int i = 0; i < 8; ++i) { xa[i] = (__bf16)(a + i); xb[i] = (__bf16)(b - i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MF == 0) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            else acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(c0), "v"(c1));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][3];
    for (int i = 0; i < 12; ++i) s += v[i];
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
        rec[wv].hwid = 0;
        rec[wv].role = 0;
        rec[wv].cycles = t1 - t0;
    }
    if (s == 12345.678f) sink[0] = s;
}

static Rec *d_rec;
static float *d_sink;
static std::vector<Rec> h_rec;

struct PairResult {
    float ms;
    double cycA, cycB;       // mean s_memtime span per trip of the waves of each role
    int nA, nB, simds, mixed;
};

template <int WPS>
static PairResult run_pair(int roleA, int roleB, int prioA, int prioB, int n, int V8) {
    const int grid = 256 * WPS;               // WPS workgroups of 4 waves per CU => WPS waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    PairResult r{};
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(pair_kernel<WPS>, dim3(grid), dim3(256), 0, 0, roleA, roleB, prioA, prioB, n, V8, d_rec, d_sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&r.ms, e0, e1);
    }
    const int waves = grid * 4;
    hipMemcpy(h_rec.data(), d_rec, waves * sizeof(Rec), hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    std::map<unsigned, unsigned> simd_roles;   // (everything above the wave slot) -> bit mask of the roles seen there
    for (int i = 0; i < waves; ++i) {
        const bool odd = h_rec[i].hwid & 1;
        (odd ? sb : sa) += (double)h_rec[i].cycles;
        (odd ? r.nB : r.nA)++;
        simd_roles[h_rec[i].hwid & 0x000FFF30u] |= 1u << (odd ? 1 : 0);   // simd (5:4), cu (11:8), sh (12), se (15:13), xcc (19:16)
    }
    r.cycA = r.nA ? sa / r.nA / n : 0;
    r.cycB = r.nB ? sb / r.nB / n : 0;
    r.simds = (int)simd_roles.size();
    for (auto &kv : simd_roles) r.mixed += kv.second == 3u;
    return r;
}

template <int K, int KIND, int MF>
static double run_shadow(int n) {
    double best = 1e30;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((shadow_kernel<K, KIND, MF>), dim3(256), dim3(256), 0, 0, n, d_rec, d_sink);
        hipDeviceSynchronize();
        hipMemcpy(h_rec.data(), d_rec, 1024 * sizeof(Rec), hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < 1024; ++i) s += (double)h_rec[i].cycles;
        best = std::min(best, s / 1024 / n);
    }
    return best;
}

static const char *rname(int r) {
    static const char *n[] = {"f32mfma", "bf16mfma", "valu", "exp", "idle"};
    return n[r];
}

template <int WPS>
static void pair_table(int n, int V8fma, int V8exp) {
    printf("\n== %d waves per SIMD (even slots role A, odd slots role B); cycles = s_memtime ticks per trip of a wave of that role ==\n", WPS);
    printf("%-10s %-10s %5s %5s | %9s | %9s %9s | %5s %5s %6s %6s\n", "roleA", "roleB", "prioA", "prioB", "kernel_ms", "cyc/tripA",
           "cyc/tripB", "nA", "nB", "simds", "mixed");
    struct Cfg { int a, b, pa, pb; };
    const Cfg cfgs[] = {
        {R_F32MFMA, R_IDLE, 0, 0},  {R_BF16MFMA, R_IDLE, 0, 0}, {R_VALU, R_IDLE, 0, 0},     {R_EXP, R_IDLE, 0, 0},
        {R_F32MFMA, R_F32MFMA, 0, 0}, {R_BF16MFMA, R_BF16MFMA, 0, 0}, {R_VALU, R_VALU, 0, 0}, {R_EXP, R_EXP, 0, 0},
        {R_F32MFMA, R_VALU, 0, 0},  {R_F32MFMA, R_VALU, 3, 0},  {R_F32MFMA, R_VALU, 0, 3},
        {R_BF16MFMA, R_VALU, 0, 0}, {R_BF16MFMA, R_VALU, 3, 0}, {R_BF16MFMA, R_VALU, 0, 3},
        {R_F32MFMA, R_EXP, 0, 0},   {R_F32MFMA, R_EXP, 3, 0},   {R_F32MFMA, R_EXP, 0, 3},
        {R_F32MFMA, R_BF16MFMA, 0, 0}, {R_VALU, R_EXP, 0, 0},
    };
    for (const Cfg &c : cfgs) {
        const int V8 = (c.a == R_EXP || c.b == R_EXP) && !(c.a == R_VALU || c.b == R_VALU) ? V8exp : V8fma;
        // (valu + exp pairing: both loops use V8fma trips of 8, the exp one is then simply longer; read its solo row)
        const PairResult r = run_pair<WPS>(c.a, c.b, c.pa, c.pb, n, V8);
        printf("%-10s %-10s %5d %5d | %9.3f | %9.1f %9.1f | %5d %5d %6d %6d\n", rname(c.a), rname(c.b), c.pa, c.pb, r.ms, r.cycA,
               r.cycB, r.nA, r.nB, r.simds, r.mixed);
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4000;
    hipMalloc(&d_rec, 4096 * 4 * sizeof(Rec));
    hipMalloc(&d_sink, 64);
    h_rec.resize(4096 * 4);
    // calibration: how many groups of 8 fmas / exps take ~256 cycles when the wave has the SIMD to itself
    const PairResult f = run_pair<2>(R_VALU, R_IDLE, 0, 0, n, 8), e = run_pair<2>(R_EXP, R_IDLE, 0, 0, n, 8);
    const PairResult m = run_pair<2>(R_F32MFMA, R_IDLE, 0, 0, n, 8);
    const int V8fma = (int)(m.cycA / (f.cycA / 8) + 0.5), V8exp = (int)(m.cycA / (e.cycA / 8) + 0.5);
    printf("calibration (solo wave per SIMD): f32mfma trip %.1f ticks, 64 v_fma_f32 %.1f, 64 v_exp_f32 %.1f  =>  V8(fma) = %d, V8(exp) = %d groups of 8 per trip\n",
           m.cycA, f.cycA, e.cycA, V8fma, V8exp);
    pair_table<2>(n, V8fma, V8exp);
    pair_table<4>(n, V8fma, V8exp);

    printf("\n== one wave per SIMD: 8 MFMAs per trip, K independent VALU instructions behind each; ticks per trip ==\n");
    printf("%-28s", "K =");
    const int ks[] = {0, 1, 2, 3, 4, 6, 8, 12};
    for (int k : ks) printf(" %7d", k);
    printf("\n");
#define ROW(label, KIND, MF)                                                                                  \
    printf("%-28s %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f\n", label, run_shadow<0, KIND, MF>(n),          \
           run_shadow<1, KIND, MF>(n), run_shadow<2, KIND, MF>(n), run_shadow<3, KIND, MF>(n),                 \
           run_shadow<4, KIND, MF>(n), run_shadow<6, KIND, MF>(n), run_shadow<8, KIND, MF>(n), run_shadow<12, KIND, MF>(n))
    ROW("f32 16x16x4   + v_fma_f32", 0, 0);
    ROW("f32 16x16x4   + v_exp_f32", 1, 0);
    ROW("bf16 16x16x32 + v_fma_f32", 0, 1);
    ROW("bf16 16x16x32 + v_exp_f32", 1, 1);
    return 0;
}
