// front_common.hpp -- pieces shared by the two forms of the fp32 frontend: kernel_front_f43.hip (throughput: one wave = one
// 16-chunk tile, weights through an LDS ring) and kernel_front_lat.hip (latency: the four waves of a workgroup share one tile).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_wave.hpp"

namespace vad {
namespace {

#define VAD_INLINE __attribute__((always_inline))
#ifndef VAD_F43_EF
#define VAD_F43_EF 1             // the F(4,3) input transform reads the frames through E = x3 - x1, F = x2 - x0 (one fma for t0, t3, t4,
#endif                           // t5); 0: straight from x0..x3 (three fma's each; the form rounds 1-2 used, A/B: tools/variants.py)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

using lds_f32x4 = __attribute__((address_space(3))) const f32x4;
__device__ __forceinline__ f32x4 lds4(unsigned byte_addr) { return *reinterpret_cast<lds_f32x4 *>(byte_addr); }

// The coefficients of the F(4,3) input transform as OPAQUE scalars (SGPRs the compiler cannot see through, fresh for
// every GEMM that uses them).  Left to the optimiser the 6 x 33 transformed inputs would be computed once and kept, or
// merged between row parts -- registers this kernel does not have -- so every combination is written as fma's with
// these scalars: formed right in front of the MFMA that consumes it, one to three VALU instructions per k-step.  (Not
// inline-asm arithmetic: the hazard recogniser does not see inside asm and a VALU result consumed by the very next MFMA
// needs its wait states.)
struct Coef {
    float p1, m1, a, b;
};
template <unsigned A_BITS, unsigned B_BITS>                  // a, b as fp32 bit patterns
__device__ __forceinline__ Coef opaque_coef() {
    Coef k;
    asm volatile("s_mov_b32 %0, 1.0\n\ts_mov_b32 %1, -1.0\n\ts_mov_b32 %2, %4\n\ts_mov_b32 %3, %5"
                 : "=s"(k.p1), "=s"(k.m1), "=s"(k.a), "=s"(k.b) : "n"(A_BITS), "n"(B_BITS));
    return k;
}
[[maybe_unused]] constexpr unsigned kF2 = 0x40000000u, kFm2 = 0xC0000000u, kF4 = 0x40800000u, kFm4 = 0xC0800000u, kFm5 = 0xC0A00000u,
                   kFm125 = 0xBFA00000u, kF3 = 0x40400000u, kFm3 = 0xC0400000u, kFm025 = 0xBE800000u;

template <int M>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[M], const float *bias_lds, const Lane &ln) {
#pragma unroll
    for (int m = 0; m < M; ++m)
        acc[m] = *reinterpret_cast<const f32x4 *>(bias_lds + 16 * m + 4 * ln.g);
}
// acc += bias.  Encoders 0 and 1 start their accumulators from ZERO and take the bias here, behind the last product: a one-accumulator
// fp32 chain rounds every product-add at the ulp of what the accumulator holds, and a bias of up to 19 held from the start makes that
// ulp ~1e-6 for all 130-390 adds of the chain whatever the signal's size -- measured against float64 (tools/gx_error_study.py,
// profiles/r05_state_rows.md), the gate pre-activations carry 5 x more rounding error on quiet speech that way and 1.4 x more on
// full-level speech than the reference's evaluation order does; with the bias last both are level with it.  (Encoders 2, 3 and W_ih
// gain nothing measurable and keep the bias in the accumulator's initial value.)
// (Written like nyq_update below -- two row blocks per round, a scheduling barrier behind each: in any other shape the register
//  allocator of the 16 kHz kernel, 8 registers under its budget, spills 160.)
template <int M>
__device__ __forceinline__ void add_bias(f32x4 (&acc)[M], const float *bias_lds, const Lane &ln) {
    static_assert(M % 2 == 0, "row blocks in pairs");
#pragma unroll
    for (int m = 0; m < M; m += 2) {
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bias_lds + 16 * m + 4 * ln.g);
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bias_lds + 16 * (m + 1) + 4 * ln.g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[m][r] += b0[r];
            acc[m + 1][r] += b1[r];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int M>
__device__ __forceinline__ void zero(f32x4 (&acc)[M]) {
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
}
template <int M>
__device__ __forceinline__ void relu(f32x4 (&acc)[M]) {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][r] = fmaxf(acc[m][r], 0.f);
}
// Nyquist bin of one frame applied to RB output blocks: Y[row] += w_nyq[tap][row] * |Y_nyq| (exact fp32 fma)
template <int RB>
__device__ __forceinline__ void nyq_update(f32x4 (&Y)[RB], float xn, const float *wn_lds, const Lane &ln) {
#pragma unroll
    for (int m = 0; m < RB; m += 2) {
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wn_lds + 16 * m + 4 * ln.g);
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wn_lds + 16 * (m + 1) + 4 * ln.g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Y[m][r] = fmaf(w0[r], xn, Y[m][r]);
            Y[m + 1][r] = fmaf(w1[r], xn, Y[m + 1][r]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- encoder 1's program for one output row block: 40 blocks = 10 units x 4 k-groups, in the order the one-wave program
// accumulates them (kernel_front_f43.hip, "encoder 1 fed part by part"; layout.hpp w4_e1) -------------------------------------
struct E1Blk {
    int unit, kg, acc, frame, rbg;       // image unit, k-group, 0: out 0 (Z0) 1: out 1 (Z1), STFT frame and global row block of the B operand
};
constexpr E1Blk e1_blk(int Q, int idx) {
    const int n = idx / 4, kg = idx % 4;
    if (Q == 32) {
        // per part: [tap1 <- y0 | tap2 <- y1] -> out 0, [tap0 <- y1 | tap1 <- y2] -> out 1, and after every odd part
        // [tap2 <- y3 of part p-1 | tap2 <- y3 of part p] -> out 1; k-groups 0, 1 carry the first tensor's two row blocks
        const int parts[10] = {0, 0, 1, 1, 1, 2, 2, 3, 3, 3}, us[10] = {0, 1, 0, 1, 2, 0, 1, 0, 1, 2};
        const int p = parts[n], u = us[n], first = kg < 2;
        const int frame = u == 0 ? (first ? 0 : 1) : u == 1 ? (first ? 1 : 2) : 3;
        const int src = u == 2 ? (first ? p - 1 : p) : p;
        return E1Blk{vadl::w4_e1(p, u, 32), kg, u == 0 ? 0 : 1, frame, 2 * src + (kg & 1)};
    }
    // 8 kHz: per part tap1 <- y0, tap2 <- y1 (out 0); tap0 <- y1, tap1 <- y2, tap2 <- y3 (out 1); k-group = the part's row block
    const int p = n / 5, u = n % 5;
    const int fr[5] = {0, 1, 1, 2, 3}, ac[5] = {0, 0, 1, 1, 1};
    return E1Blk{vadl::w4_e1(p, u, 16), kg, ac[u], fr[u], 4 * p + kg};
}

}  // namespace
}  // namespace vad
