// kernel_front_b9.hip -- the throughput frontend (same function and same program as kernel_front_f43.hip: framing, reflect pad,
// window, 4 x real FFT magnitude, encoder 0 as one F(4,3) tile, encoders 1-3, W_ih) with every matrix product evaluated as EXACT
// bf16 x 9 piece products on the bf16 matrix pipe -- opt-in (option "front_mma" = "bf16x9"), the arithmetic of kernel_rec_b9.hip.
//
// Why.  A v_mfma_f32_16x16x4_f32 owns its SIMD's vector issue for 32 cycles (profiles/r03a_issue_pipes2.md): the fp32 frontend's
// 3 456 MFMAs and ~6 500 VALU instructions per tile ADD, and it sits at 0.86 of that sum.  v_mfma_f32_16x16x32_bf16 does 8x the
// MACs in half the cycles and runs BESIDE the VALU -- beside its PLAIN instructions: packed fp32 ones (v_pk_fma_f32, v_pk_add_f32,
// v_pk_mul_f32) wait for the matrix pipe (profiles/r03p_pipes3.md), so this file is built without them (__graft_entry__.EXTRA_FLAGS).
// Measured: 3.75-3.90 ms per C2 launch against the fp32 kernel's 4.33 (the clock drops to 1.99 GHz, the operand splitting doubles
// the VALU work, two waves per SIMD co-execute a fifth of the matrix pipe's busy time: profiles/r03p_front_bf16x9.md).
//
// Arithmetic.  Every fp32 operand is the exact sum of three bf16 pieces (8 significand bits each, fp32's exponent range), so a
// product of two operands is the sum of NINE piece products, each exact in fp32; the MFMA accumulates them in fp32.  Nothing is
// narrower than fp32; what changes against the fp32 chain is the order and the number of roundings in the accumulation.  The FFT,
// the Winograd input / output transforms, the Nyquist update, biases and ReLUs are the same fp32 VALU code as in the fp32 kernel.
// Weights are split on the host (layout.hpp "bf16 x 9 frontend image": the fp32 program unit for unit, 24 KiB units); activations
// are split by the lane that holds them, right before they are used as a B operand (8 values -> 3 x 4 registers of bf16 pairs).
//
// Structure: as kernel_front_f43.hip (one wave = 16 chunks; here 8 waves per workgroup, 1 workgroup per CU; the weight image streamed
// through a 3-slot LDS ring, barrier in the middle of a unit, fragment reads carried across unit boundaries).  A K32 step carries
// what two fp32 k-groups carried: slot (g, e) <-> fp32 k-step 8 kp + e at k = g, so the chain / mag layouts are unchanged.  Per step
// (1 K32 x 2 row blocks): 6 A fragments (3 pieces x 2 row blocks, 16 B per lane each) and 18 MFMAs, issued piece by piece so that
// only two fragments are live and two in flight.  The ring is 3 x 24 KiB; the FFT's tables (window, twiddles: used before the first
// request into slot 2) live in slot 2, so that the workgroup needs 79 KB and two 4-wave workgroups per CU stay possible (VAD_B9_WAVES).
// (reference: the same lines as kernel_front_f43.hip.)
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_wave.hpp"
#include "front_common.hpp"

namespace vad {
namespace {

#ifndef VAD_B9_WAVES
#define VAD_B9_WAVES 8             // waves (= tiles) per workgroup sharing one weight ring: 8 = one workgroup per CU, half the L2 -> LDS
#endif                             // traffic of two 4-wave workgroups (front 3.94 -> 3.92 ms, the recurrence launched behind it 1.31 -> 1.22 ms)
constexpr int kWaves = VAD_B9_WAVES, kShare = 24 * 1024 / kWaves;       // a wave's share of a unit's DMA
constexpr int kUnitBytes = (int)vadl::kW9UnitHalfs * 2;     // 24 blocks of 1 KiB
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using f32x2 = float __attribute__((ext_vector_type(2)));
using bf8 = __bf16 __attribute__((ext_vector_type(8)));
using bf2 = __bf16 __attribute__((ext_vector_type(2)));
using lds_u32x4 = __attribute__((address_space(3))) const u32x4;
__device__ __forceinline__ u32x4 lds4u(unsigned byte_addr) { return *reinterpret_cast<lds_u32x4 *>(byte_addr); }
__device__ __forceinline__ f32x4 mfma_b(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
// (x0, x1) -> three dwords, each holding the bf16 piece of x0 in its low and of x1 in its high half; x = p0 + p1 + p2 exactly
__device__ __forceinline__ void split3(float x0, float x1, unsigned &p0, unsigned &p1, unsigned &p2) {
#pragma clang fp contract(off)      // the remainders are exact differences: nothing may be fused into them
    f32x2 r{x0, x1};
    unsigned out[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bf2 h = __builtin_convertvector(r, bf2);                 // v_cvt_pk_bf16_f32: round to nearest even
        const unsigned bits = __builtin_bit_cast(unsigned, h);
        out[k] = bits;
        const f32x2 back{__uint_as_float(bits << 16), __uint_as_float(bits & 0xffff0000u)};
        r = r - back;
    }
    p0 = out[0];
    p1 = out[1];
    p2 = out[2];
}

// ---- the weight ring -------------------------------------------------------------------------------------------------
struct Ring {
    unsigned a_cur, a_nxt, a_far;       // LDS byte address of this lane's first A fragment in the slot of unit u, u+1, u+2
    unsigned d_cur, d_nxt, d_far;       // wave-uniform: where this wave's share of a unit lands in those slots
    const char *src;                    // wave-uniform: this wave's share of the next unit to request
    unsigned voff;                      // lane * 16
    u32x4 c0, c1;                       // A fragments (row block 0, 1) of the next (step, piece)
};

// Request this wave's share (6 x 1 KiB) of the next unit into the slot everyone has left.  LDS destination = M0 (wave-uniform
// base) + instruction offset + lane*16; the instruction offset reaches 4 KiB, so the share goes as two groups of three.
__device__ __forceinline__ void ring_request(Ring &r) {
    if (VAD_ABLATE & 8) return;
    unsigned keep_m0;                      // M0 is restored: the compiler may keep its own value there
    if constexpr (kWaves == 4) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %4\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %4 offset:2048\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep_m0) : "v"(r.voff), "s"(r.src), "s"(r.d_far), "s"(r.src + 3072), "s"(r.d_far + 3072u) : "memory");
    } else {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep_m0) : "v"(r.voff), "s"(r.src), "s"(r.d_far) : "memory");
    }
    r.src += kUnitBytes;
}
__device__ __forceinline__ void ring_rotate(Ring &r) {
    const unsigned a = r.a_cur, d = r.d_cur;
    r.a_cur = r.a_nxt; r.a_nxt = r.a_far; r.a_far = a;
    r.d_cur = r.d_nxt; r.d_nxt = r.d_far; r.d_far = d;
}

// the 8 B values f(0..7) a lane holds for a K32 step -> three registers of bf16 pairs per piece
template <class F>
__device__ __forceinline__ void split_step(u32x4 (&bp)[3], F f) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f(e);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned p0, p1, p2;
        if (VAD_ABLATE & 64) {                 // timing only: no split
            p0 = __float_as_uint(v[2 * d]);
            p1 = __float_as_uint(v[2 * d + 1]);
            p2 = p0 ^ p1;
        } else
        split3(v[2 * d], v[2 * d + 1], p0, p1, p2);
        bp[0][d] = p0;
        bp[1][d] = p1;
        bp[2][d] = p2;
    }
}

// A segment of M row blocks x KG fp32 k-groups = KG/2 K32 steps (whole units):  acc[m] += A[m][:, k] * B[k][:];  bfun(s) = the
// B-operand value this lane holds for fp32 k-step s (compile-time s), exactly as in the fp32 kernel.  A step = 1 K32 step x 2 row
// blocks = 18 MFMAs; a unit = 4 steps = 12 (step, piece) sub-steps of 6 MFMAs.  The fragments of sub-step q+1 are read from LDS
// before the MFMAs of sub-step q are issued; at the start of a unit's third step the workgroup makes the NEXT unit visible and
// requests the one after it into the slot everyone has left.  The 8 B values of a K32 step are split into pieces when its first
// row-block pair begins and serve all M/2 pairs.  (Forming the NEXT step's pieces between the MFMAs of the current one -- a software
// pipeline through sched_group_barrier, [MFMA][3 VALU] x 18 -- was built and measured: no gain, 4.23 against 4.14 ms, and 12 more
// live registers than the 16 kHz kernel has)
template <int M, int KG, int AFTER, class BF>
__device__ __forceinline__ void gemm_b(f32x4 (&acc)[M], BF bfun, Ring &r) {
    constexpr int H = M / 2, NSTEPS = (KG / 2) * H, NU = NSTEPS / 4;
    static_assert(KG % 2 == 0 && NSTEPS % 4 == 0 && M % 2 == 0, "segments are whole units");
    u32x4 bp[3];
    static_for<0, NU>([&](auto uc) VAD_INLINE {
        constexpr int u = decltype(uc)::value, after = (NU - 1 - u) + AFTER;
        static_for<0, 4>([&](auto sc_) VAD_INLINE {
            constexpr int st = decltype(sc_)::value, i = u * 4 + st, kp = i / H, mp = 2 * (i % H);
            if constexpr (st == 2 && after >= 1) {
                // this wave's share of the next unit has landed; then everyone's, and everyone has left the previous unit.
                // A bare s_barrier (no lgkmcnt(0) fence): the fragment reads in flight belong to the current slot
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(VAD_ABLATE & 1)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (after >= 2) ring_request(r);
            }
            if constexpr (i % H == 0) split_step(bp, [&](int e) VAD_INLINE { return bfun(kp * 8 + e); });
            static_for<0, 3>([&](auto pc_) VAD_INLINE {
                constexpr int pa = decltype(pc_)::value, q = st * 3 + pa;
                u32x4 n0 = r.c0, n1 = r.c1;
                if constexpr (VAD_ABLATE & 128) {          // timing only: no fragment reads
                } else if constexpr (q + 1 < 12) {
                    n0 = lds4u(r.a_cur + (2 * (q + 1)) * 1024);
                    n1 = lds4u(r.a_cur + (2 * (q + 1) + 1) * 1024);
                } else if constexpr (after >= 1) {
                    n0 = lds4u(r.a_nxt);
                    n1 = lds4u(r.a_nxt + 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pb = 0; pb < ((VAD_ABLATE & 256) ? 1 : 3); ++pb) {      // (256: timing only, a third of the MFMAs)
                    acc[mp + 0] = mfma_b(r.c0, bp[pb], acc[mp + 0]);
                    acc[mp + 1] = mfma_b(r.c1, bp[pb], acc[mp + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                r.c0 = n0;
                r.c1 = n1;
            });
        });
        ring_rotate(r);
    });
}

template <int Q, typename PcmT, int DEC>
__global__ void __launch_bounds__(64 * kWaves, 8 / kWaves) front_b9_kernel(const FrontArgs a) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int TABF = (tb.total + 3) / 4 * 4;
    constexpr int RB = w_rb(Q), P = w_parts(Q), KG0 = Q / 4;
    constexpr int PB = Q == 32 ? 2 : 1;                   // row parts per loop body (16 kHz: an even and an odd one)
    static_assert(w4_tail0(Q) + 20 == w4_units(Q) && P % PB == 0, "program mismatch");
    // LDS: [biases + head + Nyquist weights: NS floats][ring 3 x 24 KiB]; the FFT's tables (window, twiddles) sit in ring slot 2 until
    // the first request into it (middle of unit 0: behind a barrier every wave reaches only after its FFT)
    constexpr int NS = tb.window + (tb.total - tb.w_nyq), NF = tb.w_nyq - tb.window, UF = kUnitBytes / 4;
    static_assert(tb.window % 4 == 0 && tb.w_nyq % 4 == 0 && NS % 4 == 0 && NF <= UF && TABF == tb.total, "table split");
    __shared__ __attribute__((aligned(16))) float lds[NS + 3 * UF];
    float *tab = lds;                                     // + off for off < tb.window
    float *tabn = lds + tb.window - tb.w_nyq;             // + tb.w_nyq + ... for the Nyquist weights
    float *tabf = lds + NS + 2 * UF - tb.window;          // + tb.window / tb.tw1 / tb.tw2 for the FFT

    Lane ln;
    ln.lane = threadIdx.x & 63;
    ln.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ln.g = ln.lane >> 4;
    ln.j = ln.lane & 15;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    long wt = (long)blockIdx.x * kWaves + ln.wave;
    ln.tile_valid = wt < total;
    if (!ln.tile_valid) wt = total - 1;
    ln.tl = wt % a.nt;
    ln.st = wt / a.nt;
    ln.t = a.t0 + ln.tl;
    const long bb = ln.st * 16 + ln.j;
    ln.b = (int)(bb < a.B ? bb : a.B - 1);
    ln.from_tail = a.tail != nullptr && ln.t == a.T - 1;
    ln.sgnA = ln.g < 2 ? 1.f : -1.f;
    ln.sgnB = (ln.g & 1) ? -1.f : 1.f;

#if VAD_TRACE
#define B9_TRACE(i) do { if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 16 + 8] = __builtin_readcyclecounter();
#else
#define B9_TRACE(i) do { } while (0)
#endif
    B9_TRACE(0);
    Ring ring;
    {
        const unsigned slot0 = (unsigned)(size_t)((__attribute__((address_space(3))) float *)(lds + NS));
        ring.voff = ln.lane * 16;
        ring.a_cur = slot0 + ring.voff;
        ring.a_nxt = ring.a_cur + kUnitBytes;
        ring.a_far = ring.a_cur + 2 * kUnitBytes;
        // the two priming requests go to slots 0 and 1: start rotated by two, so that "far" is slot 0 first, then slot 1
        ring.d_far = slot0 + (unsigned)ln.wave * (unsigned)kShare;
        ring.d_cur = ring.d_far + kUnitBytes;
        ring.d_nxt = ring.d_far + 2 * kUnitBytes;
        ring.src = reinterpret_cast<const char *>(a.wfront) + ln.wave * kShare;
        ring_request(ring);                               // unit 0 -> slot 0
        {   const unsigned d = ring.d_far; ring.d_far = ring.d_cur; ring.d_cur = ring.d_nxt; ring.d_nxt = d; }
        ring_request(ring);                               // unit 1 -> slot 1
        {   const unsigned d = ring.d_far; ring.d_far = ring.d_cur; ring.d_cur = ring.d_nxt; ring.d_nxt = d; }
        // now d_far = slot 2 (unit 2's), d_cur = slot 0, d_nxt = slot 1
    }
    {   // tables -> LDS: all loads of a thread are issued before the first is stored
        static_assert(tb.total % 4 == 0, "tables are copied as 16-byte vectors");
        constexpr int NT = 64 * kWaves, NV = tb.total / 4, PER = (NV + NT - 1) / NT;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.tables);
        f32x4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * NT;
            v[k] = src[i < NV ? i : NV - 1];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * NT;
            float *base = 4 * i < tb.window ? tab : 4 * i < tb.w_nyq ? tabf : tabn;
            if (i < NV) reinterpret_cast<f32x4 *>(base)[i] = v[k];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // units 0 and 1 (and the tables) have landed
    __syncthreads();
    B9_TRACE(1);

    // ---- the 4 frames: one FFT body ------------------------------------------------------------------------------------
    // The four magnitude arrays are a shift register: every iteration moves the frames down one place and transforms the
    // next frame into the top one, so that the loop body sees three live arrays (like the last frame of straight-line
    // code), not four loop-carried ones -- 33 registers this kernel does not have -- for 99 v_mov per iteration.
    float X0[Q + 1], X1[Q + 1], X2[Q + 1], X3[Q + 1];
#pragma unroll
    for (int k = 0; k <= Q; ++k) X1[k] = X2[k] = X3[k] = 0.f;
#pragma clang loop unroll(disable)
    for (int v = 0; v < 4; ++v) {
        // (an array is moved only once it holds a frame: 6 array moves per tile instead of 12)
        if (v >= 3) {
#pragma unroll
            for (int k = 0; k <= Q; ++k) X0[k] = X1[k];
        }
        if (v >= 2) {
#pragma unroll
            for (int k = 0; k <= Q; ++k) X1[k] = X2[k];
        }
        if (v >= 1) {
#pragma unroll
            for (int k = 0; k <= Q; ++k) X2[k] = X3[k];
        }
        fft_frame<Q, PcmT, DEC>(X3, v, a, tabf, ln);
    }
    B9_TRACE(2);
    // |Y_nyq| of chunk j lives in lane group 0 (X[Q]); every lane of the chunk needs it
    const float xn0 = __shfl(X0[Q], ln.j), xn1 = __shfl(X1[Q], ln.j), xn2 = __shfl(X2[Q], ln.j), xn3 = __shfl(X3[Q], ln.j);

    // The input transform reads the frames through E = x3 - x1 and F = x2 - x0 (kept in place of x3 and x0): t3/t4 = E +- 2F,
    // t0 = E - 4 x1, t5 = -F - x2/4 are one fma each, t1 = (E + 4F) - 3(x1 + x2) and t2 = (E - 4F) + 3(x2 - x1) three.
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        X3[k] = X3[k] - X1[k];
        X0[k] = X2[k] - X0[k];
    }
    float poison;
    {   float p0 = 0.f, p1 = 0.f;                         // non-finite input: fft_wave.hpp poison_acc
        poison_acc<Q>(p0, p1, X3);
        poison_acc<Q>(p0, p1, X0);
        poison = poison_nyq(p0, p1, xn0, xn1, xn2, xn3);
    }
    ring.c0 = lds4u(ring.a_cur);
    ring.c1 = lds4u(ring.a_cur + 1024);

    // ---- encoder 0 as one F(4,3) tile, encoder 1 fed part by part ----------------------------------------------------------
    // (E = X3, F = X0, x1 = X1, x2 = X2 from here on.)
    f32x4 Z0[4], Z1[4];
    zero<4>(Z0);                                           // (the bias comes last: front_common.hpp add_bias)
    zero<4>(Z1);
#pragma clang loop unroll(disable)
    for (int it = 0; it < P / PB; ++it) {
        f32x4 Ykeep[RB];                                   // 16 kHz: y3 of the even part waits for the odd part's
        static_for<0, PB>([&](auto pc) VAD_INLINE {
            constexpr int pb = decltype(pc)::value;
            const int row0 = 16 * RB * (it * PB + pb);
            const float *wn = tabn + tb.w_nyq + row0;       // [tap][row]
            f32x4 Y0[RB], Y1[RB], Y2[RB], Y3[RB];          // m1, m2, m3, m4, then the four frame outputs
            zero<RB>(Y0);
            zero<RB>(Y1);
            zero<RB>(Y2);
            zero<RB>(Y3);
            const Coef ka = opaque_coef<kF4, kFm3>(), kb = opaque_coef<kFm4, kF3>(), kc = opaque_coef<kF2, kFm2>(),
                       kd = opaque_coef<kFm4, kFm025>();
            auto t1 = [&](int s) VAD_INLINE { return fmaf(fmaf(X2[s], ka.p1, X1[s]), ka.b, fmaf(X0[s], ka.a, X3[s])); };   // (E + 4F) - 3(x1 + x2)
            auto t2 = [&](int s) VAD_INLINE { return fmaf(fmaf(X1[s], kb.m1, X2[s]), kb.b, fmaf(X0[s], kb.a, X3[s])); };   // (E - 4F) + 3(x2 - x1)
            auto t3 = [&](int s) VAD_INLINE { return fmaf(X0[s], kc.a, X3[s]); };                                          // E + 2F
            auto t4 = [&](int s) VAD_INLINE { return fmaf(X0[s], kc.b, X3[s]); };                                          // E - 2F
            auto t0 = [&](int s) VAD_INLINE { return fmaf(X1[s], kd.a, X3[s]); };                                          // x3 - 5 x1 = E - 4 x1
            auto t5 = [&](int s) VAD_INLINE { return fmaf(X2[s], kd.b, -X0[s]); };                                         // x0 - 1.25 x2 = -F - x2/4
            gemm_b<RB, KG0, 2>(Y0, t1, ring);
            gemm_b<RB, KG0, 2>(Y1, t2, ring);
            gemm_b<RB, KG0, 2>(Y2, t3, ring);
            gemm_b<RB, KG0, 2>(Y3, t4, ring);
#pragma unroll
            for (int m = 0; m < RB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sm = Y0[m][r] + Y1[m][r], df = Y0[m][r] - Y1[m][r];
                    const float s2 = Y2[m][r] + Y3[m][r], d2 = Y2[m][r] - Y3[m][r];
                    Y0[m][r] = sm + s2;
                    Y1[m][r] = fmaf(2.f, d2, df);
                    Y2[m][r] = fmaf(4.f, s2, sm);
                    Y3[m][r] = fmaf(8.f, d2, df);
                }
            gemm_b<RB, KG0, 2>(Y0, t0, ring);
            gemm_b<RB, KG0, 2>(Y3, t5, ring);
            nyq_update<RB>(Y0, xn0, wn + 128, ln);
            nyq_update<RB>(Y0, xn1, wn + 256, ln);
            nyq_update<RB>(Y1, xn0, wn, ln);
            nyq_update<RB>(Y1, xn1, wn + 128, ln);
            nyq_update<RB>(Y1, xn2, wn + 256, ln);
            nyq_update<RB>(Y2, xn1, wn, ln);
            nyq_update<RB>(Y2, xn2, wn + 128, ln);
            nyq_update<RB>(Y2, xn3, wn + 256, ln);
            nyq_update<RB>(Y3, xn2, wn, ln);
            nyq_update<RB>(Y3, xn3, wn + 128, ln);
            add_bias<RB>(Y0, tab + tb.b_e0 + row0, ln);
            add_bias<RB>(Y1, tab + tb.b_e0 + row0, ln);
            add_bias<RB>(Y2, tab + tb.b_e0 + row0, ln);
            add_bias<RB>(Y3, tab + tb.b_e0 + row0, ln);
            relu<RB>(Y0);
            relu<RB>(Y1);
            relu<RB>(Y2);
            relu<RB>(Y3);
            if constexpr (Q == 32) {
                // 8 k-steps (one K32 step) per (tap, part): two taps share a unit
                auto two = [](const f32x4 (&A)[RB], const f32x4 (&B)[RB], int s) VAD_INLINE {
                    return s < 8 ? A[s >> 2][s & 3] : B[(s - 8) >> 2][s & 3];
                };
                auto e1a = [&](int s) VAD_INLINE { return two(Y0, Y1, s); };         // out 0: tap 1 <- y0 | tap 2 <- y1
                auto e1b = [&](int s) VAD_INLINE { return two(Y1, Y2, s); };         // out 1: tap 0 <- y1 | tap 1 <- y2
                auto e1c = [&](int s) VAD_INLINE { return two(Ykeep, Y3, s); };      // out 1: tap 2 <- y3, both parts
                gemm_b<4, 4, 2>(Z0, e1a, ring);
                if constexpr (pb == 0) {
                    gemm_b<4, 4, 2>(Z1, e1b, ring);
#pragma unroll
                    for (int m = 0; m < RB; ++m) Ykeep[m] = Y3[m];
                } else {
                    gemm_b<4, 4, 2>(Z1, e1b, ring);
                    gemm_b<4, 4, 2>(Z1, e1c, ring);
                }
            } else {
                auto o0 = [&](int s) VAD_INLINE { return Y0[s >> 2][s & 3]; };
                auto o1 = [&](int s) VAD_INLINE { return Y1[s >> 2][s & 3]; };
                auto o2 = [&](int s) VAD_INLINE { return Y2[s >> 2][s & 3]; };
                auto o3 = [&](int s) VAD_INLINE { return Y3[s >> 2][s & 3]; };
                gemm_b<4, 4, 2>(Z0, o0, ring);         // out 0, tap 1 <- y0
                gemm_b<4, 4, 2>(Z0, o1, ring);         // out 0, tap 2 <- y1
                gemm_b<4, 4, 2>(Z1, o1, ring);         // out 1, tap 0 <- y1
                gemm_b<4, 4, 2>(Z1, o2, ring);         // out 1, tap 1 <- y2
                gemm_b<4, 4, 2>(Z1, o3, ring);         // out 1, tap 2 <- y3
            }
        });
    }
    add_bias<4>(Z0, tab + tb.b_e1, ln);
    add_bias<4>(Z1, tab + tb.b_e1, ln);
    relu<4>(Z0);
    relu<4>(Z1);
    B9_TRACE(3);

    // ---- enc2 (T 2 -> 1, stride 2: taps 1,2 see enc1 outputs 0,1), enc3 (T = 1: centre tap only), W_ih -----------------
    f32x4 Vv[4];
    auto bZ0 = [&](int s) VAD_INLINE { return Z0[s >> 2][s & 3]; };
    auto bZ1 = [&](int s) VAD_INLINE { return Z1[s >> 2][s & 3]; };
    auto bV = [&](int s) VAD_INLINE { return Vv[s >> 2][s & 3]; };
    init_bias<4>(Vv, tab + tb.b_e2, ln);
    gemm_b<4, 4, 2>(Vv, bZ0, ring);
    gemm_b<4, 4, 2>(Vv, bZ1, ring);
    relu<4>(Vv);
    f32x4 Fe[8];
    auto bF = [&](int s) VAD_INLINE { return Fe[s >> 2][s & 3]; };
    init_bias<8>(Fe, tab + tb.b_e3, ln);
    gemm_b<8, 4, 2>(Fe, bV, ring);
    relu<8>(Fe);
    poison_into(Fe[0], poison);
    B9_TRACE(4);

    // LSTM input-gate pre-activations, one gate (8 row blocks) at a time, stored in D-fragment order: gates 0..2 share a
    // loop body, the last gate knows that the program ends
    float *gxt = a.gx + ((size_t)(ln.st * a.nt + ln.tl) * 32) * 256 + ln.lane * 4;
    const float *bg = tab + tb.b_g;
#pragma clang loop unroll(disable)
    for (int q = 0; q < 3; ++q) {
        f32x4 G[8];
        init_bias<8>(G, bg, ln);
        gemm_b<8, 8, 2>(G, bF, ring);
        if (ln.tile_valid) {
#pragma unroll
            for (int m = 0; m < 8; ++m) *reinterpret_cast<f32x4 *>(gxt + (size_t)m * 256) = G[m];
        }
        gxt += 8 * 256;
        bg += 128;
    }
    {
        f32x4 G[8];
        init_bias<8>(G, bg, ln);
        gemm_b<8, 8, 0>(G, bF, ring);
        if (ln.tile_valid) {
#pragma unroll
            for (int m = 0; m < 8; ++m) *reinterpret_cast<f32x4 *>(gxt + (size_t)m * 256) = G[m];
        }
    }
#if VAD_TRACE
    B9_TRACE(5);
    if (a.trace && threadIdx.x == 0) {
        a.trace[(size_t)blockIdx.x * 16 + 9] = __builtin_readcyclecounter();
        a.trace[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        a.trace[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
#endif
}

}  // namespace

template <typename PcmT>
hipError_t launch_front_b9(int sr, const FrontArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    const unsigned grid = (unsigned)((total + kWaves - 1) / kWaves);
    // a.dec == 2, 3: 32 / 48 kHz input, decimation folded into the loads (fft_wave.hpp load_slice; 16 kHz net only)
    if (a.dec > 1 && (sr != 16000 || a.dec > 3)) return hipErrorInvalidValue;
    if (a.dec == 3) hipLaunchKernelGGL((front_b9_kernel<32, PcmT, 3>), dim3(grid), dim3(64 * kWaves), 0, s, a);
    else if (a.dec == 2) hipLaunchKernelGGL((front_b9_kernel<32, PcmT, 2>), dim3(grid), dim3(64 * kWaves), 0, s, a);
    else if (sr == 16000) hipLaunchKernelGGL((front_b9_kernel<32, PcmT, 1>), dim3(grid), dim3(64 * kWaves), 0, s, a);
    else hipLaunchKernelGGL((front_b9_kernel<16, PcmT, 1>), dim3(grid), dim3(64 * kWaves), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_front_b9<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_front_b9<int16_t>(int, const FrontArgs &, hipStream_t);

}  // namespace vad
