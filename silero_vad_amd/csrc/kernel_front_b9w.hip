// kernel_front_b9w.hip -- the bf16 x 9 frontend in its WIDE form: the function, the arithmetic and the BITS of kernel_front_b9.hip
// (framing, reflect pad, window, 4 x real FFT magnitude, encoder 0 as one F(4,3) tile, encoders 1-3, W_ih; every matrix product as
// nine exact bf16 piece products on v_mfma_f32_16x16x32_bf16, fp32 accumulation), with the loop nest of encoder 0 turned inside out.
//
// Why.  The narrow form walks encoder 0 row part by row part (32 rows at 16 kHz) because a 243-register wave has room for one
// part's accumulators beside the tile's 132 magnitudes -- and so forms the F(4,3) input transform of every K32 step and splits it
// into bf16 pieces once PER PART: 4 x at 16 kHz.  That recomputation is more than half of the 13.9 k VALU instructions per tile
// that the narrow kernel fails to hide beside its 62 k matrix-pipe cycles (profiles/r03p_front_bf16x9.md: the two add).  Here a
// wave owns a SIMD alone (one 4-wave workgroup per CU, up to 512 registers per lane: arch VGPRs for magnitudes, operands and
// fragments, accumulation VGPRs for the 128 + 32 accumulator registers): encoder 0 runs matrix by matrix over ALL 128 rows -- one
// input transform and one split per K32 step, 72 MFMAs behind each -- and the NEXT step's pieces are formed between the MFMAs of
// the current one (the VALU rides beside the bf16 matrix pipe: profiles/r03a_issue_pipes2.md, r03p_pipes3.md; no packed fp32).
//
// Same bits.  An accumulator sees exactly the MFMAs it sees in the narrow program, in the same order (K32 steps ascending; per
// step piece pa of A against pieces 0, 1, 2 of B); m1..m4 are folded into the four frame outputs by the same expressions, m0 and
// m5 accumulate onto y0 and y3 through the C operand, encoder 1 consumes the parts in the narrow program's order.  The weight
// image is a permutation of the narrow image's 1 KiB fragments (layout.hpp "WIDE program").  tests: test_front_b9_wide_equals_narrow.
// (reference: the same lines as kernel_front_f43.hip.)
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_wave.hpp"
#include "front_common.hpp"

namespace vad {
namespace {

constexpr int kWavesW = 4;                                  // one workgroup per CU, one wave (= tile) per SIMD
constexpr int kUnitBytesW = (int)vadl::kW9UnitHalfs * 2;    // 24 fragments of 1 KiB
constexpr int kShareW = kUnitBytesW / kWavesW;              // a wave's share of a unit's DMA: 6 KiB
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
// pair d (values 2d, 2d + 1) of a K32 step's 8 B values -> dword d of the three piece registers
template <class F>
__device__ __forceinline__ void split_pair(u32x4 (&bp)[3], int d, F f) {
    unsigned p0, p1, p2;
    if (VAD_ABLATE & 64) {                 // timing only: no split
        p0 = __float_as_uint(f(2 * d));
        p1 = __float_as_uint(f(2 * d + 1));
        p2 = p0 ^ p1;
    } else
    split3(f(2 * d), f(2 * d + 1), p0, p1, p2);
    bp[0][d] = p0;
    bp[1][d] = p1;
    bp[2][d] = p2;
}
template <class F>
__device__ __forceinline__ void split_step(u32x4 (&bp)[3], F f) {
#pragma unroll
    for (int d = 0; d < 4; ++d) split_pair(bp, d, f);
}

// ---- the weight ring (3 slots of one 24 KiB unit, shared by the workgroup's 4 waves) ---------------------------------------------
struct RingW {
    unsigned a_cur, a_nxt, a_far;       // LDS byte address of this lane's first A fragment in the slot of unit u, u+1, u+2
    unsigned d_cur, d_nxt, d_far;       // wave-uniform: where this wave's share of a unit lands in those slots
    const char *src;                    // wave-uniform: this wave's share of the next unit to request
    unsigned voff;                      // lane * 16
    u32x4 c0, c1;                       // the two A fragments of the next sub-step
};
// this wave's share (6 x 1 KiB) of the next unit -> the slot everyone has left.  LDS destination = M0 + instruction offset + lane * 16
__device__ __forceinline__ void ring_request(RingW &r) {
    if (VAD_ABLATE & 8) return;
    unsigned keep_m0;                      // M0 is restored: the compiler may keep its own value there
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %4\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %4 offset:2048\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(r.voff), "s"(r.src), "s"(r.d_far), "s"(r.src + 3072), "s"(r.d_far + 3072u) : "memory");
    r.src += kUnitBytesW;
}
__device__ __forceinline__ void ring_rotate(RingW &r) {
    const unsigned a = r.a_cur, d = r.d_cur;
    r.a_cur = r.a_nxt; r.a_nxt = r.a_far; r.a_far = a;
    r.d_cur = r.d_nxt; r.d_nxt = r.d_far; r.d_far = d;
}
// in the middle of a unit: this wave's share of the next unit has landed; then everyone's, and everyone has left the previous unit
template <int AFTER>
__device__ __forceinline__ void ring_mid(RingW &r) {
    if constexpr (AFTER >= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(VAD_ABLATE & 1)) __builtin_amdgcn_s_barrier();      // (bare: the fragment reads in flight belong to the current slot)
        asm volatile("" ::: "memory");
        if constexpr (AFTER >= 2) ring_request(r);
    }
}

// One unit of the WIDE encoder-0 program: a K32 step of one F(4,3) matrix against all 8 row blocks = 12 sub-steps (A piece pa x
// row-block pair) of 6 MFMAs; `bp` holds the step's B pieces.  `next(d)` forms pair d of the NEXT unit's pieces: it is called in the
// sub-steps 1..4, inside the scheduling region of that sub-step's MFMAs, so that its VALU instructions issue beside the matrix pipe.
template <int AFTER, class NEXT>
__device__ __forceinline__ void unit_w(f32x4 (&acc)[8], const u32x4 (&bp)[3], NEXT next, RingW &r) {
    static_for<0, 12>([&](auto qc) VAD_INLINE {
        constexpr int q = decltype(qc)::value, pair = q % 4;
        if constexpr (q == 6) ring_mid<AFTER>(r);
        u32x4 n0 = r.c0, n1 = r.c1;
        if constexpr (VAD_ABLATE & 128) {          // timing only: no fragment reads
        } else if constexpr (q + 1 < 12) {
            n0 = lds4u(r.a_cur + (2 * (q + 1)) * 1024);
            n1 = lds4u(r.a_cur + (2 * (q + 1) + 1) * 1024);
        } else if constexpr (AFTER >= 1) {
            n0 = lds4u(r.a_nxt);
            n1 = lds4u(r.a_nxt + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (q >= 1 && q <= 4) next(q - 1);
#pragma unroll
        for (int pb = 0; pb < 3; ++pb) {
            acc[2 * pair + 0] = mfma_b(r.c0, bp[pb], acc[2 * pair + 0]);
            acc[2 * pair + 1] = mfma_b(r.c1, bp[pb], acc[2 * pair + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        r.c0 = n0;
        r.c1 = n1;
    });
    ring_rotate(r);
}

// A segment of the narrow program (encoders 1-3, W_ih: unit format [step 4][piece 3][row block 2][lane][8]), as kernel_front_b9.hip:
// M row blocks x KG fp32 k-groups = KG/2 K32 steps; a step = 1 K32 step x 2 row blocks = 18 MFMAs; the 8 B values of a K32 step are
// split when its first row-block pair begins and serve all M/2 pairs.
template <int M, int KG, int AFTER, class BF>
__device__ __forceinline__ void gemm_b(f32x4 (&acc)[M], BF bfun, RingW &r) {
    constexpr int H = M / 2, NSTEPS = (KG / 2) * H, NU = NSTEPS / 4;
    static_assert(KG % 2 == 0 && NSTEPS % 4 == 0 && M % 2 == 0, "segments are whole units");
    u32x4 bp[3];
    static_for<0, NU>([&](auto uc) VAD_INLINE {
        constexpr int u = decltype(uc)::value, after = (NU - 1 - u) + AFTER;
        static_for<0, 4>([&](auto sc_) VAD_INLINE {
            constexpr int st = decltype(sc_)::value, i = u * 4 + st, kp = i / H, mp = 2 * (i % H);
            if constexpr (st == 2) ring_mid<after>(r);
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
                for (int pb = 0; pb < 3; ++pb) {
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
__global__ void __launch_bounds__(64 * kWavesW, 1) front_b9w_kernel(const FrontArgs a) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int RB = w_rb(Q), P = w_parts(Q), KP = w9w_kp(Q);
    static_assert(w9w_tail0(Q) + 20 == w9w_units(Q), "program mismatch");
    // LDS: [biases + head + Nyquist weights: NS floats][ring 3 x 24 KiB]; the FFT's tables (window, twiddles) sit in ring slot 2 until
    // the first request into it (middle of unit 0: behind a barrier every wave reaches only after its FFT)
    constexpr int NS = tb.window + (tb.total - tb.w_nyq), NF = tb.w_nyq - tb.window, UF = kUnitBytesW / 4;
    static_assert(tb.window % 4 == 0 && tb.w_nyq % 4 == 0 && NS % 4 == 0 && NF <= UF && tb.total % 4 == 0, "table split");
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
    long wt = (long)blockIdx.x * kWavesW + ln.wave;
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

    RingW ring;
    {
        const unsigned slot0 = (unsigned)(size_t)((__attribute__((address_space(3))) float *)(lds + NS));
        ring.voff = ln.lane * 16;
        ring.a_cur = slot0 + ring.voff;
        ring.a_nxt = ring.a_cur + kUnitBytesW;
        ring.a_far = ring.a_cur + 2 * kUnitBytesW;
        // the two priming requests go to slots 0 and 1: start rotated by two, so that "far" is slot 0 first, then slot 1
        ring.d_far = slot0 + (unsigned)ln.wave * (unsigned)kShareW;
        ring.d_cur = ring.d_far + kUnitBytesW;
        ring.d_nxt = ring.d_far + 2 * kUnitBytesW;
        ring.src = reinterpret_cast<const char *>(a.wfront) + ln.wave * kShareW;
        ring_request(ring);                               // unit 0 -> slot 0
        {   const unsigned d = ring.d_far; ring.d_far = ring.d_cur; ring.d_cur = ring.d_nxt; ring.d_nxt = d; }
        ring_request(ring);                               // unit 1 -> slot 1
        {   const unsigned d = ring.d_far; ring.d_far = ring.d_cur; ring.d_cur = ring.d_nxt; ring.d_nxt = d; }
        // now d_far = slot 2 (unit 2's), d_cur = slot 0, d_nxt = slot 1
    }
    {   // tables -> LDS: all loads of a thread are issued before the first is stored
        constexpr int NT = 64 * kWavesW, NV = tb.total / 4, PER = (NV + NT - 1) / NT;
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

    // ---- the 4 frames ------------------------------------------------------------------------------------------------------
    // one FFT body for the four frames (code size: the instruction cache is 64 KB per two CUs): the magnitude arrays are a shift
    // register, every iteration moves the frames down one place and transforms the next frame into the top one
    float X0[Q + 1], X1[Q + 1], X2[Q + 1], X3[Q + 1];
#pragma unroll
    for (int k = 0; k <= Q; ++k) X1[k] = X2[k] = X3[k] = 0.f;
#pragma clang loop unroll(disable)
    for (int v = 0; v < 4; ++v) {
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
    // |Y_nyq| of chunk j lives in lane group 0 (X[Q]); every lane of the chunk needs it
    const float xn0 = __shfl(X0[Q], ln.j), xn1 = __shfl(X1[Q], ln.j), xn2 = __shfl(X2[Q], ln.j), xn3 = __shfl(X3[Q], ln.j);
    // The input transform reads the frames through E = x3 - x1 and F = x2 - x0 (kept in place of x3 and x0), as the narrow kernel does
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        X3[k] = X3[k] - X1[k];
        X0[k] = X2[k] - X0[k];
    }
    ring.c0 = lds4u(ring.a_cur);
    ring.c1 = lds4u(ring.a_cur + 1024);

    // t_j at fp32 k-step s, j in program order (U1, U2, U3, U4, U0, U5): the narrow kernel's expressions, term for term
    auto tval = [&](auto jc, int s) VAD_INLINE -> float {
        constexpr int j = decltype(jc)::value;
        if constexpr (j == 0) return fmaf(fmaf(X2[s], 1.0f, X1[s]), -3.0f, fmaf(X0[s], 4.0f, X3[s]));       // (E + 4F) - 3(x1 + x2)
        else if constexpr (j == 1) return fmaf(fmaf(X1[s], -1.0f, X2[s]), 3.0f, fmaf(X0[s], -4.0f, X3[s])); // (E - 4F) + 3(x2 - x1)
        else if constexpr (j == 2) return fmaf(X0[s], 2.0f, X3[s]);                                          // E + 2F
        else if constexpr (j == 3) return fmaf(X0[s], -2.0f, X3[s]);                                         // E - 2F
        else if constexpr (j == 4) return fmaf(X1[s], -4.0f, X3[s]);                                         // E - 4 x1
        else return fmaf(X2[s], -0.25f, -X0[s]);                                                             // -F - x2 / 4
    };

    // ---- encoder 0: six matrices over all 128 rows -----------------------------------------------------------------------------
    f32x4 Y0[8], Y1[8], Y2[8], Y3[8];                      // m1, m2, m3, m4, then the four frame outputs
    init_bias<8>(Y0, tab + tb.b_e0, ln);
    zero<8>(Y1);
    zero<8>(Y2);
    zero<8>(Y3);
    u32x4 bpA[3], bpB[3];                                  // the pieces of the current and of the next K32 step
    split_step(bpA, [&](int e) VAD_INLINE { return tval(std::integral_constant<int, 0>{}, e); });
    static_for<0, 6 * KP>([&](auto ic) VAD_INLINE {
        constexpr int i = decltype(ic)::value, j = i / KP, jn = (i + 1) / KP, kn = (i + 1) % KP;
        auto &cur = (i & 1) ? bpB : bpA;
        auto &nxt = (i & 1) ? bpA : bpB;
        auto next = [&](int d) VAD_INLINE {
            if constexpr (i + 1 < 6 * KP)
                split_pair(nxt, d, [&](int e) VAD_INLINE { return tval(std::integral_constant<int, jn>{}, kn * 8 + e); });
        };
        if constexpr (i == 4 * KP) {
            // m1..m4 are complete: fold them into the four frame outputs (the narrow kernel's expressions)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sm = Y0[m][r] + Y1[m][r], df = Y0[m][r] - Y1[m][r];
                    const float s2 = Y2[m][r] + Y3[m][r], d2 = Y2[m][r] - Y3[m][r];
                    Y0[m][r] = sm + s2;
                    Y1[m][r] = fmaf(2.f, d2, df);
                    Y2[m][r] = fmaf(4.f, s2, sm);
                    Y3[m][r] = fmaf(8.f, d2, df);
                }
        }
        if constexpr (j == 0 || j == 4) unit_w<2>(Y0, cur, next, ring);       // m1; m0 onto y0
        else if constexpr (j == 1) unit_w<2>(Y1, cur, next, ring);
        else if constexpr (j == 2) unit_w<2>(Y2, cur, next, ring);
        else unit_w<2>(Y3, cur, next, ring);                                  // m4; m5 onto y3
    });
    {   // Nyquist bin (exact fp32 rank-1 updates), ReLU
        const float *wn = tabn + tb.w_nyq;                  // [tap][row]
        nyq_update<8>(Y0, xn0, wn + 128, ln);
        nyq_update<8>(Y0, xn1, wn + 256, ln);
        nyq_update<8>(Y1, xn0, wn, ln);
        nyq_update<8>(Y1, xn1, wn + 128, ln);
        nyq_update<8>(Y1, xn2, wn + 256, ln);
        nyq_update<8>(Y2, xn1, wn, ln);
        nyq_update<8>(Y2, xn2, wn + 128, ln);
        nyq_update<8>(Y2, xn3, wn + 256, ln);
        nyq_update<8>(Y3, xn2, wn, ln);
        nyq_update<8>(Y3, xn3, wn + 128, ln);
        relu<8>(Y0);
        relu<8>(Y1);
        relu<8>(Y2);
        relu<8>(Y3);
    }

    // ---- encoder 1, part by part in the narrow program's order ---------------------------------------------------------------
    f32x4 Z0[4], Z1[4];
    init_bias<4>(Z0, tab + tb.b_e1, ln);
    init_bias<4>(Z1, tab + tb.b_e1, ln);
    static_for<0, P>([&](auto pc) VAD_INLINE {
        constexpr int p = decltype(pc)::value, m0 = RB * p;
        if constexpr (Q == 32) {
            // 8 k-steps (one K32 step) per (tap, part): two taps share a unit
            auto e1a = [&](int s) VAD_INLINE { return s < 8 ? Y0[m0 + (s >> 2)][s & 3] : Y1[m0 + ((s - 8) >> 2)][s & 3]; };   // out 0: tap 1 <- y0 | tap 2 <- y1
            auto e1b = [&](int s) VAD_INLINE { return s < 8 ? Y1[m0 + (s >> 2)][s & 3] : Y2[m0 + ((s - 8) >> 2)][s & 3]; };   // out 1: tap 0 <- y1 | tap 1 <- y2
            gemm_b<4, 4, 2>(Z0, e1a, ring);
            gemm_b<4, 4, 2>(Z1, e1b, ring);
            if constexpr (p & 1) {                          // out 1: tap 2 <- y3 of the even part before | of this part
                auto e1c = [&](int s) VAD_INLINE { return s < 8 ? Y3[m0 - RB + (s >> 2)][s & 3] : Y3[m0 + ((s - 8) >> 2)][s & 3]; };
                gemm_b<4, 4, 2>(Z1, e1c, ring);
            }
        } else {
            auto o0 = [&](int s) VAD_INLINE { return Y0[m0 + (s >> 2)][s & 3]; };
            auto o1 = [&](int s) VAD_INLINE { return Y1[m0 + (s >> 2)][s & 3]; };
            auto o2 = [&](int s) VAD_INLINE { return Y2[m0 + (s >> 2)][s & 3]; };
            auto o3 = [&](int s) VAD_INLINE { return Y3[m0 + (s >> 2)][s & 3]; };
            gemm_b<4, 4, 2>(Z0, o0, ring);         // out 0, tap 1 <- y0
            gemm_b<4, 4, 2>(Z0, o1, ring);         // out 0, tap 2 <- y1
            gemm_b<4, 4, 2>(Z1, o1, ring);         // out 1, tap 0 <- y1
            gemm_b<4, 4, 2>(Z1, o2, ring);         // out 1, tap 1 <- y2
            gemm_b<4, 4, 2>(Z1, o3, ring);         // out 1, tap 2 <- y3
        }
    });
    relu<4>(Z0);
    relu<4>(Z1);

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

    // LSTM input-gate pre-activations, one gate (8 row blocks) at a time, stored in D-fragment order
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
}

}  // namespace

template <typename PcmT>
hipError_t launch_front_b9w(int sr, const FrontArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    const unsigned grid = (unsigned)((total + kWavesW - 1) / kWavesW);
    // a.dec == 2, 3: 32 / 48 kHz input, decimation folded into the loads (fft_wave.hpp load_slice; 16 kHz net only)
    if (a.dec > 1 && (sr != 16000 || a.dec > 3)) return hipErrorInvalidValue;
    if (a.dec == 3) hipLaunchKernelGGL((front_b9w_kernel<32, PcmT, 3>), dim3(grid), dim3(64 * kWavesW), 0, s, a);
    else if (a.dec == 2) hipLaunchKernelGGL((front_b9w_kernel<32, PcmT, 2>), dim3(grid), dim3(64 * kWavesW), 0, s, a);
    else if (sr == 16000) hipLaunchKernelGGL((front_b9w_kernel<32, PcmT, 1>), dim3(grid), dim3(64 * kWavesW), 0, s, a);
    else hipLaunchKernelGGL((front_b9w_kernel<16, PcmT, 1>), dim3(grid), dim3(64 * kWavesW), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_front_b9w<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_front_b9w<int16_t>(int, const FrontArgs &, hipStream_t);

}  // namespace vad
