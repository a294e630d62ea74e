// kernel_front_f43.hip -- the product's fp32 frontend: the time-parallel part of the Silero-VAD hot path
//   PCM -> framing + right reflect pad -> Hann window -> 4 x real FFT magnitude -> 4 x ReLU(Conv1d k=3)
//       -> W_ih * feat + (b_ih + b_hh)  => gx
// (same function as kernel_front.hip / kernel_front_wino.hip) with two changes of form:
//
//  1. Encoder 0 -- a k = 3, stride-1 conv over the chunk's 4 STFT frames, more than half of the matrix work -- is ONE
//     Winograd F(4,3) tile: 6 GEMMs of [128 x 4Q] instead of 10 tap-GEMMs (8 with two F(2,3) tiles).  layout.hpp,
//     "F(4,3) Winograd frontend image", has the algebra; per 16-chunk tile the kernel issues 3 456 (16 kHz) /
//     2 688 (8 kHz) v_mfma_f32_16x16x4_f32 instead of 4 480 / 3 456.  Everything stays fp32: the transform changes
//     WHICH fp32 sums are formed (like the rFFT does for the STFT), not their precision; the transformed weights are
//     formed in double on the host and rounded once.
//
//  2. The code is a set of LOOPS -- one FFT body for the 4 frames, one body for a pair of encoder-0 row parts, one body
//     for three of the four LSTM gates -- instead of ~100 KB of straight-line code.  The instruction cache (64 KB per
//     two CUs) cannot hold the straight-line form: it is re-streamed from L2 by every workgroup generation, which is
//     harmless where the sequential prefetcher keeps up (MFMA-dense code consumes 8 bytes per 32 cycles) and is not
//     where it does not (the VALU-dense FFT), and on about one MI355X box in ten the L2 instruction-fetch latency is
//     long enough to cost 1.4 ms per launch (profiles/r02i_pmc: SQ_IFETCH_LEVEL x3.5, SQC_ICACHE_MISSES_DUPLICATE x79,
//     everything else identical).  In loop form the kernel fits the cache.
//
// (reference: JIT!/vad/model/vad_annotator.py:58-67 framing, JIT!/vad/utils/pytorch_stft.py:17-34 STFT,
//  JIT!/vad/utils/model_utils.py:19-25 encoder -- encoder 0 is JIT!/torch/nn/modules/conv/___torch_mangle_10.py:29 --,
//  the W_ih half of aten::lstm_cell JIT!/torch/nn/modules/rnn.py:69.)
//
// Structure (one wave = 16 chunks, 4 waves per workgroup, 2 workgroups per CU, as in the other frontends):
//   * the weight image is the program: whole 16 KiB units in the order they are consumed, streamed through a 3-slot
//     LDS ring by global_load_lds_dwordx4.  The ring state (slot addresses, next source unit) is runtime data that
//     rotates with every unit, so that loop bodies can be reused at any ring phase;
//   * the barrier that makes unit u+1 visible sits in the middle of unit u, and A-fragment reads are carried across unit
//     boundaries ("seamless" pipeline);
//   * per row part (RB = 64/Q row blocks of encoder 0's 128 output rows): m1..m4 are accumulated, folded into the four
//     frame outputs, m0 and m5 are accumulated straight onto y0 and y3 through the MFMA's C operand; the Nyquist bin is
//     applied directly on the VALU (rank 1); the part's ReLU'd outputs are encoder 1's B operand for the part's 16 RB
//     input channels and are consumed at once (the chain layout of layout.hpp: no data movement between layers).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_wave.hpp"
#include "exact_front.hpp"
#include "front_common.hpp"

namespace vad {
namespace {

#ifndef VAD_BV_BATCH
#define VAD_BV_BATCH 1           // 1: a step's four B operands are formed first, then its 8 MFMAs issue back to back (gemm_r)
#endif
constexpr int kUnitBytes = (int)vadl::kWUnitFloats * 4;      // 16 blocks of 1 KiB
// ---- the weight ring -------------------------------------------------------------------------------------------------
struct Ring {
    unsigned a_cur, a_nxt, a_far;       // LDS byte address of this lane's first A fragment in the slot of unit u, u+1, u+2
    unsigned d_cur, d_nxt, d_far;       // wave-uniform: where this wave's share of a unit lands in those slots
    const float *src;                   // wave-uniform: this wave's share of the next unit to request
    unsigned voff;                      // lane * 16
    f32x4 c0, c1;                       // A fragments of the next step
};

// Request this wave's share (4 x 1 KiB) of the next unit into the slot everyone has left.  LDS destination = M0
// (wave-uniform base) + instruction offset + lane*16, the layout global_load_lds requires.  Issued from inline asm on
// purpose (kernel_front.hip ring_issue: the compiler would degrade every lgkmcnt wait while it knows of a pending LDS-DMA).
__device__ __forceinline__ void ring_request(Ring &r) {
    if (VAD_ABLATE & 8) return;
    unsigned keep_m0;                      // M0 is restored: the compiler may keep its own value there
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(r.voff), "s"(r.src), "s"(r.d_far) : "memory");
    r.src += vadl::kWUnitFloats;
}
__device__ __forceinline__ void ring_rotate(Ring &r) {
    const unsigned a = r.a_cur, d = r.d_cur;
    r.a_cur = r.a_nxt; r.a_nxt = r.a_far; r.a_far = a;
    r.d_cur = r.d_nxt; r.d_nxt = r.d_far; r.d_far = d;
}

// A segment of M row blocks x KG k-groups ([k-group][row block] blocks, whole units):
//     acc[m] += A[m][:, k] * B[k][:]   for all 4 KG k-steps;  bfun(s) = B-operand register of k-step s (compile-time s).
// A step = 2 row blocks x 4 k-steps = 8 MFMAs; a unit = 8 steps.  The A fragments of step i+1 are read from LDS before
// the MFMAs of step i are issued; in the middle of every unit the workgroup makes the NEXT unit visible (own share
// landed -> barrier) and requests the one after it into the slot everyone has left.  AFTER = program units that follow
// the segment (2 = "at least two"): the last unit of the program has nothing to wait for, the last two nothing to request.
template <int M, int KG, int AFTER, class BF>
__device__ __forceinline__ void gemm_r(f32x4 (&acc)[M], BF bfun, Ring &r) {
    constexpr int NSTEPS = KG * (M / 2), NU = NSTEPS / 8;
    static_assert(NSTEPS % 8 == 0 && M % 2 == 0, "segments are whole units");
    static_for<0, NU>([&](auto uc) VAD_INLINE {
        constexpr int u = decltype(uc)::value, after = (NU - 1 - u) + AFTER;
        static_for<0, 8>([&](auto sc_) VAD_INLINE {
            constexpr int st = decltype(sc_)::value;
            if constexpr (st == 4 && after >= 1) {
                // this wave's share of the next unit has landed; then everyone's, and everyone has left the previous unit.
                // A bare s_barrier (no lgkmcnt(0) fence): the fragment reads in flight belong to the current slot
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(VAD_ABLATE & 1)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (after >= 2) ring_request(r);
            }
            f32x4 n0 = r.c0, n1 = r.c1;
            if constexpr (st + 1 < 8) {
                n0 = lds4(r.a_cur + (2 * (st + 1)) * 1024);
                n1 = lds4(r.a_cur + (2 * (st + 1) + 1) * 1024);
            } else if constexpr (after >= 1) {
                n0 = lds4(r.a_nxt);
                n1 = lds4(r.a_nxt + 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int i = u * 8 + st, kg = i / (M / 2), mp = 2 * (i % (M / 2));
#if VAD_BV_BATCH
            // A VALU instruction between two fp32 MFMAs costs its issue plus a ~10-cycle switch (profiles/r03a_issue_pipes2.md):
            // form the step's four B operands first, then issue its 8 MFMAs back to back -- one switch per step instead of four
            float bv[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bv[ks] = bfun(kg * 4 + ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[mp + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.c0[ks], bv[ks], acc[mp + 0], 0, 0, 0);
                acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.c1[ks], bv[ks], acc[mp + 1], 0, 0, 0);
            }
#else
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float bv = bfun(kg * 4 + ks);
                acc[mp + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.c0[ks], bv, acc[mp + 0], 0, 0, 0);
                acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.c1[ks], bv, acc[mp + 1], 0, 0, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            r.c0 = n0;
            r.c1 = n1;
        });
        ring_rotate(r);
    });
}

#ifndef VAD_F43_GEMM_PRIO
#define VAD_F43_GEMM_PRIO 0        // A/B: issue priority of a wave from the end of its FFT passes on
#endif
#ifndef VAD_F43_FFT_PRIO
#define VAD_F43_FFT_PRIO 0         // A/B (tools/variants.py fftprio*): issue priority of a wave while it transforms a frame
#endif
template <int Q, typename PcmT, int DEC>
__global__ void __launch_bounds__(256, 2) front_f43_kernel(const FrontArgs a) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int TABF = (tb.total + 3) / 4 * 4;
    constexpr int RB = w_rb(Q), P = w_parts(Q), KG0 = Q / 4;
    constexpr int PB = Q == 32 ? 2 : 1;                   // row parts per loop body (16 kHz: an even and an odd one)
    static_assert(w4_tail0(Q) + 20 == w4_units(Q) && P % PB == 0, "program mismatch");
    __shared__ __attribute__((aligned(16))) float lds[TABF + 3 * (kUnitBytes / 4)];
    float *tab = lds;

    Lane ln;
    ln.lane = threadIdx.x & 63;
    ln.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ln.g = ln.lane >> 4;
    ln.j = ln.lane & 15;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    long wt = (long)blockIdx.x * 4 + ln.wave;
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
    // bring-up (tools/trace_f43.py): 32 slots of shader-clock timestamps per WAVE (16 + v: frame v's pass begins) -- 0 start, 1 tables + units 0, 1 in LDS, 2 + 2 v / 3 + 2 v samples
    // of frame v arrived / its FFT done, 10 encoders 0 + 1 done, 11 encoders 2 + 3 done, 12 end; 13 HW_ID, 14 XCC_ID
    VAD_WAVE_STAMP(ln, 0);
#endif

    Ring ring;
    {
        const unsigned slot0 = (unsigned)(size_t)((__attribute__((address_space(3))) float *)(lds + TABF));
        ring.voff = ln.lane * 16;
        ring.a_cur = slot0 + ring.voff;
        ring.a_nxt = ring.a_cur + kUnitBytes;
        ring.a_far = ring.a_cur + 2 * kUnitBytes;
        // the two priming requests go to slots 0 and 1: start rotated by two, so that "far" is slot 0 first, then slot 1
        ring.d_far = slot0 + (unsigned)ln.wave * 4096u;
        ring.d_cur = ring.d_far + kUnitBytes;
        ring.d_nxt = ring.d_far + 2 * kUnitBytes;
        ring.src = a.wfront + ln.wave * 1024;
        ring_request(ring);                               // unit 0 -> slot 0
        {   const unsigned d = ring.d_far; ring.d_far = ring.d_cur; ring.d_cur = ring.d_nxt; ring.d_nxt = d; }
        ring_request(ring);                               // unit 1 -> slot 1
        {   const unsigned d = ring.d_far; ring.d_far = ring.d_cur; ring.d_cur = ring.d_nxt; ring.d_nxt = d; }
        // now d_far = slot 2 (unit 2's), d_cur = slot 0, d_nxt = slot 1
    }
    {   // tables -> LDS: all loads of a thread are issued before the first is stored
        static_assert(tb.total % 4 == 0, "tables are copied as 16-byte vectors");
        constexpr int NV = tb.total / 4, PER = (NV + 255) / 256;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.tables);
        f32x4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 256;
            v[k] = src[i < NV ? i : NV - 1];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < NV) reinterpret_cast<f32x4 *>(tab)[i] = v[k];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // units 0 and 1 (and the tables) have landed
    __syncthreads();
    VAD_WAVE_STAMP(ln, 1);

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
        if (VAD_F43_FFT_PRIO) __builtin_amdgcn_s_setprio(VAD_F43_FFT_PRIO);
        fft_frame<Q, PcmT, DEC>(X3, v, a, tab, ln);
        if (VAD_F43_FFT_PRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (VAD_F43_GEMM_PRIO) __builtin_amdgcn_s_setprio(VAD_F43_GEMM_PRIO);
    // |Y_nyq| of chunk j lives in lane group 0 (X[Q]); every lane of the chunk needs it
    const float xn0 = __shfl(X0[Q], ln.j), xn1 = __shfl(X1[Q], ln.j), xn2 = __shfl(X2[Q], ln.j), xn3 = __shfl(X3[Q], ln.j);
    // chunks with an exactly silent frame beside one that is not (exact_front.hpp): listed for the fix-up pass (kernel_exact.hip), which
    // overwrites their columns of gx.  Four compares and scalar code; the branch is taken where digital silence begins or ends.
    if (!VAD_NO_EXACT && a.exact_list != nullptr) {
        const SilentMasks sm = silent_chunks(X0[0], X1[0], X2[0], X3[0]);
        const long left = (long)a.B - ln.st * 16;
        const unsigned take = (sm.edge | (a.gx_silent != nullptr ? sm.silent : 0u)) & (left >= 16 ? 0xffffu : ((1u << left) - 1u));
        if (take != 0 && ln.tile_valid) {             // wave-uniform; one atomic per tile, the lanes of the listed chunks fill their slots
            int base = 0;
            if (ln.lane == 0) base = atomicAdd(a.exact_list, __builtin_popcount(take));
            base = __shfl(base, 0);
            if (ln.g == 0 && ((take >> ln.j) & 1))
                a.exact_list[2 + base + __builtin_popcount(take & ((1u << ln.j) - 1u))] =
                    (int)(wt * 16 + ln.j) | (((sm.silent >> ln.j) & 1) ? kExactSilentBit : 0);
        }
    }

#if VAD_F43_EF
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
#else
    float poison;
    {   float p0 = 0.f, p1 = 0.f;
        poison_acc<Q>(p0, p1, X0);
        poison_acc<Q>(p0, p1, X1);
        poison_acc<Q>(p0, p1, X2);
        poison_acc<Q>(p0, p1, X3);
        poison = poison_nyq(p0, p1, xn0, xn1, xn2, xn3);
    }
#endif
    ring.c0 = lds4(ring.a_cur);
    ring.c1 = lds4(ring.a_cur + 1024);

    // ---- encoder 0 as one F(4,3) tile, encoder 1 fed part by part ----------------------------------------------------------
    f32x4 Z0[4], Z1[4];
    zero<4>(Z0);                                           // (the bias comes last: front_common.hpp add_bias)
    zero<4>(Z1);
#pragma clang loop unroll(disable)
    for (int it = 0; it < P / PB; ++it) {
        f32x4 Ykeep[RB];                                   // 16 kHz: y3 of the even part waits for the odd part's
        static_for<0, PB>([&](auto pc) VAD_INLINE {
            constexpr int pb = decltype(pc)::value;
            const int row0 = 16 * RB * (it * PB + pb);
            const float *wn = tab + tb.w_nyq + row0;       // [tap][row]
            f32x4 Y0[RB], Y1[RB], Y2[RB], Y3[RB];          // m1, m2, m3, m4, then the four frame outputs
            zero<RB>(Y0);
            zero<RB>(Y1);
            zero<RB>(Y2);
            zero<RB>(Y3);
#if VAD_F43_EF
            // (E = X3, F = X0, x1 = X1, x2 = X2 from here on)
            {   const Coef k = opaque_coef<kF4, kFm3>();
                gemm_r<RB, KG0, 2>(Y0, [&](int s) VAD_INLINE {
                    return fmaf(fmaf(X2[s], k.p1, X1[s]), k.b, fmaf(X0[s], k.a, X3[s])); }, ring);        // (E + 4F) - 3(x1 + x2)
                const Coef k2 = opaque_coef<kFm4, kF3>();
                gemm_r<RB, KG0, 2>(Y1, [&](int s) VAD_INLINE {
                    return fmaf(fmaf(X1[s], k2.m1, X2[s]), k2.b, fmaf(X0[s], k2.a, X3[s])); }, ring);     // (E - 4F) + 3(x2 - x1)
            }
            {   const Coef k = opaque_coef<kF2, kFm2>();
                gemm_r<RB, KG0, 2>(Y2, [&](int s) VAD_INLINE { return fmaf(X0[s], k.a, X3[s]); }, ring);  // E + 2F
                gemm_r<RB, KG0, 2>(Y3, [&](int s) VAD_INLINE { return fmaf(X0[s], k.b, X3[s]); }, ring);  // E - 2F
            }
#else
            {   const Coef k = opaque_coef<kFm4, kF4>();
                gemm_r<RB, KG0, 2>(Y0, [&](int s) VAD_INLINE {
                    return fmaf(fmaf(X1[s], k.p1, X0[s]), k.a, fmaf(X3[s], k.p1, X2[s])); }, ring);       // (x2+x3) - 4(x0+x1)
                gemm_r<RB, KG0, 2>(Y1, [&](int s) VAD_INLINE {
                    return fmaf(fmaf(X1[s], k.m1, X0[s]), k.b, fmaf(X2[s], k.m1, X3[s])); }, ring);       // (x3-x2) + 4(x0-x1)
            }
            {   const Coef k = opaque_coef<kF2, kFm2>();
                gemm_r<RB, KG0, 2>(Y2, [&](int s) VAD_INLINE {
                    return fmaf(fmaf(X0[s], k.m1, X2[s]), k.a, fmaf(X1[s], k.m1, X3[s])); }, ring);       // (x3-x1) + 2(x2-x0)
                const Coef k2 = opaque_coef<kF2, kFm2>();
                gemm_r<RB, KG0, 2>(Y3, [&](int s) VAD_INLINE {
                    return fmaf(fmaf(X0[s], k2.m1, X2[s]), k2.b, fmaf(X1[s], k2.m1, X3[s])); }, ring);    // (x3-x1) - 2(x2-x0)
            }
#endif
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
#if VAD_F43_EF
            {   const Coef k = opaque_coef<kFm4, kFm025>();
                gemm_r<RB, KG0, 2>(Y0, [&](int s) VAD_INLINE { return fmaf(X1[s], k.a, X3[s]); }, ring);              // x3 - 5 x1 = E - 4 x1
                gemm_r<RB, KG0, 2>(Y3, [&](int s) VAD_INLINE { return fmaf(X2[s], k.b, -X0[s]); }, ring);             // x0 - 1.25 x2 = -F - x2/4
            }
#else
            {   const Coef k = opaque_coef<kFm5, kFm125>();
                gemm_r<RB, KG0, 2>(Y0, [&](int s) VAD_INLINE { return fmaf(X1[s], k.a, X3[s]); }, ring);  // x3 - 5 x1
                gemm_r<RB, KG0, 2>(Y3, [&](int s) VAD_INLINE { return fmaf(X2[s], k.b, X0[s]); }, ring);  // x0 - 1.25 x2
            }
#endif
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
                // 8 k-steps per (tap, part): two taps share a unit
                auto two = [](const f32x4 (&A)[RB], const f32x4 (&B)[RB], int s) VAD_INLINE {
                    return s < 8 ? A[s >> 2][s & 3] : B[(s - 8) >> 2][s & 3];
                };
                gemm_r<4, 4, 2>(Z0, [&](int s) VAD_INLINE { return two(Y0, Y1, s); }, ring);     // out 0: tap 1 <- y0 | tap 2 <- y1
                gemm_r<4, 4, 2>(Z1, [&](int s) VAD_INLINE { return two(Y1, Y2, s); }, ring);     // out 1: tap 0 <- y1 | tap 1 <- y2
                if constexpr (pb == 0) {
#pragma unroll
                    for (int m = 0; m < RB; ++m) Ykeep[m] = Y3[m];
                } else {
                    gemm_r<4, 4, 2>(Z1, [&](int s) VAD_INLINE { return two(Ykeep, Y3, s); }, ring);   // out 1: tap 2 <- y3, both parts
                }
            } else {
                auto one = [](const f32x4 (&A)[RB], int s) VAD_INLINE { return A[s >> 2][s & 3]; };
                gemm_r<4, 4, 2>(Z0, [&](int s) VAD_INLINE { return one(Y0, s); }, ring);         // out 0, tap 1 <- y0
                gemm_r<4, 4, 2>(Z0, [&](int s) VAD_INLINE { return one(Y1, s); }, ring);         // out 0, tap 2 <- y1
                gemm_r<4, 4, 2>(Z1, [&](int s) VAD_INLINE { return one(Y1, s); }, ring);         // out 1, tap 0 <- y1
                gemm_r<4, 4, 2>(Z1, [&](int s) VAD_INLINE { return one(Y2, s); }, ring);         // out 1, tap 1 <- y2
                gemm_r<4, 4, 2>(Z1, [&](int s) VAD_INLINE { return one(Y3, s); }, ring);         // out 1, tap 2 <- y3
            }
        });
    }
    add_bias<4>(Z0, tab + tb.b_e1, ln);
    add_bias<4>(Z1, tab + tb.b_e1, ln);
    relu<4>(Z0);
    relu<4>(Z1);
    VAD_WAVE_STAMP(ln, 10);

    // ---- enc2 (T 2 -> 1, stride 2: taps 1,2 see enc1 outputs 0,1), enc3 (T = 1: centre tap only), W_ih -----------------
    f32x4 Vv[4];
    auto bZ0 = [&](int s) VAD_INLINE { return Z0[s >> 2][s & 3]; };
    auto bZ1 = [&](int s) VAD_INLINE { return Z1[s >> 2][s & 3]; };
    auto bV = [&](int s) VAD_INLINE { return Vv[s >> 2][s & 3]; };
    init_bias<4>(Vv, tab + tb.b_e2, ln);
    gemm_r<4, 4, 2>(Vv, bZ0, ring);
    gemm_r<4, 4, 2>(Vv, bZ1, ring);
    relu<4>(Vv);
    f32x4 Fe[8];
    auto bF = [&](int s) VAD_INLINE { return Fe[s >> 2][s & 3]; };
    init_bias<8>(Fe, tab + tb.b_e3, ln);
    gemm_r<8, 4, 2>(Fe, bV, ring);
    relu<8>(Fe);
    poison_into(Fe[0], poison);
    VAD_WAVE_STAMP(ln, 11);

    // LSTM input-gate pre-activations, one gate (8 row blocks) at a time, stored in D-fragment order: gates 0..2 share a
    // loop body, the last gate knows that the program ends
    float *gxt = a.gx + ((size_t)(ln.st * a.nt + ln.tl) * 32) * 256 + ln.lane * 4;
    const float *bg = tab + tb.b_g;
#pragma clang loop unroll(disable)
    for (int q = 0; q < 3; ++q) {
        f32x4 G[8];
        init_bias<8>(G, bg, ln);
        gemm_r<8, 8, 2>(G, bF, ring);
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
        gemm_r<8, 8, 0>(G, bF, ring);
        if (ln.tile_valid) {
#pragma unroll
            for (int m = 0; m < 8; ++m) *reinterpret_cast<f32x4 *>(gxt + (size_t)m * 256) = G[m];
        }
    }
#if VAD_TRACE
    VAD_WAVE_STAMP(ln, 12);
    if (a.trace && ln.lane == 0) {
        long long *tr = a.trace + ((size_t)blockIdx.x * 4 + ln.wave) * 32;
#pragma unroll
        for (int k = 0; k < 20; ++k) tr[k] = (long long)ln.ts[k];
        tr[13] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
        tr[14] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
        tr[15] = (long long)wall_clock64();
    }
#endif
}

}  // namespace

template <typename PcmT>
hipError_t launch_front_f43(int sr, const FrontArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    const unsigned grid = (unsigned)((total + 3) / 4);
    // a.dec == 2, 3: 32 / 48 kHz input, decimation folded into the loads (fft_wave.hpp load_slice; 16 kHz net only)
    if (a.dec > 1 && (sr != 16000 || a.dec > 3)) return hipErrorInvalidValue;
    if (a.dec == 3) hipLaunchKernelGGL((front_f43_kernel<32, PcmT, 3>), dim3(grid), dim3(256), 0, s, a);
    else if (a.dec == 2) hipLaunchKernelGGL((front_f43_kernel<32, PcmT, 2>), dim3(grid), dim3(256), 0, s, a);
    else if (sr == 16000) hipLaunchKernelGGL((front_f43_kernel<32, PcmT, 1>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((front_f43_kernel<16, PcmT, 1>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_front_f43<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_front_f43<int16_t>(int, const FrontArgs &, hipStream_t);

}  // namespace vad
