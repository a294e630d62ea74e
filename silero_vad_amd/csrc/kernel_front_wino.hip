// kernel_front_wino.hip -- A/B form (option enc0=winograd2; the product is kernel_front_f43.hip) of the time-parallel part
// of the Silero-VAD hot path (same function as kernel_front.hip:
//   PCM -> framing + right reflect pad -> Hann window -> 4 x real FFT magnitude -> 4 x ReLU(Conv1d k=3)
//       -> W_ih * feat + (b_ih + b_hh)  => gx)
// with encoder 0 -- 57 % of the kernel's matrix work -- evaluated as two Winograd F(2,3) transforms over the frame
// pairs (0,1) and (2,3) instead of five tap-GEMMs per pair: 4 GEMMs of [128 x 4Q] per pair, 3 968 instead of 4 480
// v_mfma_f32_16x16x4_f32 per 16-chunk tile (layout.hpp, "Winograd frontend image", has the algebra).
//
// (reference: JIT!/vad/model/vad_annotator.py:58-67 framing, JIT!/vad/utils/pytorch_stft.py:17-34 STFT,
//  JIT!/vad/utils/model_utils.py:19-25 encoder -- encoder 0 is JIT!/torch/nn/modules/conv/___torch_mangle_10.py:29 --,
//  the W_ih half of aten::lstm_cell JIT!/torch/nn/modules/rnn.py:69.)
//
// Everything is fp32 (exact v_mfma_f32_16x16x4_f32 chains and fp32 VALU adds): Winograd changes WHICH fp32 sums are
// formed, like the rFFT does for the STFT, not their precision.  GA = (g0+g1+g2)/2 and GB = (g0-g1+g2)/2 are formed
// in double on the host and rounded once.
//
// Structure (one wave = 16 chunks, 4 waves per workgroup, 2 workgroups per CU, as in kernel_front.hip):
//   * per frame pair and per row part (RB = 64/Q row blocks of enc0's 128 output rows) four accumulator sets:
//       a_first  = bias + G0 * (d0 - d2)        P = GA * (d1 + d2)
//       a_second = bias + G2 * (d3 - d1)        Q = GB * (d2 - d1)
//       y_first = a_first + P + Q,  y_second = a_second + P - Q   (+ the Nyquist bin, applied directly on the VALU)
//     the input combinations are formed on the fly from the magnitudes in registers (one v_add/v_sub per k-step);
//   * row parts keep the live accumulators at 32 registers; two parts (one K half of encoder 1) are ReLU'd and kept
//     as encoder 1's B operand, then consumed by encoder 1's taps for that K half;
//   * the weight stream is a sequence of whole 16 KiB units (make_wsched) through a 3-slot LDS ring fed by
//     global_load_lds_dwordx4; unit indices, slots and offsets are compile-time constants.  The workgroup barrier
//     that makes unit u+1 visible sits in the middle of unit u and fragment reads are carried across unit
//     boundaries ("seamless" pipeline, as in kernel_front.hip).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_wave.hpp"

namespace vad {
namespace {

constexpr int kUnitFloats = (int)vadl::kWUnitFloats;      // 16 blocks of 1 KiB
#define VAD_INLINE __attribute__((always_inline))

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// The input combinations of the Winograd transform are needed once per row part.  Left to the optimiser they are
// computed once and kept (4 x 33 registers per frame pair, which this kernel does not have).  They are therefore
// written as fma's with an OPAQUE +-1 (an SGPR the compiler cannot see through, fresh per row part): x + 1*y and
// y - 1*x are exact, cost the one VALU instruction an add would, and cannot be merged across parts.  (Not inline-asm
// adds: the compiler's hazard recogniser does not see inside asm, and a VALU result consumed by the very next MFMA
// needs its wait states.)
struct Unit {
    float one, mone;
};
__device__ __forceinline__ Unit opaque_unit() {
    Unit u;
    asm volatile("s_mov_b32 %0, 1.0\n\ts_mov_b32 %1, -1.0" : "=s"(u.one), "=s"(u.mone));
    return u;
}

struct WRing {
    float *slots;            // LDS, 3 x kUnitFloats
    const float *image;      // global: the Winograd image
    f32x4 c0, c1;            // A fragments of the next step, carried across unit and segment boundaries
};

// One unit (16 x 1 KiB): wave w copies the contiguous blocks [4w, 4w+4).  LDS destination = M0 (wave-uniform base) +
// instruction offset + lane*16, the layout global_load_lds requires.  Issued from inline asm on purpose (see
// kernel_front.hip ring_issue: the compiler would degrade every lgkmcnt wait while it knows of a pending LDS-DMA).
template <int Q, int PU>
__device__ __forceinline__ void unit_issue(const WRing &r, const Lane &ln) {
    constexpr vadl::WSched sc = vadl::make_wsched(Q);
    static_assert(PU >= 0 && PU < sc.n, "program unit out of range");
    if (VAD_ABLATE & 8) return;
    const float *src = r.image + (long)sc.unit[PU] * kUnitFloats + ln.wave * 1024;      // wave-uniform
    const unsigned voff = ln.lane * 16;                                                  // bytes
    const unsigned dst = (unsigned)(size_t)((__attribute__((address_space(3))) float *)(r.slots + (PU % 3) * kUnitFloats))
                         + (unsigned)ln.wave * 4096u;
    unsigned keep_m0;                      // M0 is restored: the compiler may keep its own value there
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(voff), "s"(src), "s"(dst) : "memory");
}

// A segment of M row blocks x KG k-groups ([k-group][row block] blocks, whole units), starting at program unit U0:
//     acc[m] += A[m][:, k] * B[k][:]   for all 4 KG k-steps;  bfun(s) = B-operand register of k-step s (compile-time s).
// A step = 2 row blocks x 4 k-steps = 8 MFMAs; a unit = 8 steps.  The A fragments of step i+1 are read from LDS before
// the MFMAs of step i are issued; in the middle of every unit the workgroup makes the NEXT unit visible (own share
// landed -> barrier) and requests the one after it into the slot everyone has left.
template <int Q, int U0, int M, int KG, bool CARRY_IN, bool CARRY_OUT, class BF>
__device__ __forceinline__ void gemm_w(f32x4 (&acc)[M], BF bfun, WRing &ring, const Lane &ln) {
    constexpr int NSTEPS = KG * (M / 2), NU = NSTEPS / 8, NPROG = vadl::w_program_units(Q);
    static_assert(NSTEPS % 8 == 0 && M % 2 == 0, "segments are whole units");
    static_for<0, NU>([&](auto uc) VAD_INLINE {
        constexpr int u = decltype(uc)::value, PU = U0 + u;
        const f32x4 *A = reinterpret_cast<const f32x4 *>(ring.slots + (PU % 3) * kUnitFloats) + ln.lane;
        const f32x4 *An = reinterpret_cast<const f32x4 *>(ring.slots + ((PU + 1) % 3) * kUnitFloats) + ln.lane;
        if (!CARRY_IN && u == 0) {
            ring.c0 = A[0];
            ring.c1 = A[64];
        }
        static_for<0, 8>([&](auto sc_) VAD_INLINE {
            constexpr int st = decltype(sc_)::value;
            if constexpr (st == 4 && PU + 1 < NPROG) {
                // this wave's share of unit PU + 1 has landed; then everyone's, and everyone has left unit PU - 1.  A bare
                // s_barrier (no lgkmcnt(0) fence): the fragment reads in flight belong to the current slot
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(VAD_ABLATE & 1)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (PU + 2 < NPROG) unit_issue<Q, PU + 2>(ring, ln);
            }
            f32x4 n0 = ring.c0, n1 = ring.c1;
            if constexpr (st + 1 < 8) {
                n0 = A[(2 * (st + 1)) * 64];
                n1 = A[(2 * (st + 1) + 1) * 64];
            } else if constexpr (u + 1 < NU || CARRY_OUT) {
                n0 = An[0];
                n1 = An[64];
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int i = u * 8 + st, kg = i / (M / 2), mp = 2 * (i % (M / 2));
            // (the compiler places each combination right in front of the MFMA pair that uses it and pays a 2-cycle wait
            //  state per VALU -> MFMA hand-over; forcing them ahead with a scheduling barrier costs far more: 5.99 vs 4.98 ms)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float bv = bfun(kg * 4 + ks);
                acc[mp + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring.c0[ks], bv, acc[mp + 0], 0, 0, 0);
                acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring.c1[ks], bv, acc[mp + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            ring.c0 = n0;
            ring.c1 = n1;
        });
    });
}

template <int M>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[M], const float *bias_lds, const Lane &ln) {
#pragma unroll
    for (int m = 0; m < M; ++m)
        acc[m] = *reinterpret_cast<const f32x4 *>(bias_lds + 16 * m + 4 * ln.g);
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

// Program unit at which a piece of the program starts (constexpr mirror of layout.hpp make_wsched)
template <int Q>
struct Prog {
    static constexpr int P = vadl::w_parts(Q), PH = P / 2;
    static constexpr int pair_units(int pair) { return 2 * (4 * PH + (pair == 0 ? 3 : 2)); }
    static constexpr int pair0(int pair) { return pair == 0 ? 0 : pair_units(0); }
    static constexpr int half0(int pair, int h) { return pair0(pair) + h * (4 * PH + (pair == 0 ? 3 : 2)); }
    static constexpr int e0(int pair, int h, int pp, int j) { return half0(pair, h) + 4 * pp + j; }
    static constexpr int e1(int pair, int h, int i) { return half0(pair, h) + 4 * PH + i; }
    static constexpr int tail0 = pair_units(0) + pair_units(1);       // E2T1, E2T2, E3 (2), IH (16)
};

template <int Q, typename PcmT, int DEC>
__global__ void __launch_bounds__(256, 2) front_wino_kernel(const FrontArgs a) {
    using namespace vadl;
    using PG = Prog<Q>;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int TABF = (tb.total + 3) / 4 * 4;
    constexpr int RB = w_rb(Q), P = w_parts(Q), PH = P / 2, KG0 = Q / 4, NPROG = w_program_units(Q);
    static_assert(make_wsched(Q).n == NPROG && PG::tail0 + 20 == NPROG, "program / schedule mismatch");
    __shared__ __attribute__((aligned(16))) float lds[TABF + 3 * kUnitFloats];
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

    WRing ring{lds + TABF, a.wfront, f32x4{}, f32x4{}};
    unit_issue<Q, 0>(ring, ln);                   // prime: units 0 and 1
    unit_issue<Q, 1>(ring, ln);
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

    float X0[Q + 1], X1[Q + 1], X2[Q + 1], X3[Q + 1];
    // non-finite input: fft_wave.hpp poison_acc (all four frames, raw).  No register to spare here either: the running value lives in the
    // thread's own LDS word between the passes (same thread reads and writes it: no barrier)
    __shared__ float pz_lds[256];
    auto pz_add = [&](const float (&X)[Q + 1], bool first) VAD_INLINE {
        float p = first ? 0.f : pz_lds[threadIdx.x];
        poison_acc<Q>(p, p, X);
        pz_lds[threadIdx.x] = fmaf(X[Q], 0.f, p);            // (the Nyquist magnitude: lane group 0; B[k][j] of that lane is enough)
    };
    fft_pass<Q, 0, PcmT, DEC>(X0, a, tab, ln);
    fft_pass<Q, 1, PcmT, DEC>(X1, a, tab, ln);
    fft_pass<Q, 2, PcmT, DEC>(X2, a, tab, ln);
    pz_add(X0, true);
    pz_add(X1, false);
    pz_add(X2, false);

    const float *wn = tab + tb.w_nyq;                  // [tap][row]
    // |Y_nyq| of chunk j lives in lane group 0 (X[Q]); every lane of the chunk needs it
    auto nyq = [&](const float (&X)[Q + 1]) VAD_INLINE { return __shfl(X[Q], ln.j); };

    f32x4 Z0[4], Z1[4];
    init_bias<4>(Z0, tab + tb.b_e1, ln);
    init_bias<4>(Z1, tab + tb.b_e1, ln);

    // ---- frame pair (0, 1): d = (0, x0, x1, x2) -> y0 (enc1 out 0 tap 1), y1 (out 0 tap 2, out 1 tap 0) --------------
    static_for<0, 2>([&](auto hc) VAD_INLINE {
        constexpr int h = decltype(hc)::value;
        f32x4 Ya[4], Yb[4];                              // ReLU(y0), ReLU(y1), rows 64h .. 64h+63
        static_for<0, PH>([&](auto pc) VAD_INLINE {
            constexpr int pp = decltype(pc)::value, part = h * PH + pp, row0 = 16 * RB * part;
            constexpr bool first = h == 0 && pp == 0;
            f32x4 a0[RB], a1[RB], Pm[RB], Qm[RB];
            init_bias<RB>(a0, tab + tb.b_e0 + row0, ln);
            init_bias<RB>(a1, tab + tb.b_e0 + row0, ln);
            zero<RB>(Pm);
            zero<RB>(Qm);
            const Unit k = opaque_unit();
            gemm_w<Q, PG::e0(0, h, pp, WG0), RB, KG0, !first, true>(a0, [&](int s) VAD_INLINE { return X1[s] * k.mone; }, ring, ln);
            gemm_w<Q, PG::e0(0, h, pp, WGA), RB, KG0, true, true>(Pm, [&](int s) VAD_INLINE { return fmaf(X1[s], k.one, X0[s]); }, ring, ln);
            gemm_w<Q, PG::e0(0, h, pp, WGB), RB, KG0, true, true>(Qm, [&](int s) VAD_INLINE { return fmaf(X0[s], k.mone, X1[s]); }, ring, ln);
            gemm_w<Q, PG::e0(0, h, pp, WG2), RB, KG0, true, true>(a1, [&](int s) VAD_INLINE { return fmaf(X0[s], k.mone, X2[s]); }, ring, ln);
            nyq_update<RB>(a0, nyq(X0), wn + 128 + row0, ln);
            nyq_update<RB>(a0, nyq(X1), wn + 256 + row0, ln);
            nyq_update<RB>(a1, nyq(X0), wn + row0, ln);
            nyq_update<RB>(a1, nyq(X1), wn + 128 + row0, ln);
            nyq_update<RB>(a1, nyq(X2), wn + 256 + row0, ln);
#pragma unroll
            for (int m = 0; m < RB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pq = Pm[m][r] + Qm[m][r], pmq = Pm[m][r] - Qm[m][r];
                    Ya[pp * RB + m][r] = fmaxf(a0[m][r] + pq, 0.f);
                    Yb[pp * RB + m][r] = fmaxf(a1[m][r] + pmq, 0.f);
                }
        });
        auto bYa = [&](int s) VAD_INLINE { return Ya[s >> 2][s & 3]; };
        auto bYb = [&](int s) VAD_INLINE { return Yb[s >> 2][s & 3]; };
        gemm_w<Q, PG::e1(0, h, 0), 4, 4, true, true>(Z0, bYa, ring, ln);            // out 0, tap 1 <- y0
        gemm_w<Q, PG::e1(0, h, 1), 4, 4, true, true>(Z0, bYb, ring, ln);            // out 0, tap 2 <- y1
        gemm_w<Q, PG::e1(0, h, 2), 4, 4, true, h == 0>(Z1, bYb, ring, ln);          // out 1, tap 0 <- y1 (last before FFT 3)
    });

    fft_pass<Q, 3, PcmT, DEC>(X3, a, tab, ln);
    pz_add(X3, false);

    // ---- frame pair (2, 3): d = (x1, x2, x3, 0) -> y2 (enc1 out 1 tap 1), y3 (out 1 tap 2) ---------------------------
    static_for<0, 2>([&](auto hc) VAD_INLINE {
        constexpr int h = decltype(hc)::value;
        f32x4 Ya[4], Yb[4];
        static_for<0, PH>([&](auto pc) VAD_INLINE {
            constexpr int pp = decltype(pc)::value, part = h * PH + pp, row0 = 16 * RB * part;
            constexpr bool first = h == 0 && pp == 0;
            f32x4 a0[RB], a1[RB], Pm[RB], Qm[RB];
            init_bias<RB>(a0, tab + tb.b_e0 + row0, ln);
            init_bias<RB>(a1, tab + tb.b_e0 + row0, ln);
            zero<RB>(Pm);
            zero<RB>(Qm);
            const Unit k = opaque_unit();
            gemm_w<Q, PG::e0(1, h, pp, WG0), RB, KG0, !first, true>(a0, [&](int s) VAD_INLINE { return fmaf(X3[s], k.mone, X1[s]); }, ring, ln);
            gemm_w<Q, PG::e0(1, h, pp, WGA), RB, KG0, true, true>(Pm, [&](int s) VAD_INLINE { return fmaf(X3[s], k.one, X2[s]); }, ring, ln);
            gemm_w<Q, PG::e0(1, h, pp, WGB), RB, KG0, true, true>(Qm, [&](int s) VAD_INLINE { return fmaf(X2[s], k.mone, X3[s]); }, ring, ln);
            gemm_w<Q, PG::e0(1, h, pp, WG2), RB, KG0, true, true>(a1, [&](int s) VAD_INLINE { return X2[s] * k.mone; }, ring, ln);
            nyq_update<RB>(a0, nyq(X1), wn + row0, ln);
            nyq_update<RB>(a0, nyq(X2), wn + 128 + row0, ln);
            nyq_update<RB>(a0, nyq(X3), wn + 256 + row0, ln);
            nyq_update<RB>(a1, nyq(X2), wn + row0, ln);
            nyq_update<RB>(a1, nyq(X3), wn + 128 + row0, ln);
#pragma unroll
            for (int m = 0; m < RB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pq = Pm[m][r] + Qm[m][r], pmq = Pm[m][r] - Qm[m][r];
                    Ya[pp * RB + m][r] = fmaxf(a0[m][r] + pq, 0.f);
                    Yb[pp * RB + m][r] = fmaxf(a1[m][r] + pmq, 0.f);
                }
        });
        auto bYa = [&](int s) VAD_INLINE { return Ya[s >> 2][s & 3]; };
        auto bYb = [&](int s) VAD_INLINE { return Yb[s >> 2][s & 3]; };
        gemm_w<Q, PG::e1(1, h, 0), 4, 4, true, true>(Z1, bYa, ring, ln);            // out 1, tap 1 <- y2
        gemm_w<Q, PG::e1(1, h, 1), 4, 4, true, true>(Z1, bYb, ring, ln);            // out 1, tap 2 <- y3
    });
    relu<4>(Z0);
    relu<4>(Z1);

    // ---- enc2 (T 2 -> 1, stride 2: taps 1,2 see enc1 outputs 0,1), enc3 (T = 1: centre tap only), W_ih -----------------
    f32x4 Vv[4];
    auto bZ0 = [&](int s) VAD_INLINE { return Z0[s >> 2][s & 3]; };
    auto bZ1 = [&](int s) VAD_INLINE { return Z1[s >> 2][s & 3]; };
    auto bV = [&](int s) VAD_INLINE { return Vv[s >> 2][s & 3]; };
    constexpr int T0 = PG::tail0;
    init_bias<4>(Vv, tab + tb.b_e2, ln);
    gemm_w<Q, T0 + 0, 4, 4, true, true>(Vv, bZ0, ring, ln);
    gemm_w<Q, T0 + 1, 4, 4, true, true>(Vv, bZ1, ring, ln);
    relu<4>(Vv);
    f32x4 Fe[8];
    auto bF = [&](int s) VAD_INLINE { return Fe[s >> 2][s & 3]; };
    init_bias<8>(Fe, tab + tb.b_e3, ln);
    gemm_w<Q, T0 + 2, 8, 4, true, true>(Fe, bV, ring, ln);
    relu<8>(Fe);
    poison_into(Fe[0], pz_lds[threadIdx.x]);

    // LSTM input-gate pre-activations, one gate (8 row blocks) at a time, stored in D-fragment order
    float *gxt = a.gx + ((size_t)(ln.st * a.nt + ln.tl) * 32) * 256 + ln.lane * 4;
    static_for<0, 4>([&](auto qc) VAD_INLINE {
        constexpr int q = decltype(qc)::value;
        f32x4 G[8];
        init_bias<8>(G, tab + tb.b_g + 128 * q, ln);
        gemm_w<Q, T0 + 4 + 4 * q, 8, 8, true, q != 3>(G, bF, ring, ln);
        if (ln.tile_valid) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
                *reinterpret_cast<f32x4 *>(gxt + (size_t)(8 * q + m) * 256) = G[m];
        }
    });
}

}  // namespace

template <typename PcmT>
hipError_t launch_front_wino(int sr, const FrontArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    const unsigned grid = (unsigned)((total + 3) / 4);
    // a.dec == 2: 32 kHz input, decimation folded into the loads (fft_wave.hpp load_slice; 16 kHz net only)
    if (sr == 16000 && a.dec == 2) hipLaunchKernelGGL((front_wino_kernel<32, PcmT, 2>), dim3(grid), dim3(256), 0, s, a);
    else if (a.dec > 1) return hipErrorInvalidValue;
    else if (sr == 16000) hipLaunchKernelGGL((front_wino_kernel<32, PcmT, 1>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((front_wino_kernel<16, PcmT, 1>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_front_wino<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_front_wino<int16_t>(int, const FrontArgs &, hipStream_t);

}  // namespace vad
