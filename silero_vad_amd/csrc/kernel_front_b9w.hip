// kernel_front_b9w.hip -- the bf16 x 9 frontend as TWO kernels that run side by side: the function, the arithmetic and the BITS of
// kernel_front_b9.hip (framing, reflect pad, window, 4 x real FFT magnitude, encoder 0 as one F(4,3) tile, encoders 1-3, W_ih; every
// matrix product as nine exact bf16 piece products on v_mfma_f32_16x16x32_bf16, fp32 accumulation).
//
//   fft_mags_kernel   PCM -> the four STFT magnitude frames of every 16-chunk tile, left in HBM in the register order the GEMM kernel
//                     loads them in ("mag layout", 33 KB per tile at 16 kHz).  Plain fp32 VALU work, ~100 registers, several waves
//                     per SIMD; launched on the engine's side stream, slab by slab AHEAD of the GEMM kernel.
//   front_b9g_kernel  magnitudes -> gate pre-activations.  GEMM only, persistent (one 4-wave workgroup per CU walks its tiles; the
//                     weight ring never restarts), ONE wave per SIMD with up to ~380 registers: encoder 0 runs matrix by matrix over
//                     ALL 128 rows, so the F(4,3) input transform of a K32 step is formed and split into bf16 pieces ONCE (the narrow
//                     kernel: once per 32-row part, 4 x at 16 kHz -- more than half of its 13.9 k VALU instructions per tile), the
//                     next step's pieces are formed between the MFMAs of the current one, and the next tile's magnitudes are
//                     requested as soon as encoder 0 has read the current ones.
//
// Why two kernels.  The narrow kernel's VALU work does not hide beside its matrix work (profiles/r03p_front_bf16x9.md); a fused wide
// form with one wave per SIMD removes the recomputation and exposes every latency instead (profiles/r04c_front_b9_wide_fused.md: 4.58
// ms against 3.83).  Waves of ONE kernel share one register budget, so "a big GEMM wave and a small FFT wave on every SIMD" cannot be
// one kernel -- but the SIMDs interleave co-resident waves of DIFFERENT kernels just the same (tools/coexec_diag.py: 2.86 ms of dense
// fp32 VALU work on every SIMD costs the GEMM path 0.4 ms).  The FFT kernel of slab s + 1 runs in the shadows of the GEMM kernel of
// slab s; the price is the magnitudes' round trip through HBM (2 x 2 KB per chunk, at an HBM load of ~40 %).
//
// Same bits.  The FFT code is the narrow kernel's (fft_wave.hpp, built without packed fp32 like it); an accumulator sees exactly the
// MFMAs it sees in the narrow program, in the same order (K32 steps ascending; per step piece pa of A against pieces 0, 1, 2 of B);
// m1..m4 are folded into the four frame outputs by the same expressions, m0 and m5 accumulate onto y0 and y3 through the C operand,
// encoder 1 consumes the parts in the narrow program's order.  The weight image is a permutation of the narrow image's 1 KiB
// fragments (layout.hpp "WIDE program").  tests: test_front_b9_wide_equals_narrow.
// (reference: the same lines as kernel_front_f43.hip.)
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_wave.hpp"
#include "front_common.hpp"

namespace vad {
namespace {

#ifndef VAD_B9G_VGPRS
#define VAD_B9G_VGPRS 344          // the GEMM kernel's register budget: beside it a SIMD must hold a 168-register wave of the FFT kernel
#endif
#define VAD_B9G_REGS __attribute__((amdgpu_num_vgpr(VAD_B9G_VGPRS)))
#ifndef VAD_B9G_DEPTH
#define VAD_B9G_DEPTH 2            // A fragments are read this many sub-steps (of 6 MFMAs = 96 cycles) ahead
#endif
#ifndef VAD_B9G_PRIO
#define VAD_B9G_PRIO 3
#endif
#ifndef VAD_FFT_WG_PER_CU
#define VAD_FFT_WG_PER_CU 4        // the FFT kernel's register budget as workgroups (of 4 waves) per CU: 4 -> 128 registers
#endif
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
// The program of a tile is the image's U units in order; the kernel is persistent, so after the last unit of a tile comes unit 0 of the
// workgroup's next tile: the ring never drains and `src` wraps around the image.
struct RingW {
    unsigned a_cur, a_nxt, a_far;       // LDS byte address of this lane's first A fragment in the slot of unit u, u+1, u+2
    unsigned d_cur, d_nxt, d_far;       // wave-uniform: where this wave's share of a unit lands in those slots
    const char *src, *base;             // wave-uniform: this wave's share of the next unit to request / of unit 0
    int left;                           // wave-uniform: units until `src` wraps
    int units;
    unsigned voff;                      // lane * 16
    u32x4 c0, c1;                       // the two A fragments of the next sub-step
    u32x4 d0, d1;                       // ... and of the one after it (VAD_B9G_DEPTH 2)
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
    if (--r.left == 0) {
        r.src = r.base;
        r.left = r.units;
    }
}
__device__ __forceinline__ void ring_rotate(RingW &r) {
    const unsigned a = r.a_cur, d = r.d_cur;
    r.a_cur = r.a_nxt; r.a_nxt = r.a_far; r.a_far = a;
    r.d_cur = r.d_nxt; r.d_nxt = r.d_far; r.d_far = d;
}
// In the middle of a unit: this wave's share of the next unit has landed; then everyone's, and everyone has left the previous unit; the
// unit after it is requested into the slot that is free now.  YOUNGER = vector-memory LOADS this wave issued after that share's request
// and that may still be in flight (loads return in order: the magnitude prefetch); 0 where stores may be among them.
template <int YOUNGER = 0>
__device__ __forceinline__ void ring_mid(RingW &r) {
    if constexpr (YOUNGER == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(YOUNGER) : "memory");
    if (!(VAD_ABLATE & 1)) __builtin_amdgcn_s_barrier();      // (bare: the fragment reads in flight belong to the current slot)
    asm volatile("" ::: "memory");
    ring_request(r);
}

// One unit of the WIDE encoder-0 program: a K32 step of one F(4,3) matrix against all 8 row blocks = 12 sub-steps (A piece pa x
// row-block pair) of 6 MFMAs; `bp` holds the step's B pieces.  `next(d)` forms pair d of the NEXT unit's pieces: it is called in the
// sub-steps 1..4, inside the scheduling region of that sub-step's MFMAs, so that its VALU instructions issue beside the matrix pipe.
template <class NEXT>
__device__ __forceinline__ void unit_w(f32x4 (&acc)[8], const u32x4 (&bp)[3], NEXT next, RingW &r) {
    static_for<0, 12>([&](auto qc) VAD_INLINE {
        constexpr int q = decltype(qc)::value, pair = q % 4;
        if constexpr (q == 6) ring_mid(r);
        constexpr int qa = q + VAD_B9G_DEPTH;      // the sub-step whose fragments are requested now
        u32x4 n0 = r.c0, n1 = r.c1;
        if constexpr (VAD_ABLATE & 128) {          // timing only: no fragment reads
        } else if constexpr (qa < 12) {
            n0 = lds4u(r.a_cur + (2 * qa) * 1024);
            n1 = lds4u(r.a_cur + (2 * qa + 1) * 1024);
        } else {
            n0 = lds4u(r.a_nxt + (2 * (qa - 12)) * 1024);
            n1 = lds4u(r.a_nxt + (2 * (qa - 12) + 1) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (q >= 1 && q <= 4) next(q - 1);
#pragma unroll
        for (int pb = 0; pb < 3; ++pb) {
            acc[2 * pair + 0] = mfma_b(r.c0, bp[pb], acc[2 * pair + 0]);
            acc[2 * pair + 1] = mfma_b(r.c1, bp[pb], acc[2 * pair + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (VAD_B9G_DEPTH == 2) {
            r.c0 = r.d0; r.c1 = r.d1; r.d0 = n0; r.d1 = n1;
        } else {
            r.c0 = n0; r.c1 = n1;
        }
    });
    ring_rotate(r);
}

// A segment of the narrow program (encoders 1-3, W_ih: unit format [step 4][piece 3][row block 2][lane][8]), as kernel_front_b9.hip:
// M row blocks x KG fp32 k-groups = KG/2 K32 steps; a step = 1 K32 step x 2 row blocks = 18 MFMAs; the 8 B values of a K32 step are
// split when its first row-block pair begins and serve all M/2 pairs.  YOUNGER: see ring_mid (first unit of the segment only).
template <int M, int KG, int YOUNGER = 0, class BF>
__device__ __forceinline__ void gemm_b(f32x4 (&acc)[M], BF bfun, RingW &r) {
    constexpr int H = M / 2, NSTEPS = (KG / 2) * H, NU = NSTEPS / 4;
    static_assert(KG % 2 == 0 && NSTEPS % 4 == 0 && M % 2 == 0, "segments are whole units");
    u32x4 bp[3];
    static_for<0, NU>([&](auto uc) VAD_INLINE {
        constexpr int u = decltype(uc)::value;
        static_for<0, 4>([&](auto sc_) VAD_INLINE {
            constexpr int st = decltype(sc_)::value, i = u * 4 + st, kp = i / H, mp = 2 * (i % H);
            if constexpr (st == 2) ring_mid<u == 0 ? YOUNGER : 0>(r);
            if constexpr (i % H == 0) split_step(bp, [&](int e) VAD_INLINE { return bfun(kp * 8 + e); });
            static_for<0, 3>([&](auto pc_) VAD_INLINE {
                constexpr int pa = decltype(pc_)::value, q = st * 3 + pa;
                constexpr int qa = q + VAD_B9G_DEPTH;
                u32x4 n0 = r.c0, n1 = r.c1;
                if constexpr (VAD_ABLATE & 128) {          // timing only: no fragment reads
                } else if constexpr (qa < 12) {
                    n0 = lds4u(r.a_cur + (2 * qa) * 1024);
                    n1 = lds4u(r.a_cur + (2 * qa + 1) * 1024);
                } else {
                    n0 = lds4u(r.a_nxt + (2 * (qa - 12)) * 1024);
                    n1 = lds4u(r.a_nxt + (2 * (qa - 12) + 1) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pb = 0; pb < 3; ++pb) {
                    acc[mp + 0] = mfma_b(r.c0, bp[pb], acc[mp + 0]);
                    acc[mp + 1] = mfma_b(r.c1, bp[pb], acc[mp + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (VAD_B9G_DEPTH == 2) {
                    r.c0 = r.d0; r.c1 = r.d1; r.d0 = n0; r.d1 = n1;
                } else {
                    r.c0 = n0; r.c1 = n1;
                }
            });
        });
        ring_rotate(r);
    });
}

// ---- the magnitudes in HBM -------------------------------------------------------------------------------------------------
// Per 16-chunk tile: [frame 4][k4 Q/4][lane 64][4] floats -- register X_f[4 k4 + e] of lane l at ((f Q/4 + k4) 64 + l) 4 + e, so that
// a wave stores / loads 1 KiB per instruction -- then [frame 4][chunk 16] Nyquist magnitudes (mag layout keeps them in lane group 0).
template <int Q> constexpr long kMagTileFloats = 4L * Q * 64 + 64;

// PCM -> magnitudes: one wave = one tile, the narrow kernel's FFT code frame by frame (one body, four trips).
template <int Q, typename PcmT, int DEC>
__global__ void __launch_bounds__(256, VAD_FFT_WG_PER_CU) fft_mags_kernel(const FrontArgs a, float *mags, long tile0, long ntiles) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int NF = tb.w_nyq - tb.window;               // window + twiddles
    __shared__ __attribute__((aligned(16))) float lds[NF];
    const float *tabf = lds - tb.window;                   // + tb.window / tb.tw1 / tb.tw2
    Lane ln;
    ln.lane = threadIdx.x & 63;
    ln.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ln.g = ln.lane >> 4;
    ln.j = ln.lane & 15;
    long wt = tile0 + (long)blockIdx.x * 4 + ln.wave;
    ln.tile_valid = wt < tile0 + ntiles;
    if (!ln.tile_valid) wt = tile0 + ntiles - 1;
    ln.tl = wt % a.nt;
    ln.st = wt / a.nt;
    ln.t = a.t0 + ln.tl;
    const long bb = ln.st * 16 + ln.j;
    ln.b = (int)(bb < a.B ? bb : a.B - 1);
    ln.from_tail = a.tail != nullptr && ln.t == a.T - 1;
    ln.sgnA = ln.g < 2 ? 1.f : -1.f;
    ln.sgnB = (ln.g & 1) ? -1.f : 1.f;
    {
        static_assert(NF % 4 == 0 && tb.window % 4 == 0, "tables are copied as 16-byte vectors");
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.tables + tb.window);
        for (int i = threadIdx.x; i < NF / 4; i += 256) reinterpret_cast<f32x4 *>(lds)[i] = src[i];
    }
    __syncthreads();
    float *out = mags + wt * kMagTileFloats<Q> + ln.lane * 4;
#pragma clang loop unroll(disable)
    for (int v = 0; v < 4; ++v) {
        float X[Q + 1];
        fft_frame<Q, PcmT, DEC>(X, v, a, tabf, ln);
        if (ln.tile_valid) {
#pragma unroll
            for (int k = 0; k < Q / 4; ++k)
                *reinterpret_cast<f32x4 *>(out + (size_t)(v * (Q / 4) + k) * 256) = f32x4{X[4 * k], X[4 * k + 1], X[4 * k + 2], X[4 * k + 3]};
            if (ln.g == 0) mags[wt * kMagTileFloats<Q> + 4L * Q * 64 + v * 16 + ln.j] = X[Q];
        }
    }
}

template <int Q>
__device__ __forceinline__ void load_mags(float (&X0)[Q + 1], float (&X1)[Q + 1], float (&X2)[Q + 1], float (&X3)[Q + 1],
                                          const float *mags, long wt, const Lane &ln) {
    const float *in = mags + wt * kMagTileFloats<Q> + ln.lane * 4;
    auto one = [&](float (&X)[Q + 1], int f) VAD_INLINE {
#pragma unroll
        for (int k = 0; k < Q / 4; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (size_t)(f * (Q / 4) + k) * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) X[4 * k + e] = v[e];
        }
        X[Q] = mags[wt * kMagTileFloats<Q> + 4L * Q * 64 + f * 16 + ln.j];      // (every lane of the chunk reads it)
    };
    one(X0, 0);
    one(X1, 1);
    one(X2, 2);
    one(X3, 3);
}

// magnitudes -> gate pre-activations; persistent: workgroup w walks the tiles tile0 + 4 w + wave + i * 4 * gridDim.x
template <int Q>
__global__ void __launch_bounds__(64 * kWavesW, 1) VAD_B9G_REGS front_b9g_kernel(const FrontArgs a, const float *mags, long tile0, long ntiles) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int RB = w_rb(Q), P = w_parts(Q), KP = w9w_kp(Q), U = w9w_units(Q);
    static_assert(w9w_tail0(Q) + 20 == U, "program mismatch");
    // LDS: [biases + head + Nyquist weights: NS floats][ring 3 x 24 KiB]
    constexpr int NS = tb.window + (tb.total - tb.w_nyq), UF = kUnitBytesW / 4;
    static_assert(tb.window % 4 == 0 && tb.w_nyq % 4 == 0 && NS % 4 == 0 && tb.total % 4 == 0, "table split");
    __shared__ __attribute__((aligned(16))) float lds[NS + 3 * UF];
    float *tab = lds;                                     // + off for off < tb.window
    float *tabn = lds + tb.window - tb.w_nyq;             // + tb.w_nyq + ... for the Nyquist weights

    Lane ln;
    ln.lane = threadIdx.x & 63;
    ln.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ln.g = ln.lane >> 4;
    ln.j = ln.lane & 15;
    const long tile_end = tile0 + ntiles, stride = 4L * gridDim.x;
    const long trips = (ntiles - 4L * blockIdx.x + stride - 1) / stride;      // the same for the four waves of the workgroup

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
        ring.base = ring.src = reinterpret_cast<const char *>(a.wfront) + ln.wave * kShareW;
        ring.left = ring.units = U;
        ring_request(ring);                               // unit 0 -> slot 0
        {   const unsigned d = ring.d_far; ring.d_far = ring.d_cur; ring.d_cur = ring.d_nxt; ring.d_nxt = d; }
        ring_request(ring);                               // unit 1 -> slot 1
        {   const unsigned d = ring.d_far; ring.d_far = ring.d_cur; ring.d_cur = ring.d_nxt; ring.d_nxt = d; }
        // now d_far = slot 2 (unit 2's), d_cur = slot 0, d_nxt = slot 1
    }
    {   // tables -> LDS (biases, head, Nyquist weights; the FFT's tables stay with the FFT kernel)
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
            if (i < NV && (4 * i < tb.window || 4 * i >= tb.w_nyq)) reinterpret_cast<f32x4 *>(4 * i < tb.window ? tab : tabn)[i] = v[k];
        }
    }
    long wt = tile0 + 4L * blockIdx.x + ln.wave;
    float X0[Q + 1], X1[Q + 1], X2[Q + 1], X3[Q + 1];
    load_mags<Q>(X0, X1, X2, X3, mags, wt < tile_end ? wt : tile_end - 1, ln);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // units 0 and 1, the tables and the first tile's magnitudes have landed
    __syncthreads();
    ring.c0 = lds4u(ring.a_cur);
    ring.c1 = lds4u(ring.a_cur + 1024);
    if (VAD_B9G_DEPTH == 2) {
        ring.d0 = lds4u(ring.a_cur + 2048);
        ring.d1 = lds4u(ring.a_cur + 3072);
    }
    if (VAD_B9G_PRIO) __builtin_amdgcn_s_setprio(VAD_B9G_PRIO);     // ahead of the FFT kernel's waves on the same SIMD

#pragma clang loop unroll(disable)
    for (long trip = 0; trip < trips; ++trip) {
        ln.tile_valid = wt < tile_end;
        const long wc = ln.tile_valid ? wt : tile_end - 1;
        ln.tl = wc % a.nt;
        ln.st = wc / a.nt;
        const float xn0 = X0[Q], xn1 = X1[Q], xn2 = X2[Q], xn3 = X3[Q];     // |Y_nyq| of the lane's chunk, frame by frame
        // The input transform reads the frames through E = x3 - x1 and F = x2 - x0 (kept in place of x3 and x0), as the narrow kernel does
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            X3[k] = X3[k] - X1[k];
            X0[k] = X2[k] - X0[k];
        }
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

        // ---- encoder 0: six matrices over all 128 rows -------------------------------------------------------------------------
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
                for (int m = 0; m < 8; ++m) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sm = Y0[m][r] + Y1[m][r], df = Y0[m][r] - Y1[m][r];
                        const float s2 = Y2[m][r] + Y3[m][r], d2 = Y2[m][r] - Y3[m][r];
                        Y0[m][r] = sm + s2;
                        Y1[m][r] = fmaf(2.f, d2, df);
                        Y2[m][r] = fmaf(4.f, s2, sm);
                        Y3[m][r] = fmaf(8.f, d2, df);
                    }
                    __builtin_amdgcn_sched_barrier(0);     // block by block, in place: 16 registers in flight, not 128
                }
            }
            if constexpr (j == 0 || j == 4) unit_w(Y0, cur, next, ring);          // m1; m0 onto y0
            else if constexpr (j == 1) unit_w(Y1, cur, next, ring);
            else if constexpr (j == 2) unit_w(Y2, cur, next, ring);
            else unit_w(Y3, cur, next, ring);                                     // m4; m5 onto y3
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

        // ---- encoder 1, part by part in the narrow program's order ------------------------------------------------------------
        f32x4 Z0[4], Z1[4];
        init_bias<4>(Z0, tab + tb.b_e1, ln);
        init_bias<4>(Z1, tab + tb.b_e1, ln);
        static_for<0, P>([&](auto pc) VAD_INLINE {
            constexpr int p = decltype(pc)::value, m0 = RB * p;
            if constexpr (Q == 32) {
                // 8 k-steps (one K32 step) per (tap, part): two taps share a unit
                auto e1a = [&](int s) VAD_INLINE { return s < 8 ? Y0[m0 + (s >> 2)][s & 3] : Y1[m0 + ((s - 8) >> 2)][s & 3]; };   // out 0: tap 1 <- y0 | tap 2 <- y1
                auto e1b = [&](int s) VAD_INLINE { return s < 8 ? Y1[m0 + (s >> 2)][s & 3] : Y2[m0 + ((s - 8) >> 2)][s & 3]; };   // out 1: tap 0 <- y1 | tap 1 <- y2
                gemm_b<4, 4>(Z0, e1a, ring);
                gemm_b<4, 4>(Z1, e1b, ring);
                if constexpr (p & 1) {                          // out 1: tap 2 <- y3 of the even part before | of this part
                    auto e1c = [&](int s) VAD_INLINE { return s < 8 ? Y3[m0 - RB + (s >> 2)][s & 3] : Y3[m0 + ((s - 8) >> 2)][s & 3]; };
                    gemm_b<4, 4>(Z1, e1c, ring);
                }
            } else {
                auto o0 = [&](int s) VAD_INLINE { return Y0[m0 + (s >> 2)][s & 3]; };
                auto o1 = [&](int s) VAD_INLINE { return Y1[m0 + (s >> 2)][s & 3]; };
                auto o2 = [&](int s) VAD_INLINE { return Y2[m0 + (s >> 2)][s & 3]; };
                auto o3 = [&](int s) VAD_INLINE { return Y3[m0 + (s >> 2)][s & 3]; };
                gemm_b<4, 4>(Z0, o0, ring);            // out 0, tap 1 <- y0
                gemm_b<4, 4>(Z0, o1, ring);            // out 0, tap 2 <- y1
                gemm_b<4, 4>(Z1, o1, ring);            // out 1, tap 0 <- y1
                gemm_b<4, 4>(Z1, o2, ring);            // out 1, tap 1 <- y2
                gemm_b<4, 4>(Z1, o3, ring);            // out 1, tap 2 <- y3
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
        // encoder 0's outputs are dead too: the next tile's magnitudes are requested now and arrive while encoders 2, 3 and W_ih run
        // (20 units; Q + 4 loads, all younger than the ring's pending request: the next barrier does not wait for them)
        wt += stride;
        __builtin_amdgcn_sched_barrier(0);             // (not earlier: the registers are encoder 1's until here)
        load_mags<Q>(X0, X1, X2, X3, mags, wt < tile_end ? wt : tile_end - 1, ln);
        __builtin_amdgcn_sched_barrier(0);
        gemm_b<4, 4, Q + 4>(Vv, bZ0, ring);
        gemm_b<4, 4>(Vv, bZ1, ring);
        relu<4>(Vv);
        f32x4 Fe[8];
        auto bF = [&](int s) VAD_INLINE { return Fe[s >> 2][s & 3]; };
        init_bias<8>(Fe, tab + tb.b_e3, ln);
        gemm_b<8, 4>(Fe, bV, ring);
        relu<8>(Fe);

        // LSTM input-gate pre-activations, one gate (8 row blocks) at a time, stored in D-fragment order
        float *gxt = a.gx + ((size_t)(ln.st * a.nt + ln.tl) * 32) * 256 + ln.lane * 4;
        const float *bg = tab + tb.b_g;
#pragma clang loop unroll(disable)
        for (int q = 0; q < 4; ++q) {
            f32x4 G[8];
            init_bias<8>(G, bg, ln);
            gemm_b<8, 8>(G, bF, ring);
            if (ln.tile_valid) {
#pragma unroll
                for (int m = 0; m < 8; ++m) *reinterpret_cast<f32x4 *>(gxt + (size_t)m * 256) = G[m];
            }
            gxt += 8 * 256;
            bg += 128;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last (surplus) requests must not outlive the workgroup's LDS
    __syncthreads();
}

}  // namespace

// magnitudes of the launch's tiles [tile0, tile0 + ntiles) (tile index = stream tile * a.nt + slab-relative step) -> `mags`
template <typename PcmT>
hipError_t launch_fft_mags(int sr, const FrontArgs &a, float *mags, long tile0, long ntiles, hipStream_t s) {
    if (ntiles <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((ntiles + 3) / 4);
    // a.dec == 2, 3: 32 / 48 kHz input, decimation folded into the loads (fft_wave.hpp load_slice; 16 kHz net only)
    if (a.dec > 1 && (sr != 16000 || a.dec > 3)) return hipErrorInvalidValue;
    if (a.dec == 3) hipLaunchKernelGGL((fft_mags_kernel<32, PcmT, 3>), dim3(grid), dim3(256), 0, s, a, mags, tile0, ntiles);
    else if (a.dec == 2) hipLaunchKernelGGL((fft_mags_kernel<32, PcmT, 2>), dim3(grid), dim3(256), 0, s, a, mags, tile0, ntiles);
    else if (sr == 16000) hipLaunchKernelGGL((fft_mags_kernel<32, PcmT, 1>), dim3(grid), dim3(256), 0, s, a, mags, tile0, ntiles);
    else hipLaunchKernelGGL((fft_mags_kernel<16, PcmT, 1>), dim3(grid), dim3(256), 0, s, a, mags, tile0, ntiles);
    return hipGetLastError();
}
template hipError_t launch_fft_mags<float>(int, const FrontArgs &, float *, long, long, hipStream_t);
template hipError_t launch_fft_mags<int16_t>(int, const FrontArgs &, float *, long, long, hipStream_t);

// magnitudes -> gate pre-activations for the same tile range; `cus` workgroups at most (one per CU)
hipError_t launch_front_b9g(int sr, const FrontArgs &a, const float *mags, long tile0, long ntiles, int cus, hipStream_t s) {
    if (ntiles <= 0) return hipSuccess;
    const long groups = (ntiles + kWavesW - 1) / kWavesW;
    const unsigned grid = (unsigned)(groups < cus ? groups : cus);
    if (sr == 16000) hipLaunchKernelGGL((front_b9g_kernel<32>), dim3(grid), dim3(64 * kWavesW), 0, s, a, mags, tile0, ntiles);
    else hipLaunchKernelGGL((front_b9g_kernel<16>), dim3(grid), dim3(64 * kWavesW), 0, s, a, mags, tile0, ntiles);
    return hipGetLastError();
}
long mag_tile_floats(int sr) { return sr == 16000 ? kMagTileFloats<32> : kMagTileFloats<16>; }

}  // namespace vad
