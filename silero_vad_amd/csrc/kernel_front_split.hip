// kernel_front_split.hip -- the time-parallel part of the path (same function as kernel_front.hip:
// PCM -> framing -> 4 x real-FFT magnitude -> 4 x ReLU(Conv1d k=3) -> W_ih * feat + b  => gx), with
// every matrix product evaluated as an fp16 x 3 "split" product on the f16 matrix cores:
//
//     a = a_hi + a_lo,  b = b_hi + b_lo   (hi = fp16(x), lo = fp16(x - hi), round to nearest)
//     a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi          (dropped term a_lo b_lo < 2^-22 |a b|)
//
// accumulated in fp32 by v_mfma_f32_16x16x32_f16.  Operands keep 22 significant bits, sums are fp32:
// measured against the reference this is indistinguishable from the exact-fp32 chain (both ~2e-6 on
// the speech probability, tests/test_gpu_parity.py), while gfx950 runs f16 MFMA at 16x the rate of
// f32 MFMA (MI355X_MICROARCH.md: 2.5 PFLOP/s vs 157 TFLOP/s), i.e. 16/3 = 5.3x fewer matrix-pipe
// cycles for the same contraction.
//
// (reference: JIT!/vad/model/vad_annotator.py:58-67 framing, JIT!/vad/utils/pytorch_stft.py:17-34
//  STFT, JIT!/vad/utils/model_utils.py:19-25 encoder, the W_ih half of aten::lstm_cell
//  JIT!/torch/nn/modules/rnn.py:69.)
//
// What changes with the matrix pipe 5x faster is what bounds the kernel: LDS bandwidth for the A
// (weight) fragments.  Hence, relative to kernel_front.hip:
//   * an A fragment read from LDS is used for TWO frames where the conv taps allow it (enc0 is
//     evaluated for frame pairs (0,1) and (2,3); 1.67 uses per fragment on average);
//   * enc0 is evaluated in two halves of its output rows and enc1 in the matching halves of its
//     K dimension, which keeps only 2 x 4 accumulator blocks of enc0 live;
//   * the Nyquist bin (the 129th / 65th input channel of enc0, alone in a fifth K step) is applied as
//     an fp32 rank-1 VALU update instead of an almost empty MFMA step;
//   * activations are converted to (hi, lo) half pairs once, when produced (5 VALU per 2 values), and
//     live in registers as the packed B operands of the next layer (chain layout, layout.hpp);
//   * the weight image streams L2 -> LDS through a ring of 4 x 16 KiB slots by a static schedule
//     (make_sched), each unit requested three units ahead, with counted vmcnt waits (sring_wait).
// BUILD REQUIREMENT -- no packed-fp32 VALU instructions in this translation unit
// (__graft_entry__.py: -Xclang -target-feature -Xclang -packed-fp32-ops, and it disassembles the object
// to check).  Measured on MI355X / ROCm 7.2: when one wave of a SIMD runs v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 (the in-wave FFT of fft_wave.hpp compiles to them) while ANOTHER wave of the same SIMD
// has v_mfma_f32_16x16x32_f16 in flight, the MFMA results are corrupted now and then: with two
// workgroups per CU -- whose FFT and MFMA phases are not barrier-locked to each other -- about 3 % of
// the 16-chunk tiles of a launch came out wrong and differently on every run, independent of how the
// weights reached the matrix pipe (LDS-DMA ring, register copies or plain global loads), of builtin
// vs inline-asm MFMAs and of wait-state padding; one workgroup per CU (all waves in the same phase) or
// the same code without packed-fp32 instructions is bit-stable over > 10^6 tiles
// (tools/split_stress.py, tools/variants.py `nopk*`, `at2_lds1`, `w8`).  Scalar fp32 VALU costs nothing
// here: beside MFMAs the packed forms are no faster (MI355X_MICROARCH.md, "price of one filler").
// kernel_front.hip (f32 MFMA, which shares the VALU's issue pipe and therefore never overlaps it) is
// not affected and keeps the packed FFT.
//
// Range: fp16 overflows at 65504.  |pcm| <= 1 keeps every activation below ~5.3e3 on all inputs tried
// (sines, square waves, noise, speech); a lane that nevertheless sees |x| > 65000 poisons its chunk's
// gx with NaN, which kernel_rec_split.hip propagates to the probability, so an out-of-range input is
// reported as NaN (the host wrapper reruns it with precision=fp32) and never as a wrong number.
#include <hip/hip_runtime.h>

#ifndef VAD_SPLIT_NT_PCM
#define VAD_SPLIT_NT_PCM 0         // non-temporal hint on the PCM loads / the gx stores (measured: both cost time in
#endif                             // this kernel -- frames overlap, so PCM lines are re-read; kept as knobs)
#ifndef VAD_SPLIT_NT_GX
#define VAD_SPLIT_NT_GX 0
#endif
#define VAD_NT_PCM VAD_SPLIT_NT_PCM
#include "fft_wave.hpp"

namespace vad {
namespace {

using h8 = _Float16 __attribute__((ext_vector_type(8)));
using h2 = _Float16 __attribute__((ext_vector_type(2)));

#ifndef VAD_SPLIT_SLOT_BLOCKS
#define VAD_SPLIT_SLOT_BLOCKS 16   // 1-KiB blocks per ring slot = per unit = 8 (u, mblock) pairs
#endif
#ifndef VAD_SPLIT_SLOTS
#define VAD_SPLIT_SLOTS 4          // ring slots; a unit is requested VAD_SPLIT_SLOTS - 1 units ahead of its use
#endif
#ifndef VAD_SPLIT_WAVES
#define VAD_SPLIT_WAVES 4          // waves (= 16-chunk tiles) per workgroup
#endif
constexpr int kWV = VAD_SPLIT_WAVES;
constexpr int kSB = VAD_SPLIT_SLOT_BLOCKS;
constexpr int kSlotWords = kSB * 256;
constexpr int kSlots = VAD_SPLIT_SLOTS, kAhead = kSlots - 1;

// Program order of the weight stream, as ring units (word offsets into the split image).  The kernel
// walks its segments in exactly this order; gemm_split() cross-checks every unit it consumes.
struct Sched {
    int n;
    int off[96];
};
constexpr Sched make_sched(int Q) {
    using namespace vadl;
    Sched sc{};
    int n = 0;
    const int a_order[6] = {SE0 + 0, SE0 + 1, SE0 + 2, SE1 + 1, SE1 + 2, SE1 + 0};   // frames 0,1 (per row half)
    const int b_order[5] = {SE0 + 0, SE0 + 1, SE0 + 2, SE1 + 1, SE1 + 2};            // frames 2,3
    int segs[40] = {};
    int ns = 0;
    for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 6; ++i) segs[ns++] = a_order[i] + 6 * h;
    for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 5; ++i) segs[ns++] = b_order[i] + 6 * h;
    segs[ns++] = SE2T1; segs[ns++] = SE2T2; segs[ns++] = SE3T1;
    for (int q = 0; q < 4; ++q) segs[ns++] = SIH0 + q;
    for (int i = 0; i < ns; ++i) {
        const int units = (int)(sseg_words(segs[i], Q) / kSlotWords);
        for (int u = 0; u < units; ++u) sc.off[n++] = (int)(sseg_offset(segs[i], Q) + (long)u * kSlotWords);
    }
    sc.n = n;
    return sc;
}
__device__ const Sched kSched32 = make_sched(32);
__device__ const Sched kSched16 = make_sched(16);
constexpr float kHalfLimit = 65000.f;

struct SRing {
    unsigned *slots;           // LDS, kSlots x kSlotWords
    const unsigned *w;         // global split image
    const Sched *sched;
    int unit;                  // units consumed so far (wave-uniform)
    int bad;                   // a consumed unit was not the scheduled one (programming error): poison
};
struct Bop {                   // B operand of one K32 step: 8 halves hi, 8 halves lo
    u32x4 hi, lo;
};

template <int BLOCKS>
__device__ __forceinline__ void sring_issue(const SRing &r, long woff, int slot, const Lane &ln) {
    // BLOCKS x 1 KiB; wave w copies the CONTIGUOUS blocks [w n, (w+1) n), n = BLOCKS / kWV <= 4: one M0
    // value and one source base per wave and unit, the instruction's immediate offset (which advances
    // the global and the LDS address alike) selects the block.  global_load_lds is issued from asm for
    // the reason given in kernel_front.hip (ring_issue): it keeps the compiler's lgkmcnt waits fine-grained.
    constexpr int PW = BLOCKS / kWV;
    static_assert(BLOCKS % kWV == 0 && BLOCKS <= kSB && (PW == 4 || PW == 2), "blocks per wave and unit");
    if (VAD_ABLATE & 8) return;
    const unsigned *src = r.w + woff + (long)ln.wave * PW * 256;
    const unsigned voff = ln.lane * 16;
    const unsigned dst = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned *)(r.slots + slot * kSlotWords))
                         + (unsigned)ln.wave * PW * 1024u;
    unsigned keep_m0;                      // M0 is restored: the compiler may keep its own value there
    if (PW == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep_m0)
                     : "v"(voff), "s"(src), "s"(dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep_m0)
                     : "v"(voff), "s"(src), "s"(dst)
                     : "memory");
}
// Unit `ring.unit` has landed for THIS wave.  Every wave issues exactly kSB / kWV LDS-DMA instructions
// per unit and `later` younger units are in flight, so the wait is counted: loads (PCM, DMA) complete
// in order among themselves; stores (gx, ctx) may complete out of order but only ever ADD to the
// counter, so they can delay this wait, never release it early.
__device__ __forceinline__ void sring_wait(int later) {
    static_assert(kSB / kWV == 4 && kAhead <= 3, "wait counts below assume 4 DMA instructions per wave and unit");
    if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void sring_request(const SRing &r, int unit, const Lane &ln);

__device__ __forceinline__ void sring_request(const SRing &r, int unit, const Lane &ln) {
    sring_issue<kSB>(r, r.sched->off[unit], unit % kSlots, ln);
}

__device__ __forceinline__ f32x4 mfma_h(u32x4 a, u32x4 b, f32x4 c) {
    if (VAD_ABLATE & 64) {                 // timing experiment: no matrix pipe, operands stay live
        c[0] += __uint_as_float((a[0] ^ b[0]) & 0x3fffffffu);
        return c;
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// One segment = U K32-steps x M row blocks = U M / 8 ring units of 8 (u, mblock) pairs, consumed in the
// order of make_sched() (seg_off, the segment's offset in the image, is only used to cross-check that).
// NUSE accumulator sets share every A fragment:  acc0 += A * b0,  (NUSE == 2:) acc1 += A * b1.
// b*(u) returns the packed B operand of K32 step u (compile-time u).
template <int M, int U, int NUSE, class BF0, class BF1>
__device__ __forceinline__ void gemm_split(f32x4 (&acc0)[M], BF0 b0, f32x4 (&acc1)[M], BF1 b1, SRing &ring,
                                           long seg_off, const Lane &ln) {
    constexpr int PAIRS = U * M, PPU = kSB / 2, NU = PAIRS / PPU;
    static_assert(PPU % M == 0 && M % 2 == 0 && PAIRS % PPU == 0,
                  "a unit holds whole K32 steps; steps take two row blocks; segments are whole units");
    constexpr int np = PPU;
#pragma unroll
    for (int un = 0; un < NU; ++un) {
        const int later = ring.sched->n - 1 - ring.unit;
        sring_wait(later < kAhead - 1 ? later : kAhead - 1);
        if (!(VAD_ABLATE & 1)) __syncthreads();        // unit landed for every wave; slot of unit-1 is free
        if (ring.sched->off[ring.unit] != (int)(seg_off + (long)un * kSlotWords)) ring.bad = 1;
        if (ring.unit + kAhead < ring.sched->n) sring_request(ring, ring.unit + kAhead, ln);
        const int slot = ring.unit % kSlots;
        // A step = two row blocks of one K32 step: fragments (hi, lo) x 2, 6 (12) MFMAs ordered so that
        // MFMAs on the same accumulator are never adjacent.  The fragments of step i+1 are read from
        // LDS before the MFMAs of step i issue (explicit double buffer, as in kernel_front.hip).
        const u32x4 *A = reinterpret_cast<const u32x4 *>(ring.slots + slot * kSlotWords) + ln.lane;
        if (VAD_ABLATE & 128) A = reinterpret_cast<const u32x4 *>(ring.slots) + ln.lane;   // timing: see below
        u32x4 c0 = A[0], c1 = A[64], c2 = A[128], c3 = A[192];
#pragma unroll
        for (int st = 0; st < PPU / 2; ++st) {
            if (st < np / 2) {
                u32x4 n0 = c0, n1 = c1, n2 = c2, n3 = c3;
                if ((VAD_ABLATE & 128) && st > 0) {
                    // timing experiment: one fragment read per unit instead of one per step
                } else if (st + 1 < np / 2) {
                    n0 = A[(4 * (st + 1) + 0) * 64];
                    n1 = A[(4 * (st + 1) + 1) * 64];
                    n2 = A[(4 * (st + 1) + 2) * 64];
                    n3 = A[(4 * (st + 1) + 3) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const int p = un * PPU + 2 * st, u = p / M, m = p % M;
                const Bop x0 = b0(u);
                acc0[m] = mfma_h(c0, x0.hi, acc0[m]);
                acc0[m + 1] = mfma_h(c2, x0.hi, acc0[m + 1]);
                if (NUSE == 2) {
                    const Bop x1 = b1(u);
                    acc1[m] = mfma_h(c0, x1.hi, acc1[m]);
                    acc1[m + 1] = mfma_h(c2, x1.hi, acc1[m + 1]);
                    acc0[m] = mfma_h(c0, x0.lo, acc0[m]);
                    acc0[m + 1] = mfma_h(c2, x0.lo, acc0[m + 1]);
                    acc1[m] = mfma_h(c0, x1.lo, acc1[m]);
                    acc1[m + 1] = mfma_h(c2, x1.lo, acc1[m + 1]);
                    acc0[m] = mfma_h(c1, x0.hi, acc0[m]);
                    acc0[m + 1] = mfma_h(c3, x0.hi, acc0[m + 1]);
                    acc1[m] = mfma_h(c1, x1.hi, acc1[m]);
                    acc1[m + 1] = mfma_h(c3, x1.hi, acc1[m + 1]);
                } else {
                    acc0[m] = mfma_h(c0, x0.lo, acc0[m]);
                    acc0[m + 1] = mfma_h(c2, x0.lo, acc0[m + 1]);
                    acc0[m] = mfma_h(c1, x0.hi, acc0[m]);
                    acc0[m + 1] = mfma_h(c3, x0.hi, acc0[m + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            }
        }
        ring.unit++;
    }
}

template <int M>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[M], const float *bias_lds, const Lane &ln) {
#pragma unroll
    for (int m = 0; m < M; ++m)
        acc[m] = *reinterpret_cast<const f32x4 *>(bias_lds + 16 * m + 4 * ln.g);
}

// (x0, x1) -> packed halves hi, lo; mx tracks the largest magnitude converted
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hi, unsigned &lo, float &mx) {
#pragma clang fp contract(off)      // lo must come from the ROUNDED x (a caller's multiply must not fuse in):
    const f32x2 v{x0, x1};              // the same x, reloaded from HBM by a later call, has to split alike
    const h2 h = __builtin_convertvector(v, h2);                 // v_cvt_pk_f16_f32 (RTN)
    const f32x2 r = v - __builtin_convertvector(h, f32x2);       // exact
    const h2 l = __builtin_convertvector(r, h2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
    mx = fmaxf(mx, fmaxf(fabsf(x0), fabsf(x1)));                 // v_max3_f32 |.| |.|
}
// ReLU + split of NB D-fragment blocks into NB/2 K32-step operands (chain layout)
template <int NB>
__device__ __forceinline__ void relu_pack(const f32x4 (&D)[NB], Bop (&B)[NB / 2], float &mx) {
#pragma unroll
    for (int u = 0; u < NB / 2; ++u) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x4 d = D[2 * u + (p >> 1)];
            const int r = 2 * (p & 1);
            unsigned hi, lo;
            split2(fmaxf(d[r], 0.f), fmaxf(d[r + 1], 0.f), hi, lo, mx);
            B[u].hi[p] = hi;
            B[u].lo[p] = lo;
        }
    }
}
// mag layout -> K32-step operands: slot (g, e) of step u = X[8u + e]
template <int Q>
__device__ __forceinline__ void pack_mag(const float (&X)[Q + 1], Bop (&B)[Q / 8], float &mx) {
#pragma unroll
    for (int u = 0; u < Q / 8; ++u)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned hi, lo;
            split2(X[8 * u + 2 * p], X[8 * u + 2 * p + 1], hi, lo, mx);
            B[u].hi[p] = hi;
            B[u].lo[p] = lo;
        }
}
// Nyquist bin: Y[row] += w_nyq[tap][row] * |Y_nyq| of the chunk, exact fp32
__device__ __forceinline__ void nyq_update(f32x4 (&Y)[4], float xn, const float *wn_lds, const Lane &ln) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(wn_lds + 16 * m + 4 * ln.g);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[m][r] = fmaf(w[r], xn, Y[m][r]);
    }
}

template <int Q, typename PcmT>
__global__ void __launch_bounds__(64 * kWV, 2) front_split_kernel(const FrontArgs a) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int TABF = (tb.total + 3) / 4 * 4;
    constexpr int U0 = Q / 8;
    __shared__ __attribute__((aligned(16))) float lds[TABF + kSlots * kSlotWords];
    float *tab = lds;

    Lane ln;
    ln.lane = threadIdx.x & 63;
    ln.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ln.g = ln.lane >> 4;
    ln.j = ln.lane & 15;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    long wt = (long)blockIdx.x * kWV + ln.wave;
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

    SRing ring{reinterpret_cast<unsigned *>(lds + TABF), reinterpret_cast<const unsigned *>(a.wfront),
               Q == 32 ? &kSched32 : &kSched16, 0, 0};
    auto off = [](int s) { return sseg_offset(s, Q); };

#pragma unroll
    for (int u = 0; u < kAhead; ++u) sring_request(ring, u, ln);      // prime the ring
    if (!(VAD_ABLATE & 256)) {             // (256: timing experiment, tables left uninitialised)
        // tables -> LDS: all loads of a thread are issued before the first is stored (one memory latency,
        // not one per 1 KiB; the plain loop cost 4 % of the kernel)
        static_assert(tb.total % 4 == 0, "tables are copied as 16-byte vectors");
        constexpr int NV = tb.total / 4, PER = (NV + 64 * kWV - 1) / (64 * kWV);
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.tables);
        f32x4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 64 * kWV;
            v[k] = src[i < NV ? i : NV - 1];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 64 * kWV;
            if (i < NV) reinterpret_cast<f32x4 *>(tab)[i] = v[k];
        }
    }
    __syncthreads();

    float mx = 0.f;
    Bop X0[U0], X1[U0], X2[U0], X3[U0];
    float xn0, xn1, xn2, xn3;
    {
        float Xf[Q + 1];
        fft_pass<Q, 0, PcmT>(Xf, a, tab, ln);
        pack_mag<Q>(Xf, X0, mx);
        xn0 = __shfl(Xf[Q], ln.j);
        fft_pass<Q, 1, PcmT>(Xf, a, tab, ln);
        pack_mag<Q>(Xf, X1, mx);
        xn1 = __shfl(Xf[Q], ln.j);
        fft_pass<Q, 2, PcmT>(Xf, a, tab, ln);
        pack_mag<Q>(Xf, X2, mx);
        xn2 = __shfl(Xf[Q], ln.j);
    }
    mx = fmaxf(mx, fmaxf(xn0, fmaxf(xn1, xn2)));
    auto bX0 = [&](int u) { return X0[u]; };
    auto bX1 = [&](int u) { return X1[u]; };
    auto bX2 = [&](int u) { return X2[u]; };
    auto bX3 = [&](int u) { return X3[u]; };

#ifndef VAD_SPLIT_PRIO
#define VAD_SPLIT_PRIO 0           // s_setprio level of the GEMM phases (the FFT passes run at 0)
#endif
    if (VAD_SPLIT_PRIO) __builtin_amdgcn_s_setprio(VAD_SPLIT_PRIO);
    f32x4 Z0[4], Z1[4];
    init_bias<4>(Z0, tab + tb.b_e1, ln);
    init_bias<4>(Z1, tab + tb.b_e1, ln);
    const float *wn = tab + tb.w_nyq;

    // ---- frames 0 and 1 of enc0 (rows 64h..64h+63) -> enc1 out 0 taps 1,2 and out 1 tap 0 ------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 Ya[4], Yb[4];                              // enc0 frame 0, frame 1
        init_bias<4>(Ya, tab + tb.b_e0 + 64 * h, ln);
        init_bias<4>(Yb, tab + tb.b_e0 + 64 * h, ln);
        gemm_split<4, U0, 1>(Yb, bX0, Yb, bX0, ring, off(SE0 + 6 * h + 0), ln);
        gemm_split<4, U0, 2>(Ya, bX0, Yb, bX1, ring, off(SE0 + 6 * h + 1), ln);
        gemm_split<4, U0, 2>(Ya, bX1, Yb, bX2, ring, off(SE0 + 6 * h + 2), ln);
        nyq_update(Yb, xn0, wn + 0 * 128 + 64 * h, ln);
        nyq_update(Ya, xn0, wn + 1 * 128 + 64 * h, ln);
        nyq_update(Yb, xn1, wn + 1 * 128 + 64 * h, ln);
        nyq_update(Ya, xn1, wn + 2 * 128 + 64 * h, ln);
        nyq_update(Yb, xn2, wn + 2 * 128 + 64 * h, ln);
        Bop Pa[2], Pb[2];
        relu_pack<4>(Ya, Pa, mx);
        relu_pack<4>(Yb, Pb, mx);
        auto bPa = [&](int u) { return Pa[u]; };
        auto bPb = [&](int u) { return Pb[u]; };
        gemm_split<4, 2, 1>(Z0, bPa, Z0, bPa, ring, off(SE1 + 6 * h + 1), ln);
        gemm_split<4, 2, 1>(Z0, bPb, Z0, bPb, ring, off(SE1 + 6 * h + 2), ln);
        gemm_split<4, 2, 1>(Z1, bPb, Z1, bPb, ring, off(SE1 + 6 * h + 0), ln);
    }

    {
        if (VAD_SPLIT_PRIO) __builtin_amdgcn_s_setprio(0);
        float Xf[Q + 1];
        fft_pass<Q, 3, PcmT>(Xf, a, tab, ln);
        pack_mag<Q>(Xf, X3, mx);
        xn3 = __shfl(Xf[Q], ln.j);
        mx = fmaxf(mx, xn3);
        if (VAD_SPLIT_PRIO) __builtin_amdgcn_s_setprio(VAD_SPLIT_PRIO);
    }

    // ---- frames 2 and 3 of enc0 -> enc1 out 1 taps 1,2 -------------------------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 Ya[4], Yb[4];                              // enc0 frame 2, frame 3
        init_bias<4>(Ya, tab + tb.b_e0 + 64 * h, ln);
        init_bias<4>(Yb, tab + tb.b_e0 + 64 * h, ln);
        gemm_split<4, U0, 2>(Ya, bX1, Yb, bX2, ring, off(SE0 + 6 * h + 0), ln);
        gemm_split<4, U0, 2>(Ya, bX2, Yb, bX3, ring, off(SE0 + 6 * h + 1), ln);
        gemm_split<4, U0, 1>(Ya, bX3, Ya, bX3, ring, off(SE0 + 6 * h + 2), ln);
        nyq_update(Ya, xn1, wn + 0 * 128 + 64 * h, ln);
        nyq_update(Yb, xn2, wn + 0 * 128 + 64 * h, ln);
        nyq_update(Ya, xn2, wn + 1 * 128 + 64 * h, ln);
        nyq_update(Yb, xn3, wn + 1 * 128 + 64 * h, ln);
        nyq_update(Ya, xn3, wn + 2 * 128 + 64 * h, ln);
        Bop Pa[2], Pb[2];
        relu_pack<4>(Ya, Pa, mx);
        relu_pack<4>(Yb, Pb, mx);
        auto bPa = [&](int u) { return Pa[u]; };
        auto bPb = [&](int u) { return Pb[u]; };
        gemm_split<4, 2, 1>(Z1, bPa, Z1, bPa, ring, off(SE1 + 6 * h + 1), ln);
        gemm_split<4, 2, 1>(Z1, bPb, Z1, bPb, ring, off(SE1 + 6 * h + 2), ln);
    }

    // ---- enc2 (stride 2, taps 1,2 see enc1 outputs 0,1), enc3 (centre tap), W_ih ----------------------
    Bop Q0[2], Q1[2];
    relu_pack<4>(Z0, Q0, mx);
    relu_pack<4>(Z1, Q1, mx);
    auto bQ0 = [&](int u) { return Q0[u]; };
    auto bQ1 = [&](int u) { return Q1[u]; };
    f32x4 Vv[4];
    init_bias<4>(Vv, tab + tb.b_e2, ln);
    gemm_split<4, 2, 1>(Vv, bQ0, Vv, bQ0, ring, off(SE2T1), ln);
    gemm_split<4, 2, 1>(Vv, bQ1, Vv, bQ1, ring, off(SE2T2), ln);
    Bop Pv[2];
    relu_pack<4>(Vv, Pv, mx);
    auto bPv = [&](int u) { return Pv[u]; };
    f32x4 Fe[8];
    init_bias<8>(Fe, tab + tb.b_e3, ln);
    gemm_split<8, 2, 1>(Fe, bPv, Fe, bPv, ring, off(SE3T1), ln);
    Bop Pf[4];
    relu_pack<8>(Fe, Pf, mx);
    auto bPf = [&](int u) { return Pf[u]; };

    const bool bad = !(mx < kHalfLimit) || ring.bad;     // out of fp16 range (or NaN input): poison
    const float nanv = __builtin_nanf("");
    float *gxt = a.gx + ((size_t)(ln.st * a.nt + ln.tl) * 32) * 256 + ln.lane * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 G[8];
        init_bias<8>(G, tab + tb.b_g + 128 * q, ln);
        gemm_split<8, 4, 1>(G, bPf, G, bPf, ring, off(SIH0 + q), ln);
        if (ln.tile_valid) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                f32x4 v = G[m];
                if (bad) v = f32x4{nanv, nanv, nanv, nanv};
                f32x4 *dst = reinterpret_cast<f32x4 *>(gxt + (size_t)(8 * q + m) * 256);
                if (VAD_SPLIT_NT_GX) __builtin_nontemporal_store(v, dst);
                else *dst = v;
            }
        }
    }
}

}  // namespace

template <typename PcmT>
hipError_t launch_front_split(int sr, const FrontArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    const unsigned grid = (unsigned)((total + kWV - 1) / kWV);
    if (sr == 16000) hipLaunchKernelGGL((front_split_kernel<32, PcmT>), dim3(grid), dim3(64 * kWV), 0, s, a);
    else hipLaunchKernelGGL((front_split_kernel<16, PcmT>), dim3(grid), dim3(64 * kWV), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_front_split<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_front_split<int16_t>(int, const FrontArgs &, hipStream_t);

}  // namespace vad
