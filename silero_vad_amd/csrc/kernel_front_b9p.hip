// kernel_front_b9p.hip -- the bf16 x 9 frontend in its PAIR form: the function, the arithmetic and the BITS of kernel_front_b9.hip
// (framing, reflect pad, window, 4 x real FFT magnitude, encoder 0 as one F(4,3) tile, encoders 1-3, W_ih; every matrix product as
// nine exact bf16 piece products on v_mfma_f32_16x16x32_bf16, fp32 accumulation) with a 16-chunk tile shared by TWO waves.
//
// Why.  The narrow form forms the F(4,3) input transform of a K32 step and splits it into bf16 pieces once per 32-row part (4 x at
// 16 kHz) because one 243-register wave holds one part's accumulators beside the tile's 132 magnitudes; that recomputation is more
// than half of its VALU work, which does not hide beside its matrix work (profiles/r03p_front_bf16x9.md).  A wave that owns the whole
// tile and all 128 rows avoids it but needs 433 registers, is alone on its SIMD and exposes every latency (profiles/r04c).  Here the
// two waves of a pair split the ROWS of every layer (wave h: rows 64 h .. 64 h + 63 of encoder 0, half of every later layer) and the
// K32 STEPS of the operand work: a wave keeps only the magnitudes of its own K32 steps (64 registers at 16 kHz), forms and splits the
// operand of a step once per PAIR, and hands the three pieces to its partner through LDS (3 KiB per step); later layers' activations
// are split by the wave that holds them and exchanged the same way.  Accumulators halve (64 + 16 registers), two such waves share a
// SIMD, the split work per tile is the wide form's, and a wave's 36 MFMAs per unit stand beside a split only every other unit.
//
// Same bits: every accumulator sees exactly the MFMAs it sees in the narrow program, in the same order (K32 steps ascending; per step
// piece pa of A against pieces 0, 1, 2 of B); fold, Nyquist update, biases and ReLU are the narrow kernel's expressions on the wave's own
// rows.  The weight image is the WIDE program's (layout.hpp): encoder 0 matrix by matrix over all 8 row blocks, then the narrow units.
// tests: test_front_b9_pair_equals_narrow.  (reference: the same lines as kernel_front_f43.hip.)
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_wave.hpp"
#include "front_common.hpp"

namespace vad {
namespace {

constexpr int kWavesP = 8;                                  // 4 tiles x 2 waves: one workgroup per CU, two waves per SIMD
constexpr int kUnitBytesP = (int)vadl::kW9UnitHalfs * 2;    // 24 fragments of 1 KiB
constexpr int kShareP = kUnitBytesP / kWavesP;              // a wave's share of a unit's DMA: 3 KiB
constexpr int kSlotBytes = 3072;                            // one K32 step's B operand as three pieces: [piece 3][lane 64][16 B]
constexpr int kPairBytes = 16 * 1024;                       // a pair's exchange area: magnitudes during the FFT phase, then 4 piece slots
using f32x2 = float __attribute__((ext_vector_type(2)));
using bf8 = __bf16 __attribute__((ext_vector_type(8)));
using bf2 = __bf16 __attribute__((ext_vector_type(2)));
using lds_u32x4 = __attribute__((address_space(3))) const u32x4;
using lds_u32x4_w = __attribute__((address_space(3))) u32x4;
using lds_f32x4_w = __attribute__((address_space(3))) f32x4;
__device__ __forceinline__ u32x4 lds4u(unsigned byte_addr) { return *reinterpret_cast<lds_u32x4 *>(byte_addr); }
__device__ __forceinline__ void lds4u_store(unsigned byte_addr, u32x4 v) { *reinterpret_cast<lds_u32x4_w *>(byte_addr) = v; }
__device__ __forceinline__ void lds4f_store(unsigned byte_addr, f32x4 v) { *reinterpret_cast<lds_f32x4_w *>(byte_addr) = v; }
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
// the 8 B values f(0..7) a lane holds for a K32 step -> its three pieces, stored to a piece slot (byte address incl. lane * 16)
template <class F>
__device__ __forceinline__ void split_to_slot(unsigned slot_addr, F f) {
    u32x4 bp[3];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
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
#pragma unroll
    for (int k = 0; k < 3; ++k) lds4u_store(slot_addr + k * 1024, bp[k]);
}
struct Pieces {
    u32x4 p[3];
};
__device__ __forceinline__ Pieces read_slot(unsigned slot_addr) {
    Pieces b;
#pragma unroll
    for (int k = 0; k < 3; ++k) b.p[k] = (VAD_ABLATE & 256) ? u32x4{slot_addr, 1u, 2u, 3u + (unsigned)k} : lds4u(slot_addr + k * 1024);   // (256: timing only)
    return b;
}

// ---- the weight ring (3 slots of one 24 KiB unit, shared by the workgroup's 8 waves) ---------------------------------------------
struct RingP {
    unsigned a_cur, a_nxt, a_far;       // LDS byte address of lane's offset in the slot of unit u, u+1, u+2
    unsigned d_cur, d_nxt, d_far;       // wave-uniform: where this wave's share of a unit lands in those slots
    const char *src;                    // wave-uniform: this wave's share of the next unit to request
    unsigned voff;                      // lane * 16
    u32x4 c0, c1;                       // the two A fragments of the next sub-step
};
__device__ __forceinline__ void ring_request(RingP &r) {
    if (VAD_ABLATE & 8) return;
    unsigned keep_m0;                      // M0 is restored: the compiler may keep its own value there
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(r.voff), "s"(r.src), "s"(r.d_far) : "memory");
    r.src += kUnitBytesP;
}
__device__ __forceinline__ void ring_rotate(RingP &r) {
    const unsigned a = r.a_cur, d = r.d_cur;
    r.a_cur = r.a_nxt; r.a_nxt = r.a_far; r.a_far = a;
    r.d_cur = r.d_nxt; r.d_nxt = r.d_far; r.d_far = d;
}
// in the middle of a unit: this wave's share of the next unit has landed; then everyone's, and everyone has left the previous unit.  The
// barrier is also the pair's: piece slots written before it are the partner's to read behind it (hence the LDS fence in front of it)
template <int AFTER>
__device__ __forceinline__ void ring_mid(RingP &r) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!(VAD_ABLATE & 1)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (AFTER >= 2) ring_request(r);
}
// a barrier outside the ring's rhythm (a layer's activations are complete: their pieces have been written, the partner may read)
__device__ __forceinline__ void pair_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(VAD_ABLATE & 1)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- units -------------------------------------------------------------------------------------------------------------------
// A wave consumes HALF of every 24 KiB unit -- the fragments of its own rows -- in 6 sub-steps of 6 MFMAs (2 row blocks x 3 B pieces):
//   KW  encoder-0 unit of the wide program [piece 3][row block 8]: sub-step q = (pa, pr), row blocks 4 h + 2 pr, + 1; one B block
//   K4  narrow unit of an M = 4 segment [step 4 = (kp 2, row-block pair 2)][piece 3][row block 2]: the wave's pair is mh = h, sub-step
//       q = (kp, pa): B block kp, always the wave's two accumulator blocks
//   K8  narrow unit of an M = 8 segment [step 4 = row-block pair][piece 3][row block 2] (one K32 step): the wave's pairs are 2 h, 2 h + 1,
//       sub-step q = (s2, pa): accumulator blocks 2 s2, + 1; one B block
enum Kind { KW = 0, K4 = 1, K8 = 2 };
__host__ __device__ constexpr int hstride(int kind) { return kind == KW ? 4096 : kind == K4 ? 6144 : 12288; }
__host__ __device__ constexpr int qoff(int kind, int q) {       // byte offset of the sub-step's first fragment (second: + 1024), without h
    return kind == KW ? ((q / 2) * 8 + 2 * (q % 2)) * 1024 : kind == K4 ? ((2 * (q / 3)) * 3 + q % 3) * 2048 : q * 2048;
}
__host__ __device__ constexpr int q_acc(int kind, int q) { return kind == KW ? q % 2 : kind == K4 ? 0 : q / 3; }     // accumulator pair
__host__ __device__ constexpr int q_blk(int kind, int q) { return kind == K4 ? q / 3 : 0; }                           // B block

// One unit.  acc: the wave's accumulator blocks of the segment (pairs as q_acc says); b0, b1: the unit's B blocks (K4 uses both);
// early(): work placed in front of the first sub-step's MFMAs (the owner's split of a LATER unit: its stores reach LDS before the
// barrier in the middle of this unit); late(): work placed behind that barrier (reading the next unit's pieces).
template <int KIND, int NEXT, int AFTER, int NACC, class EARLY, class LATE>
__device__ __forceinline__ void unit(f32x4 (&acc)[NACC], const Pieces &b0, const Pieces &b1, EARLY early, LATE late, RingP &r, int h) {
    const unsigned base = r.a_cur + (unsigned)(h * hstride(KIND));
    static_for<0, 6>([&](auto qc) VAD_INLINE {
        constexpr int q = decltype(qc)::value, ap = 2 * q_acc(KIND, q);
        static_assert(ap + 1 < NACC, "accumulator pair");
        if constexpr (q == 3) {
            if constexpr (AFTER >= 1) ring_mid<AFTER>(r);
            else pair_barrier();
            late();
        }
        u32x4 n0 = r.c0, n1 = r.c1;
        if constexpr (VAD_ABLATE & 128) {          // timing only: no fragment reads
        } else if constexpr (q + 1 < 6) {
            n0 = lds4u(base + qoff(KIND, q + 1));
            n1 = lds4u(base + qoff(KIND, q + 1) + 1024);
        } else if constexpr (AFTER >= 1) {
            const unsigned nb = r.a_nxt + (unsigned)(h * hstride(NEXT));
            n0 = lds4u(nb + qoff(NEXT, 0));
            n1 = lds4u(nb + qoff(NEXT, 0) + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (q == 0) early();
        const Pieces &b = q_blk(KIND, q) ? b1 : b0;
#pragma unroll
        for (int pb = 0; pb < 3; ++pb) {
            acc[ap + 0] = mfma_b(r.c0, b.p[pb], acc[ap + 0]);
            acc[ap + 1] = mfma_b(r.c1, b.p[pb], acc[ap + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        r.c0 = n0;
        r.c1 = n1;
    });
    ring_rotate(r);
}

template <int Q, typename PcmT, int DEC>
__global__ void __launch_bounds__(64 * kWavesP, 1) front_b9p_kernel(const FrontArgs a) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int RB = w_rb(Q), P = w_parts(Q), KP = w9w_kp(Q), QH = Q / 2, NE0 = 6 * KP;
    static_assert(w9w_tail0(Q) + 20 == w9w_units(Q), "program mismatch");
    // LDS: [biases + head + Nyquist weights: NS floats][ring 3 x 24 KiB][4 pairs x 16 KiB exchange / piece slots][Nyquist exchange];
    // the FFT's tables (window, twiddles) sit in ring slot 2 until the first request into it (middle of unit 0)
    constexpr int NS = tb.window + (tb.total - tb.w_nyq), NF = tb.w_nyq - tb.window, UF = kUnitBytesP / 4;
    static_assert(tb.window % 4 == 0 && tb.w_nyq % 4 == 0 && NS % 4 == 0 && NF <= UF && tb.total % 4 == 0, "table split");
    __shared__ __attribute__((aligned(16))) float lds[NS + 3 * UF + 4 * (kPairBytes / 4) + 4 * 2 * 2 * 64];
    float *tab = lds;                                     // + off for off < tb.window
    float *tabn = lds + tb.window - tb.w_nyq;             // + tb.w_nyq + ... for the Nyquist weights
    float *tabf = lds + NS + 2 * UF - tb.window;          // + tb.window / tb.tw1 / tb.tw2 for the FFT

    Lane ln;
    ln.lane = threadIdx.x & 63;
    ln.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ln.g = ln.lane >> 4;
    ln.j = ln.lane & 15;
    // pair = tile of the workgroup, h = which half of the rows / which K32 steps; the two waves that share a SIMD (w, w + 4) differ in h
    const int pair = ln.wave >> 1, h = (ln.wave ^ (ln.wave >> 2)) & 1;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    long wt = (long)blockIdx.x * 4 + pair;
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

    RingP ring;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) float *)lds);
    const unsigned slot0 = lds0 + NS * 4;
    const unsigned xch = slot0 + 3 * kUnitBytesP + (unsigned)pair * kPairBytes + (unsigned)ln.lane * 16;      // this lane's column of the pair's area
    const unsigned xnq = slot0 + 3 * kUnitBytesP + 4 * kPairBytes + (unsigned)pair * (2 * 2 * 64 * 4) + (unsigned)ln.lane * 4;
    {
        ring.voff = ln.lane * 16;
        ring.a_cur = slot0 + ring.voff;
        ring.a_nxt = ring.a_cur + kUnitBytesP;
        ring.a_far = ring.a_cur + 2 * kUnitBytesP;
        // the two priming requests go to slots 0 and 1: start rotated by two, so that "far" is slot 0 first, then slot 1
        ring.d_far = slot0 + (unsigned)ln.wave * (unsigned)kShareP;
        ring.d_cur = ring.d_far + kUnitBytesP;
        ring.d_nxt = ring.d_far + 2 * kUnitBytesP;
        ring.src = reinterpret_cast<const char *>(a.wfront) + ln.wave * kShareP;
        ring_request(ring);                               // unit 0 -> slot 0
        {   const unsigned d = ring.d_far; ring.d_far = ring.d_cur; ring.d_cur = ring.d_nxt; ring.d_nxt = d; }
        ring_request(ring);                               // unit 1 -> slot 1
        {   const unsigned d = ring.d_far; ring.d_far = ring.d_cur; ring.d_cur = ring.d_nxt; ring.d_nxt = d; }
        // now d_far = slot 2 (unit 2's), d_cur = slot 0, d_nxt = slot 1
    }
    {   // tables -> LDS: all loads of a thread are issued before the first is stored
        constexpr int NT = 64 * kWavesP, NV = tb.total / 4, PER = (NV + NT - 1) / NT;
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

    // ---- the wave's two frames (2 h, 2 h + 1), then the exchange: a wave keeps the magnitudes of ITS K32 steps of all four frames -----
    // own K32 steps: kp = h (mod 2); local index i < Q/2 <-> k-step s = 8 (2 (i / 8) + h) + i % 8
    float XH0[QH], XH1[QH], XH2[QH], XH3[QH];
    float xn0, xn1, xn2, xn3;
    {
        float XA[Q + 1], XB[Q + 1];
#pragma unroll
        for (int k = 0; k <= Q; ++k) XB[k] = 0.f;
#pragma clang loop unroll(disable)
        for (int v = 0; v < 2; ++v) {
#pragma unroll
            for (int k = 0; k <= Q; ++k) XA[k] = XB[k];
            fft_frame<Q, PcmT, DEC>(XB, 2 * h + v, a, tabf, ln);
        }
        // the partner's K32 steps of my two frames -> the pair's area [dst half][frame of the source wave 2][vec Q/8][lane]; mine stay
        const unsigned wr = xch + (unsigned)(1 - h) * (kPairBytes / 2);
        auto give = [&](const float (&X)[Q + 1], int fl) VAD_INLINE {
#pragma unroll
            for (int v4 = 0; v4 < QH / 4; ++v4) {
                f32x4 o, p;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * v4 + e, s0 = 8 * (2 * (i / 8)) + i % 8;          // the k-step for h = 0; + 8 for h = 1
                    o[e] = X[s0];
                    p[e] = X[s0 + 8];
                }
                lds4f_store(wr + (unsigned)((fl * (QH / 4) + v4) * 1024), h ? o : p);   // the OTHER half's steps
            }
        };
        give(XA, 0);
        give(XB, 1);
        // Nyquist magnitudes (mag layout keeps them in lane group 0; every lane of the chunk needs them, and so does the partner)
        const float na = __shfl(XA[Q], ln.j), nb = __shfl(XB[Q], ln.j);
        *reinterpret_cast<__attribute__((address_space(3))) float *>(xnq + (unsigned)(h * 2 * 64 * 4)) = na;
        *reinterpret_cast<__attribute__((address_space(3))) float *>(xnq + (unsigned)(h * 2 * 64 * 4 + 256)) = nb;
        pair_barrier();
        const unsigned rd = xch + (unsigned)h * (kPairBytes / 2);
        float YA[QH], YB[QH];                               // the partner's frames, my K32 steps
#pragma unroll
        for (int v4 = 0; v4 < QH / 4; ++v4) {
            const f32x4 u = *reinterpret_cast<__attribute__((address_space(3))) const f32x4 *>(rd + (unsigned)(v4 * 1024));
            const f32x4 w = *reinterpret_cast<__attribute__((address_space(3))) const f32x4 *>(rd + (unsigned)((QH / 4 + v4) * 1024));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                YA[4 * v4 + e] = u[e];
                YB[4 * v4 + e] = w[e];
            }
        }
        const float pa_ = *reinterpret_cast<__attribute__((address_space(3))) const float *>(xnq + (unsigned)((1 - h) * 2 * 64 * 4));
        const float pb_ = *reinterpret_cast<__attribute__((address_space(3))) const float *>(xnq + (unsigned)((1 - h) * 2 * 64 * 4 + 256));
#pragma unroll
        for (int i = 0; i < QH; ++i) {
            const int s0 = 8 * (2 * (i / 8)) + i % 8;
            const float ma = h ? XA[s0 + 8] : XA[s0], mb = h ? XB[s0 + 8] : XB[s0];   // my frames, my steps
            XH0[i] = h ? YA[i] : ma;                       // frame 0 = wave 0's first frame
            XH1[i] = h ? YB[i] : mb;
            XH2[i] = h ? ma : YA[i];
            XH3[i] = h ? mb : YB[i];
        }
        xn0 = h ? pa_ : na;
        xn1 = h ? pb_ : nb;
        xn2 = h ? na : pa_;
        xn3 = h ? nb : pb_;
    }
    // The input transform reads the frames through E = x3 - x1 and F = x2 - x0 (kept in place of x3 and x0), as the narrow kernel does
#pragma unroll
    for (int k = 0; k < QH; ++k) {
        XH3[k] = XH3[k] - XH1[k];
        XH0[k] = XH2[k] - XH0[k];
    }
    // t_j at the wave's local k-step i, j in program order (U1, U2, U3, U4, U0, U5): the narrow kernel's expressions, term for term
    auto tval = [&](auto jc, int i) VAD_INLINE -> float {
        constexpr int j = decltype(jc)::value;
        if constexpr (j == 0) return fmaf(fmaf(XH2[i], 1.0f, XH1[i]), -3.0f, fmaf(XH0[i], 4.0f, XH3[i]));       // (E + 4F) - 3(x1 + x2)
        else if constexpr (j == 1) return fmaf(fmaf(XH1[i], -1.0f, XH2[i]), 3.0f, fmaf(XH0[i], -4.0f, XH3[i])); // (E - 4F) + 3(x2 - x1)
        else if constexpr (j == 2) return fmaf(XH0[i], 2.0f, XH3[i]);                                            // E + 2F
        else if constexpr (j == 3) return fmaf(XH0[i], -2.0f, XH3[i]);                                           // E - 2F
        else if constexpr (j == 4) return fmaf(XH1[i], -4.0f, XH3[i]);                                           // E - 4 x1
        else return fmaf(XH2[i], -0.25f, -XH0[i]);                                                               // -F - x2 / 4
    };
    // the pieces of encoder-0 unit c = (j, kp): formed by the wave that owns K32 step kp, into piece slot c & 1
    auto e0_split = [&](auto cc) VAD_INLINE {
        constexpr int c = decltype(cc)::value, j = c / KP, kp = c % KP;
        if (h == (kp & 1))
            split_to_slot(xch + (unsigned)((c & 1) * kSlotBytes),
                          [&](int e) VAD_INLINE { return tval(std::integral_constant<int, j>{}, 8 * (kp / 2) + e); });
    };
    // (the exchange area is the piece slots' from here on: every wave has read its magnitudes -- they are behind the loads above --
    //  before the barrier that follows the first split)
    e0_split(std::integral_constant<int, 0>{});
    pair_barrier();

    // ---- encoder 0: six matrices over the wave's 64 rows ----------------------------------------------------------------------
    const float *bias0 = tab + tb.b_e0 + 64 * h;
    f32x4 Y0[4], Y1[4], Y2[4], Y3[4];                      // m1, m2, m3, m4, then the four frame outputs (rows 64 h + 16 m + 4 g + r)
    init_bias<4>(Y0, bias0, ln);
    zero<4>(Y1);
    zero<4>(Y2);
    zero<4>(Y3);
    ring.c0 = lds4u(ring.a_cur + (unsigned)(h * hstride(KW)) + qoff(KW, 0));
    ring.c1 = lds4u(ring.a_cur + (unsigned)(h * hstride(KW)) + qoff(KW, 0) + 1024);
    Pieces bcur = read_slot(xch), bnxt = bcur;
    static_for<0, NE0>([&](auto cc) VAD_INLINE {
        constexpr int c = decltype(cc)::value, j = c / KP;
        auto early = [&]() VAD_INLINE {
            if constexpr (c + 1 < NE0) e0_split(std::integral_constant<int, c + 1>{});
        };
        auto late = [&]() VAD_INLINE {
            if constexpr (c + 1 < NE0) bnxt = read_slot(xch + (unsigned)(((c + 1) & 1) * kSlotBytes));
        };
        if constexpr (c == 4 * KP) {
            // m1..m4 are complete: fold them into the four frame outputs (the narrow kernel's expressions)
#pragma unroll
            for (int m = 0; m < 4; ++m)
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
        constexpr int NEXT = c + 1 < NE0 ? KW : K4;
        if constexpr (j == 0 || j == 4) unit<KW, NEXT, 2>(Y0, bcur, bcur, early, late, ring, h);        // m1; m0 onto y0
        else if constexpr (j == 1) unit<KW, NEXT, 2>(Y1, bcur, bcur, early, late, ring, h);
        else if constexpr (j == 2) unit<KW, NEXT, 2>(Y2, bcur, bcur, early, late, ring, h);
        else unit<KW, NEXT, 2>(Y3, bcur, bcur, early, late, ring, h);                                   // m4; m5 onto y3
        bcur = bnxt;
    });
    {   // Nyquist bin (exact fp32 rank-1 updates) on the wave's rows, ReLU
        const float *wn = tabn + tb.w_nyq + 64 * h;         // [tap][row]
        nyq_update<4>(Y0, xn0, wn + 128, ln);
        nyq_update<4>(Y0, xn1, wn + 256, ln);
        nyq_update<4>(Y1, xn0, wn, ln);
        nyq_update<4>(Y1, xn1, wn + 128, ln);
        nyq_update<4>(Y1, xn2, wn + 256, ln);
        nyq_update<4>(Y2, xn1, wn, ln);
        nyq_update<4>(Y2, xn2, wn + 128, ln);
        nyq_update<4>(Y2, xn3, wn + 256, ln);
        nyq_update<4>(Y3, xn2, wn, ln);
        nyq_update<4>(Y3, xn3, wn + 128, ln);
        relu<4>(Y0);
        relu<4>(Y1);
        relu<4>(Y2);
        relu<4>(Y3);
    }

    // ---- encoder 1: the narrow program's units, part by part; a unit's two B blocks come from the wave that holds the part's rows ----
    // piece slots: unit d of the layer uses slots 2 (d & 1), + 1; the first unit's pieces go in front of a barrier of their own, every
    // later unit's are written during the unit before it
    constexpr int NE1 = Q == 32 ? 10 : 10;
    // unit d -> (part, which of the part's units)
    auto e1_write = [&](auto dc) VAD_INLINE {
        constexpr int d = decltype(dc)::value;
        constexpr int part = Q == 32 ? (d < 2 ? 0 : d < 5 ? 1 : d < 7 ? 2 : 3) : d / 5;
        constexpr int idx = Q == 32 ? (d < 2 ? d : d < 5 ? d - 2 : d < 7 ? d - 5 : d - 7) : d % 5;
        constexpr int owner = part * RB / 4;                // the wave that holds rows 16 RB part ..
        constexpr int m0 = (part * RB) % 4;                 // ... as its accumulator blocks m0 ..
        const unsigned s0 = xch + (unsigned)((2 * (d & 1)) * kSlotBytes), s1 = s0 + kSlotBytes;
        if (h != owner) return;
        if constexpr (Q == 32) {
            // two taps share a unit: K32 step 0 <- the first tap's frame, K32 step 1 <- the second's (8 k-steps = the part's 2 row blocks)
            auto blk = [&](const f32x4 (&A)[4], int mm, unsigned slot) VAD_INLINE {
                split_to_slot(slot, [&](int e) VAD_INLINE { return A[mm + (e >> 2)][e & 3]; });
            };
            if constexpr (idx == 0) { blk(Y0, m0, s0); blk(Y1, m0, s1); }                 // out 0: tap 1 <- y0 | tap 2 <- y1
            else if constexpr (idx == 1) { blk(Y1, m0, s0); blk(Y2, m0, s1); }            // out 1: tap 0 <- y1 | tap 1 <- y2
            else { blk(Y3, m0 - RB, s0); blk(Y3, m0, s1); }                               // out 1: tap 2 <- y3 of the even part | of this part
        } else {
            // 16 k-steps per tap and part: K32 step kp <- the part's row blocks 2 kp, 2 kp + 1
            auto both = [&](const f32x4 (&A)[4]) VAD_INLINE {
                split_to_slot(s0, [&](int e) VAD_INLINE { return A[m0 + (e >> 2)][e & 3]; });
                split_to_slot(s1, [&](int e) VAD_INLINE { return A[m0 + 2 + (e >> 2)][e & 3]; });
            };
            if constexpr (idx == 0) both(Y0);              // out 0, tap 1 <- y0
            else if constexpr (idx == 1 || idx == 2) both(Y1);   // out 0, tap 2 <- y1 ; out 1, tap 0 <- y1
            else if constexpr (idx == 3) both(Y2);         // out 1, tap 1 <- y2
            else both(Y3);                                 // out 1, tap 2 <- y3
        }
    };
    // which accumulator a unit of encoder 1 adds to: out 0 (Z0) or out 1 (Z1)
    f32x4 Z0[2], Z1[2];                                    // the wave's 2 of encoder 1's 4 row blocks: rows 32 h + 16 m + ...
    init_bias<2>(Z0, tab + tb.b_e1 + 32 * h, ln);
    init_bias<2>(Z1, tab + tb.b_e1 + 32 * h, ln);
    e1_write(std::integral_constant<int, 0>{});
    pair_barrier();
    Pieces b0 = read_slot(xch), b1 = read_slot(xch + kSlotBytes), n0p = b0, n1p = b1;
    static_for<0, NE1>([&](auto dc) VAD_INLINE {
        constexpr int d = decltype(dc)::value;
        constexpr int idx = Q == 32 ? (d < 2 ? d : d < 5 ? d - 2 : d < 7 ? d - 5 : d - 7) : d % 5;
        constexpr bool to_z0 = Q == 32 ? idx == 0 : idx < 2;
        auto early = [&]() VAD_INLINE {
            if constexpr (d + 1 < NE1) e1_write(std::integral_constant<int, d + 1>{});
        };
        auto late = [&]() VAD_INLINE {
            if constexpr (d + 1 < NE1) {
                n0p = read_slot(xch + (unsigned)((2 * ((d + 1) & 1)) * kSlotBytes));
                n1p = read_slot(xch + (unsigned)((2 * ((d + 1) & 1) + 1) * kSlotBytes));
            }
        };
        if constexpr (to_z0) unit<K4, K4, 2>(Z0, b0, b1, early, late, ring, h);
        else unit<K4, K4, 2>(Z1, b0, b1, early, late, ring, h);
        b0 = n0p;
        b1 = n1p;
    });
    relu<2>(Z0);
    relu<2>(Z1);

    // ---- encoder 2 (T 2 -> 1, stride 2: taps 1, 2 see encoder 1's outputs 0, 1): K32 step kp <- row blocks 2 kp, + 1 = wave kp's -----
    auto own_block = [&](const f32x4 (&A)[2], unsigned slot) VAD_INLINE {
        split_to_slot(slot, [&](int e) VAD_INLINE { return A[e >> 2][e & 3]; });
    };
    f32x4 Vv[2];
    init_bias<2>(Vv, tab + tb.b_e2 + 32 * h, ln);
    own_block(Z0, xch + (unsigned)(h * kSlotBytes));                       // unit 0: slots 0, 1
    own_block(Z1, xch + (unsigned)((2 + h) * kSlotBytes));                 // unit 1: slots 2, 3
    pair_barrier();
    {
        const Pieces p0 = read_slot(xch), p1 = read_slot(xch + kSlotBytes);
        const Pieces p2 = read_slot(xch + 2 * kSlotBytes), p3 = read_slot(xch + 3 * kSlotBytes);
        auto none = [&]() VAD_INLINE {};
        unit<K4, K4, 2>(Vv, p0, p1, none, none, ring, h);
        unit<K4, K8, 2>(Vv, p2, p3, none, none, ring, h);
    }
    relu<2>(Vv);
    // ---- encoder 3 (T = 1: centre tap only): 64 -> 128, the wave's 4 row blocks; K32 step kp <- Vv of wave kp ---------------------------
    f32x4 Fe[4];
    init_bias<4>(Fe, tab + tb.b_e3 + 64 * h, ln);
    pair_barrier();                                        // (everyone has read encoder 2's pieces)
    own_block(Vv, xch + (unsigned)(h * kSlotBytes));
    pair_barrier();
    {
        const Pieces p0 = read_slot(xch), p1 = read_slot(xch + kSlotBytes);
        auto none = [&]() VAD_INLINE {};
        unit<K8, K8, 2>(Fe, p0, p0, none, none, ring, h);
        unit<K8, K8, 2>(Fe, p1, p1, none, none, ring, h);
    }
    relu<4>(Fe);

    // ---- W_ih: 128 -> 512, one gate (8 row blocks, the wave's 4) at a time; K32 step kp <- Fe row blocks 2 kp, + 1 = wave kp / 2's ------
    pair_barrier();                                        // (everyone has read encoder 3's pieces)
    split_to_slot(xch + (unsigned)((2 * h) * kSlotBytes), [&](int e) VAD_INLINE { return Fe[e >> 2][e & 3]; });
    split_to_slot(xch + (unsigned)((2 * h + 1) * kSlotBytes), [&](int e) VAD_INLINE { return Fe[2 + (e >> 2)][e & 3]; });
    pair_barrier();
    const Pieces w0 = read_slot(xch), w1 = read_slot(xch + kSlotBytes), w2 = read_slot(xch + 2 * kSlotBytes), w3 = read_slot(xch + 3 * kSlotBytes);
    // gx in D-fragment order: row block 8 q + 4 h + m of the tile's 32
    float *gxt = a.gx + ((size_t)(ln.st * a.nt + ln.tl) * 32 + 4 * h) * 256 + ln.lane * 4;
    const float *bg = tab + tb.b_g + 64 * h;
    auto none = [&]() VAD_INLINE {};
#pragma clang loop unroll(disable)
    for (int q = 0; q < 3; ++q) {
        f32x4 G[4];
        init_bias<4>(G, bg, ln);
        unit<K8, K8, 2>(G, w0, w0, none, none, ring, h);
        unit<K8, K8, 2>(G, w1, w1, none, none, ring, h);
        unit<K8, K8, 2>(G, w2, w2, none, none, ring, h);
        unit<K8, K8, 2>(G, w3, w3, none, none, ring, h);
        if (ln.tile_valid) {
#pragma unroll
            for (int m = 0; m < 4; ++m) *reinterpret_cast<f32x4 *>(gxt + (size_t)m * 256) = G[m];
        }
        gxt += 8 * 256;
        bg += 128;
    }
    {
        f32x4 G[4];
        init_bias<4>(G, bg, ln);
        unit<K8, K8, 2>(G, w0, w0, none, none, ring, h);
        unit<K8, K8, 2>(G, w1, w1, none, none, ring, h);
        unit<K8, K8, 1>(G, w2, w2, none, none, ring, h);
        unit<K8, K8, 0>(G, w3, w3, none, none, ring, h);
        if (ln.tile_valid) {
#pragma unroll
            for (int m = 0; m < 4; ++m) *reinterpret_cast<f32x4 *>(gxt + (size_t)m * 256) = G[m];
        }
    }
}

}  // namespace

template <typename PcmT>
hipError_t launch_front_b9p(int sr, const FrontArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    const unsigned grid = (unsigned)((total + 3) / 4);
    // a.dec == 2, 3: 32 / 48 kHz input, decimation folded into the loads (fft_wave.hpp load_slice; 16 kHz net only)
    if (a.dec > 1 && (sr != 16000 || a.dec > 3)) return hipErrorInvalidValue;
    if (a.dec == 3) hipLaunchKernelGGL((front_b9p_kernel<32, PcmT, 3>), dim3(grid), dim3(64 * kWavesP), 0, s, a);
    else if (a.dec == 2) hipLaunchKernelGGL((front_b9p_kernel<32, PcmT, 2>), dim3(grid), dim3(64 * kWavesP), 0, s, a);
    else if (sr == 16000) hipLaunchKernelGGL((front_b9p_kernel<32, PcmT, 1>), dim3(grid), dim3(64 * kWavesP), 0, s, a);
    else hipLaunchKernelGGL((front_b9p_kernel<16, PcmT, 1>), dim3(grid), dim3(64 * kWavesP), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_front_b9p<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_front_b9p<int16_t>(int, const FrontArgs &, hipStream_t);

}  // namespace vad
