// kernel_front_lat.hip -- the LATENCY form of the fp32 frontend: the same function as kernel_front_f43.hip
//   PCM -> framing + right reflect pad -> Hann window -> 4 x real FFT magnitude -> 4 x ReLU(Conv1d k=3)
//       -> W_ih * feat + (b_ih + b_hh)  => gx
// and the same arithmetic, bit for bit, for launches that are too small to fill the chip the throughput way.
// (reference: JIT!/vad/model/vad_annotator.py:58-67 framing, JIT!/vad/utils/pytorch_stft.py:17-34 STFT,
//  JIT!/vad/utils/model_utils.py:19-25 encoder, the W_ih half of aten::lstm_cell JIT!/torch/nn/modules/rnn.py:69.)
//
// Why.  kernel_front_f43.hip gives one 16-chunk tile to ONE wave, which walks the whole 54-unit weight program on its own:
// 3 456 MFMAs x 32 cycles + the FFT = 135 k cycles = 56 us at best, 90 us in practice with one wave per SIMD.  With tens
// of thousands of tiles that is the right shape (two waves per SIMD interleave, four tiles share one weight stream).  One
// step of 8 192 live streams (BASELINE configs[4]) is 512 tiles, a `model(chunk, sr)` call of an unmodified caller is ONE
// (src/silero_vad/utils_vad.py:328, :528): three quarters of the chip's SIMDs idle while the rest take 90 us.
//
// How.  One workgroup = 4 waves = ONE tile; the output rows of every layer are split over the 4 waves (encoder 0: 32 of its
// 128 rows each -- a "row part" of the F(4,3) program at 16 kHz --, encoder 1 and 2: one 16-row block each, encoder 3: two,
// W_ih: one LSTM gate each), so a wave issues 864 MFMAs instead of 3 456.  Between layers the activations (the next layer's
// B operand) are exchanged through LDS in MFMA D-fragment order -- which IS the B-operand order of the chain layout
// (layout.hpp), so a lane reads back exactly the 16 bytes another lane of the same (g, j) wrote -- and each wave sums over
// ALL input channels in the order the one-wave program does: same products, same order per accumulator, identical bits
// (tests/test_gpu_parity.py::test_latency_frontend_is_bit_identical).  The 4 STFT frames are split the same way: wave v
// transforms frame v, the magnitudes are exchanged through LDS.
// Measured (one MI355X, tools/lat_time.py): one step of 8 192 streams 84 -> 54 us, of <= 4 096 streams 84 -> 30 us, a B = 1..16 call
// 80 -> 27 us.  Of the 27 us of one tile ~6 are the wave's STFT frame (PCM latency + a VALU-bound FFT nobody overlaps), ~3 A-fragment
// waits, ~18 the MFMA + VALU issue time itself (timing-only ablations: variants lat_nofft / lat_noload / lat_neither); two workgroups
// per CU (variant lat_wg2) do not help, because all workgroups of a step run in phase.
//   One workgroup per CU (launch bound 256 x 1): with a single wave per SIMD there is nobody to hide latency, so the A
// fragments do not go through an LDS ring and its barriers; each wave streams exactly the blocks it needs from the L2-resident
// image straight into registers, kD = 8 blocks (32 MFMAs, ~1 000 cycles) ahead (deeper measured no faster: tools/variants.py lat_d*),
// and everything a wave waits for first (its PCM slice, the head of its stream, the tables) is requested before anything else.
#include <hip/hip_runtime.h>

#include "activations.hpp"
#include "exact_front.hpp"
#include "front_common.hpp"

namespace vad {
namespace {

#ifndef VAD_LAT_DEPTH
#define VAD_LAT_DEPTH 8
#endif
constexpr int kD = VAD_LAT_DEPTH;        // A-fragment prefetch distance, in 1 KiB blocks (one block = 4 MFMAs = 128 pipe cycles)
#define IC(x) (decltype(x)::value)      // the value of an integral_constant argument, as a constant expression

struct Pipe {
    f32x4 q[kD];
};

#ifndef VAD_LAT_ABLATE
#define VAD_LAT_ABLATE 0                 // timing experiments only (wrong results): 1 no A-fragment loads, 2 no FFT
#endif
__device__ __forceinline__ f32x4 ld_blk(const float *lane_base, long blk) {      // lane_base already holds lane * 4
    if (VAD_LAT_ABLATE & 1) return f32x4{1e-3f, 2e-3f, -1e-3f, 5e-4f};
    return *reinterpret_cast<const f32x4 *>(lane_base + blk * 256);
}

// One segment of a wave's program: blocks START .. START + NB - 1 of the wave's block stream, consumed in order; NBS
// consecutive blocks form a step that shares its four B operands (bvec(step)); block i of the segment accumulates into
// acc(i).  load(g) requests block g of the STREAM (any g, also beyond the segment: the FIFO runs kD blocks ahead, across
// segment boundaries, and never restarts cold).
template <int START, int NB, int NBS, class AccF, class BF, class LoadF>
__device__ __forceinline__ void run_segment(Pipe &pp, AccF acc, BF bvec, LoadF load) {
    static_assert(NB % NBS == 0, "segments are whole steps");
    static_for<0, NB / NBS>([&](auto sc) VAD_INLINE {
        constexpr int st = decltype(sc)::value;
        f32x4 a[NBS];
        static_for<0, NBS>([&](auto bc) VAD_INLINE {
            constexpr int b = decltype(bc)::value, gi = START + st * NBS + b;
            a[b] = pp.q[gi % kD];
            pp.q[gi % kD] = load(std::integral_constant<int, gi + kD>{});
        });
        const f32x4 bv = bvec(std::integral_constant<int, st>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            static_for<0, NBS>([&](auto bc) VAD_INLINE {
                constexpr int b = decltype(bc)::value;
                f32x4 &c = acc(std::integral_constant<int, st * NBS + b>{});
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][ks], bv[ks], c, 0, 0, 0);
            });
        __builtin_amdgcn_sched_barrier(0);
    });
}

#ifndef VAD_LAT_WG_PER_CU
#define VAD_LAT_WG_PER_CU 1
#endif
template <int Q, typename PcmT, int DEC, bool CELL>
__global__ void __launch_bounds__(256, VAD_LAT_WG_PER_CU) front_lat_kernel(const FrontArgs a, const CellArgs cell) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int TABF = (tb.total + 3) / 4 * 4;
    constexpr int RB = w_rb(Q), KG0 = Q / 4, T0 = w4_tail0(Q);
    constexpr int NB0 = 2 * KG0;                                 // blocks of one encoder-0 matrix for one wave (2 row blocks)
    // the wave's block stream: encoder 0 (6 matrices), encoder 1 (40 blocks), encoder 2 (8), encoder 3 (8), W_ih (64)
    // and, in the fused single-step form (CELL), W_hh of the wave's gate (64)
    constexpr int S_E1 = 6 * NB0, S_E2 = S_E1 + 40, S_E3 = S_E2 + 8, S_IH = S_E3 + 8, S_HH = S_IH + 64, S_END = S_HH + (CELL ? 64 : 0);
    __shared__ __attribute__((aligned(16))) float tab[TABF];
    // (the STFT magnitudes are dead once every wave has them in registers: encoder 0's output takes their place)
    constexpr int XSF = 4 * (Q + 1) * 64, YBF = 4 * 8 * 256;
    __shared__ __attribute__((aligned(16))) float xy[XSF > YBF ? XSF : YBF];
    float (*xs)[Q + 1][64] = reinterpret_cast<float (*)[Q + 1][64]>(xy);          // [frame][k][lane], frame v by wave v
    float (*ybuf)[8][256] = reinterpret_cast<float (*)[8][256]>(xy);              // encoder 0: [frame][row block][lane][4]
    __shared__ __attribute__((aligned(16))) float zbuf[2][4][256];   // encoder 1: [out][row block][lane][4]
    __shared__ __attribute__((aligned(16))) float vbuf[4][256];      // encoder 2
    __shared__ __attribute__((aligned(16))) float febuf[8][256];     // encoder 3

    Lane ln;
    ln.lane = threadIdx.x & 63;
    ln.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ln.g = ln.lane >> 4;
    ln.j = ln.lane & 15;
    const int w = ln.wave;
    const long wt = blockIdx.x;                                  // one tile per workgroup (grid = tiles)
    ln.tile_valid = true;
    ln.tl = wt % a.nt;
    ln.st = wt / a.nt;
    ln.t = a.t0 + ln.tl;
    const long bb = ln.st * 16 + ln.j;
    ln.b = (int)(bb < a.B ? bb : a.B - 1);
    ln.from_tail = a.tail != nullptr && ln.t == a.T - 1;
    ln.sgnA = ln.g < 2 ? 1.f : -1.f;
    ln.sgnB = (ln.g & 1) ? -1.f : 1.f;

    // this wave's slice of encoder 0: rows 32 w .. 32 w + 31 = row blocks rb0, rb0 + 1 of row part `part`
    const int part = Q == 32 ? w : (w >> 1), rb0 = Q == 32 ? 0 : 2 * (w & 1);
    const int u_e0 = part == 0 ? w4_part0(0, Q) : part == 1 ? w4_part0(1, Q) : part == 2 ? w4_part0(2, Q) : w4_part0(3, Q);
    const float *lane_w = a.wfront + ln.lane * 4;
    const float *e0 = lane_w + (size_t)u_e0 * 4096 + rb0 * 256;              // + j * 4096 + (kg * RB + r) * 256
    const float *e3 = lane_w + (size_t)(T0 + 2) * 4096 + (2 * w) * 256;      // encoder 3: [kg 4][rb 8], this wave's row blocks 2w, 2w + 1
    const float *ih = lane_w + (size_t)(T0 + 4 + 4 * w) * 4096;              // W_ih, gate w: [kg 8][rb 8], 64 consecutive blocks
    const float *hh = CELL ? cell.whh_lat + ln.lane * 4 + (size_t)w * 64 * 256 : nullptr;   // W_hh, gate w, the same order
    // block g of this wave's stream (g is a compile-time constant at every call site)
    auto gload = [&](auto gc) VAD_INLINE -> f32x4 {
        constexpr int g = IC(gc);
        if constexpr (g < S_E1) {
            constexpr int j = g / NB0, i = g % NB0;
            return ld_blk(e0, (long)j * 16 + (i >> 1) * RB + (i & 1));
        } else if constexpr (g < S_E2) {
            constexpr E1Blk eb = e1_blk(Q, g - S_E1);
            return ld_blk(lane_w, (long)eb.unit * 16 + eb.kg * 4 + w);
        } else if constexpr (g < S_E3) {
            constexpr int i = g - S_E2;                           // encoder 2: 2 units x 4 k-groups, row block w
            return ld_blk(lane_w, (long)(T0 + i / 4) * 16 + (i % 4) * 4 + w);
        } else if constexpr (g < S_IH) {
            constexpr int i = g - S_E3;
            return ld_blk(e3, (long)(i >> 1) * 8 + (i & 1));
        } else if constexpr (g < S_HH) {
            return ld_blk(ih, g - S_IH);
        } else if constexpr (g < S_END) {
            return ld_blk(hh, g - S_HH);
        } else {
            return f32x4{0.f, 0.f, 0.f, 0.f};                     // past the end of the program
        }
    };

    // ---- everything this wave will wait for first is requested first: its PCM slice, the head of its weight stream, the tables
    float pcm_s[2 * Q];
    if (!(VAD_LAT_ABLATE & 2)) load_slice<Q, PcmT, DEC>(pcm_s, a, ln, w);          // wave v owns STFT frame v
    f32x4 Hp[CELL ? 8 : 1];                                      // fused step: h_{t-1}, units 16 m + 4 g + r of stream j
    if constexpr (CELL) {
        const float *hsrc = cell.state + (size_t)ln.b * 128 + 4 * ln.g;
#pragma unroll
        for (int m = 0; m < 8; ++m) Hp[m] = *reinterpret_cast<const f32x4 *>(hsrc + 16 * m);
    }
    Pipe pp;
    static_for<0, kD>([&](auto ic) VAD_INLINE { pp.q[IC(ic)] = gload(ic); });
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
    __syncthreads();

    // ---- STFT: wave v transforms frame v; everybody reads all four ------------------------------------------------------------
    float X0[Q + 1], X1[Q + 1], X2[Q + 1], X3[Q + 1];
    {
        float Xm[Q + 1];
        if (VAD_LAT_ABLATE & 2) {
#pragma unroll
            for (int k = 0; k <= Q; ++k) Xm[k] = (float)(ln.lane + k) * 1e-3f;
        } else {
            fft_math<Q>(Xm, pcm_s, tab, ln);
        }
#pragma unroll
        for (int k = 0; k <= Q; ++k) xs[w][k][ln.lane] = Xm[k];
    }
    auto lds_barrier = [&]() VAD_INLINE {                       // (not __syncthreads: that would drain the A-fragment stream too)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    lds_barrier();
#pragma unroll
    for (int k = 0; k <= Q; ++k) {
        X0[k] = xs[0][k][ln.lane];
        X1[k] = xs[1][k][ln.lane];
        X2[k] = xs[2][k][ln.lane];
        X3[k] = xs[3][k][ln.lane];
    }
    lds_barrier();                                               // everybody has read xs: its memory becomes ybuf
    // |Y_nyq| of chunk j lives in lane group 0 (X[Q]); every lane of the chunk needs it
    const float xn0 = __shfl(X0[Q], ln.j), xn1 = __shfl(X1[Q], ln.j), xn2 = __shfl(X2[Q], ln.j), xn3 = __shfl(X3[Q], ln.j);
    // chunks with an exactly silent frame beside one that is not (exact_front.hpp): their gx is recomputed below, in double (every
    // wave holds all four frames: the same mask in all of them)
    unsigned exact_mask = 0, silent_mask = 0;
    if (!VAD_NO_EXACT && a.exact_net != nullptr) {
        const SilentMasks sm = silent_chunks(X0[0], X1[0], X2[0], X3[0]);
        const long left = (long)a.B - ln.st * 16;
        const unsigned rows = left >= 16 ? 0xffffu : ((1u << left) - 1u);
        exact_mask = sm.edge & rows;
        silent_mask = a.gx_silent != nullptr ? sm.silent & rows : 0u;
    }
    // the F(4,3) input transform reads the frames through E = x3 - x1 and F = x2 - x0 (kept in place of x3 and x0)
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        X3[k] = X3[k] - X1[k];
        X0[k] = X2[k] - X0[k];
    }
    float poison;
    {   float p0 = 0.f, p1 = 0.f;                         // non-finite input: fft_wave.hpp poison_acc (every wave holds all four frames)
        poison_acc<Q>(p0, p1, X3);
        poison_acc<Q>(p0, p1, X0);
        poison = poison_nyq(p0, p1, xn0, xn1, xn2, xn3);
    }

    // ---- encoder 0, this wave's 32 rows: one F(4,3) tile (kernel_front_f43.hip has the algebra) -------------------------------
    {
        const int row0 = 32 * w;
        const float *wn = tab + tb.w_nyq + row0;                 // [tap][row]
        f32x4 Y0[2], Y1[2], Y2[2], Y3[2];
        zero<2>(Y0);                                             // (the bias comes last: front_common.hpp add_bias)
        zero<2>(Y1);
        zero<2>(Y2);
        zero<2>(Y3);
        auto seg0 = [&](auto jc, f32x4 (&Y)[2], auto bfun) VAD_INLINE {
            run_segment<IC(jc) * NB0, NB0, 2>(pp, [&](auto i) VAD_INLINE -> f32x4 & { return Y[IC(i) & 1]; },
                                              [&](auto kg) VAD_INLINE {
                                                  return f32x4{bfun(4 * IC(kg)), bfun(4 * IC(kg) + 1), bfun(4 * IC(kg) + 2), bfun(4 * IC(kg) + 3)}; },
                                              gload);
        };
        {   const Coef k = opaque_coef<kF4, kFm3>();
            seg0(std::integral_constant<int, 0>{}, Y0, [&](int s) VAD_INLINE {
                return fmaf(fmaf(X2[s], k.p1, X1[s]), k.b, fmaf(X0[s], k.a, X3[s])); });                  // (E + 4F) - 3(x1 + x2)
            const Coef k2 = opaque_coef<kFm4, kF3>();
            seg0(std::integral_constant<int, 1>{}, Y1, [&](int s) VAD_INLINE {
                return fmaf(fmaf(X1[s], k2.m1, X2[s]), k2.b, fmaf(X0[s], k2.a, X3[s])); });               // (E - 4F) + 3(x2 - x1)
        }
        {   const Coef k = opaque_coef<kF2, kFm2>();
            seg0(std::integral_constant<int, 2>{}, Y2, [&](int s) VAD_INLINE { return fmaf(X0[s], k.a, X3[s]); });   // E + 2F
            seg0(std::integral_constant<int, 3>{}, Y3, [&](int s) VAD_INLINE { return fmaf(X0[s], k.b, X3[s]); });   // E - 2F
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sm = Y0[m][r] + Y1[m][r], df = Y0[m][r] - Y1[m][r];
                const float s2 = Y2[m][r] + Y3[m][r], d2 = Y2[m][r] - Y3[m][r];
                Y0[m][r] = sm + s2;
                Y1[m][r] = fmaf(2.f, d2, df);
                Y2[m][r] = fmaf(4.f, s2, sm);
                Y3[m][r] = fmaf(8.f, d2, df);
            }
        {   const Coef k = opaque_coef<kFm4, kFm025>();
            seg0(std::integral_constant<int, 4>{}, Y0, [&](int s) VAD_INLINE { return fmaf(X1[s], k.a, X3[s]); });   // x3 - 5 x1 = E - 4 x1
            seg0(std::integral_constant<int, 5>{}, Y3, [&](int s) VAD_INLINE { return fmaf(X2[s], k.b, -X0[s]); });  // x0 - 1.25 x2 = -F - x2/4
        }
        nyq_update<2>(Y0, xn0, wn + 128, ln);
        nyq_update<2>(Y0, xn1, wn + 256, ln);
        nyq_update<2>(Y1, xn0, wn, ln);
        nyq_update<2>(Y1, xn1, wn + 128, ln);
        nyq_update<2>(Y1, xn2, wn + 256, ln);
        nyq_update<2>(Y2, xn1, wn, ln);
        nyq_update<2>(Y2, xn2, wn + 128, ln);
        nyq_update<2>(Y2, xn3, wn + 256, ln);
        nyq_update<2>(Y3, xn2, wn, ln);
        nyq_update<2>(Y3, xn3, wn + 128, ln);
        add_bias<2>(Y0, tab + tb.b_e0 + row0, ln);
        add_bias<2>(Y1, tab + tb.b_e0 + row0, ln);
        add_bias<2>(Y2, tab + tb.b_e0 + row0, ln);
        add_bias<2>(Y3, tab + tb.b_e0 + row0, ln);
        relu<2>(Y0);
        relu<2>(Y1);
        relu<2>(Y2);
        relu<2>(Y3);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            *reinterpret_cast<f32x4 *>(&ybuf[0][2 * w + m][ln.lane * 4]) = Y0[m];
            *reinterpret_cast<f32x4 *>(&ybuf[1][2 * w + m][ln.lane * 4]) = Y1[m];
            *reinterpret_cast<f32x4 *>(&ybuf[2][2 * w + m][ln.lane * 4]) = Y2[m];
            *reinterpret_cast<f32x4 *>(&ybuf[3][2 * w + m][ln.lane * 4]) = Y3[m];
        }
    }
    lds_barrier();

    // ---- encoder 1: output row block w of both outputs, over all 128 input channels in the one-wave program's order ----------
    // (all of encoder 0's output first: 32 registers' worth of LDS reads in one go instead of one exposed LDS latency per block --
    //  the 132 registers of the magnitudes are free by now)
    f32x4 Yall[4][8];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int m = 0; m < 8; ++m) Yall[f][m] = *reinterpret_cast<const f32x4 *>(&ybuf[f][m][ln.lane * 4]);
    f32x4 Z[2];
    Z[0] = f32x4{0.f, 0.f, 0.f, 0.f};                            // (the bias comes last: front_common.hpp add_bias)
    Z[1] = Z[0];
    run_segment<S_E1, 40, 1>(pp, [&](auto i) VAD_INLINE -> f32x4 & { constexpr E1Blk eb = e1_blk(Q, IC(i)); return Z[eb.acc]; },
                             [&](auto i) VAD_INLINE { constexpr E1Blk eb = e1_blk(Q, IC(i)); return Yall[eb.frame][eb.rbg]; },
                             gload);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        Z[o] += *reinterpret_cast<const f32x4 *>(tab + tb.b_e1 + 16 * w + 4 * ln.g);
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[o][r] = fmaxf(Z[o][r], 0.f);
        *reinterpret_cast<f32x4 *>(&zbuf[o][w][ln.lane * 4]) = Z[o];
    }
    lds_barrier();

    // ---- encoder 2 (T 2 -> 1, stride 2: taps 1, 2 see encoder-1 outputs 0, 1): output row block w --------------------------------
    f32x4 Zall[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int m = 0; m < 4; ++m) Zall[o][m] = *reinterpret_cast<const f32x4 *>(&zbuf[o][m][ln.lane * 4]);
    f32x4 V1[1];
    V1[0] = *reinterpret_cast<const f32x4 *>(tab + tb.b_e2 + 16 * w + 4 * ln.g);
    run_segment<S_E2, 8, 1>(pp, [&](auto) VAD_INLINE -> f32x4 & { return V1[0]; },
                            [&](auto i) VAD_INLINE { return Zall[IC(i) / 4][IC(i) % 4]; }, gload);
#pragma unroll
    for (int r = 0; r < 4; ++r) V1[0][r] = fmaxf(V1[0][r], 0.f);
    *reinterpret_cast<f32x4 *>(&vbuf[w][ln.lane * 4]) = V1[0];
    lds_barrier();

    // ---- encoder 3 (T = 1: centre tap only): output row blocks 2w, 2w + 1 ---------------------------------------------------------------
    f32x4 Vall[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) Vall[m] = *reinterpret_cast<const f32x4 *>(&vbuf[m][ln.lane * 4]);
    f32x4 Fw[2];
    Fw[0] = *reinterpret_cast<const f32x4 *>(tab + tb.b_e3 + 32 * w + 4 * ln.g);
    Fw[1] = *reinterpret_cast<const f32x4 *>(tab + tb.b_e3 + 32 * w + 16 + 4 * ln.g);
    run_segment<S_E3, 8, 2>(pp, [&](auto i) VAD_INLINE -> f32x4 & { return Fw[IC(i) & 1]; },
                            [&](auto kg) VAD_INLINE { return Vall[IC(kg)]; }, gload);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Fw[m][r] = fmaxf(Fw[m][r], 0.f);
        *reinterpret_cast<f32x4 *>(&febuf[2 * w + m][ln.lane * 4]) = Fw[m];
    }
    lds_barrier();

    // ---- LSTM input-gate pre-activations of gate w (8 row blocks), stored in D-fragment order ------------------------------------------
    f32x4 Fe[8], G[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) Fe[m] = *reinterpret_cast<const f32x4 *>(&febuf[m][ln.lane * 4]);
    poison_into(Fe[0], poison);
    init_bias<8>(G, tab + tb.b_g + 128 * w, ln);
    run_segment<S_IH, 64, 8>(pp, [&](auto i) VAD_INLINE -> f32x4 & { return G[IC(i) & 7]; },
                             [&](auto kg) VAD_INLINE { return Fe[IC(kg)]; }, gload);
    if ((silent_mask >> ln.j) & 1) {                               // a chunk of zeros behind a silent context: the net's constant
        const float *gs = a.gx_silent + 128 * w + 4 * ln.g;
#pragma unroll
        for (int m = 0; m < 8; ++m) G[m] = *reinterpret_cast<const f32x4 *>(gs + 16 * m);
    }
    if (exact_mask != 0) {
        // workgroup-uniform and rare: the tile's EXACT chunks one after the other, all 256 threads on each (exact_front.hpp); the lanes of
        // chunk j then take gate w's 128 rows from the workspace in place of the chain's.  (The workspace lies over the exchange buffers:
        // every wave is past its last read of them.)
        ExactWs<Q> &ws = *reinterpret_cast<ExactWs<Q> *>(xy);
        static_assert(sizeof(ExactWs<Q>) <= sizeof(xy), "the exact workspace aliases the magnitude exchange buffer");
        __shared__ RefNet net;
        __syncthreads();
        if (threadIdx.x == 0) net = *a.exact_net;
        __syncthreads();
#pragma clang loop unroll(disable)
        for (int j = 0; j < 16; ++j) {
            if (!((exact_mask >> j) & 1)) continue;
            exact_gx<Q, PcmT, DEC>(a, net, ln.st * 16 + j, ln.t, ws);
            if (ln.j == j) {
#pragma unroll
                for (int m = 0; m < 8; ++m) G[m] = *reinterpret_cast<const f32x4 *>(&ws.gx[128 * w + 16 * m + 4 * ln.g]);
            }
            __syncthreads();
        }
    }
    if constexpr (!CELL) {
        float *gxt = a.gx + ((size_t)(ln.st * a.nt + ln.tl) * 32 + 8 * w) * 256 + ln.lane * 4;
#pragma unroll
        for (int m = 0; m < 8; ++m) *reinterpret_cast<f32x4 *>(gxt + (size_t)m * 256) = G[m];
    } else {
        // ---- the LSTM cell and the head, fused (one step: a.nt == 1).  Same sums in the same order as kernel_rec.hip: the gate
        // accumulators continue from W_ih x + b into W_hh h_{t-1}, k-groups ascending; same pointwise formulas; the head's 8 partial
        // sums (one per 16-unit row block) are added in the same order.  (reference: aten::lstm_cell, JIT!/torch/nn/modules/rnn.py:69)
        __shared__ __attribute__((aligned(16))) float gbuf[4][8][256];          // gates: [gate][row block][lane][4]
        __shared__ float pb[8 * 16];                                              // head partial sums: [row block][stream]
        const bool valid = bb < a.B && (cell.present == nullptr || cell.present[ln.b] != 0);   // (an absent row keeps its state)
        run_segment<S_HH, 64, 8>(pp, [&](auto i) VAD_INLINE -> f32x4 & { return G[IC(i) & 7]; },
                                 [&](auto kg) VAD_INLINE { return Hp[IC(kg)]; }, gload);
#pragma unroll
        for (int m = 0; m < 8; ++m) *reinterpret_cast<f32x4 *>(&gbuf[w][m][ln.lane * 4]) = G[m];
        lds_barrier();
        // wave w finishes the 32 hidden units of row blocks 2w, 2w + 1
        const float bo = tab[tb.b_out];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int rb = 2 * w + m;
            const f32x4 gi = *reinterpret_cast<const f32x4 *>(&gbuf[0][rb][ln.lane * 4]);
            const f32x4 gf = *reinterpret_cast<const f32x4 *>(&gbuf[1][rb][ln.lane * 4]);
            const f32x4 gg = *reinterpret_cast<const f32x4 *>(&gbuf[2][rb][ln.lane * 4]);
            const f32x4 go = *reinterpret_cast<const f32x4 *>(&gbuf[3][rb][ln.lane * 4]);
            const size_t soff = (size_t)ln.b * 128 + 16 * rb + 4 * ln.g;
            f32x4 c = *reinterpret_cast<const f32x4 *>(cell.state + (size_t)a.B * 128 + soff);
            const f32x4 wo = 0.5f * *reinterpret_cast<const f32x4 *>(tab + tb.w_out + 16 * rb + 4 * ln.g);   // halved: relu2_f (activations.hpp)
            f32x4 h;
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = sigmoid_f(gi[r]), fg = sigmoid_f(gf[r]), gt = tanh_f(gg[r]);
                const float cn = fmaf(fg, c[r], ig * gt);
                c[r] = cn;
                h[r] = sigmoid_f(go[r]) * tanh_f(cn);
                part = fmaf(wo[r], relu2_f(h[r]), part);
            }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            if (ln.g == 0) pb[rb * 16 + ln.j] = part;
            if (valid) {
                *reinterpret_cast<f32x4 *>(cell.state + soff) = h;
                *reinterpret_cast<f32x4 *>(cell.state + (size_t)a.B * 128 + soff) = c;
            }
        }
        lds_barrier();
        if (w == 0 && ln.g == 0 && valid) {
            float p = bo;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) p += pb[ww * 16 + ln.j];
            cell.probs[(size_t)bb * cell.ldp + a.t0] = sigmoid_f(p);
        }
    }
}

}  // namespace

template <typename PcmT, bool CELL>
static hipError_t launch_lat(int sr, const FrontArgs &a, const CellArgs &c, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    if (total > 0x7fffffffL || (CELL && a.nt != 1)) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)total;
    if (a.dec > 1 && (sr != 16000 || a.dec > 3)) return hipErrorInvalidValue;
    if (a.dec == 3) hipLaunchKernelGGL((front_lat_kernel<32, PcmT, 3, CELL>), dim3(grid), dim3(256), 0, s, a, c);
    else if (a.dec == 2) hipLaunchKernelGGL((front_lat_kernel<32, PcmT, 2, CELL>), dim3(grid), dim3(256), 0, s, a, c);
    else if (sr == 16000) hipLaunchKernelGGL((front_lat_kernel<32, PcmT, 1, CELL>), dim3(grid), dim3(256), 0, s, a, c);
    else hipLaunchKernelGGL((front_lat_kernel<16, PcmT, 1, CELL>), dim3(grid), dim3(256), 0, s, a, c);
    return hipGetLastError();
}
template <typename PcmT>
hipError_t launch_front_lat(int sr, const FrontArgs &a, hipStream_t s) {
    return launch_lat<PcmT, false>(sr, a, CellArgs{}, s);
}
template <typename PcmT>
hipError_t launch_step_lat(int sr, const FrontArgs &a, const CellArgs &c, hipStream_t s) {
    return launch_lat<PcmT, true>(sr, a, c, s);
}
template hipError_t launch_front_lat<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_front_lat<int16_t>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_step_lat<float>(int, const FrontArgs &, const CellArgs &, hipStream_t);
template hipError_t launch_step_lat<int16_t>(int, const FrontArgs &, const CellArgs &, hipStream_t);

}  // namespace vad
