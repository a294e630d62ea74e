// kernel_step_one.hip -- ONE step of a handful of streams, one workgroup per STREAM: the step every unmodified caller of the
// reference issues, `model(chunk, sr).item()` with B = 1 once per 32 ms (src/silero_vad/utils_vad.py:324-336 get_speech_timestamps,
// :528 VADIterator.__call__; native twin examples/cpp/silero-vad-onnx.cpp:103-142).
//
// Why.  The latency frontend (kernel_front_lat.hip) gives a 16-stream tile to a 4-wave workgroup: its MFMAs compute 16 columns, and
// a B = 1 call uses one of them -- 1 120 MFMAs x 32 cycles per wave (15 us of matrix pipe) for 1/16 of their result, 34 us per call.
// Here the same sums are formed for ONE column on the VALU: every output row of every layer is one sequential fmaf chain over its
// inputs IN THE ORDER the MFMA program adds them -- k-groups in program order, inside a k-group the four k-steps, inside a k-step
// the instruction's k = lane group 0..3 (v_mfma_f32_16x16x4_f32 is exactly that chain: profiles/r03v_mfma_order.md) -- read from the
// SAME packed weight images (the A fragment of lane (g, i) at k-step ks IS W[row i][k = (kg, ks, g)]), with the same in-wave FFT, the
// same transforms, bias placement, ReLUs, activations and head summation.  Identical bits to the tile kernels
// (tests/test_gpu_parity.py::test_one_stream_step_is_bit_identical), as kernel_rec_small.hip is to kernel_rec.hip.
//
// Shape.  256 threads.  STFT: wave v transforms frame v with fft_wave.hpp's code (its 16 columns all carry this stream; lane group g of
// column 0 delivers the magnitudes of k = (s, g)).  Then layer by layer through LDS, activations stored in CHAIN order (channel
// 16 kg + 4 g + ks at position 16 kg + 4 ks + g), one thread per output row (two chains per thread where a layer has 512 rows or two
// independent accumulators).  A chain's weights are 16-byte vectors of the fragment image, all of a layer's vectors requested before
// its first fmaf.  Bound by streaming 1.1 MB of weights from L2 through one CU (~7 us) plus the FFT, not by the matrix pipe.
#include <hip/hip_runtime.h>

#include "activations.hpp"
#include "exact_front.hpp"
#include "front_common.hpp"

namespace vad {
namespace {

// A chain's weights requested AHEAD of their use: the 16-byte vectors of NKG k-groups of one row of the fragment image.  `w` points at
// the row's 16 bytes of block (kg = 0, lane group 0): lane group g is 16 lanes = 64 floats further, k-group kg is `kg_stride` floats
// further.  load() only issues the requests; run() is the chain: acc += W[row][(kg, ks, g)] * x[16 kg + 4 ks + g], k-groups in order,
// inside a k-group the four k-steps, inside a k-step the lane groups -- the order in which the MFMA program adds them.
template <int NKG>
struct WSet {
    f32x4 a[NKG][4];
    __device__ __forceinline__ void load(const float *w, long kg_stride) {
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg)
#pragma unroll
            for (int g = 0; g < 4; ++g) a[kg][g] = *reinterpret_cast<const f32x4 *>(w + kg * kg_stride + g * 64);
    }
    __device__ __forceinline__ float run(float acc, const float *x) const {
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc = fmaf(a[kg][g][ks], x[16 * kg + 4 * ks + g], acc);
        return acc;
    }
};
// two chains at once (independent accumulators), each over its own input
template <int NKG>
__device__ __forceinline__ void run2(const WSet<NKG> &A, const WSet<NKG> &B, float &acc0, float &acc1, const float *x0, const float *x1) {
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc0 = fmaf(A.a[kg][g][ks], x0[16 * kg + 4 * ks + g], acc0);
                acc1 = fmaf(B.a[kg][g][ks], x1[16 * kg + 4 * ks + g], acc1);
            }
}
#define VAD_PIN() __builtin_amdgcn_sched_barrier(0)
// A workgroup barrier for data exchanged through LDS that does NOT wait for the global loads in flight (__syncthreads would drain
// the weight requests that were issued ahead on purpose)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// the same between the lanes of ONE wave (a wave's LDS operations execute in order)
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// bring-up (option trace_ptr, tools/b1_phase_trace.py): thread 0 of workgroup 0 leaves the shader clock at the phase boundaries
#define VAD_STAMP(k) do { if (a.trace != nullptr && threadIdx.x == 0 && blockIdx.x == 0) a.trace[(k)] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ int chain_pos(int ch) { return (ch & ~15) + 4 * (ch & 3) + ((ch >> 2) & 3); }   // channel -> position

// encoder 1's 40-block program (front_common.hpp e1_blk) split by accumulator: the rank of block idx among the blocks of its own
// accumulator (out 0: 16 blocks, out 1: 24)
constexpr int e1_rank(int Q, int idx) {
    int r = 0;
    for (int i = 0; i < idx; ++i) r += e1_blk(Q, i).acc == e1_blk(Q, idx).acc ? 1 : 0;
    return r;
}
// The chains are cut in two and wave v forms half (v >> 1) of the chains of output (v & 1): per wave the blocks it owns, in chain order,
// as offsets (floats) of the block's k-group in the weight image and of its 16 inputs in e0c[frame][position].
struct E1Seg {
    int n, woff[12], xoff[12];
};
struct E1Tab {
    E1Seg s[4];
};
constexpr E1Tab make_e1_tab(int Q) {
    E1Tab t{};
    for (int wv = 0; wv < 4; ++wv) {
        const int O = wv & 1, P = wv >> 1, n = O ? 24 : 16, lo = P ? n / 2 : 0, hi = P ? n : n / 2;
        t.s[wv].n = hi - lo;
        for (int idx = 0; idx < 40; ++idx) {
            const E1Blk eb = e1_blk(Q, idx);
            const int rk = e1_rank(Q, idx);
            if (eb.acc != O || rk < lo || rk >= hi) continue;
            t.s[wv].woff[rk - lo] = (eb.unit * 16 + eb.kg * 4) * 256;
            t.s[wv].xoff[rk - lo] = eb.frame * 128 + 16 * eb.rbg;
        }
    }
    return t;
}
__device__ constexpr E1Tab kE1Tab32 = make_e1_tab(32), kE1Tab16 = make_e1_tab(16);

// one radix-2 DIF stage of span N over the lane's Q values (fft_wave.hpp fft_inlane, one iteration of its outer loop)
template <int Q, int N>
__device__ __forceinline__ void fft_stage(f32x2 (&z)[Q]) {
    constexpr int half = N >> 1;
#pragma unroll
    for (int b0 = 0; b0 < Q; b0 += N) {
#pragma unroll
        for (int jx = 0; jx < half; ++jx) {
            const int i0 = b0 + jx, i1 = i0 + half;
            const f32x2 u = z[i0], v = z[i1];
            z[i0] = u + v;
            const f32x2 d = u - v;
            const int tw = jx * (32 / N);
            if (tw == 0) z[i1] = d;
            else if (tw == 8) z[i1] = swap2(d) * f32x2{1.0f, -1.0f};
            else z[i1] = cmul(d, kCos32[tw], -kSin32[tw]);
        }
    }
}

template <int Q, typename PcmT, bool CELL>
__global__ void __launch_bounds__(256) step_one_kernel(const FrontArgs a, const CellArgs cell) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int TABF = (tb.total + 3) / 4 * 4;
    constexpr int RB = w_rb(Q), KG0 = Q / 4, T0 = w4_tail0(Q), NQ = 4 * Q;     // NQ: k values of encoder 0 without the Nyquist bin
    constexpr int SL = 2 * Q, H = Q / 2, LG = ilog2(Q);
    __shared__ __attribute__((aligned(16))) float tab[TABF];
    __shared__ __attribute__((aligned(16))) float mag[4][NQ];       // [frame][(s, g)]: |Y| of k = (s, g), chain order
    __shared__ float nyq[4];
    __shared__ __attribute__((aligned(16))) float tin[6][NQ];       // encoder 0's six transformed inputs (U1, U2, U3, U4, U0, U5 order)
    __shared__ __attribute__((aligned(16))) float my[4][128];       // m1..m4 of every row
    __shared__ __attribute__((aligned(16))) float e0c[4][128];      // encoder 0 output per frame, chain order
    __shared__ __attribute__((aligned(16))) float e1p[2][64];       // encoder 1: the first half of each chain
    __shared__ __attribute__((aligned(16))) float e1c[2][64];
    __shared__ __attribute__((aligned(16))) float e2c[64];
    __shared__ __attribute__((aligned(16))) float fec[128];         // encoder 3 output ("feat"), chain order, poison applied
    __shared__ __attribute__((aligned(16))) float hc[128];          // h_{t-1}, chain order
    __shared__ __attribute__((aligned(16))) float gates[512];
    __shared__ __attribute__((aligned(16))) float hnew[128];
    __shared__ float pb[8], pg[32];
    __shared__ float poison_g[4];
    __shared__ __attribute__((aligned(16))) ExactWs<Q> ws;          // (its first 8 KB double as the STFT's exchange buffers)
    __shared__ RefNet net;
    static_assert(sizeof(ExactWs<Q>) >= 4 * 4 * SL * sizeof(float) + 4 * 4 * Q * sizeof(f32x2), "STFT exchange buffers alias the exact workspace");

    const int tid = threadIdx.x;
    const long b = blockIdx.x;                                      // this workgroup's stream
    Lane ln;
    ln.lane = tid & 63;
    ln.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ln.g = ln.lane >> 4;
    ln.j = ln.lane & 15;
    ln.tile_valid = true;
    ln.tl = 0;
    ln.st = b >> 4;
    ln.t = a.t0;
    ln.b = (int)b;                                                  // every column of the wave carries this stream
    ln.from_tail = a.tail != nullptr && ln.t == a.T - 1;
    ln.sgnA = ln.g < 2 ? 1.f : -1.f;
    ln.sgnB = (ln.g & 1) ? -1.f : 1.f;
    const int w = ln.wave;

    VAD_STAMP(0);
    // this thread's rows of encoder 0 (row r; the two matrices of its half first, requested in eight batches between the STFT's steps:
    // 256 KB pass the CU's one vector-memory path in ~4 000 cycles, and a wave that asks for all of it at once sits in the queue instead
    // of computing)
    const int r = tid & 127, half = tid >> 7, i16 = r & 15;
    const int part = r / (16 * RB), rbl = (r % (16 * RB)) / 16;
    const int u0 = part == 0 ? w4_part0(0, Q) : part == 1 ? w4_part0(1, Q) : part == 2 ? w4_part0(2, Q) : w4_part0(3, Q);
    const float *wb = a.wfront + (size_t)u0 * 4096 + rbl * 256 + i16 * 4;
    WSet<KG0> W0a, W0b, W0c;
    auto ask0 = [&](auto kc) VAD_INLINE {
        constexpr int PER = 2 * KG0 * 4 / 8, k = decltype(kc)::value;
        static_for<0, PER>([&](auto vc) VAD_INLINE {
            constexpr int v = k * PER + decltype(vc)::value, set = v / (KG0 * 4), kg = (v % (KG0 * 4)) / 4, g = v % 4;
            const f32x4 x = *reinterpret_cast<const f32x4 *>(wb + (size_t)(2 * half + set) * 4096 + kg * (RB * 256) + g * 64);
            if constexpr (set == 0) W0a.a[kg][g] = x;
            else W0b.a[kg][g] = x;
        });
        VAD_PIN();
    };
    // ---- STFT: wave v, frame v.  fft_wave.hpp's arithmetic, value for value, but spread over the wave: there the 16 columns of a wave
    // are 16 streams and the four lane groups of a column share one frame; here all 16 columns carry THIS stream, so column j forms only
    // the cross-lane butterflies of the pair (q, q + Q/2), q = j, and only the bins k = j, j + 16 of the real-FFT split -- the lanes
    // exchange through LDS (the wave's own 2 KB), every lane runs the in-lane Q-point FFT.
    float *sbuf = reinterpret_cast<float *>(&ws) + (w * 4 + ln.g) * SL;                                     // [wave][g][2Q] samples
    f32x2 *zbuf = reinterpret_cast<f32x2 *>(reinterpret_cast<float *>(&ws) + 4 * 4 * SL) + w * 4 * Q;       // [wave][g][Q]
    // The context for the next call is slice 8 (frame 3, lane group 2).  With a second buffer load_slice stores it as the tile kernels do;
    // IN PLACE (ctx_out == ctx_in: what vad_step hands over -- no second buffer, no 256-byte copy operation behind the kernel) it is
    // taken from the exchange buffer once every wave has its samples and stored behind the frontend, for a stream that has a chunk this tick.
    const bool ctx_inplace = a.ctx_out != nullptr && a.ctx_out == a.ctx_in;
    {
        float pcm_s[SL];
        FrontArgs al = a;
        if (ctx_inplace) al.ctx_out = nullptr;
        load_slice<Q, PcmT, 1>(pcm_s, al, ln, w);
        VAD_PIN();
        {   // tables -> LDS
            constexpr int NV = tb.total / 4;
            const f32x4 *src = reinterpret_cast<const f32x4 *>(a.tables);
            for (int i = tid; i < NV; i += 256) reinterpret_cast<f32x4 *>(tab)[i] = src[i];
        }
        if (CELL && tid < 128) hc[chain_pos(tid)] = cell.state[(size_t)b * 128 + tid];
        if (ln.j == 0) {
#pragma unroll
            for (int k = 0; k < SL / 4; ++k)
                reinterpret_cast<f32x4 *>(sbuf)[k] = f32x4{pcm_s[4 * k], pcm_s[4 * k + 1], pcm_s[4 * k + 2], pcm_s[4 * k + 3]};
        }
    }
    lds_barrier();
    VAD_STAMP(1);
    f32x4 ctx_new{};                                               // (kept in registers: the double-precision route reads the OLD context)
    if (ctx_inplace && tid < SL / 4) ctx_new = reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(&ws) + (3 * 4 + 2) * SL)[tid];
    {
        const int q = ln.j & (H - 1);
        const float *win = tab + tb.window + SL * ln.g;
        f32x2 zl = *reinterpret_cast<const f32x2 *>(sbuf + 2 * q) * *reinterpret_cast<const f32x2 *>(win + 2 * q);
        f32x2 zh = *reinterpret_cast<const f32x2 *>(sbuf + 2 * (q + H)) * *reinterpret_cast<const f32x2 *>(win + 2 * (q + H));
        const f32x4 *tw1 = reinterpret_cast<const f32x4 *>(tab + tb.tw1 + ln.g * Q * 4);
        const f32x4 tl = tw1[q], th = tw1[q + H];
        ask0(std::integral_constant<int, 0>{});
        const f32x2 rotA = (ln.g & 1) ? f32x2{0.f, 0.f} : f32x2{1.f, 1.f};
        const f32x2 rotB = (ln.g & 1) ? f32x2{1.f, -1.f} : f32x2{0.f, 0.f};
        trade32(zl, zh);
        {
            const f32x2 u = zl, v = zh, d = u - v;
            zl = u + v;
            zh = __builtin_elementwise_fma(swap2(d), rotB, d * rotA);
        }
        trade32(zl, zh);
        trade16(zl, zh);
        {
            const f32x2 u = zl, v = zh;
            zl = u + v;
            zh = u - v;
        }
        trade16(zl, zh);
        zl = __builtin_elementwise_fma(swap2(zl), f32x2{tl[0], tl[1]}, zl * f32x2{tl[2], tl[2]});
        zh = __builtin_elementwise_fma(swap2(zh), f32x2{th[0], th[1]}, zh * f32x2{th[2], th[2]});
        zbuf[ln.g * Q + q] = zl;
        zbuf[ln.g * Q + q + H] = zh;
        ask0(std::integral_constant<int, 1>{});
        wave_lds_sync();
        f32x2 z[Q];
#pragma unroll
        for (int k = 0; k < Q / 2; ++k) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(zbuf + ln.g * Q)[k];
            z[2 * k] = f32x2{v[0], v[1]};
            z[2 * k + 1] = f32x2{v[2], v[3]};
        }
        ask0(std::integral_constant<int, 2>{});
        if constexpr (Q == 32) {
            fft_stage<Q, 32>(z);
            ask0(std::integral_constant<int, 3>{});
        } else {
            ask0(std::integral_constant<int, 3>{});
        }
        fft_stage<Q, 16>(z);
        ask0(std::integral_constant<int, 4>{});
        fft_stage<Q, 8>(z);
        ask0(std::integral_constant<int, 5>{});
        fft_stage<Q, 4>(z);
        ask0(std::integral_constant<int, 6>{});
        fft_stage<Q, 2>(z);
        ask0(std::integral_constant<int, 7>{});
        wave_lds_sync();                                            // (every lane has read the wave's z)
        if (ln.j == 0) {
#pragma unroll
            for (int k = 0; k < Q / 2; ++k)
                reinterpret_cast<f32x4 *>(zbuf + ln.g * Q)[k] = f32x4{z[2 * k].x, z[2 * k].y, z[2 * k + 1].x, z[2 * k + 1].y};
        }
        if (ln.lane == 0) nyq[w] = fabsf(z[0].x - z[0].y);          // Nyquist: Re Z0 - Im Z0 of lane group 0
        wave_lds_sync();
        // real-FFT split of bin k' = j (+ 16): Y = E + W O from Z[k'] and the conjugate of its partner bin, which lane group 0 holds itself
        // and the others find in lane group g (g = 1) or g ^ 1 (g = 2, 3)
        const f32x4 *tw2 = reinterpret_cast<const f32x4 *>(tab + tb.tw2 + ln.g * Q * 4);
        const int srcg = ln.g >= 2 ? (ln.g ^ 1) : ln.g;
#pragma unroll
        for (int i = 0; i < (Q + 15) / 16; ++i) {
            const int k = (ln.j + 16 * i) & (Q - 1);
            const int bk = (int)(__builtin_bitreverse32((unsigned)k) >> (32 - LG));
            const int ks = (int)(__builtin_bitreverse32((unsigned)((Q - k) & (Q - 1))) >> (32 - LG));
            const int kr = (int)(__builtin_bitreverse32((unsigned)(Q - 1 - k)) >> (32 - LG));
            const f32x2 u = zbuf[ln.g * Q + bk];
            const f32x2 p = zbuf[ln.g == 0 ? ks : srcg * Q + kr];
            const f32x2 av = __builtin_elementwise_fma(p, f32x2{1.0f, -1.0f}, u);     // Z + conj(Zp)
            const f32x2 ev = __builtin_elementwise_fma(p, f32x2{-1.0f, 1.0f}, u);     // Z - conj(Zp)
            const f32x4 t = tw2[k];                                                  // (c, -c, s, 0)
            const f32x2 y = __builtin_elementwise_fma(swap2(ev), f32x2{t[0], t[1]}, __builtin_elementwise_fma(ev, f32x2{t[2], t[2]}, av));
            const f32x2 yy = y * y;
            mag[w][4 * k + ln.g] = 0.5f * __builtin_amdgcn_sqrtf(yy.x + yy.y);
        }
    }
    lds_barrier();
    VAD_STAMP(2);
    // exact_front.hpp: silent frames beside frames that are not (every thread reads the four frames' bins 0..3 itself)
    int how = 0;
    if (!VAD_NO_EXACT && a.exact_net != nullptr) {
        bool any = false, all = true;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const f32x4 m = *reinterpret_cast<const f32x4 *>(&mag[f][0]);
            const bool z = m[0] == 0.f && m[1] == 0.f && m[2] == 0.f && m[3] == 0.f;
            any = any || z;
            all = all && z;
        }
        how = all ? (a.gx_silent != nullptr ? 2 : 0) : any ? 1 : 0;
    }
    how = __builtin_amdgcn_readfirstlane(how);
    WSet<8> Wh0, Wh1;                                               // W_hh rows tid, tid + 256
    const float *wh0 = nullptr, *wh1 = nullptr;
    if constexpr (CELL) {
        const int q0 = tid >> 7, m = (tid >> 4) & 7, i = tid & 15;
        wh0 = cell.whh_lat + ((size_t)q0 * 64 + m) * 256 + i * 4;
        wh1 = cell.whh_lat + ((size_t)(q0 + 2) * 64 + m) * 256 + i * 4;
    }
    float g0, g1;                                                   // gate rows tid and tid + 256: bias + W_ih feat
    VAD_STAMP(3);
    if (how != 0) {
        if (how == 1) {
            if (tid == 0) net = *a.exact_net;
            __syncthreads();
            exact_gx<Q, PcmT, 1>(a, net, b, ln.t, ws);
            __syncthreads();
            g0 = ws.gx[tid];
            g1 = ws.gx[tid + 256];
        } else {
            g0 = a.gx_silent[tid];
            g1 = a.gx_silent[tid + 256];
        }
        if constexpr (CELL) {
            Wh0.load(wh0, 8 * 256);
            VAD_PIN();
        }
    } else {
        W0c.load(wb + (size_t)(4 + half) * 4096, RB * 256);         // the fifth / sixth matrix of this half
        VAD_PIN();
        // ---- encoder 0 as one F(4,3) tile: the six transformed inputs of every k (kernel_front_lat.hip has the algebra) -----------
        if (tid < NQ) {
            const float x0 = mag[0][tid], x1 = mag[1][tid], x2 = mag[2][tid], x3 = mag[3][tid];
            const float E = x3 - x1, F = x2 - x0;
            tin[0][tid] = fmaf(fmaf(x2, 1.0f, x1), -3.0f, fmaf(F, 4.0f, E));     // (E + 4F) - 3(x1 + x2)
            tin[1][tid] = fmaf(fmaf(x1, -1.0f, x2), 3.0f, fmaf(F, -4.0f, E));    // (E - 4F) + 3(x2 - x1)
            tin[2][tid] = fmaf(F, 2.0f, E);
            tin[3][tid] = fmaf(F, -2.0f, E);
            tin[4][tid] = fmaf(x1, -4.0f, E);
            tin[5][tid] = fmaf(x2, -0.25f, -F);
        }
        if (tid >= 192 && tid < 196) {                              // non-finite input (fft_wave.hpp): lane group g's own poison value
            const int g = tid - 192;
            float p0 = 0.f, p1 = 0.f;
            for (int s = 0; s < Q; s += 2) {                        // E first, then F, two chains -- as poison_acc does
                p0 = fmaf(mag[3][4 * s + g] - mag[1][4 * s + g], 0.f, p0);
                p1 = fmaf(mag[3][4 * (s + 1) + g] - mag[1][4 * (s + 1) + g], 0.f, p1);
            }
            for (int s = 0; s < Q; s += 2) {
                p0 = fmaf(mag[2][4 * s + g] - mag[0][4 * s + g], 0.f, p0);
                p1 = fmaf(mag[2][4 * (s + 1) + g] - mag[0][4 * (s + 1) + g], 0.f, p1);
            }
            poison_g[g] = poison_nyq(p0, p1, nyq[0], nyq[1], nyq[2], nyq[3]);
        }
        lds_barrier();
        VAD_STAMP(4);
        // From here to the gates the waves own different rows.  Encoder 1 (64 rows x two outputs, the 40-block program of front_common.hpp
        // e1_blk) has its chains cut in two -- wave v forms half (v >> 1) of the chains of output (v & 1), 8 or 12 blocks a thread (E1Seg),
        // the first halves handed over through LDS; encoder 2 (taps 1, 2 see encoder-1 outputs 0, 1) is wave 0's, encoder 3 (centre tap ->
        // feat, with the non-finite poison in the k = (0, 0, g) inputs) waves 0's and 2's (row 64 (v >> 1) + lane); then the gate rows
        // tid, tid + 256 of W_ih and of W_hh.  Every weight is asked for a phase or more ahead, in the order of use.
        const E1Seg &sg = (Q == 32 ? kE1Tab32 : kE1Tab16).s[w];
        const bool odd = (w & 1) != 0, second = w >= 2;
        const int e1wr = ln.lane >> 4, e1i = ln.lane & 15;
        const float *e1w = a.wfront + e1wr * 256 + e1i * 4;
        f32x4 wq[12][4];
        auto ask = [&](auto &W, const float *p) VAD_INLINE {
            VAD_PIN();
            W.load(p, 8 * 256);
            VAD_PIN();
        };
        {
            // m1, m2 (half 0) | m3, m4 (half 1): two independent chains per thread
            float ma = 0.f, mb = 0.f;
            run2<KG0>(W0a, W0b, ma, mb, tin[2 * half], tin[2 * half + 1]);
            my[2 * half][r] = ma;
            my[2 * half + 1][r] = mb;
        }
        VAD_PIN();
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
#pragma unroll
            for (int g = 0; g < 4; ++g) wq[sl][g] = *reinterpret_cast<const f32x4 *>(e1w + sg.woff[sl] + g * 64);
        if (odd) {
#pragma unroll
            for (int sl = 8; sl < 12; ++sl)
#pragma unroll
                for (int g = 0; g < 4; ++g) wq[sl][g] = *reinterpret_cast<const f32x4 *>(e1w + sg.woff[sl] + g * 64);
        }
        VAD_PIN();
        lds_barrier();
        VAD_STAMP(5);
        {
            const float m1 = my[0][r], m2 = my[1][r], m3 = my[2][r], m4 = my[3][r];
            const float sm = m1 + m2, df = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
            // half 0 finishes frames 0 and 1, half 1 frames 3 and 2
            float ya, yb;                                           // ya: the frame that takes a fifth / sixth product chain
            if (half == 0) {
                ya = W0c.run(sm + s2, tin[4]);                                           // y0 += U0 t0
                yb = fmaf(2.f, d2, df);                                                  // y1
            } else {
                ya = W0c.run(fmaf(8.f, d2, df), tin[5]);                                 // y3 += U5 t5
                yb = fmaf(4.f, s2, sm);                                                  // y2
            }
            const float *wn = tab + tb.w_nyq + r;                   // [tap][row]
            const float bias = tab[tb.b_e0 + r];
            if (half == 0) {
                ya = fmaf(wn[128], nyq[0], ya);
                ya = fmaf(wn[256], nyq[1], ya);
                yb = fmaf(wn[0], nyq[0], yb);
                yb = fmaf(wn[128], nyq[1], yb);
                yb = fmaf(wn[256], nyq[2], yb);
                e0c[0][chain_pos(r)] = fmaxf(ya + bias, 0.f);
                e0c[1][chain_pos(r)] = fmaxf(yb + bias, 0.f);
            } else {
                ya = fmaf(wn[0], nyq[2], ya);
                ya = fmaf(wn[128], nyq[3], ya);
                yb = fmaf(wn[0], nyq[1], yb);
                yb = fmaf(wn[128], nyq[2], yb);
                yb = fmaf(wn[256], nyq[3], yb);
                e0c[3][chain_pos(r)] = fmaxf(ya + bias, 0.f);
                e0c[2][chain_pos(r)] = fmaxf(yb + bias, 0.f);
            }
        }
        VAD_PIN();
        f32x4 w2[8][4];
        WSet<4> W3;
        const int row3 = 64 * (w >> 1) + ln.lane;
        if (w == 0) {
#pragma unroll
            for (int bi = 0; bi < 8; ++bi)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    w2[bi][g] = *reinterpret_cast<const f32x4 *>(e1w + ((size_t)(T0 + bi / 4) * 16 + (bi % 4) * 4) * 256 + g * 64);
        }
        if (!odd) W3.load(a.wfront + (size_t)(T0 + 2) * 4096 + (row3 >> 4) * 256 + (row3 & 15) * 4, 8 * 256);
        VAD_PIN();
        lds_barrier();
        VAD_STAMP(6);
        auto e1_run = [&](float z) VAD_INLINE -> float {
            const float *x0 = &e0c[0][0];
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const float *x = x0 + sg.xoff[sl];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int g = 0; g < 4; ++g) z = fmaf(wq[sl][g][ks], x[4 * ks + g], z);
            }
            if (odd) {
#pragma unroll
                for (int sl = 8; sl < 12; ++sl) {
                    const float *x = x0 + sg.xoff[sl];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int g = 0; g < 4; ++g) z = fmaf(wq[sl][g][ks], x[4 * ks + g], z);
                }
            }
            return z;
        };
        WSet<8> Wi0, Wi1;
        const float *wi0, *wi1;
        {
            const int q0 = tid >> 7, m = (tid >> 4) & 7, i = tid & 15;
            wi0 = a.wfront + (size_t)(T0 + 4 + 4 * q0) * 4096 + m * 256 + i * 4;
            wi1 = a.wfront + (size_t)(T0 + 4 + 4 * (q0 + 2)) * 4096 + m * 256 + i * 4;
        }
        if (!second) e1p[w & 1][ln.lane] = e1_run(0.f);
        lds_barrier();
        if (second) e1c[w & 1][chain_pos(ln.lane)] = fmaxf(e1_run(e1p[w & 1][ln.lane]) + tab[tb.b_e1 + ln.lane], 0.f);
        ask(Wi0, wi0);
        ask(Wi1, wi1);
        lds_barrier();
        if (w == 0) {
            float v = tab[tb.b_e2 + tid];
#pragma unroll
            for (int bi = 0; bi < 8; ++bi) {
                const float *x = e1c[bi / 4] + 16 * (bi % 4);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int g = 0; g < 4; ++g) v = fmaf(w2[bi][g][ks], x[4 * ks + g], v);
            }
            e2c[chain_pos(tid)] = fmaxf(v, 0.f);
        }
        lds_barrier();
        if (!odd) {
            float f = W3.run(tab[tb.b_e3 + row3], e2c);
            f = fmaxf(f, 0.f);
            const int pos = chain_pos(row3);
            if (pos < 4) f = __uint_as_float(__float_as_uint(f) | __float_as_uint(poison_g[pos]));     // positions 0..3 = (kg 0, ks 0, g)
            fec[pos] = f;
        }
        lds_barrier();
        VAD_STAMP(7);
        // ---- W_ih: gate rows tid and tid + 256.  The first row's W_hh is asked for BETWEEN the k-groups: a vector-memory request waits until
        // the CU's memory path has room for it, so a block of 32 requests stops the wave for as long as the path needs for them (~4 000
        // cycles); spread over the chain they are taken while the wave computes.
        g0 = tab[tb.b_g + tid];
        g1 = tab[tb.b_g + tid + 256];
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
            if constexpr (CELL) {
#pragma unroll
                for (int g = 0; g < 4; ++g) Wh0.a[kg][g] = *reinterpret_cast<const f32x4 *>(wh0 + kg * (8 * 256) + g * 64);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    g0 = fmaf(Wi0.a[kg][g][ks], fec[16 * kg + 4 * ks + g], g0);
                    g1 = fmaf(Wi1.a[kg][g][ks], fec[16 * kg + 4 * ks + g], g1);
                }
            VAD_PIN();
        }
    }
    if (ctx_inplace && tid < SL / 4) {
        bool here = true;
        if constexpr (CELL) here = cell.present == nullptr || cell.present[b] != 0;
        if (here) reinterpret_cast<f32x4 *>(a.ctx_out + (size_t)b * SL)[tid] = ctx_new;
    }
    if constexpr (!CELL) {
        // gx[tile][row block 32][lane 64][4] (layout.hpp): row 16 mb + 4 g + r of column j at lane 16 g + j, element r
        float *gxt = a.gx + (size_t)(b >> 4) * 32 * 256;
        const int j = (int)(b & 15);
        gxt[((size_t)(tid >> 4) * 64 + ((tid >> 2) & 3) * 16 + j) * 4 + (tid & 3)] = g0;
        gxt[((size_t)((tid + 256) >> 4) * 64 + ((tid >> 2) & 3) * 16 + j) * 4 + (tid & 3)] = g1;
    } else {
        // ---- the LSTM cell: the gate chains continue into W_hh h_{t-1}; pointwise; head (kernel_front_lat.hip's order) ---------------------
        VAD_PIN();
        VAD_STAMP(8);
        // (the second row's W_hh between the k-groups of the first row's chain, likewise)
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
#pragma unroll
            for (int g = 0; g < 4; ++g) Wh1.a[kg][g] = *reinterpret_cast<const f32x4 *>(wh1 + kg * (8 * 256) + g * 64);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int g = 0; g < 4; ++g) g0 = fmaf(Wh0.a[kg][g][ks], hc[16 * kg + 4 * ks + g], g0);
            VAD_PIN();
        }
        gates[tid] = g0;
        g1 = Wh1.run(g1, hc);
        gates[tid + 256] = g1;
        lds_barrier();
        VAD_STAMP(9);
        const bool present = cell.present == nullptr || cell.present[b] != 0;
        if (tid < 128) {
            const float c0 = cell.state[((size_t)a.B + b) * 128 + tid];
            const float ig = sigmoid_f(gates[tid]), fg = sigmoid_f(gates[128 + tid]), gt = tanh_f(gates[256 + tid]);
            const float cn = fmaf(fg, c0, ig * gt);
            const float h = sigmoid_f(gates[384 + tid]) * tanh_f(cn);
            hnew[tid] = h;
            if (present) {
                cell.state[(size_t)b * 128 + tid] = h;
                cell.state[((size_t)a.B + b) * 128 + tid] = cn;
            }
        }
        lds_barrier();
        if (tid < 32) {                                             // (row block rb, lane group g): units 16 rb + 4 g + r, r ascending
            const int rb = tid >> 2, g = tid & 3;
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) part = fmaf(0.5f * tab[tb.w_out + 16 * rb + 4 * g + r], relu2_f(hnew[16 * rb + 4 * g + r]), part);
            pg[tid] = part;
        }
        lds_barrier();
        if (tid < 8) pb[tid] = (pg[4 * tid] + pg[4 * tid + 1]) + (pg[4 * tid + 2] + pg[4 * tid + 3]);
        lds_barrier();
        if (tid == 0 && present) {
            float p = tab[tb.b_out];
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) p += pb[ww];
            cell.probs[(size_t)b * cell.ldp + a.t0] = sigmoid_f(p);
        }
        VAD_STAMP(10);
    }
}

}  // namespace

template <typename PcmT, bool CELL>
static hipError_t launch_one(int sr, const FrontArgs &a, const CellArgs &c, hipStream_t s) {
    if (a.B <= 0) return hipSuccess;
    if (a.nt != 1 || a.T != 1 || a.dec > 1) return hipErrorInvalidValue;
    if (sr == 16000) hipLaunchKernelGGL((step_one_kernel<32, PcmT, CELL>), dim3((unsigned)a.B), dim3(256), 0, s, a, c);
    else hipLaunchKernelGGL((step_one_kernel<16, PcmT, CELL>), dim3((unsigned)a.B), dim3(256), 0, s, a, c);
    return hipGetLastError();
}
template <typename PcmT>
hipError_t launch_step_one(int sr, const FrontArgs &a, const CellArgs &c, hipStream_t s) {
    return launch_one<PcmT, true>(sr, a, c, s);
}
template <typename PcmT>
hipError_t launch_front_one(int sr, const FrontArgs &a, hipStream_t s) {
    return launch_one<PcmT, false>(sr, a, CellArgs{}, s);
}
template hipError_t launch_step_one<float>(int, const FrontArgs &, const CellArgs &, hipStream_t);
template hipError_t launch_step_one<int16_t>(int, const FrontArgs &, const CellArgs &, hipStream_t);
template hipError_t launch_front_one<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_front_one<int16_t>(int, const FrontArgs &, hipStream_t);

}  // namespace vad
