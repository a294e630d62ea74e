// kernel_front.hip -- the time-parallel part of the Silero-VAD hot path as ONE fused kernel:
//
//   PCM (fp32 or int16, HBM) -> framing + right reflect pad -> Hann window -> 4 x real FFT
//   magnitude -> 4 x ReLU(Conv1d k=3) -> W_ih * feat + (b_ih + b_hh)  => gx (LSTM input gates)
//
// (reference: JIT!/vad/model/vad_annotator.py:58-67 framing, JIT!/vad/utils/pytorch_stft.py:17-34
//  STFT, JIT!/vad/utils/model_utils.py:19-25 encoder, the W_ih half of aten::lstm_cell
//  JIT!/torch/nn/modules/rnn.py:69.)
//
// Design (gfx950 / CDNA4, fp32 everywhere -- SURVEY.md section 0.4):
//   * one WAVE owns 16 chunks (16 streams x one time step) end to end; lane (g, j) = (l>>4, l&15),
//     column j of every MFMA is chunk j.  A workgroup is 4 such waves; two workgroups share a CU
//     so that one's VALU (FFT) phases overlap the other's MFMA phases.
//   * STFT is a 4-lane cooperative real FFT, not the reference's dense 258x256 DFT-basis conv:
//     the 4 lanes of a chunk each load a 2Q-sample slice of the frame, do a radix-4 butterfly
//     ACROSS lanes (two wave shuffles), then a Q-point complex FFT entirely in registers, then the
//     real-FFT split.  Lane group g ends up holding bins 4k'+P[g] -- any fixed bin->(register,
//     group) placement is acceptable because ...
//   * ... every matrix product runs on v_mfma_f32_16x16x4_f32 with activations kept in registers
//     in "chain layout" (layout.hpp): the D fragment of one layer IS the B operand of the next,
//     the k-order permutation being folded into the packed weights on the host.  Activations
//     never touch LDS or HBM between the FFT and the gx store.
//   * weights (0.65 MB packed, L2 resident) stream global -> LDS with global_load_lds_dwordx4
//     (no VGPR round trip) through a 2-slot ring shared by the 4 waves, one k-group (4 k-steps x
//     all row blocks, <= 8 KiB) per slot, and reach the MFMA A operand by ds_read_b128.
//   * zero-padding taps of the convs are skipped (enc0 10/12, enc1 5/6, enc2 2/3, enc3 1/3 taps).
#include <hip/hip_runtime.h>

#include "device_api.hpp"
#include "layout.hpp"

namespace vad {
namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));

constexpr int kRingSlotFloats = 2048;   // 8 KiB: one k-group of an 8-row-block segment

// W_32^j = cos - i sin, j < 16 (fp32-rounded from double)
__device__ constexpr float kCos32[16] = {1.0f, 0.98078525f, 0.9238795f, 0.8314696f, 0.70710677f,
    0.55557024f, 0.38268343f, 0.19509032f, 0.0f, -0.19509032f, -0.38268343f, -0.55557024f,
    -0.70710677f, -0.8314696f, -0.9238795f, -0.98078525f};
__device__ constexpr float kSin32[16] = {0.0f, 0.19509032f, 0.38268343f, 0.55557024f, 0.70710677f,
    0.8314696f, 0.9238795f, 0.98078525f, 1.0f, 0.98078525f, 0.9238795f, 0.8314696f, 0.70710677f,
    0.55557024f, 0.38268343f, 0.19509032f};

constexpr int bitrev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x / 2); }

// ---- per-lane context ---------------------------------------------------------------------------
struct Lane {
    int lane, g, j, wave;
    long t;                  // absolute time step of this wave's tile
    long tl;                 // slab-relative
    long st;                 // stream tile
    int b;                   // stream of this lane (clamped to B-1)
    bool tile_valid;         // wave-uniform
    bool from_tail;          // wave-uniform: this is the last, partial chunk -> read a.tail
    float sgnA, sgnB;        // +-1 butterfly signs for the cross-lane radix-4
};

// ---- weight ring ----------------------------------------------------------------------------------
struct Ring {
    float *slots;            // LDS, 2 x kRingSlotFloats
    const float *wfront;     // global
    int unit;                // units consumed so far (wave-uniform)
};

template <int M>
__device__ __forceinline__ void ring_issue(const Ring &r, long goff, int slot, const Lane &ln) {
    // M blocks of 1 KiB; wave w copies blocks w, w+4.  LDS destination = wave-uniform base + lane*16.
#pragma unroll
    for (int blk = 0; blk < M; blk += 4) {
        const int bb = blk + ln.wave;
        const float *src = r.wfront + goff + (long)bb * 256 + ln.lane * 4;
        float *dst = r.slots + slot * kRingSlotFloats + bb * 256;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
}

// One segment = KG k-groups of M row blocks.  acc[m] += A_seg[m][:, k] * B[k][:]  for all k-steps.
// bfun(s) must return the B-operand register of k-step s (compile-time s).
// NEXT_M / next_off describe the unit that follows this segment in program order (prefetch).
template <int M, int KS, int NEXT_M, class BF>
__device__ __forceinline__ void gemm_seg(f32x4 (&acc)[M], BF bfun, Ring &ring, long seg_off,
                                         long next_off, const Lane &ln) {
    constexpr int KG = (KS + 3) / 4;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        __syncthreads();     // unit `ring.unit` has landed for every wave; the other slot is free
        const int slot = ring.unit & 1;
        if (kg + 1 < KG) ring_issue<M>(ring, seg_off + (long)(kg + 1) * M * 256, slot ^ 1, ln);
        else if (NEXT_M > 0) ring_issue<(NEXT_M > 0 ? NEXT_M : 4)>(ring, next_off, slot ^ 1, ln);
        const f32x4 *A = reinterpret_cast<const f32x4 *>(ring.slots + slot * kRingSlotFloats) + ln.lane;
#pragma unroll
        for (int mp = 0; mp < M; mp += 2) {
            const f32x4 a0 = A[(mp + 0) * 64];
            const f32x4 a1 = A[(mp + 1) * 64];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (kg * 4 + ks < KS) {
                    const float bv = bfun(kg * 4 + ks);
                    acc[mp + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ks], bv, acc[mp + 0], 0, 0, 0);
                    acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], bv, acc[mp + 1], 0, 0, 0);
                }
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
template <int M>
__device__ __forceinline__ void relu(f32x4 (&acc)[M]) {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][r] = fmaxf(acc[m][r], 0.f);
}

// ---- PCM slice loads --------------------------------------------------------------------------------
__device__ __forceinline__ void cvt8(const u32x4 v, float *o) {       // 8 x int16 -> float / 32768
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lo = (int)(v[k] << 16) >> 16, hi = (int)v[k] >> 16;
        o[2 * k] = (float)lo * (1.0f / 32768.0f);
        o[2 * k + 1] = (float)hi * (1.0f / 32768.0f);
    }
}
template <int SL>
__device__ __forceinline__ void load_vec(const float *p, float (&s)[SL]) {
#pragma unroll
    for (int k = 0; k < SL / 4; ++k) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(p)[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[4 * k + e] = v[e];
    }
}
template <int SL>
__device__ __forceinline__ void load_vec(const int16_t *p, float (&s)[SL]) {
#pragma unroll
    for (int k = 0; k < SL / 8; ++k) cvt8(reinterpret_cast<const u32x4 *>(p)[k], &s[8 * k]);
}

// slice V of the lane: s[i] = x[2Q*(2V+g) + i], x = ctx | chunk (| reflected tail for V==3,g==3).
// Vector loads only: the engine guarantees 16-byte aligned rows, and hands the (zero padded) last
// chunk of every stream in `tail` when L is not a multiple of the chunk size.
template <int Q, int V, typename PcmT>
__device__ __forceinline__ void load_slice(float (&s)[2 * Q], const FrontArgs &a, const Lane &ln) {
    constexpr int SL = 2 * Q, N = 16 * Q;
    const PcmT *row = reinterpret_cast<const PcmT *>(a.pcm) + (size_t)ln.b * a.ld;
    const int sigma = 2 * V + ln.g;
    const int sg = (V == 3 && sigma > 8) ? 8 : sigma;
    const long p0 = (long)SL * (8 * ln.t - 1 + sg);          // stream-absolute index of s[0]
    const PcmT *src = row + p0;
    const PcmT *esrc = row + ((long)N * ln.t + N - SL - 1);   // x[16Q-1], for the reflect pad
    if (ln.from_tail) {                                       // wave-uniform
        const PcmT *trow = reinterpret_cast<const PcmT *>(a.tail) + (size_t)ln.b * N;
        if (sg > 0) src = trow + SL * (sg - 1);
        esrc = trow + (N - SL - 1);
    }
    if (V == 0 && ln.t == 0 && ln.g == 0) load_vec<SL>(a.ctx_in + (size_t)ln.b * SL, s);
    else load_vec<SL>(src, s);
    if (V == 3) {
        // context for the next call = last C = 2Q samples of the (zero padded) last chunk = slice 8
        if (a.ctx_out && ln.t == a.T - 1 && ln.g == 2 && ln.tile_valid &&
            (ln.st * 16 + ln.j) < a.B) {
            float *o = a.ctx_out + (size_t)ln.b * SL;
#pragma unroll
            for (int k = 0; k < SL / 4; ++k)
                reinterpret_cast<f32x4 *>(o)[k] = f32x4{s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]};
        }
        // right reflect pad: lanes g == 3 need x[18Q-2-i] = slice8[2Q-2-i] (i < 2Q-1), x[16Q-1] (i = 2Q-1)
        const bool rev = ln.g == 3;
        float extra = 0.f;
        if (rev) extra = load_pcm(esrc);
#pragma unroll
        for (int i = 0; i < SL / 2 - 1; ++i) {
            const int k = SL - 2 - i;                          // i <-> k, i < k
            const float lo = s[i], hi = s[k];
            s[i] = rev ? hi : lo;
            s[k] = rev ? lo : hi;
        }
        s[SL - 1] = rev ? extra : s[SL - 1];
    }
}

// ---- in-register Q-point complex FFT, DIF radix-2, output index bit-reversed ---------------------
template <int Q>
__device__ __forceinline__ void fft_inlane(float (&re)[Q], float (&im)[Q]) {
#pragma unroll
    for (int n = Q; n >= 2; n >>= 1) {
        const int half = n >> 1;
#pragma unroll
        for (int b0 = 0; b0 < Q; b0 += n) {
#pragma unroll
            for (int jx = 0; jx < half; ++jx) {
                const int i0 = b0 + jx, i1 = i0 + half;
                const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
                re[i0] = ar + br;
                im[i0] = ai + bi;
                const float dr = ar - br, di = ai - bi;
                const int tw = jx * (32 / n);                  // W_n^jx = W_32^(jx*32/n)
                if (tw == 0) { re[i1] = dr; im[i1] = di; }
                else if (tw == 8) { re[i1] = di; im[i1] = -dr; }                 // * (-i)
                else {
                    const float c = kCos32[tw], sn = kSin32[tw];               // * (c - i sn)
                    re[i1] = fmaf(dr, c, di * sn);
                    im[i1] = fmaf(di, c, -(dr * sn));
                }
            }
        }
    }
}

// One frame (V) of 16 chunks: X[s] (s < Q): |Y[4s + P[g]]|;  X[Q]: |Y[4Q]| in group 0, 0 elsewhere.
template <int Q, int V, typename PcmT>
__device__ __forceinline__ void fft_pass(float (&X)[Q + 1], const FrontArgs &a, const float *tab_lds,
                                         const Lane &ln) {
    constexpr int SL = 2 * Q;
    constexpr vadl::Tab tb = vadl::make_tab(8 * Q, Q);
    __builtin_amdgcn_sched_barrier(0);     // keep each pass's loads inside the pass (register budget)
    float s[SL];
    load_slice<Q, V, PcmT>(s, a, ln);

    float re[Q], im[Q];
    {   // window (same taps for every frame: the lane's slice always sits at 2Q g inside the frame)
        const f32x4 *w = reinterpret_cast<const f32x4 *>(tab_lds + tb.window + SL * ln.g);
#pragma unroll
        for (int k = 0; k < SL / 4; ++k) {
            const f32x4 wv = w[k];
            re[2 * k] = s[4 * k] * wv[0];
            im[2 * k] = s[4 * k + 1] * wv[1];
            re[2 * k + 1] = s[4 * k + 2] * wv[2];
            im[2 * k + 1] = s[4 * k + 3] * wv[3];
        }
    }
    // radix-4 across the 4 lanes of a chunk.  Stage A pairs g <-> g^2, stage B pairs g <-> g^1.
    const float *tw1 = tab_lds + tb.tw1 + ln.g * Q * 2;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        float xr = re[q], xi = im[q];
        const float pr = __shfl_xor(xr, 32), pi = __shfl_xor(xi, 32);
        xr = fmaf(ln.sgnA, xr, pr);                // g<2: own + partner ; g>=2: partner - own
        xi = fmaf(ln.sgnA, xi, pi);
        if (ln.g == 3) { const float tr = xr; xr = xi; xi = -tr; }     // * (-i)
        const float qr = __shfl_xor(xr, 16), qi = __shfl_xor(xi, 16);
        xr = fmaf(ln.sgnB, xr, qr);                // g even: own + partner ; g odd: partner - own
        xi = fmaf(ln.sgnB, xi, qi);
        const float c = tw1[2 * q], sn = tw1[2 * q + 1];               // * W_4Q^(P[g] q)
        re[q] = fmaf(xr, c, -(xi * sn));
        im[q] = fmaf(xr, sn, xi * c);
    }
    fft_inlane<Q>(re, im);
    // real-FFT split: Y[k] = E + W_8Q^k O from Z[k] and conj Z[4Q - k]
    constexpr int LG = ilog2(Q);
    const float *tw2 = tab_lds + tb.tw2 + ln.g * Q * 2;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const float ur = re[bitrev(k, LG)], ui = im[bitrev(k, LG)];
        const int ks = bitrev((Q - k) % Q, LG), kr = bitrev(Q - 1 - k, LG);
        const float xr = __shfl_xor(re[kr], 16), xi = __shfl_xor(im[kr], 16);
        float pr = ln.g >= 2 ? xr : re[kr], pi = ln.g >= 2 ? xi : im[kr];
        pr = ln.g == 0 ? re[ks] : pr;
        pi = ln.g == 0 ? im[ks] : pi;
        const float ar = ur + pr, ai = ui - pi;                // Z + conj(Zp)
        const float dr = ui + pi, di = pr - ur;                // -i (Z - conj(Zp))
        const float c = tw2[2 * k], sn = tw2[2 * k + 1];
        const float yr = ar + fmaf(dr, c, -(di * sn));
        const float yi = ai + fmaf(dr, sn, di * c);
        X[k] = 0.5f * __builtin_amdgcn_sqrtf(fmaf(yr, yr, yi * yi));
    }
    X[Q] = ln.g == 0 ? fabsf(re[0] - im[0]) : 0.f;             // Nyquist: Re Z0 - Im Z0
}

// ---- the kernel ------------------------------------------------------------------------------------
template <int Q, typename PcmT>
__global__ void __launch_bounds__(256, 2) front_kernel(const FrontArgs a) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int TABF = (tb.total + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float lds[TABF + 2 * kRingSlotFloats];
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

    Ring ring{lds + TABF, a.wfront, 0};
    constexpr long o_e0t0 = seg_offset(E0T0, Q), o_e0t1 = seg_offset(E0T1, Q), o_e0t2 = seg_offset(E0T2, Q);
    constexpr long o_e1t0 = seg_offset(E1T0, Q), o_e1t1 = seg_offset(E1T1, Q), o_e1t2 = seg_offset(E1T2, Q);
    constexpr long o_e2t1 = seg_offset(E2T1, Q), o_e2t2 = seg_offset(E2T2, Q), o_e3t1 = seg_offset(E3T1, Q);

    ring_issue<8>(ring, o_e0t1, 0, ln);          // first unit of the program: E0T1 k-group 0
    for (int i = threadIdx.x; i < tb.total; i += 256) tab[i] = a.tables[i];
    __syncthreads();

    float X0[Q + 1], X1[Q + 1], X2[Q + 1], X3[Q + 1];
    fft_pass<Q, 0, PcmT>(X0, a, tab, ln);
    fft_pass<Q, 1, PcmT>(X1, a, tab, ln);
    fft_pass<Q, 2, PcmT>(X2, a, tab, ln);

    auto bX0 = [&](int s) { return X0[s]; };
    auto bX1 = [&](int s) { return X1[s]; };
    auto bX2 = [&](int s) { return X2[s]; };
    auto bX3 = [&](int s) { return X3[s]; };

    f32x4 Y[8], Z0[4], Z1[4];
    auto bY = [&](int s) { return Y[s >> 2][s & 3]; };
    constexpr int KS0 = Q + 1;

    // enc0 frame 0 (taps 1,2; tap 0 is the left zero pad)  -> enc1 out 0 tap 1
    init_bias<8>(Y, tab + tb.b_e0, ln);
    gemm_seg<8, KS0, 8>(Y, bX0, ring, o_e0t1, o_e0t2, ln);
    gemm_seg<8, KS0, 4>(Y, bX1, ring, o_e0t2, o_e1t1, ln);
    relu<8>(Y);
    init_bias<4>(Z0, tab + tb.b_e1, ln);
    gemm_seg<4, 32, 8>(Z0, bY, ring, o_e1t1, o_e0t0, ln);
    // enc0 frame 1 -> enc1 out 0 tap 2, out 1 tap 0
    init_bias<8>(Y, tab + tb.b_e0, ln);
    gemm_seg<8, KS0, 8>(Y, bX0, ring, o_e0t0, o_e0t1, ln);
    gemm_seg<8, KS0, 8>(Y, bX1, ring, o_e0t1, o_e0t2, ln);
    gemm_seg<8, KS0, 4>(Y, bX2, ring, o_e0t2, o_e1t2, ln);
    relu<8>(Y);
    gemm_seg<4, 32, 4>(Z0, bY, ring, o_e1t2, o_e1t0, ln);
    init_bias<4>(Z1, tab + tb.b_e1, ln);
    gemm_seg<4, 32, 8>(Z1, bY, ring, o_e1t0, o_e0t0, ln);

    fft_pass<Q, 3, PcmT>(X3, a, tab, ln);

    // enc0 frame 2 -> enc1 out 1 tap 1
    init_bias<8>(Y, tab + tb.b_e0, ln);
    gemm_seg<8, KS0, 8>(Y, bX1, ring, o_e0t0, o_e0t1, ln);
    gemm_seg<8, KS0, 8>(Y, bX2, ring, o_e0t1, o_e0t2, ln);
    gemm_seg<8, KS0, 4>(Y, bX3, ring, o_e0t2, o_e1t1, ln);
    relu<8>(Y);
    gemm_seg<4, 32, 8>(Z1, bY, ring, o_e1t1, o_e0t0, ln);
    // enc0 frame 3 (taps 0,1; tap 2 is the right zero pad) -> enc1 out 1 tap 2
    init_bias<8>(Y, tab + tb.b_e0, ln);
    gemm_seg<8, KS0, 8>(Y, bX2, ring, o_e0t0, o_e0t1, ln);
    gemm_seg<8, KS0, 4>(Y, bX3, ring, o_e0t1, o_e1t2, ln);
    relu<8>(Y);
    gemm_seg<4, 32, 4>(Z1, bY, ring, o_e1t2, o_e2t1, ln);
    relu<4>(Z0);
    relu<4>(Z1);

    // enc2 (T 2 -> 1, stride 2: taps 1,2 see frames 0,1), enc3 (T = 1: centre tap only)
    f32x4 Vv[4];
    auto bZ0 = [&](int s) { return Z0[s >> 2][s & 3]; };
    auto bZ1 = [&](int s) { return Z1[s >> 2][s & 3]; };
    auto bV = [&](int s) { return Vv[s >> 2][s & 3]; };
    init_bias<4>(Vv, tab + tb.b_e2, ln);
    gemm_seg<4, 16, 4>(Vv, bZ0, ring, o_e2t1, o_e2t2, ln);
    gemm_seg<4, 16, 8>(Vv, bZ1, ring, o_e2t2, o_e3t1, ln);
    relu<4>(Vv);
    f32x4 Fe[8];
    auto bF = [&](int s) { return Fe[s >> 2][s & 3]; };
    init_bias<8>(Fe, tab + tb.b_e3, ln);
    gemm_seg<8, 16, 8>(Fe, bV, ring, o_e3t1, seg_offset(IH0, Q), ln);
    relu<8>(Fe);

    // LSTM input-gate pre-activations, one gate (8 row blocks) at a time, stored in D-fragment order
    float *gxt = a.gx + ((size_t)(ln.st * a.nt + ln.tl) * 32) * 256 + ln.lane * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 G[8];
        init_bias<8>(G, tab + tb.b_g + 128 * q, ln);
        if (q < 3) gemm_seg<8, 32, 8>(G, bF, ring, seg_offset(IH0 + q, Q), seg_offset(IH0 + q + 1, Q), ln);
        else gemm_seg<8, 32, 0>(G, bF, ring, seg_offset(IH3, Q), 0, ln);
        if (ln.tile_valid) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
                *reinterpret_cast<f32x4 *>(gxt + (size_t)(8 * q + m) * 256) = G[m];
        }
    }
}

__global__ void unpack_gx_kernel(const float *gx, float *out, int B, long T) {
    // out[b][t][row] ; gx[st][t][mb][lane][r] with row = 16 mb + 4 g + r, b = 16 st + j
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * T * 512;
    if (idx >= total) return;
    const int row = (int)(idx % 512);
    const long t = (idx / 512) % T;
    const long b = idx / (512 * T);
    const int mb = row >> 4, g = (row >> 2) & 3, r = row & 3, j = (int)(b & 15);
    const long st = b >> 4;
    out[idx] = gx[(((st * T + t) * 32 + mb) * 64 + (g * 16 + j)) * 4 + r];
}

}  // namespace

template <typename PcmT>
hipError_t launch_front(int sr, const FrontArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    const unsigned grid = (unsigned)((total + 3) / 4);
    if (sr == 16000) hipLaunchKernelGGL((front_kernel<32, PcmT>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((front_kernel<16, PcmT>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_front<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_front<int16_t>(int, const FrontArgs &, hipStream_t);

hipError_t launch_unpack_gx(const float *gx, float *out, int B, long T, hipStream_t s) {
    const long total = (long)B * T * 512;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_gx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gx, out, B, T);
    return hipGetLastError();
}

}  // namespace vad
