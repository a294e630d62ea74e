// kernel_rec_split.hip -- the time-sequential part of the path (same function as kernel_rec.hip: the
// W_hh half of the LSTM cell, the pointwise update and the ReLU -> 1x1 conv -> sigmoid head) with
// W_hh * h evaluated as an fp16 x 3 split product on v_mfma_f32_16x16x32_f16 (see
// kernel_front_split.hip for the arithmetic; h is in (-1, 1), so there is no range question here).
// (reference: aten::lstm_cell, JIT!/torch/nn/modules/rnn.py:69, gate order i,f,g,o;
//  JIT!/vad/model/vad_annotator.py:170-187; head JIT!/torch/nn/modules/container/___torch_mangle_7.py).
//
// Same persistent-RNN layout as kernel_rec.hip: one workgroup = 8 waves = 16 streams; wave w owns
// hidden units [16w, 16w+16) and keeps its slice of W_hh as (hi, lo) half fragments in 128 VGPRs for
// the whole launch.  Per step 48 MFMAs of 16 cycles replace 128 of 32 cycles; h_t is exchanged through
// a double-buffered LDS image that already is the packed (hi, lo) B operand of the next step.
// A NaN/Inf cell state (poisoned gx, kernel_front_split.hip) is propagated to the probability.
// Built without packed-fp32 VALU instructions, like kernel_front_split.hip (see the note there).
#include <hip/hip_runtime.h>

#include "activations.hpp"
#include "device_api.hpp"
#include "layout.hpp"

namespace vad {
namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x2 = float __attribute__((ext_vector_type(2)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using u32x2 = unsigned __attribute__((ext_vector_type(2)));
using h8 = _Float16 __attribute__((ext_vector_type(8)));
using h2 = _Float16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_h(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hi, unsigned &lo) {
#pragma clang fp contract(off)      // lo must come from the ROUNDED x (a caller's multiply must not fuse in):
    const f32x2 v{x0, x1};              // the same x, reloaded from HBM by a later call, has to split alike
    const h2 h = __builtin_convertvector(v, h2);
    const f32x2 r = v - __builtin_convertvector(h, f32x2);
    const h2 l = __builtin_convertvector(r, h2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

#ifndef VAD_REC_DEPTH
#define VAD_REC_DEPTH 2
#endif
constexpr int kDepth = VAD_REC_DEPTH;      // gx prefetch distance in time steps

template <int NTAB_WOUT, int NTAB_BOUT>
__global__ void __launch_bounds__(512, 2) rec_split_kernel(const RecArgs a) {
    // [buf][u 4][hi|lo][lane 64][4 words]: word pair (w & 1) of lane l in step u = w >> 1 is written by
    // wave w; a reader gets the 8 halves of its K32-step operand with one ds_read_b128.
    __shared__ __attribute__((aligned(16))) unsigned hbuf[2][4 * 2 * 256];
    __shared__ float pbuf[2][8 * 16];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const long st = blockIdx.x;
    const long b = st * 16 + j;
    const bool valid = b < a.B;
    const long bc = valid ? b : a.B - 1;

    u32x4 Ah[4][4], Al[4][4];
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(a.whh) + (size_t)w * 4 * 4 * 2 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                Ah[q][u] = src[((q * 4 + u) * 2 + 0) * 64];
                Al[q][u] = src[((q * 4 + u) * 2 + 1) * 64];
            }
    }
    const f32x4 wo = *reinterpret_cast<const f32x4 *>(a.tables + NTAB_WOUT + 16 * w + 4 * g);
    const float bo = a.tables[NTAB_BOUT];

    const size_t soff = (size_t)bc * 128 + 16 * w + 4 * g;
    f32x4 h = *reinterpret_cast<const f32x4 *>(a.state + soff);
    f32x4 c = *reinterpret_cast<const f32x4 *>(a.state + (size_t)a.B * 128 + soff);
    const int hw = (((w >> 1) * 2) * 64 + lane) * 4 + (w & 1) * 2;     // word index of this lane's hi pair
    auto publish = [&](int buf) {
        unsigned h0, l0, h1, l1;
        split2(h[0], h[1], h0, l0);
        split2(h[2], h[3], h1, l1);
        const u32x2 hi{h0, h1}, lo{l0, l1};
        *reinterpret_cast<u32x2 *>(&hbuf[buf][hw]) = hi;
        *reinterpret_cast<u32x2 *>(&hbuf[buf][hw + 256]) = lo;
    };
    publish(0);

    const f32x4 *gx = reinterpret_cast<const f32x4 *>(a.gx) + ((size_t)st * a.nt * 32) * 64 + lane;
    f32x4 gpre[kDepth][4];
#pragma unroll
    for (int d = 0; d < kDepth; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            gpre[d][q] = __builtin_nontemporal_load(gx + ((size_t)(d < a.nt ? d : 0) * 32 + 8 * q + w) * 64);
    __syncthreads();

    for (long t = 0; t < a.nt; ++t) {
        const int cur = (int)(t & 1);
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = gpre[0][q];
#pragma unroll
        for (int d = 0; d + 1 < kDepth; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) gpre[d][q] = gpre[d + 1][q];
        {
            const long tn = t + kDepth < a.nt ? t + kDepth : a.nt - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                gpre[kDepth - 1][q] = __builtin_nontemporal_load(gx + ((size_t)tn * 32 + 8 * q + w) * 64);   // streamed once
        }
        // gates += W_hh h_{t-1}
        const u32x4 *hb = reinterpret_cast<const u32x4 *>(&hbuf[cur][0]) + lane;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32x4 bh = hb[(u * 2 + 0) * 64], bl = hb[(u * 2 + 1) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma_h(Ah[q][u], bh, acc[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma_h(Ah[q][u], bl, acc[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma_h(Al[q][u], bh, acc[q]);
        }
        // pointwise LSTM + head partial
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = sigmoid_f(acc[0][r]), fg = sigmoid_f(acc[1][r]);
            const float gg = tanh_f(acc[2][r]), og = sigmoid_f(acc[3][r]);
            const float cn = fmaf(fg, c[r], ig * gg);
            c[r] = cn;
            h[r] = og * tanh_f(cn);
            part = fmaf(wo[r], fmaxf(h[r], 0.f), part);
            part = fmaf(0.f, cn, part);                  // NaN/Inf cell state -> NaN probability
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        publish(cur ^ 1);
        if (g == 0) pbuf[cur][w * 16 + j] = part;
        __syncthreads();
        if (w == 0 && g == 0) {
            float p = bo;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) p += pbuf[cur][ww * 16 + j];
            if (valid) a.probs[(size_t)b * a.ldp + a.t0 + t] = sigmoid_f(p);
        }
    }
    if (valid) {
        *reinterpret_cast<f32x4 *>(a.state + soff) = h;
        *reinterpret_cast<f32x4 *>(a.state + (size_t)a.B * 128 + soff) = c;
    }
}

// bring-up probe: one v_mfma_f32_16x16x32_f16 on caller-supplied fragments (tests pin the slot pairing
// and the subnormal behaviour the split arithmetic relies on)
__global__ void mfma_f16_probe_kernel(const u32x4 *a, const u32x4 *b, f32x4 *d) {
    const int lane = threadIdx.x;
    d[lane] = mfma_h(a[lane], b[lane], f32x4{0.f, 0.f, 0.f, 0.f});
}

}  // namespace

hipError_t launch_mfma_f16_probe(const void *a, const void *b, float *d, hipStream_t s) {
    hipLaunchKernelGGL(mfma_f16_probe_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const u32x4 *>(a),
                       reinterpret_cast<const u32x4 *>(b), reinterpret_cast<f32x4 *>(d));
    return hipGetLastError();
}

hipError_t launch_rec_split(int sr, const RecArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((a.B + 15) / 16);
    if (sr == 16000)
        hipLaunchKernelGGL((rec_split_kernel<vadl::tab16.w_out, vadl::tab16.b_out>), dim3(grid), dim3(512), 0, s, a);
    else
        hipLaunchKernelGGL((rec_split_kernel<vadl::tab8.w_out, vadl::tab8.b_out>), dim3(grid), dim3(512), 0, s, a);
    return hipGetLastError();
}

}  // namespace vad
