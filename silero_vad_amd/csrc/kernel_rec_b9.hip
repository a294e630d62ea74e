// kernel_rec_b9.hip -- the recurrence (same function as kernel_rec.hip: the W_hh half of the LSTM cell, the pointwise update
// and the ReLU -> 1x1 conv -> sigmoid head) with W_hh * h evaluated as EXACT bf16 x 9 products on the bf16 matrix pipe.
// (reference: aten::lstm_cell, JIT!/torch/nn/modules/rnn.py:69, gate order i,f,g,o; JIT!/vad/model/vad_annotator.py:170-187;
//  head JIT!/torch/nn/modules/container/___torch_mangle_7.py:10-19.)
//
// Why.  On gfx950 a v_mfma_f32_16x16x4_f32 owns its SIMD's vector issue for 32 cycles: the recurrence's 256 MFMAs and ~1 600
// VALU cycles per step and SIMD ADD (profiles/r03a_issue_pipes2.md), and the fp32 kernel sits at 0.94 of that floor.  The
// bf16 matrix pipe is 16x faster per MAC and runs BESIDE the VALU.
//
// Arithmetic.  Every fp32 operand is the exact sum of three bf16 pieces (8 significand bits each, fp32's exponent range:
// x = p0 + p1 + p2 with p0 = bf16(x), p1 = bf16(x - p0), p2 = x - p0 - p1; nothing is dropped, nothing underflows), so
// w * h = sum of the NINE piece products, each of which is exact in fp32 (8 x 8 significand bits); v_mfma_f32_16x16x32_bf16
// accumulates them in fp32.  Against kernel_rec.hip, whose fp32 MFMA chain rounds every product once and every partial sum
// once, this forms the same sum from exact products and rounds only in the accumulation: it is not narrower than fp32 -- but
// it is a different summation, so it is an option ("rec" = "bf16x9"), not the default, until the comparison against a float64
// evaluation of the recurrence says otherwise (tests/test_gpu_parity.py::test_rec_bf16x9_against_float64).
// Weights are split on the host (layout.hpp "bf16 x 9 recurrent image"); h_t is split by the lanes that produce it and
// exchanged through LDS already in B-operand form.
//
// Layout: one workgroup = 8 waves = 16 streams, one workgroup per CU; wave w owns hidden units [16w, 16w+16).  Its W_hh slice
// is 3 pieces x 64 VGPRs: pieces 0 and 1 stay in 128 VGPRs for the whole launch, piece 2 of all 8 waves stays in 128 KiB of
// LDS (the CU's register file cannot hold 384 KiB of weights plus two waves per SIMD).  Per step and wave: 4 gates x 4 K32
// steps x 9 = 144 MFMAs of 16 cycles (2 304 cycles against 4 096 for the 128 fp32 MFMAs) and the activations beside them.
// Measured at C2: 0.79 ms against 1.14 ms for the fp32 recurrence (3.1 us per step against a 1.95 us matrix-pipe floor: what is
// left is the per-step barrier, the LDS round trip of h and 28 KiB of LDS operand reads per wave and step).  A variant with the two
// waves of a SIMD half a step apart (as tried for the fp32 kernel) measured 0.87 ms and was not kept.
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
using bf8 = __bf16 __attribute__((ext_vector_type(8)));
using bf2 = __bf16 __attribute__((ext_vector_type(2)));

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

template <int NTAB_WOUT, int NTAB_BOUT>
__global__ void __launch_bounds__(512, 1) rec_b9_kernel(const RecArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned w2s[8 * 4 * 4 * 64 * 4];   // W_hh piece 2: [wave][gate][u][lane][4 dwords]
    __shared__ __attribute__((aligned(16))) unsigned hb[2][3 * 4 * 64 * 4];     // h pieces: [buf][piece][u][lane][4 dwords]
    __shared__ float pbuf[2][8 * 16];                                             // head partial sums: [buf][wave][stream]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const long st = blockIdx.x;
    const long b = st * 16 + j;
    const long bc = b < a.B ? b : a.B - 1;
    const bool valid = b < a.B && (a.present == nullptr || a.present[bc] != 0);          // (an absent row keeps its state: vad_step_present)

    // W_hh slice: pieces 0, 1 -> registers, piece 2 -> LDS.  Image [wave][piece][gate][u][lane][8 bf16] (16 B per lane)
    const u32x4 *img = reinterpret_cast<const u32x4 *>(a.whh) + (size_t)w * 3 * 16 * 64 + lane;
    u32x4 A0[4][4], A1[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            A0[q][u] = img[(0 * 16 + q * 4 + u) * 64];
            A1[q][u] = img[(1 * 16 + q * 4 + u) * 64];
            *reinterpret_cast<u32x4 *>(&w2s[(((w * 4 + q) * 4 + u) * 64 + lane) * 4]) = img[(2 * 16 + q * 4 + u) * 64];
        }
    const f32x4 wo = 0.5f * *reinterpret_cast<const f32x4 *>(a.tables + NTAB_WOUT + 16 * w + 4 * g);   // halved: relu2_f (activations.hpp)
    const float bo = a.tables[NTAB_BOUT];

    // state: lane (g, j) holds units 16w + 4g + r of stream j (the D layout of this wave's gate rows)
    const size_t soff = (size_t)bc * 128 + 16 * w + 4 * g;
    f32x4 h = *reinterpret_cast<const f32x4 *>(a.state + soff);
    f32x4 c = *reinterpret_cast<const f32x4 *>(a.state + (size_t)a.B * 128 + soff);
    // where this lane's four h values live in the B-operand image: unit n = 16w + 4g + r <-> K32 step n / 32, slot group
    // (n % 32) / 8, element n % 8: four consecutive bf16 (8 bytes)
    const int hu = w >> 1, hg = 2 * (w & 1) + (g >> 1), he = 4 * (g & 1);
    const int hslot = ((hu * 64 + hg * 16 + j) * 4) + (he >> 1);              // dword index inside one piece's [u][lane][4]
    auto publish = [&](int buf) {
        unsigned a0, a1, a2, b0, b1, b2;
        split3(h[0], h[1], a0, a1, a2);
        split3(h[2], h[3], b0, b1, b2);
        *reinterpret_cast<u32x2 *>(&hb[buf][0 * 1024 + hslot]) = u32x2{a0, b0};
        *reinterpret_cast<u32x2 *>(&hb[buf][1 * 1024 + hslot]) = u32x2{a1, b1};
        *reinterpret_cast<u32x2 *>(&hb[buf][2 * 1024 + hslot]) = u32x2{a2, b2};
    };
    publish(0);

    const f32x4 *gx = reinterpret_cast<const f32x4 *>(a.gx) + ((size_t)st * a.nt * 32) * 64 + lane;
    f32x4 gnext[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gnext[q] = gx[(size_t)(8 * q + w) * 64];
    __syncthreads();

    for (long t = 0; t < a.nt; ++t) {
        const int cur = (int)(t & 1);
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = gnext[q];
        if (t + 1 < a.nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) gnext[q] = gx[((size_t)(t + 1) * 32 + 8 * q + w) * 64];
        }
        const bool head = t > 0 && w == 0;                   // wave-uniform: the head of step t-1 is finished under this step's MFMAs
        float ps[8];
        if (head) {
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) ps[ww] = pbuf[cur ^ 1][ww * 16 + j];
        }
        // gates += W_hh h_{t-1}: per K32 step u the nine piece products, smallest first; gates i, f, g first, the o gate last,
        // so that the pointwise work of the first three is in flight while the o gate's MFMAs issue
        u32x4 H[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int p = 0; p < 3; ++p) H[u][p] = *reinterpret_cast<const u32x4 *>(&hb[cur][((p * 4 + u) * 64 + lane) * 4]);
        auto gate = [&](int q) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32x4 W2 = *reinterpret_cast<const u32x4 *>(&w2s[(((w * 4 + q) * 4 + u) * 64 + lane) * 4]);
                acc[q] = mfma_b(W2, H[u][2], acc[q]);
                acc[q] = mfma_b(W2, H[u][1], acc[q]);
                acc[q] = mfma_b(A1[q][u], H[u][2], acc[q]);
                acc[q] = mfma_b(W2, H[u][0], acc[q]);
                acc[q] = mfma_b(A0[q][u], H[u][2], acc[q]);
                acc[q] = mfma_b(A1[q][u], H[u][1], acc[q]);
                acc[q] = mfma_b(A1[q][u], H[u][0], acc[q]);
                acc[q] = mfma_b(A0[q][u], H[u][1], acc[q]);
                acc[q] = mfma_b(A0[q][u], H[u][0], acc[q]);
            }
        };
        gate(0);
        gate(1);
        gate(2);
        if (head) {
            float p = bo;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) p += ps[ww];
            if (valid && g == 0) a.probs[(size_t)b * a.ldp + a.t0 + t - 1] = sigmoid_f(p);
        }
        float th[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = sigmoid_f(acc[0][r]), fg = sigmoid_f(acc[1][r]), gg = tanh_f(acc[2][r]);
            const float cn = fmaf(fg, c[r], ig * gg);
            c[r] = cn;
            th[r] = tanh_f(cn);
        }
        gate(3);
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            h[r] = sigmoid_f(acc[3][r]) * th[r];
            part = fmaf(wo[r], relu2_f(h[r]), part);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        publish(cur ^ 1);
        if (g == 0) pbuf[cur][w * 16 + j] = part;
        __syncthreads();
    }
    if (w == 0 && g == 0) {                                  // head of the last step
        const int last = (int)((a.nt - 1) & 1);
        float p = bo;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) p += pbuf[last][ww * 16 + j];
        if (valid) a.probs[(size_t)b * a.ldp + a.t0 + a.nt - 1] = sigmoid_f(p);
    }
    if (valid) {
        *reinterpret_cast<f32x4 *>(a.state + soff) = h;
        *reinterpret_cast<f32x4 *>(a.state + (size_t)a.B * 128 + soff) = c;
    }
}


}  // namespace

hipError_t launch_rec_b9(int sr, const RecArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((a.B + 15) / 16);
    if (sr == 16000)
        hipLaunchKernelGGL((rec_b9_kernel<vadl::tab16.w_out, vadl::tab16.b_out>), dim3(grid), dim3(512), 0, s, a);
    else
        hipLaunchKernelGGL((rec_b9_kernel<vadl::tab8.w_out, vadl::tab8.b_out>), dim3(grid), dim3(512), 0, s, a);
    return hipGetLastError();
}

}  // namespace vad
