// kernel_rec.hip -- the only time-sequential part of the path: the W_hh half of the LSTM cell,
// the LSTM pointwise update and the ReLU -> 1x1 conv -> sigmoid head
// (reference: aten::lstm_cell, JIT!/torch/nn/modules/rnn.py:69, gate order i,f,g,o;
//  JIT!/vad/model/vad_annotator.py:170-187; head JIT!/torch/nn/modules/container/___torch_mangle_7.py).
//
// Persistent-RNN layout for gfx950:
//   * one workgroup = 8 waves = 16 streams, looping over the slab's time steps;
//   * wave w owns hidden units [16w, 16w+16) and therefore the gate rows {128q + 16w + i}: its
//     slice of W_hh (4 gates x 32 k-steps = 128 MFMA A fragments) stays in 128 VGPRs for the whole
//     launch -- W_hh (256 KiB) never leaves the register file of the CU;
//   * per step: acc = gx_t (the frontend wrote it in exactly this D-fragment order), then
//     128 x v_mfma_f32_16x16x4_f32 against h_{t-1}; the pointwise LSTM update is lane-local (a lane
//     holds i,f,g,o of the same 4 units); h_t goes to a double-buffered 8 KiB LDS image that is
//     the next step's B operand for all 8 waves (chain layout, layout.hpp); ONE barrier per step;
//   * the head's 128-term dot product is reduced lane -> wave (2 shuffles) -> workgroup (LDS).
#include <hip/hip_runtime.h>

#include "activations.hpp"
#include "device_api.hpp"
#include "layout.hpp"

#ifndef VAD_REC_SKEW
#define VAD_REC_SKEW 0           // 0: rec_kernel (all waves in phase; the product), 1: rec_skew_kernel (A/B; measured slower, see there)
#endif
#ifndef VAD_REC_GATE_MAJOR
#define VAD_REC_GATE_MAJOR 1     // 0: k-group-major MFMA order, all activations after the last MFMA (A/B)
#endif

namespace vad {
namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));

template <int NTAB_WOUT, int NTAB_BOUT>
__global__ void __launch_bounds__(512, 2) rec_kernel(const RecArgs a) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][8 * 256];   // [buf][wave][lane][4]
    __shared__ float pbuf[2][8 * 16];                                  // [buf][wave][stream]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const long st = blockIdx.x;
    const long b = st * 16 + j;
    const long bc = b < a.B ? b : a.B - 1;
    const bool valid = b < a.B && (a.present == nullptr || a.present[bc] != 0);          // (an absent row keeps its state: vad_step_present)

    // W_hh slice -> registers: A[q][kg] holds k-steps 4kg..4kg+3 of gate q
    f32x4 A[4][8];
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.whh) + (size_t)w * 4 * 8 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) A[q][kg] = src[(q * 8 + kg) * 64];
    }
    const f32x4 wo = 0.5f * *reinterpret_cast<const f32x4 *>(a.tables + NTAB_WOUT + 16 * w + 4 * g);   // halved: relu2_f (activations.hpp)
    const float bo = a.tables[NTAB_BOUT];

    // state: lane (g, j) holds units 16w + 4g + r of stream j
    const size_t soff = (size_t)bc * 128 + 16 * w + 4 * g;
    f32x4 h = *reinterpret_cast<const f32x4 *>(a.state + soff);
    f32x4 c = *reinterpret_cast<const f32x4 *>(a.state + (size_t)a.B * 128 + soff);
    *reinterpret_cast<f32x4 *>(&hbuf[0][(w * 64 + lane) * 4]) = h;

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
        // gates += W_hh h_{t-1}.  The head of step t-1 is finished by wave 0 in the middle of this step's MFMAs: right
        // behind the barrier it would delay that wave's MFMAs by an LDS read + exp + rcp chain, and with one barrier per
        // step the slowest wave paces all 8 (pbuf is double buffered: step t-1's partial sums stay until step t+1's).
        const f32x4 *hb = reinterpret_cast<const f32x4 *>(&hbuf[cur][0]) + lane;
        const bool head = t > 0 && w == 0;                   // wave-uniform
        float ps[8];
        if (head) {
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) ps[ww] = pbuf[cur ^ 1][ww * 16 + j];
        }
#if VAD_REC_GATE_MAJOR
        // Gate-major order -- i, f, g first (all 8 k-groups each), the o gate last -- so that the pointwise work of the first
        // three gates, the cell update and tanh(c) are in flight while the o gate's 32 MFMAs issue; only sigmoid(o) * tanh(c)
        // is left behind the last MFMA.  The quarter-rate transcendental chains execute beside the matrix pipe, they need
        // not follow it (1.216 -> 1.13 ms; splitting further -- i, f | g | o -- was slower, 1.156).  Same sums in the same
        // order per accumulator, same pointwise formulas: bit-identical results.
        f32x4 hv[8];
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) hv[kg] = hb[kg * 64];          // units 16kg + 4g + r of stream j
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int kg = 0; kg < 8; ++kg)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q][kg][r], hv[kg][r], acc[q], 0, 0, 0);
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
#pragma unroll
        for (int kg = 0; kg < 8; ++kg)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[3][kg][r], hv[kg][r], acc[3], 0, 0, 0);
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            h[r] = sigmoid_f(acc[3][r]) * th[r];
            part = fmaf(wo[r], relu2_f(h[r]), part);
        }
#else
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
            const f32x4 hv = hb[kg * 64];                    // units 16kg + 4g + r of stream j
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q][kg][r], hv[r], acc[q], 0, 0, 0);
            if (kg == 3 && head) {
                float p = bo;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) p += ps[ww];
                if (valid && g == 0) a.probs[(size_t)b * a.ldp + a.t0 + t - 1] = sigmoid_f(p);
            }
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
            part = fmaf(wo[r], relu2_f(h[r]), part);
        }
#endif
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        *reinterpret_cast<f32x4 *>(&hbuf[cur ^ 1][(w * 64 + lane) * 4]) = h;
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


// ---- A/B form (VAD_REC_SKEW=1; NOT the product: 1.188 ms against rec_kernel's 1.136 per C2 launch, profiles/r03a_issue_pipes2.md):
// the same arithmetic, the two waves of a SIMD half a step apart -------------------------------------------------------
// rec_kernel above puts all 8 waves in phase: everybody's MFMAs, then everybody's activation tail, the barrier, the
// LDS round trip of h -- and for that tail (~1 us of a 4.4 us step) the matrix pipe of all four SIMDs idles.  Here the
// waves form two halves, A = waves 0..3 (hidden units 0..63 = k-groups 0..3 of the next step) and B = waves 4..7
// (k-groups 4..7); wave w and wave w + 4 share a SIMD.  A step's 128 MFMAs per wave are split by k-group into a first
// half (k-groups 0..3: needs only A's h) and a second half (k-groups 4..7: needs only B's h), and A runs half a step
// ahead of B.  Two phases per step, one barrier each:
//     P1(t):  A: second half of step t, then its tail (activations, h_t(A) -> LDS)      B: first half of step t
//     P2(t):  A: first half of step t+1 (needs h_t(A), written in P1(t))               B: second half of step t, tail
// so that in every phase each SIMD holds one wave that ends in a tail and one that has 64 more MFMAs to issue: the tail
// runs under the partner's MFMAs instead of under nothing.  B's operands are always one phase old, so B reads them from
// LDS a phase early and starts its MFMAs right behind the barrier while A waits for its LDS read.
// Same sums in the same order per accumulator (k-groups 0..7 ascending), same pointwise formulas, same order in the
// head's 8-term sum: bit-identical to rec_kernel (tests: test_full_size_launches_are_bit_stable, the golden vectors).
#ifndef VAD_REC_PRIO
#define VAD_REC_PRIO 1           // 1: the wave whose phase ends in a tail issues at raised priority
#endif

template <int NTAB_WOUT, int NTAB_BOUT>
__global__ void __launch_bounds__(512, 2) rec_skew_kernel(const RecArgs a) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][8 * 256];   // [buf][wave][lane][4]
    __shared__ float pbuf[2][8 * 16];                                  // [buf][wave][stream]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool isB = w >= 4;                                            // wave-uniform
    const int g = lane >> 4, j = lane & 15;
    const long st = blockIdx.x;
    const long b = st * 16 + j;
    const long bc = b < a.B ? b : a.B - 1;
    const bool valid = b < a.B && (a.present == nullptr || a.present[bc] != 0);          // (an absent row keeps its state: vad_step_present)

    f32x4 A[4][8];
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.whh) + (size_t)w * 4 * 8 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) A[q][kg] = src[(q * 8 + kg) * 64];
    }
    const f32x4 wo = 0.5f * *reinterpret_cast<const f32x4 *>(a.tables + NTAB_WOUT + 16 * w + 4 * g);   // halved: relu2_f (activations.hpp)
    const float bo = a.tables[NTAB_BOUT];

    const size_t soff = (size_t)bc * 128 + 16 * w + 4 * g;
    f32x4 h = *reinterpret_cast<const f32x4 *>(a.state + soff);
    f32x4 c = *reinterpret_cast<const f32x4 *>(a.state + (size_t)a.B * 128 + soff);
    *reinterpret_cast<f32x4 *>(&hbuf[0][(w * 64 + lane) * 4]) = h;     // h_{-1} lives in buffer 0; h_t in buffer (t + 1) & 1

    const f32x4 *gx = reinterpret_cast<const f32x4 *>(a.gx) + ((size_t)st * a.nt * 32) * 64 + lane;
    f32x4 gnext[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gnext[q] = gx[(size_t)(8 * q + w) * 64];
    f32x4 acc[4];
    f32x4 hv[4], hn[4];                                                // operands of this phase / (B only) of the next

    auto wg_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // own LDS writes (and B's early reads) have completed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto read_h = [&](f32x4 (&dst)[4], int buf, int kg0) {
        const f32x4 *hb = reinterpret_cast<const f32x4 *>(&hbuf[buf][0]) + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = hb[(kg0 + k) * 64];       // units 16 (kg0 + k) + 4g + r of stream j
    };
    // first half of a step: gates += W_hh[:, units 0..63] h(A); four independent accumulator chains
    auto first_half = [&]() {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q][kg][r], hv[kg][r], acc[q], 0, 0, 0);
    };
    // second half + tail: k-groups 4..7; i, f, g first so that their activations, the cell update and tanh(c) are in flight
    // while the o gate's MFMAs issue (as in rec_kernel's gate-major order); returns the head's partial sum
    auto second_half_tail = [&]() -> float {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q][4 + kg][r], hv[kg][r], acc[q], 0, 0, 0);
        float th[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = sigmoid_f(acc[0][r]), fg = sigmoid_f(acc[1][r]), gg = tanh_f(acc[2][r]);
            const float cn = fmaf(fg, c[r], ig * gg);
            c[r] = cn;
            th[r] = tanh_f(cn);
        }
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[3][4 + kg][r], hv[kg][r], acc[3], 0, 0, 0);
        if (VAD_REC_PRIO) __builtin_amdgcn_s_setprio(0);
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            h[r] = sigmoid_f(acc[3][r]) * th[r];
            part = fmaf(wo[r], relu2_f(h[r]), part);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        return part;
    };
    auto take_gx = [&](long tnext) {                                   // acc <- gx of the step that starts; prefetch the one after
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = gnext[q];
        if (tnext < a.nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) gnext[q] = gx[((size_t)tnext * 32 + 8 * q + w) * 64];
        }
    };

    __syncthreads();
    // prologue = P2(-1): A runs the first half of step 0; B fetches its operands for P1(0)
    if (!isB) {
        take_gx(1);
        read_h(hv, 0, 0);
        first_half();
    } else {
        read_h(hn, 0, 0);
    }
    wg_barrier();

    for (long t = 0; t < a.nt; ++t) {
        const int cur = (int)(t & 1);                                  // h_{t-1} is in hbuf[cur], h_t goes to hbuf[cur ^ 1]
        // ---------------- P1(t)
        if (!isB) {
            if (VAD_REC_PRIO) __builtin_amdgcn_s_setprio(1);
            read_h(hv, cur, 4);                                        // h_{t-1}(B), written in P2(t-1)
            const bool head = t > 0 && w == 0;                         // the head of step t-1 is finished here, under the MFMAs
            float ps[8];
            if (head) {
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) ps[ww] = pbuf[cur ^ 1][ww * 16 + j];
            }
            const float part = second_half_tail();
            if (head) {
                float p = bo;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) p += ps[ww];
                if (valid && g == 0) a.probs[(size_t)b * a.ldp + a.t0 + t - 1] = sigmoid_f(p);
            }
            *reinterpret_cast<f32x4 *>(&hbuf[cur ^ 1][(w * 64 + lane) * 4]) = h;
            if (g == 0) pbuf[cur][w * 16 + j] = part;
        } else {
            take_gx(t + 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) hv[k] = hn[k];                 // h_{t-1}(A), read a phase ago
            read_h(hn, cur, 4);                                        // h_{t-1}(B) for P2(t), written in P2(t-1)
            first_half();
        }
        wg_barrier();
        // ---------------- P2(t)
        if (!isB) {
            if (t + 1 < a.nt) {
                take_gx(t + 2);
                read_h(hv, cur ^ 1, 0);                                // h_t(A), written in P1(t)
                first_half();
            }
        } else {
            if (VAD_REC_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int k = 0; k < 4; ++k) hv[k] = hn[k];
            read_h(hn, cur ^ 1, 0);                                    // h_t(A) for P1(t+1), written in P1(t)
            const float part = second_half_tail();
            *reinterpret_cast<f32x4 *>(&hbuf[cur ^ 1][(w * 64 + lane) * 4]) = h;
            if (g == 0) pbuf[cur][w * 16 + j] = part;
        }
        wg_barrier();
    }
    if (w == 0 && g == 0) {                                            // head of the last step
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

namespace {
__global__ void activation_probe_kernel(int kind, const float *x, float *y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = kind == 0 ? sigmoid_f(x[i]) : tanh_f(x[i]);
}
}  // namespace
hipError_t launch_activation_probe(int kind, const float *x, float *y, long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(activation_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, kind, x, y, n);
    return hipGetLastError();
}

hipError_t launch_rec(int sr, const RecArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((a.B + 15) / 16);
#if VAD_REC_SKEW
    if (sr == 16000)
        hipLaunchKernelGGL((rec_skew_kernel<vadl::tab16.w_out, vadl::tab16.b_out>), dim3(grid), dim3(512), 0, s, a);
    else
        hipLaunchKernelGGL((rec_skew_kernel<vadl::tab8.w_out, vadl::tab8.b_out>), dim3(grid), dim3(512), 0, s, a);
#else
    if (sr == 16000)
        hipLaunchKernelGGL((rec_kernel<vadl::tab16.w_out, vadl::tab16.b_out>), dim3(grid), dim3(512), 0, s, a);
    else
        hipLaunchKernelGGL((rec_kernel<vadl::tab8.w_out, vadl::tab8.b_out>), dim3(grid), dim3(512), 0, s, a);
#endif
    return hipGetLastError();
}

}  // namespace vad
