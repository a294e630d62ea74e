// exact_front.hpp -- the frontend of ONE chunk evaluated in double precision by a 256-thread workgroup, for the chunks whose fp32
// evaluation is ill-conditioned: those that hold an EXACTLY silent STFT frame next to a frame that is not silent.
//
// Why.  Every fp32 evaluation of the network loses accuracy where the signal drops abruptly to exact zeros (a muted microphone, a DTX
// gap, the zero padding behind a recording's end) or comes back from them: the frame that straddles the edge has a flat, broadband
// spectrum, so encoder 0's sums cancel 129 large terms per tap, and the F(4,3) form of encoder 0 additionally produces the silent
// frames' outputs by cancelling terms of the loud frame's size.  Measured against a float64 evaluation of the network on 1 025 streams
// that all drop to zeros one sample into a chunk (tools/zero_run_study.py, profiles/r06_state_rows.md): carried (h, c) 1.2e-4 from
// float64 for the F(4,3) chain, 0.7-0.9e-4 for the tap-by-tap chain, 0.2e-4 for the reference's blocked summation -- the contract is
// 1e-4.  On continuous audio all of them sit at 0.5-1.5e-5.  A one-accumulator fp32 chain cannot be made to sum like the
// reference's vectorised dot products without accumulators the throughput kernel does not have; what it can do is notice such a chunk
// -- four v_cmp per tile -- and hand it to arithmetic that has no conditioning problem at all.
//
// The rule (a pure function of the chunk's own samples and context, so a result never depends on the batch it is computed in): a chunk
// is EXACT if at least one of its four STFT frames has |Y[0..3]| == 0 exactly in fp32 and at least one has not.  (A frame of zero samples
// has every bin exactly zero, in the FFT as in the DFT; a frame with a non-zero sample whose four lowest bins are all exactly zero is
// not something audio does, and would only buy that chunk the better arithmetic.  Four silent frames: the chunk's gx does not depend on
// any sample -- it is a constant of the net, see silent_chunks below.)  An exact chunk's gate pre-activations gx[512] are computed here:
// reference definition, dense DFT with the reference's own fp32 basis, every tap, every sum in double, ONE rounding to fp32 at the
// end (JIT!/vad/utils/pytorch_stft.py:17-34, JIT!/vad/utils/model_utils.py:19-25, the W_ih half of JIT!/torch/nn/modules/rnn.py:69).
//
// Who calls it.  kernel_front_lat.hip (latency form, fused step): in the kernel, the workgroup of the tile, before the recurrent half
// -- no extra launch on the streaming path.  kernel_front_f43.hip (throughput form; 253 of 256 registers in use) only appends the
// chunk to a list; kernel_exact.hip's fix-up kernel computes the listed chunks and overwrites their columns of the gx scratch before
// the recurrence reads it.  Same function, same thread roles, same order of every sum: identical bits on every route.
// Cost of a listed chunk: ~0.6 M double FMAs on 256 threads, 10-20 us; none for a chunk that is not listed.
#pragma once
#include <hip/hip_runtime.h>

#include "device_api.hpp"

#ifndef VAD_NO_EXACT
#define VAD_NO_EXACT 0           // 1: timing A/B only (tools/variants.py noexact): the frontends without the silent-frame test
#endif

namespace vad {
namespace {

template <int Q>
struct ExactWs {                               // LDS workspace of one chunk: 26 KB (16 kHz)
    static constexpr int N = 16 * Q, C = 2 * Q, F = 8 * Q, H = 4 * Q, K = 4 * Q + 1;
    double xp[C + N + C];                      // context | chunk | right reflect pad
    double part[2 * K * 4];                    // [real | imaginary][bin][frame]
    double mag[K * 4];                         // [bin][frame]
    double e0[128 * 4];                        // [channel][frame], like torch
    double e1[64 * 2];
    double e2[64];
    double e3[128];
    float gx[512];
};

// sample i of the chunk's input x1 = ctx | chunk, i in [-C, N), exactly as the frontends' loads see it (fft_wave.hpp load_slice): the
// carried context for the first chunk of the call, the zero-padded tail copy for a partial last chunk, every DEC-th raw sample for a
// 32 / 48 kHz input, int16 scaled by 1 / 32768
template <int Q, typename PcmT, int DEC>
__device__ __forceinline__ float exact_sample(const FrontArgs &a, long b, long t, int i) {
    constexpr int N = 16 * Q, C = 2 * Q;
    if (a.pcm == nullptr) return 0.f;                            // (a chunk of zeros: launch_exact_silent)
    if (t == 0 && i < 0) return a.ctx_in[(size_t)b * C + (C + i)];
    const PcmT *row = reinterpret_cast<const PcmT *>(a.pcm) + (size_t)b * a.ld;
    if (a.tail != nullptr && t == a.T - 1 && i >= 0)
        return load_pcm(reinterpret_cast<const PcmT *>(a.tail) + ((size_t)b * N + i) * DEC);
    return load_pcm(row + ((long)N * t + i) * DEC);
}

__device__ __forceinline__ double relu_d(double x) { return x <= 0.0 ? 0.0 : x; }      // NaN stays NaN (torch.relu = clamp_min)

// 4 consecutive floats of a weight row (rows start 4-byte aligned only: 387- and 129-float rows), as doubles
__device__ __forceinline__ void ld4(const float *p, double (&w)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (double)p[e];
}

// All 256 threads of the workgroup.  Result in ws.gx (valid after the function returns: it ends with a barrier).
// Every loop streams ONE weight row per thread from the L2-resident canonical tensors; the loops are unrolled by hand in groups of
// 8-16 elements so that a group's loads are all in flight before its first FMA (one exposed load latency per group, not per element:
// ~0.1 ms per chunk instead of ~1 ms).  Every sum is ONE double accumulator in index order: the thread mapping is part of the
// definition (identical bits wherever the function is called from).
template <int Q, typename PcmT, int DEC>
__device__ __forceinline__ void exact_gx(const FrontArgs &a, const RefNet &net, long b, long t, ExactWs<Q> &ws) {
    using W = ExactWs<Q>;
    constexpr int N = W::N, C = W::C, F = W::F, H = W::H, K = W::K, L = C + N;
    const int tid = threadIdx.x;
    for (int i = tid; i < L; i += 256) ws.xp[i] = (double)exact_sample<Q, PcmT, DEC>(a, b, t, i - C);
    __syncthreads();
    for (int j = tid; j < C; j += 256) ws.xp[L + j] = ws.xp[L - 2 - j];                   // right reflect, edge not repeated
    __syncthreads();
    // STFT: rows [0, K) of the basis are window x cos, rows [K, 2 K) window x -sin.  Thread (k, part): bin k < K - 1, part 0 = real,
    // 1 = imaginary, all four frames; the last bin (Nyquist) is done by threads 0 and 1 in a second round.
    {
        double acc[4];
        auto row_dot = [&](const float *row) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = 0.0;
#pragma clang loop unroll(disable)
            for (int n0 = 0; n0 < F; n0 += 16) {
                float w[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) w[e] = row[n0 + e];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const double c = (double)w[e];
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = fma(c, ws.xp[m * H + n0 + e], acc[m]);
                }
            }
        };
        double *part_buf = ws.part;
        const int k = tid >> 1, part = tid & 1;
        if (k < K - 1) {
            row_dot(net.basis + (size_t)(part * K + k) * F);
#pragma unroll
            for (int m = 0; m < 4; ++m) part_buf[(part * K + k) * 4 + m] = acc[m];
        }
        if (tid < 2) {
            row_dot(net.basis + (size_t)(tid * K + K - 1) * F);
#pragma unroll
            for (int m = 0; m < 4; ++m) part_buf[(tid * K + K - 1) * 4 + m] = acc[m];
        }
        __syncthreads();
        for (int i = tid; i < K * 4; i += 256) {
            const double re = part_buf[i], im = part_buf[K * 4 + i];
            ws.mag[i] = sqrt(fma(re, re, im * im));                 // (written out: nothing is left to the compiler's contraction, which
                                                                    //  may differ between the kernels this function is inlined into)
        }
    }
    __syncthreads();
    // encoder 0: K -> 128 channels, stride 1, 4 -> 4 frames.  Thread (o, half): frames 2 half, 2 half + 1 of output channel o
    {
        const int o = tid & 127, u0 = 2 * (tid >> 7);
        const float *w = net.ew[0] + (size_t)o * K * 3;
        double acc[2] = {0, 0};
        auto taps = [&](int i, double w0, double w1, double w2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int u = u0 + d;
                if (u > 0) acc[d] = fma(w0, ws.mag[i * 4 + u - 1], acc[d]);
                acc[d] = fma(w1, ws.mag[i * 4 + u], acc[d]);
                if (u < 3) acc[d] = fma(w2, ws.mag[i * 4 + u + 1], acc[d]);
            }
        };
        static_assert((K - 1) % 4 == 0, "input channels in groups of four, the last one apart");
#pragma clang loop unroll(disable)
        for (int i0 = 0; i0 < K - 1; i0 += 4) {
            float v[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) v[e] = w[3 * i0 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) taps(i0 + e, (double)v[3 * e], (double)v[3 * e + 1], (double)v[3 * e + 2]);
        }
        taps(K - 1, (double)w[3 * (K - 1)], (double)w[3 * (K - 1) + 1], (double)w[3 * (K - 1) + 2]);
        const double bias = (double)net.eb[0][o];
        ws.e0[o * 4 + u0] = relu_d(acc[0] + bias);
        ws.e0[o * 4 + u0 + 1] = relu_d(acc[1] + bias);
    }
    __syncthreads();
    // encoder 1: 128 -> 64, stride 2, 4 -> 2 frames: input frame v = 2 u + tau - 1
    if (tid < 128) {
        const int o = tid & 63, u = tid >> 6;
        const float *w = net.ew[1] + (size_t)o * 128 * 3;
        double acc = 0;
#pragma clang loop unroll(disable)
        for (int i0 = 0; i0 < 128; i0 += 4) {
            float v[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) v[e] = w[3 * i0 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tau = 0; tau < 3; ++tau) {
                    const int fr = 2 * u + tau - 1;
                    if (fr >= 0 && fr < 4) acc = fma((double)v[3 * e + tau], ws.e0[(i0 + e) * 4 + fr], acc);
                }
        }
        ws.e1[o * 2 + u] = relu_d(acc + (double)net.eb[1][o]);
    }
    __syncthreads();
    // encoder 2: 64 -> 64, stride 2, 2 -> 1 frame: taps 1, 2 see frames 0, 1
    if (tid < 64) {
        const float *w = net.ew[2] + (size_t)tid * 64 * 3;
        double acc = 0;
#pragma clang loop unroll(disable)
        for (int i0 = 0; i0 < 64; i0 += 4) {
            float v[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) v[e] = w[3 * i0 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = fma((double)v[3 * e + 1], ws.e1[(i0 + e) * 2], acc);
                acc = fma((double)v[3 * e + 2], ws.e1[(i0 + e) * 2 + 1], acc);
            }
        }
        ws.e2[tid] = relu_d(acc + (double)net.eb[2][tid]);
    }
    __syncthreads();
    // encoder 3: 64 -> 128, one frame: the centre tap only
    if (tid < 128) {
        const float *w = net.ew[3] + (size_t)tid * 64 * 3;
        double acc = 0;
#pragma clang loop unroll(disable)
        for (int i0 = 0; i0 < 64; i0 += 8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = w[3 * (i0 + e) + 1];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fma((double)v[e], ws.e2[i0 + e], acc);
        }
        ws.e3[tid] = relu_d(acc + (double)net.eb[3][tid]);
    }
    __syncthreads();
    // LSTM input half: gx = W_ih z + b_ih + b_hh, rows tid and tid + 256 (gate order i, f, g, o)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = tid + 256 * h;
        const float *w = net.w_ih + (size_t)r * 128;
        double acc = 0;
#pragma clang loop unroll(disable)
        for (int j0 = 0; j0 < 128; j0 += 16) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = w[j0 + e];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc = fma((double)v[e], ws.e3[j0 + e], acc);
        }
        ws.gx[r] = (float)(acc + (double)net.b_ih[r] + (double)net.b_hh[r]);
    }
    __syncthreads();
}

// Which of the 16 chunks of a tile hold silent frames (bit j = chunk j), from the four frames' magnitudes in "mag layout": every lane's
// X[0] is one of the chunk's bins 0..3 (fft_wave.hpp: X[s] = |Y[4 s + P[g]]|), the four lanes j, j + 16, j + 32, j + 48 hold all four.
// Four compares per lane; the rest is scalar.  Wave-uniform result.
//   edge:   some frame silent, some not -> exact_gx
//   silent: all four frames silent      -> the constant gx_silent[512] (the same double-precision evaluation of a chunk of zeros, done
//           once per net when the engine is created).  The fp32 chains are not ill-conditioned there, but they make the SAME rounding
//           error in every silent chunk, and a cell state that integrates a constant offset over a long silence drifts coherently:
//           4.6e-5 from float64 one chunk into a silence at 16 kHz, 1e-5 with the constant (tools/zero_run_study.py).
struct SilentMasks {
    unsigned edge, silent;
};
__device__ __forceinline__ SilentMasks silent_chunks(float x0, float x1, float x2, float x3) {
    auto quiet = [](float x) -> unsigned {
        const unsigned long long m = __ballot(x == 0.0f);
        return (unsigned)(m & (m >> 16) & (m >> 32) & (m >> 48)) & 0xffffu;
    };
    const unsigned z0 = quiet(x0), z1 = quiet(x1), z2 = quiet(x2), z3 = quiet(x3);
    const unsigned any = z0 | z1 | z2 | z3, all = z0 & z1 & z2 & z3;
    return SilentMasks{any & ~all, all};
}
constexpr int kExactSilentBit = 1 << 30;       // list entries: chunk id | this bit for an all-silent chunk

}  // namespace
}  // namespace vad
