// kernels_ref.hip -- impl="reference": slow, obviously-correct all-VALU kernels kept as an
// on-device A/B for the MFMA path (tests only; never the default).  One workgroup per stream,
// time loop inside, canonical (un-packed) weights, dense DFT-basis STFT exactly as the reference
// computes it (JIT!/vad/utils/pytorch_stft.py:17-34).
#include <hip/hip_runtime.h>

#include "activations.hpp"
#include "device_api.hpp"

namespace vad {

template <int N, int C, int F, int K>
__device__ void ref_chunk(const RefNet &w, const float *xp, float *sm, float *h, float *c,
                          float *prob_out) {
    constexpr int H = F / 2;
    float *mag = sm;                 // [K][4]
    float *a0 = mag + 132 * 4;       // [128][4]
    float *a1 = a0 + 128 * 4;        // [64][2]
    float *a2 = a1 + 64 * 2;         // [64]
    float *a3 = a2 + 64;             // [128]
    float *gates = a3 + 128;         // [512]
    const int tid = threadIdx.x, nt = blockDim.x;
    // non-finite input (fft_wave.hpp, "non-finite input"): a chunk with any non-finite magnitude is NaN, as in the MFMA path
    __shared__ int poisoned;
    if (tid == 0) poisoned = 0;
    __syncthreads();

    for (int idx = tid; idx < K * 4; idx += nt) {
        const int k = idx >> 2, m = idx & 3;
        const float *br = w.basis + (size_t)k * F, *bi = w.basis + (size_t)(K + k) * F;
        float re = 0.f, im = 0.f;
        for (int n = 0; n < F; ++n) {
            const float v = xp[m * H + n];
            re = fmaf(br[n], v, re);
            im = fmaf(bi[n], v, im);
        }
        const float mg = sqrtf(re * re + im * im);
        mag[k * 4 + m] = mg;
        if (!(fabsf(mg) < __builtin_inff())) poisoned = 1;            // (benign race: every writer stores 1)
    }
    __syncthreads();

    const float *in = mag;
    float *outs[4] = {a0, a1, a2, a3};
    const int cin[4] = {K, 128, 64, 64}, cout[4] = {128, 64, 64, 128}, st[4] = {1, 2, 2, 1};
    int T = 4;
    for (int l = 0; l < 4; ++l) {
        const int To = (T - 1) / st[l] + 1;
        for (int idx = tid; idx < cout[l] * To; idx += nt) {
            const int o = idx / To, u = idx % To;
            const float *wr = w.ew[l] + (size_t)o * cin[l] * 3;
            float acc = w.eb[l][o];
            for (int tau = 0; tau < 3; ++tau) {
                const int v = u * st[l] + tau - 1;
                if (v < 0 || v >= T) continue;
                for (int i = 0; i < cin[l]; ++i) acc = fmaf(wr[i * 3 + tau], in[i * T + v], acc);
            }
            outs[l][o * To + u] = relu_f(acc);
        }
        __syncthreads();
        in = outs[l];
        T = To;
    }

    for (int r = tid; r < 512; r += nt) {
        const float *wi = w.w_ih + (size_t)r * 128, *wh = w.w_hh + (size_t)r * 128;
        float a = 0.f, b = 0.f;
        for (int j = 0; j < 128; ++j) {
            a = fmaf(wi[j], a3[j], a);
            b = fmaf(wh[j], h[j], b);
        }
        gates[r] = poisoned ? __builtin_nanf("") : (a + w.b_ih[r]) + (b + w.b_hh[r]);
    }
    __syncthreads();
    if (tid < 128) {
        const int j = tid;
        const float ig = 1.f / (1.f + expf(-gates[j])), fg = 1.f / (1.f + expf(-gates[128 + j]));
        const float gg = tanhf(gates[256 + j]), og = 1.f / (1.f + expf(-gates[384 + j]));
        const float cn = fg * c[j] + ig * gg;
        c[j] = cn;
        h[j] = og * tanhf(cn);
    }
    __syncthreads();
    if (tid == 0) {
        float p = w.b_out[0];
        for (int j = 0; j < 128; ++j) p = fmaf(w.w_out[j], relu_f(h[j]), p);
        *prob_out = 1.f / (1.f + expf(-p));
    }
    __syncthreads();
}

template <int N, int C, int F, int K, typename PcmT>
__global__ void __launch_bounds__(256)
ref_forward_kernel(RefNet w, const PcmT *pcm, long ld, long L, long T, float *ctx, float *state,
                   int B, float *probs, long ldp) {
    __shared__ float xp[N + 2 * C];
    __shared__ float sm[132 * 4 + 128 * 4 + 64 * 2 + 64 + 128 + 512];
    __shared__ float h[128], c[128];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 128) {
        h[tid] = state[(size_t)b * 128 + tid];
        c[tid] = state[(size_t)B * 128 + (size_t)b * 128 + tid];
    }
    if (tid < C) xp[tid] = ctx[(size_t)b * C + tid];
    __syncthreads();
    for (long t = 0; t < T; ++t) {
        for (int i = tid; i < N; i += blockDim.x) {
            const long s = t * N + i;
            xp[C + i] = s < L ? load_pcm(pcm + (size_t)b * ld + s) : 0.f;
        }
        __syncthreads();
        if (tid < C) xp[C + N + tid] = xp[C + N - 2 - tid];        // right reflect pad
        __syncthreads();
        ref_chunk<N, C, F, K>(w, xp, sm, h, c, probs + (size_t)b * ldp + t);
        if (tid < C) xp[tid] = xp[N + tid];                         // ctx = last C samples of x1
        __syncthreads();
    }
    if (tid < 128) {
        state[(size_t)b * 128 + tid] = h[tid];
        state[(size_t)B * 128 + (size_t)b * 128 + tid] = c[tid];
    }
    if (tid < C) ctx[(size_t)b * C + tid] = xp[tid];
}

template <typename PcmT>
hipError_t launch_ref_forward(const RefNet &w, int sr, int B, long L, const PcmT *pcm, long ld,
                              float *ctx, float *state, float *probs, long ldp, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (sr == 16000) {
        const long T = (L + 511) / 512;
        hipLaunchKernelGGL((ref_forward_kernel<512, 64, 256, 129, PcmT>), dim3(B), dim3(256), 0, s, w,
                           pcm, ld, L, T, ctx, state, B, probs, ldp);
    } else {
        const long T = (L + 255) / 256;
        hipLaunchKernelGGL((ref_forward_kernel<256, 32, 128, 65, PcmT>), dim3(B), dim3(256), 0, s, w,
                           pcm, ld, L, T, ctx, state, B, probs, ldp);
    }
    return hipGetLastError();
}

// ---- sample-rate front door ---------------------------------------------------------------------------
// dst[b][i] = src[b][i * k], i < Ld: the reference's decimation of 32 / 48 / ... kHz input to 16 kHz,
// `x[:, ::step]` (JIT!/vad/model/vad_annotator.py:104-112, src/silero_vad/utils_vad.py:39-42): no
// filter, first sample kept.  HBM-bound gather; rows of dst are 16-byte aligned (ldd % 8 == 0).
template <typename PcmT>
__global__ void decimate_kernel(const PcmT *src, long lds, PcmT *dst, long ldd, long Ld, int k) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long b = blockIdx.y;
    if (i < Ld) dst[b * ldd + i] = src[b * lds + i * k];
}
template <typename PcmT>
hipError_t launch_decimate(const PcmT *src, long lds, PcmT *dst, long ldd, int B, long Ld, int k, hipStream_t s) {
    if (B <= 0 || Ld <= 0) return hipSuccess;
    hipLaunchKernelGGL((decimate_kernel<PcmT>), dim3((unsigned)((Ld + 255) / 256), (unsigned)B), dim3(256), 0, s,
                       src, lds, dst, ldd, Ld, k);
    return hipGetLastError();
}
template hipError_t launch_decimate<float>(const float *, long, float *, long, int, long, int, hipStream_t);
template hipError_t launch_decimate<int16_t>(const int16_t *, long, int16_t *, long, int, long, int, hipStream_t);

// ---- test hook: a "foreign tenant" ------------------------------------------------------------------------
// A kernel that does nothing but VALU fp32 FMAs for `iters` rounds, in one-wave workgroups with a handful of
// registers, so that its waves fit beside anything.  kind 0: packed fp32 (v_pk_fma_f32), kind 1: scalar fp32
// (v_fma_f32).  tests/test_gpu_parity.py launches it on a second stream while the engine runs, to pin down
// that the engine's results do not depend on what another tenant of the GPU is doing (test_bit_stable_under_foreign_load).
__global__ void __launch_bounds__(64) foreign_spin_kernel(float *sink, long iters, int kind) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a{1.0f + threadIdx.x * 1e-3f, 0.5f}, b{0.25f, 0.75f};
    const f2 m{0.999f, 1.0001f}, c{1e-3f, -1e-3f};
    if (kind == 0) {
        for (long i = 0; i < iters; ++i)
            asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3\n\t"
                         "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3"
                         : "+v"(a), "+v"(b) : "v"(m), "v"(c));
    } else {
        for (long i = 0; i < iters; ++i)
            asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\t"
                         "v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3"
                         : "+v"(a.x), "+v"(b.x) : "v"(m.x), "v"(c.x));
    }
    if (a.x + a.y + b.x + b.y == 123.456f) sink[0] = a.x;             // keep the chain alive
}
hipError_t launch_foreign_spin(float *sink, int blocks, long iters, int kind, hipStream_t s) {
    if (blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(foreign_spin_kernel, dim3((unsigned)blocks), dim3(64), 0, s, sink, iters, kind);
    return hipGetLastError();
}

// ---- test hook: gx (MFMA D-fragment order) -> row-major [B][T][512] for vad_debug_frontend ----------------------------
namespace {
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
hipError_t launch_unpack_gx(const float *gx, float *out, int B, long T, hipStream_t s) {
    const long total = (long)B * T * 512;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_gx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gx, out, B, T);
    return hipGetLastError();
}


template hipError_t launch_ref_forward<float>(const RefNet &, int, int, long, const float *, long,
                                              float *, float *, float *, long, hipStream_t);
template hipError_t launch_ref_forward<int16_t>(const RefNet &, int, int, long, const int16_t *,
                                                long, float *, float *, float *, long, hipStream_t);

}  // namespace vad
