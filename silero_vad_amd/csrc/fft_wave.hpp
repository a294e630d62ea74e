// fft_wave.hpp -- per-wave framing + in-wave real FFT magnitude shared by the frontend kernels
// (kernel_front_f43.hip: one wave per tile; kernel_front_lat.hip: one frame per wave of a 4-wave tile; test build: kernel_front.hip,
//  kernel_front_wino.hip).
//
// One wave owns 16 chunks; lane (g, j) = (l>>4, l&15); the 4 lanes g of a chunk cooperate on each
// of its 4 STFT frames.  Output is "mag layout" (layout.hpp): X[s], s < Q, = |Y[4s + P[g]]|, and
// X[Q] = Nyquist magnitude in lane group 0.
// (reference: JIT!/vad/model/vad_annotator.py:58-67 framing; JIT!/vad/utils/pytorch_stft.py:17-34;
//  right reflect pad JIT!/torch/nn/modules/padding/___torch_mangle_8.py:6,10.)
#pragma once
#include <hip/hip_runtime.h>

#include "device_api.hpp"
#include "layout.hpp"

#ifndef VAD_NT_PCM
#define VAD_NT_PCM 0     // 1: PCM is loaded with the non-temporal hint (read once; keep the L2 for the weights)
#endif

namespace vad {
namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));

#ifndef VAD_TRACE
#define VAD_TRACE 0              // 1: workgroups write phase timestamps to FrontArgs::trace (tools/trace_front.py)
#endif
#ifndef VAD_ABLATE
#define VAD_ABLATE 0             // timing experiments only (wrong results): 1 no barriers, 2 no FFT
#endif                           // math, 4 no PCM loads, 8 no weight-ring loads

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
#if VAD_TRACE
    mutable unsigned long long ts[20] = {};    // bring-up: shader-clock timestamps of this WAVE, kept in SGPRs (wave-uniform) and written
                                               // out at the end of the kernel -- no vector register, no memory operation in flight
#endif
};
#if VAD_TRACE
#define VAD_WAVE_STAMP(ln, slot) do { (ln).ts[(slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define VAD_WAVE_STAMP(ln, slot) do { } while (0)
#endif

// ---- PCM slice loads --------------------------------------------------------------------------------
__device__ __forceinline__ void cvt8(const u32x4 v, float *o) {       // 8 x int16 -> float / 32768
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lo = (int)(v[k] << 16) >> 16, hi = (int)v[k] >> 16;
        o[2 * k] = (float)lo * (1.0f / 32768.0f);
        o[2 * k + 1] = (float)hi * (1.0f / 32768.0f);
    }
}
// s[i] = p[i * DEC], i < SL.  DEC > 1 is the sample-rate front door folded into the load: the reference decimates
// 32 / 48 kHz input with x[:, ::DEC] (JIT!/vad/model/vad_annotator.py:104-112, src/silero_vad/utils_vad.py:39-42);
// here the lane reads exactly the samples it keeps, so the raw signal is touched once and never copied.
template <int SL, int DEC>
__device__ __forceinline__ void load_vec(const float *p, float (&s)[SL]) {
    if (DEC == 1) {
#pragma unroll
        for (int k = 0; k < SL / 4; ++k) {
            const f32x4 v = VAD_NT_PCM ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p) + k)
                                       : reinterpret_cast<const f32x4 *>(p)[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[4 * k + e] = v[e];
        }
    } else {
        // one 4-byte load per sample that is kept (immediate offsets): loading the whole DEC-times longer span with vector
        // loads and discarding would need DEC x the registers while the loads are in flight -- this kernel has none to spare
#pragma unroll
        for (int i = 0; i < SL; ++i) s[i] = p[i * DEC];
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int SL, int DEC>
__device__ __forceinline__ void load_vec(const int16_t *p, float (&s)[SL]) {
    if (DEC == 1) {
#pragma unroll
        for (int k = 0; k < SL / 8; ++k)
            cvt8(VAD_NT_PCM ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p) + k)
                            : reinterpret_cast<const u32x4 *>(p)[k], &s[8 * k]);
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {                          // two batches: half the raw values in flight at a time
#pragma unroll
            for (int i = h * SL / 2; i < (h + 1) * SL / 2; ++i) s[i] = (float)p[i * DEC] * (1.0f / 32768.0f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// slice V of the lane: s[i] = x[2Q*(2V+g) + i], x = ctx | chunk (| reflected tail for V==3,g==3).
// Vector loads only: the engine guarantees 16-byte aligned rows, and hands the (zero padded) last
// chunk of every stream in `tail` when L is not a multiple of the chunk size.
// V is an ordinary argument: a literal at the call site folds to the frame's own code, a loop variable gives one body
// for all four frames (wave-uniform branches for the context of frame 0 and the reflect pad of frame 3).
template <int Q, typename PcmT, int DEC>
__device__ __forceinline__ void load_slice(float (&s)[2 * Q], const FrontArgs &a, const Lane &ln, const int V) {
    constexpr int SL = 2 * Q, N = 16 * Q;
    const PcmT *row = reinterpret_cast<const PcmT *>(a.pcm) + (size_t)ln.b * a.ld;
    const int sigma = 2 * V + ln.g;
    const int sg = (V == 3 && sigma > 8) ? 8 : sigma;
    const long p0 = (long)SL * (8 * ln.t - 1 + sg);          // stream-absolute index of s[0] (in 16 / 8 kHz samples)
    const PcmT *src = row + p0 * DEC;
    const PcmT *esrc = row + ((long)N * ln.t + N - SL - 1) * DEC;   // x[16Q-1], for the reflect pad
    if (ln.from_tail) {                                       // wave-uniform; the tail copy keeps the raw sample spacing
        const PcmT *trow = reinterpret_cast<const PcmT *>(a.tail) + (size_t)ln.b * N * DEC;
        if (sg > 0) src = trow + SL * (sg - 1) * DEC;
        esrc = trow + (N - SL - 1) * DEC;
    }
    if (VAD_ABLATE & 16) {
        // timing experiment: same instruction count and bytes, but every instruction reads 1 KiB of
        // ONE row (lane l <- 16 B at l*16), i.e. perfectly coalesced
        const PcmT *r0 = reinterpret_cast<const PcmT *>(a.pcm) + (size_t)(ln.st * 16) * a.ld + (long)SL * 8 * ln.t;
#pragma unroll
        for (int k = 0; k < SL / 4; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(r0) + (size_t)(k & 15) * a.ld + ln.lane * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[4 * k + e] = v[e];
        }
    } else if (VAD_ABLATE & 32) {
        // timing experiment: the 4 lanes of a chunk read one contiguous 64-B segment per instruction
        const float *rr = reinterpret_cast<const float *>(row) + (long)SL * 8 * ln.t;
#pragma unroll
        for (int k = 0; k < SL / 4; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(rr + 16 * k + 4 * ln.g + 128 * V);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[4 * k + e] = v[e];
        }
    } else if (DEC == 1) {
        if (V == 0 && ln.t == 0 && ln.g == 0) load_vec<SL, 1>(a.ctx_in + (size_t)ln.b * SL, s);
        else load_vec<SL, 1>(src, s);
    } else if (V == 0) {
        // the carried context (lane group 0 of the first chunk) is stored at unit stride, the signal at stride DEC.  Two
        // divergent load sequences would hold both results in registers; instead EVERY lane runs the strided sequence
        // (lane group 0 of chunk 0 from a harmless in-bounds address) and the context is patched in by select.
        const bool from_ctx = ln.t == 0 && ln.g == 0;
        // (the harmless address: the start of the row -- or of the zero-padded tail copy when the first chunk is also the
        //  last, partial one and the row itself may be shorter than a slice)
        const PcmT *safe = ln.from_tail ? reinterpret_cast<const PcmT *>(a.tail) + (size_t)ln.b * N * DEC : row;
        load_vec<SL, DEC>(from_ctx ? safe : src, s);
        const f32x4 *c4 = reinterpret_cast<const f32x4 *>(a.ctx_in + (size_t)ln.b * SL);
#pragma unroll
        for (int k = 0; k < SL / 4; ++k) {
            const f32x4 v = c4[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[4 * k + e] = from_ctx ? v[e] : s[4 * k + e];
        }
    } else {
        load_vec<SL, DEC>(src, s);
    }
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
// Complex values are float2 (re, im) so that the butterflies compile to packed fp32 math
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two flops per lane per issue).  On gfx950 the fp32
// MFMA and the VALU share one issue pipe per SIMD (tools/ubench/overlap.hip: their times add, they do
// not overlap), so every VALU instruction saved here is kernel time saved.
using f32x2 = float __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 swap2(f32x2 v) { return f32x2{v.y, v.x}; }
// v * (c + i sn)
__device__ __forceinline__ f32x2 cmul(f32x2 v, float c, float sn) {
    return __builtin_elementwise_fma(swap2(v), f32x2{-sn, sn}, v * f32x2{c, c});
}

#ifndef VAD_XLANE_SWAP
#define VAD_XLANE_SWAP 1
#endif
// a of lanes 32..63 <-> b of lanes 0..31 (v_permlane32_swap_b32); a of odd 16-lane rows <-> b of even rows
// (v_permlane16_swap_b32).  Both registers change in place.
__device__ __forceinline__ void trade32(f32x2 &a, f32x2 &b) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
        a[e] = __uint_as_float(r[0]);
        b[e] = __uint_as_float(r[1]);
    }
}
__device__ __forceinline__ void trade16(f32x2 &a, f32x2 &b) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
        a[e] = __uint_as_float(r[0]);
        b[e] = __uint_as_float(r[1]);
    }
}

template <int Q>
__device__ __forceinline__ void fft_inlane(f32x2 (&z)[Q]) {
#pragma unroll
    for (int n = Q; n >= 2; n >>= 1) {
        const int half = n >> 1;
#pragma unroll
        for (int b0 = 0; b0 < Q; b0 += n) {
#pragma unroll
            for (int jx = 0; jx < half; ++jx) {
                const int i0 = b0 + jx, i1 = i0 + half;
                const f32x2 u = z[i0], v = z[i1];
                z[i0] = u + v;
                const f32x2 d = u - v;
                const int tw = jx * (32 / n);                  // W_n^jx = W_32^(jx*32/n)
                if (tw == 0) z[i1] = d;
                else if (tw == 8) z[i1] = swap2(d) * f32x2{1.0f, -1.0f};        // * (-i)
                else z[i1] = cmul(d, kCos32[tw], -kSin32[tw]);                   // * (c - i sn)
            }
        }
    }
}

// The math of one frame of 16 chunks from the lanes' PCM slices s[] (load_slice): X[s] (s < Q): |Y[4s + P[g]]|;  X[Q]: |Y[4Q]| in
// group 0, 0 elsewhere.
template <int Q>
__device__ __forceinline__ void fft_math(float (&X)[Q + 1], const float (&s)[2 * Q], const float *tab_lds, const Lane &ln);

// One frame (V) of 16 chunks: load the slices, then fft_math.
template <int Q, typename PcmT, int DEC = 1>
__device__ __forceinline__ void fft_frame(float (&X)[Q + 1], const int V, const FrontArgs &a, const float *tab_lds,
                                          const Lane &ln) {
    constexpr int SL = 2 * Q;
    __builtin_amdgcn_sched_barrier(0);     // keep each pass's loads inside the pass (register budget)
    VAD_WAVE_STAMP(ln, 16 + V);            // (trace builds: the pass begins -- behind the shift-register moves)
    float s[SL];
    if (VAD_ABLATE & 4) {
#pragma unroll
        for (int i = 0; i < SL; ++i) s[i] = (float)(ln.lane + i) * 1e-3f;
    } else {
        load_slice<Q, PcmT, DEC>(s, a, ln, V);
    }
#if VAD_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (trace builds: the frame's samples have arrived -> slot 2 + 2 V)
    VAD_WAVE_STAMP(ln, 2 + 2 * V);
#endif
    if (VAD_ABLATE & 2) {
#pragma unroll
        for (int k = 0; k < Q; ++k) X[k] = s[k] + s[k + Q];
        X[Q] = s[0];
        return;
    }
    fft_math<Q>(X, s, tab_lds, ln);
    VAD_WAVE_STAMP(ln, 3 + 2 * V);
}

template <int Q>
__device__ __forceinline__ void fft_math(float (&X)[Q + 1], const float (&s)[2 * Q], const float *tab_lds, const Lane &ln) {
    constexpr int SL = 2 * Q;
    constexpr vadl::Tab tb = vadl::make_tab(8 * Q, Q);

    f32x2 z[Q];
    {   // window (same taps for every frame: the lane's slice always sits at 2Q g inside the frame)
        const f32x4 *w = reinterpret_cast<const f32x4 *>(tab_lds + tb.window + SL * ln.g);
#pragma unroll
        for (int k = 0; k < SL / 4; ++k) {
            const f32x4 wv = w[k];
            z[2 * k] = f32x2{s[4 * k], s[4 * k + 1]} * f32x2{wv[0], wv[1]};
            z[2 * k + 1] = f32x2{s[4 * k + 2], s[4 * k + 3]} * f32x2{wv[2], wv[3]};
        }
    }
    // radix-4 across the 4 lanes of a chunk.  Stage A pairs g <-> g^2 (lanes 32 apart), stage B pairs g <-> g^1
    // (lanes 16 apart).  Each pair splits the butterflies between its two lanes with gfx950's register-pair swaps
    // (v_permlane32_swap / v_permlane16_swap: VALU, no LDS round trip, no lane selects): the lower lane trades its
    // upper Q/2 values for the upper lane's lower Q/2, both lanes form sums and differences of what they now hold,
    // and a second trade leaves all sums in the lower lane and all differences (lower minus upper) in the upper one.
    const f32x4 *tw1 = reinterpret_cast<const f32x4 *>(tab_lds + tb.tw1 + ln.g * Q * 4);
#if VAD_XLANE_SWAP
    constexpr int H = Q / 2;
    // lane group 3 multiplies by -i between the two stages.  Its values are the stage-A differences formed in the odd lane
    // groups (1 keeps the lower half for 3, 3 its own upper half), so only those H differences are rotated, before they are
    // traded: x*rotA + swap(x)*rotB
    const f32x2 rotA = (ln.g & 1) ? f32x2{0.f, 0.f} : f32x2{1.f, 1.f};
    const f32x2 rotB = (ln.g & 1) ? f32x2{1.f, -1.f} : f32x2{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < H; ++q) trade32(z[q], z[q + H]);
#pragma unroll
    for (int q = 0; q < H; ++q) {
        const f32x2 u = z[q], v = z[q + H], d = u - v;
        z[q] = u + v;
        z[q + H] = __builtin_elementwise_fma(swap2(d), rotB, d * rotA);
    }
#pragma unroll
    for (int q = 0; q < H; ++q) trade32(z[q], z[q + H]);
#pragma unroll
    for (int q = 0; q < H; ++q) trade16(z[q], z[q + H]);
#pragma unroll
    for (int q = 0; q < H; ++q) {
        const f32x2 u = z[q], v = z[q + H];
        z[q] = u + v;
        z[q + H] = u - v;
    }
#pragma unroll
    for (int q = 0; q < H; ++q) trade16(z[q], z[q + H]);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const f32x2 x = z[q];
        const f32x4 t = tw1[q];                        // * W_4Q^(P[g] q): (-s, s, c, 0)
        z[q] = __builtin_elementwise_fma(swap2(x), f32x2{t[0], t[1]}, x * f32x2{t[2], t[2]});
    }
#else
    const f32x2 sA{ln.sgnA, ln.sgnA}, sB{ln.sgnB, ln.sgnB};
    // lane group 3 multiplies by -i between the two stages: x*rotA + swap(x)*rotB
    const f32x2 rotA = ln.g == 3 ? f32x2{0.f, 0.f} : f32x2{1.f, 1.f};
    const f32x2 rotB = ln.g == 3 ? f32x2{1.f, -1.f} : f32x2{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        f32x2 x = z[q];
        const f32x2 p{__shfl_xor(x.x, 32), __shfl_xor(x.y, 32)};
        x = __builtin_elementwise_fma(sA, x, p);       // g<2: own + partner ; g>=2: partner - own
        x = __builtin_elementwise_fma(swap2(x), rotB, x * rotA);
        const f32x2 r{__shfl_xor(x.x, 16), __shfl_xor(x.y, 16)};
        x = __builtin_elementwise_fma(sB, x, r);       // g even: own + partner ; g odd: partner - own
        const f32x4 t = tw1[q];                        // * W_4Q^(P[g] q): (-s, s, c, 0)
        z[q] = __builtin_elementwise_fma(swap2(x), f32x2{t[0], t[1]}, x * f32x2{t[2], t[2]});
    }
#endif
    fft_inlane<Q>(z);
    // real-FFT split: Y[k] = E + W_8Q^k O from Z[k] and conj Z[4Q - k]
    constexpr int LG = ilog2(Q);
    const f32x4 *tw2 = reinterpret_cast<const f32x4 *>(tab_lds + tb.tw2 + ln.g * Q * 4);
    // partner bin 8Q - k lives in the same lane for groups 0, 1 and in lane ^ 16 for groups 2, 3
    const int src_lane4 = (ln.g >= 2 ? (ln.lane ^ 16) : ln.lane) * 4;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const f32x2 u = z[bitrev(k, LG)];
        const int ks = bitrev((Q - k) % Q, LG), kr = bitrev(Q - 1 - k, LG);
        const f32x2 own = z[kr], alt = z[ks];
        const float xr = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane4, __float_as_int(own.x)));
        const float xi = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane4, __float_as_int(own.y)));
        const f32x2 p{ln.g == 0 ? alt.x : xr, ln.g == 0 ? alt.y : xi};
        const f32x2 av = __builtin_elementwise_fma(p, f32x2{1.0f, -1.0f}, u);     // Z + conj(Zp)
        const f32x2 ev = __builtin_elementwise_fma(p, f32x2{-1.0f, 1.0f}, u);     // Z - conj(Zp)
        // y = av + (-i ev) (c + i sn) = av + ev (sn - i c)
        const f32x4 t = tw2[k];                                                  // (c, -c, s, 0)
        const f32x2 y = __builtin_elementwise_fma(swap2(ev), f32x2{t[0], t[1]},
                                                  __builtin_elementwise_fma(ev, f32x2{t[2], t[2]}, av));
        const f32x2 yy = y * y;
        X[k] = 0.5f * __builtin_amdgcn_sqrtf(yy.x + yy.y);
    }
    X[Q] = ln.g == 0 ? fabsf(z[0].x - z[0].y) : 0.f;           // Nyquist: Re Z0 - Im Z0
}

// ---- non-finite input -------------------------------------------------------------------------------------------------------
// The reference propagates NaN: torch.relu is clamp_min(0), aten::lstm_cell is plain arithmetic, so ONE NaN or Inf sample (or a
// finite one so large that |Y|^2 overflows) makes the chunk's probability NaN and leaves NaN in the carried (h, c), i.e. in
// every later chunk of that stream until reset_states() (JIT!/vad/utils/model_utils.py:19-25, JIT!/torch/nn/modules/rnn.py:69;
// recorded from the reference: tests/golden/make_golden.py protocol "nonfinite").  The ReLUs here are v_max_f32, which returns
// the operand that is NOT NaN, so the frontends carry the fact explicitly: RULE -- a chunk with ANY non-finite STFT magnitude
// is NaN.  (A NaN / Inf sample makes every bin of its frame non-finite, in the DFT as in the FFT.  An Inf magnitude alone --
// overflow -- reaches NaN in the reference through Inf - Inf in the encoder sums, which formally depends on the weights' signs;
// with these weights it always does: tests/test_oracle.py::test_any_single_overflowing_bin_is_nan.)
//   poison_acc: p stays +0 while every value is finite, becomes NaN at the first one that is not (x * 0 is NaN for NaN and
// +-Inf, +-0 otherwise; +-0 + +0 = +0).  One VALU instruction per value, two independent chains.  E = x3 - x1 and F = x2 - x0 are
// non-finite whenever one of their frames is, so the kernels feed it the 2 Q transformed values instead of the 4 Q magnitudes.
//   poison_into: ORs p's bits into ONE B operand of the W_ih GEMM, behind the last ReLU.  +0 changes nothing (bit-identical
// results for clean chunks); NaN in B[k][j] makes column j -- this chunk, and only this chunk -- NaN in all 512 rows of gx.  From
// there the recurrence carries it in (h, c) by itself; its head uses relu_f (activations.hpp).
template <int Q>
__device__ __forceinline__ void poison_acc(float &p0, float &p1, const float (&X)[Q + 1]) {
#pragma unroll
    for (int k = 0; k < Q; k += 2) {
        p0 = fmaf(X[k], 0.f, p0);
        p1 = fmaf(X[k + 1], 0.f, p1);
    }
}
__device__ __forceinline__ float poison_nyq(float p0, float p1, float xn0, float xn1, float xn2, float xn3) {
    p0 = fmaf(xn0, 0.f, p0);
    p1 = fmaf(xn1, 0.f, p1);
    p0 = fmaf(xn2, 0.f, p0);
    p1 = fmaf(xn3, 0.f, p1);
    return p0 + p1;
}
__device__ __forceinline__ void poison_into(f32x4 &v, float p) {
    v[0] = __uint_as_float(__float_as_uint(v[0]) | __float_as_uint(p));
}

template <int Q, int V, typename PcmT, int DEC = 1>
__device__ __forceinline__ void fft_pass(float (&X)[Q + 1], const FrontArgs &a, const float *tab_lds, const Lane &ln) {
    fft_frame<Q, PcmT, DEC>(X, V, a, tab_lds, ln);
}

}  // namespace
}  // namespace vad
