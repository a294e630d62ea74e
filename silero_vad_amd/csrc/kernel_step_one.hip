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

// acc += sum over NKG k-groups of W[row][(kg, ks, g)] * x[16 kg + 4 ks + g], in chain order.  `w` points at the row's 16 bytes of
// block (kg = 0, lane group 0): lane group g is 16 lanes = 64 floats further, k-group kg is `kg_stride` floats further.
template <int NKG>
__device__ __forceinline__ float chain(float acc, const float *w, long kg_stride, const float *x) {
    f32x4 a[NKG][4];
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg)
#pragma unroll
        for (int g = 0; g < 4; ++g) a[kg][g] = *reinterpret_cast<const f32x4 *>(w + kg * kg_stride + g * 64);
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc = fmaf(a[kg][g][ks], x[16 * kg + 4 * ks + g], acc);
    return acc;
}
// two chains at once (independent accumulators: twice the instruction-level parallelism), each over its own input
template <int NKG>
__device__ __forceinline__ void chain2(float &acc0, float &acc1, const float *w0, const float *w1, long kg_stride, const float *x0,
                                       const float *x1) {
    f32x4 a[NKG][4], b[NKG][4];
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a[kg][g] = *reinterpret_cast<const f32x4 *>(w0 + kg * kg_stride + g * 64);
            b[kg][g] = *reinterpret_cast<const f32x4 *>(w1 + kg * kg_stride + g * 64);
        }
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc0 = fmaf(a[kg][g][ks], x0[16 * kg + 4 * ks + g], acc0);
                acc1 = fmaf(b[kg][g][ks], x1[16 * kg + 4 * ks + g], acc1);
            }
}

// A chain's weights requested AHEAD of their use: the vectors of NKG k-groups of one row.  load() only issues the requests (pinned in
// place by a scheduling barrier); run() is chain<NKG> over them.  Every layer's weights depend on the thread, not on the data, so each
// layer is requested while the layer before it computes -- a step is ten dependent phases, and what each of them would otherwise wait
// for first is an L2 round trip.
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
// bring-up (option trace_ptr, tools/b1_phase_trace.py): thread 0 of workgroup 0 leaves the shader clock at the phase boundaries
#define VAD_STAMP(k) do { if (a.trace != nullptr && threadIdx.x == 0 && blockIdx.x == 0) a.trace[(k)] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ int chain_pos(int ch) { return (ch & ~15) + 4 * (ch & 3) + ((ch >> 2) & 3); }   // channel -> position

template <int Q, typename PcmT, bool CELL>
__global__ void __launch_bounds__(256) step_one_kernel(const FrontArgs a, const CellArgs cell) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int TABF = (tb.total + 3) / 4 * 4;
    constexpr int RB = w_rb(Q), KG0 = Q / 4, T0 = w4_tail0(Q), NQ = 4 * Q;     // NQ: k values of encoder 0 without the Nyquist bin
    __shared__ __attribute__((aligned(16))) float tab[TABF];
    __shared__ __attribute__((aligned(16))) float mag[4][NQ];       // [frame][(s, g)]: |Y| of k = (s, g), chain order
    __shared__ float nyq[4];
    __shared__ __attribute__((aligned(16))) float tin[6][NQ];       // encoder 0's six transformed inputs (U1, U2, U3, U4, U0, U5 order)
    __shared__ __attribute__((aligned(16))) float my[4][128];       // m1..m4 of every row
    __shared__ __attribute__((aligned(16))) float e0c[4][128];      // encoder 0 output per frame, chain order
    __shared__ __attribute__((aligned(16))) float e1c[2][64];
    __shared__ __attribute__((aligned(16))) float e2c[64];
    __shared__ __attribute__((aligned(16))) float fec[128];         // encoder 3 output ("feat"), chain order, poison applied
    __shared__ __attribute__((aligned(16))) float hc[128];          // h_{t-1}, chain order
    __shared__ __attribute__((aligned(16))) float gates[512];
    __shared__ __attribute__((aligned(16))) float hnew[128];
    __shared__ float pb[8], pg[32];
    __shared__ float poison_g[4];
    __shared__ ExactWs<Q> ws;
    __shared__ RefNet net;
    __shared__ int route;                                           // 0: the chains; 1: exact_gx; 2: the net's constant for a chunk of zeros

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

    // this thread's rows of encoder 0 (row r, the two matrices of its half first) -- requested before anything else
    VAD_STAMP(0);
    const int r = tid & 127, half = tid >> 7, i16 = r & 15;
    const int part = r / (16 * RB), rbl = (r % (16 * RB)) / 16;
    const int u0 = part == 0 ? w4_part0(0, Q) : part == 1 ? w4_part0(1, Q) : part == 2 ? w4_part0(2, Q) : w4_part0(3, Q);
    const float *wb = a.wfront + (size_t)u0 * 4096 + rbl * 256 + i16 * 4;
    WSet<KG0> W0a, W0b;
    // ---- STFT: wave v, frame v (fft_wave.hpp; the context for the next call is written by load_slice) -------------------------------
    {
        float pcm_s[2 * Q], Xm[Q + 1];
        load_slice<Q, PcmT, 1>(pcm_s, a, ln, w);                    // (the chunk first: the FFT waits for it and for the tables)
        VAD_PIN();
        {   // tables -> LDS
            constexpr int NV = tb.total / 4;
            const f32x4 *src = reinterpret_cast<const f32x4 *>(a.tables);
            for (int i = tid; i < NV; i += 256) reinterpret_cast<f32x4 *>(tab)[i] = src[i];
        }
        if (CELL && tid < 128) hc[chain_pos(tid)] = cell.state[(size_t)b * 128 + tid];
        W0a.load(wb + (size_t)(2 * half) * 4096, RB * 256);
        W0b.load(wb + (size_t)(2 * half + 1) * 4096, RB * 256);
        VAD_PIN();
        lds_barrier();
        VAD_STAMP(1);
        fft_math<Q>(Xm, pcm_s, tab, ln);
        VAD_STAMP(2);
        if (ln.j == 0) {
#pragma unroll
            for (int s = 0; s < Q; ++s) mag[w][4 * s + ln.g] = Xm[s];
            if (ln.g == 0) nyq[w] = Xm[Q];
        }
    }
    lds_barrier();
    if (tid == 0) {
        int r = 0;
        if (!VAD_NO_EXACT && a.exact_net != nullptr) {              // exact_front.hpp: silent frames beside frames that are not
            bool any = false, all = true;
            for (int f = 0; f < 4; ++f) {
                const bool z = mag[f][0] == 0.f && mag[f][1] == 0.f && mag[f][2] == 0.f && mag[f][3] == 0.f;
                any = any || z;
                all = all && z;
            }
            r = all ? (a.gx_silent != nullptr ? 2 : 0) : any ? 1 : 0;
            if (r == 1) net = *a.exact_net;
        }
        route = r;
    }
    lds_barrier();
    const int how = route;
    VAD_STAMP(3);
    if (how == 1) {
        exact_gx<Q, PcmT, 1>(a, net, b, ln.t, ws);
        for (int r = tid; r < 512; r += 256) gates[r] = ws.gx[r];
    } else if (how == 2) {
        for (int r = tid; r < 512; r += 256) gates[r] = a.gx_silent[r];
    } else {
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
        if (tid < 4) {                                              // non-finite input (fft_wave.hpp): lane group g's own poison value
            float p0 = 0.f, p1 = 0.f;
            for (int s = 0; s < Q; s += 2) {                        // E first, then F, two chains -- as poison_acc does
                p0 = fmaf(mag[3][4 * s + tid] - mag[1][4 * s + tid], 0.f, p0);
                p1 = fmaf(mag[3][4 * (s + 1) + tid] - mag[1][4 * (s + 1) + tid], 0.f, p1);
            }
            for (int s = 0; s < Q; s += 2) {
                p0 = fmaf(mag[2][4 * s + tid] - mag[0][4 * s + tid], 0.f, p0);
                p1 = fmaf(mag[2][4 * (s + 1) + tid] - mag[0][4 * (s + 1) + tid], 0.f, p1);
            }
            poison_g[tid] = poison_nyq(p0, p1, nyq[0], nyq[1], nyq[2], nyq[3]);
        }
        lds_barrier();
        {
            WSet<KG0> W0c;                                          // the fifth / sixth matrix of this half: requested before the first four run
            W0c.load(wb + (size_t)(4 + half) * 4096, RB * 256);
            VAD_PIN();
            // m1, m2 (half 0) | m3, m4 (half 1): two independent chains per thread
            float ma = 0.f, mb = 0.f;
            run2<KG0>(W0a, W0b, ma, mb, tin[2 * half], tin[2 * half + 1]);
            my[2 * half][r] = ma;
            my[2 * half + 1][r] = mb;
            lds_barrier();
            VAD_STAMP(4);
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
        lds_barrier();
        VAD_STAMP(5);
        // ---- encoder 1: 64 rows x two outputs, the 40-block program of front_common.hpp e1_blk ----------------------------------------
        if (tid < 128) {
            const int ro = tid & 63, o = tid >> 6, wr = ro >> 4, i = ro & 15;
            float z = 0.f;
            static_for<0, 40>([&](auto ic) VAD_INLINE {
                constexpr E1Blk eb = e1_blk(Q, decltype(ic)::value);
                if (eb.acc == o)
                    z = chain<1>(z, a.wfront + ((size_t)eb.unit * 16 + eb.kg * 4 + wr) * 256 + i * 4, 0, e0c[eb.frame] + 16 * eb.rbg);
            });
            e1c[o][chain_pos(ro)] = fmaxf(z + tab[tb.b_e1 + ro], 0.f);
        }
        lds_barrier();
        VAD_STAMP(6);
        // (the 512 gate rows' W_ih weights -- rows tid and tid + 256 -- are requested here, two layers ahead)
        WSet<8> Wi0, Wi1;
        {
            const int q0 = tid >> 7, m = (tid >> 4) & 7, i = tid & 15;
            Wi0.load(a.wfront + (size_t)(T0 + 4 + 4 * q0) * 4096 + m * 256 + i * 4, 8 * 256);
            Wi1.load(a.wfront + (size_t)(T0 + 4 + 4 * (q0 + 2)) * 4096 + m * 256 + i * 4, 8 * 256);
            VAD_PIN();
        }
        // ---- encoder 2 (taps 1, 2 see encoder-1 outputs 0, 1) -----------------------------------------------------------------------------
        if (tid < 64) {
            const int wr = tid >> 4, i = tid & 15;
            float v = tab[tb.b_e2 + tid];
#pragma unroll
            for (int bi = 0; bi < 8; ++bi)
                v = chain<1>(v, a.wfront + ((size_t)(T0 + bi / 4) * 16 + (bi % 4) * 4 + wr) * 256 + i * 4, 0, e1c[bi / 4] + 16 * (bi % 4));
            e2c[chain_pos(tid)] = fmaxf(v, 0.f);
        }
        lds_barrier();
        VAD_STAMP(7);
        // ---- encoder 3 (centre tap) -> feat, with the non-finite poison in the k = (0, 0, g) inputs -------------------------------------
        if (tid < 128) {
            const int wr = tid >> 5, m = (tid >> 4) & 1, i = tid & 15;
            float f = tab[tb.b_e3 + tid];
            f = chain<4>(f, a.wfront + (size_t)(T0 + 2) * 4096 + (2 * wr + m) * 256 + i * 4, 8 * 256, e2c);
            f = fmaxf(f, 0.f);
            const int pos = chain_pos(tid);
            if (pos < 4) f = __uint_as_float(__float_as_uint(f) | __float_as_uint(poison_g[pos]));     // positions 0..3 = (kg 0, ks 0, g)
            fec[pos] = f;
        }
        lds_barrier();
        VAD_STAMP(8);
        // ---- W_ih: gate rows tid and tid + 256 ---------------------------------------------------------------------------------------------
        {
            float g0 = tab[tb.b_g + tid], g1 = tab[tb.b_g + tid + 256];
            run2<8>(Wi0, Wi1, g0, g1, fec, fec);
            gates[tid] = g0;
            gates[tid + 256] = g1;
        }
    }
    lds_barrier();
    VAD_STAMP(9);
    if constexpr (!CELL) {
        // gx[tile][row block 32][lane 64][4] (layout.hpp): row 16 mb + 4 g + r of column j at lane 16 g + j, element r
        float *gxt = a.gx + (size_t)(b >> 4) * 32 * 256;
        const int j = (int)(b & 15);
        for (int r = tid; r < 512; r += 256) gxt[((size_t)(r >> 4) * 64 + ((r >> 2) & 3) * 16 + j) * 4 + (r & 3)] = gates[r];
    } else {
        // ---- the LSTM cell: the gate chains continue into W_hh h_{t-1}; pointwise; head (kernel_front_lat.hip's order) ---------------------
        {
            float g0 = gates[tid], g1 = gates[tid + 256];
            const int q0 = tid >> 7, m = (tid >> 4) & 7, i = tid & 15;
            chain2<8>(g0, g1, cell.whh_lat + ((size_t)q0 * 64 + m) * 256 + i * 4, cell.whh_lat + ((size_t)(q0 + 2) * 64 + m) * 256 + i * 4,
                      8 * 256, hc, hc);
            lds_barrier();
            gates[tid] = g0;
            gates[tid + 256] = g1;
        }
        lds_barrier();
        VAD_STAMP(10);
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
        VAD_STAMP(11);
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
