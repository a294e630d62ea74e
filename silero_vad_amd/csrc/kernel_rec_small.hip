// kernel_rec_small.hip -- the recurrence (same function, same bits as kernel_rec.hip) for up to 1 024 streams: the reference's own
// workflow -- one file at a time through get_speech_timestamps (src/silero_vad/utils_vad.py:324-336) -- is B = 1 with thousands of
// time steps, and rec_kernel's step costs 4.4 us whatever the batch: its 1 024 MFMAs per step multiply W_hh by a 16-column tile of
// which a single file fills one column.  Here W_hh * h is what it is for one stream, a matrix-vector product, on the VALU:
//   * 512 threads, thread (q, n) owns gate row 128 q + n: its 128 weights stay in 128 VGPRs for the whole launch, in the ORDER in
//     which rec_kernel's MFMA chain adds them -- v_mfma_f32_16x16x4_f32 is exactly fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))
//     (tools/ubench/mfma_order.hip: 1 048 576 of 1 048 576 outputs), k-groups ascending -- so that the same fmaf chain gives the SAME
//     BITS: position 16 kg + 4 r + g of the chain is hidden unit 16 kg + 4 g + r (layout.hpp "whh_rows");
//   * h_{t-1} lies in LDS in chain order and is read as broadcast 16-byte vectors; per step and stream 128 fmaf per thread;
//   * the four gates of a unit meet in LDS; 128 threads per stream do the pointwise update, one unit each, with the same formulas
//     (activations.hpp); 32 threads per stream -- thread (w, g) holds units 16 w + 4 g + r, r < 4, exactly what a lane of rec_kernel
//     holds -- reduce the head's dot product in rec_kernel's order: fmaf chain over r, ((p0 + p1) + (p2 + p3)) over g, then w = 0..7 onto
//     the bias -- one step behind, beside the next step's products (the head is not on the recurrence's critical path).
//   h reaches the fmas through DPP row broadcasts from 8 registers per lane, not through 32 broadcast LDS reads per thread and step: those
//   kept the LDS pipe busy for 2 048 of a step's ~2 300 cycles (8 waves x 32 x 8), twice what the VALU needs.  (Two rows per thread with
//   packed fmas halve the waves instead but need 256 weight registers a thread -- half of them AGPRs the VALU cannot read directly:
//   measured 1.6 x slower.)
// Every workgroup carries 1, 2 or 4 streams -- as few as put B streams on the chip's 256 CUs at once -- and a step costs 1.04 (one file;
// 1.37 with 256 of them) / 1.70 / 2.62 us against rec_kernel's 4.34 whatever the batch (tools/rec_small_time.py): a single 60 s file
// 8.1 -> 1.94 ms.
// The engine takes this kernel for B <= 1 024 (option "rec_form" = auto | mfma); above that rec_kernel's 16 streams per CU win.
// (reference: aten::lstm_cell, JIT!/torch/nn/modules/rnn.py:69, gate order i,f,g,o; JIT!/vad/model/vad_annotator.py:170-187; head
//  JIT!/torch/nn/modules/container/___torch_mangle_7.py:10-19.)
#include <hip/hip_runtime.h>

#include "activations.hpp"
#include "device_api.hpp"
#include "layout.hpp"

namespace vad {
namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));

// acc = fma(h of lane I of this 16-lane row, w, acc): v_fmac_f32 with its first operand through DPP row_newbcast (gfx90a+)
template <int I>
__device__ __forceinline__ void fmac_bcast(float &acc, float h, float w) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w), "n"(I));
}
// the 16 chain positions 16 v .. 16 v + 15 of every stream of the workgroup, the streams' chains interleaved instruction by instruction
template <int NB, int I = 0>
__device__ __forceinline__ void fmac_row(float (&acc)[NB], const float (&hreg)[NB][8], const float (&W)[128], const int v) {
    if constexpr (I < 16) {
        // (a DPP operand must not have been written by the VALU in the two instructions before: h comes straight from LDS loads, but the
        //  hazard recogniser does not look into inline assembly, so the distance is made explicit once per row)
        if constexpr (I == 0) asm volatile("s_nop 1");
#pragma unroll
        for (int j = 0; j < NB; ++j) fmac_bcast<I>(acc[j], hreg[j][v], W[16 * v + I]);
        fmac_row<NB, I + 1>(acc, hreg, W, v);
    }
}

template <int NB, int NTAB_WOUT, int NTAB_BOUT>
__global__ void __launch_bounds__(512, 1) rec_small_kernel(const RecArgs a) {
    __shared__ __attribute__((aligned(16))) float hs[2][NB][128];      // h, chain order: [16 kg + 4 r + g] = unit 16 kg + 4 g + r
    __shared__ __attribute__((aligned(16))) float gs[NB][4][128];      // gate pre-activations [stream][gate][unit]

    const int tid = threadIdx.x, q = tid >> 7, n = tid & 127;
    const long b0 = (long)blockIdx.x * NB;                 // first stream of this workgroup; NB divides 16: all of them in one gx tile
    const int nb = (int)(a.B - b0 < NB ? a.B - b0 : NB);   // streams that exist (>= 1)
    float W[128];
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.whh) + (size_t)(128 * q + n) * 32;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const f32x4 v = src[m];
#pragma unroll
            for (int e = 0; e < 4; ++e) W[4 * m + e] = v[e];
        }
    }
    // where the frontend put this row's gate pre-activation: D-fragment order, tile 0 (layout.hpp, kernel_rec.hip)
    const int w = n >> 4, g = (n >> 2) & 3, r = n & 3;
    const float *gxp = a.gx + ((size_t)(b0 >> 4) * a.nt * 32) * 256 + ((size_t)(8 * q + w) * 64 + g * 16 + (b0 & 15)) * 4 + r;   // + t * 32 * 256 + 4 j

    // pointwise role: thread (pj, pu) holds unit pu of stream pj -- one unit per thread (128 threads per stream): the five activations of a
    // unit are ~100 dependent VALU instructions, and this phase sits between two barriers on the critical path of every step
    const bool pt = tid < 128 * NB;
    const int pj = tid >> 7, pu = tid & 127, ppos = (pu & ~15) + 4 * (pu & 3) + ((pu >> 2) & 3);      // chain position of unit pu
    const long pb = b0 + (pj < nb ? pj : nb - 1);          // this thread's stream (clamped: lanes of missing streams compute, never store)
    const bool pvalid = pt && pj < nb && (a.present == nullptr || a.present[pb] != 0);   // (an absent row keeps its state: vad_step_present)
    float h = 0.f, c = 0.f;
    if (pt) {
        h = a.state[(size_t)pb * 128 + pu];
        c = a.state[((size_t)a.B + pb) * 128 + pu];
        hs[0][pj][ppos] = h;
    }
    // head role: thread (hj, hw, hg) holds units 16 hw + 4 hg + rr, rr < 4, of stream hj -- what a lane of rec_kernel holds -- and forms the
    // head's dot product in rec_kernel's order from the h the pointwise threads left in LDS.  Off the critical path: the head of step t
    // is formed at the start of step t + 1, beside the other waves' matrix-vector products.
    const bool ht = tid < 32 * NB;
    const int hj = tid >> 5, hw = (tid & 31) >> 2, hg = tid & 3;
    const long hb = b0 + (hj < nb ? hj : nb - 1);
    const bool hvalid = ht && hj < nb && hw == 0 && hg == 0 && (a.present == nullptr || a.present[hb] != 0);
    f32x4 wo = {0.f, 0.f, 0.f, 0.f};
    float bo = 0.f;
    if (ht) {
        wo = 0.5f * *reinterpret_cast<const f32x4 *>(a.tables + NTAB_WOUT + 16 * hw + 4 * hg);   // halved: relu2_f (activations.hpp)
        bo = a.tables[NTAB_BOUT];
    }
    auto head = [&](int buf, long t) {
        if (!ht) return;
        float part = 0.f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) part = fmaf(wo[rr], relu2_f(hs[buf][hj][16 * hw + 4 * rr + hg]), part);
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        float p = bo;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) p += __shfl(part, (tid & 32) + 4 * ww);
        if (hvalid) a.probs[(size_t)hb * a.ldp + a.t0 + t] = sigmoid_f(p);
    };
    float gnext[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) gnext[j] = gxp[4 * (j < nb ? j : nb - 1)];
    __syncthreads();

    for (long t = 0; t < a.nt; ++t) {
        const int cur = (int)(t & 1);
        if (t > 0) head(cur, t - 1);                       // hs[cur] = h_{t-1}: the output of the step before
        float acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = gnext[j];
        if (t + 1 < a.nt) {
#pragma unroll
            for (int j = 0; j < NB; ++j) gnext[j] = gxp[(size_t)(t + 1) * 32 * 256 + 4 * (j < nb ? j : nb - 1)];
        }
        // h as 8 registers per stream: lane l of every 16-lane row holds h[16 v + l % 16] in register v, and the fma takes its h operand
        // through the data-parallel-primitive path -- row_newbcast:i hands lane i of the row to all 16 lanes -- so the chain
        // acc = fma(h[k], W[k], acc), k ascending, costs 128 VALU instructions and 8 four-byte LDS reads per thread, not 32 broadcast
        // 16-byte reads: those were 8 waves x 32 x 8 cycles = 2 048 cycles of LDS pipe per step, twice the VALU's 1 024
        float hreg[NB][8];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int v = 0; v < 8; ++v) hreg[j][v] = hs[cur][j][16 * v + (tid & 15)];
#pragma unroll
        for (int v = 0; v < 8; ++v) fmac_row<NB>(acc, hreg, W, v);
#pragma unroll
        for (int j = 0; j < NB; ++j) gs[j][q][n] = acc[j];
        __syncthreads();
        if (pt) {
            const float ig = sigmoid_f(gs[pj][0][pu]), fg = sigmoid_f(gs[pj][1][pu]), gg = tanh_f(gs[pj][2][pu]);
            c = fmaf(fg, c, ig * gg);
            h = sigmoid_f(gs[pj][3][pu]) * tanh_f(c);
            hs[cur ^ 1][pj][ppos] = h;
        }
        __syncthreads();
    }
    head((int)(a.nt & 1), a.nt - 1);
    if (pvalid) {
        a.state[(size_t)pb * 128 + pu] = h;
        a.state[((size_t)a.B + pb) * 128 + pu] = c;
    }
}

template <int NB>
hipError_t launch_nb(int sr, const RecArgs &a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + NB - 1) / NB);
    if (sr == 16000) hipLaunchKernelGGL((rec_small_kernel<NB, vadl::tab16.w_out, vadl::tab16.b_out>), dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((rec_small_kernel<NB, vadl::tab8.w_out, vadl::tab8.b_out>), dim3(grid), dim3(512), 0, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_rec_small(int sr, const RecArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    if (a.B > kRecSmallMaxB) return hipErrorInvalidValue;
    // one workgroup per CU (512 threads, 128 weight registers each): as few streams per workgroup as fill the chip's 256 CUs once
    return a.B <= 256 ? launch_nb<1>(sr, a, s) : a.B <= 512 ? launch_nb<2>(sr, a, s) : launch_nb<4>(sr, a, s);
}

}  // namespace vad
