// kernel_present.hip -- live streams that have no chunk this tick (vad_step_present, include/silero_vad_hip.h).
//
// In the reference a stream's (h, c) and context change only when THAT stream's caller calls the model: VADIterator.__call__ runs
// one model call per chunk that arrived (src/silero_vad/utils_vad.py:507-549), the model object replaces _state / _context inside
// that call and nowhere else (JIT!/vad/model/vad_annotator.py:72,86-87), and the native loop does the same around session.run
// (examples/cpp/silero-vad-onnx.cpp:335-390).  A lock-step batch of thousands of live streams has rows whose packet is late: those
// rows must come out of the tick exactly as they went in.
//
// Division of labour.  The step kernels (kernel_front_lat.hip fused step, kernel_rec*.hip) take `present[B]` and simply do not write
// an absent row's (h, c) or probability -- one byte load per lane, nothing when the pointer is null.  The frontends know nothing about
// presence: they write the next context of EVERY row into the second context buffer (ctx_out is never the buffer they read), so an
// absent row's ctx_out holds the tail of whatever its PCM slot held.  This pass runs behind them on the same stream and finishes the
// absent rows: ctx_out[b] = ctx_in[b] (bit copy), probs[b] = VAD_PROB_ABSENT.  HBM-bound byte work: B * C * 4 bytes at most (2 MiB for
// 8 192 streams), one 16-byte vector per lane, rows of present streams are not touched.
//
// expand_rows (compact ticks of the pump, pump.hip): the delivering streams' chunks crossed the link back to back; this pass copies row
// pos[b] of that block to row b of the batch buffer the step kernels read -- HBM-bound byte work, 2 x B x N x 2 bytes at most.
#include <hip/hip_runtime.h>

#include "device_api.hpp"

namespace vad {
namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));

// one thread per 16 bytes of context: C / 4 threads per row (16 or 8), 256 threads per workgroup
__global__ void __launch_bounds__(256) carry_absent_kernel(const uint8_t *__restrict__ present, const float *__restrict__ ctx_in,
                                                           float *__restrict__ ctx_out, int vec_per_row, float *__restrict__ probs, long ldp, int B) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long b = i / vec_per_row;
    const int v = (int)(i - b * vec_per_row);
    if (b >= B || present[b]) return;
    const size_t off = (size_t)b * vec_per_row + v;
    if (ctx_out != nullptr) reinterpret_cast<f32x4 *>(ctx_out)[off] = reinterpret_cast<const f32x4 *>(ctx_in)[off];     // (null: the step kernel left the absent rows' context where it was)
    if (v == 0) probs[(size_t)b * ldp] = VAD_PROB_ABSENT;
}

// one thread per 16 bytes of a row: dst[b] <- src[pos[b]] for the rows that deliver (a compact tick of the pump: the link carried only
// those rows, back to back; the step kernels read row b of the batch buffer)
__global__ void __launch_bounds__(256) expand_rows_kernel(const uint8_t *__restrict__ present, const int32_t *__restrict__ pos,
                                                          const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, int vec_per_row, int B) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long b = i / vec_per_row;
    const int v = (int)(i - b * vec_per_row);
    if (b >= B || !present[b]) return;
    dst[(size_t)b * vec_per_row + v] = __builtin_nontemporal_load(src + (size_t)pos[b] * vec_per_row + v);
}

}  // namespace

hipError_t launch_expand_rows(const uint8_t *present, const int32_t *pos, const uint8_t *src, void *dst, long row_bytes, int B, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (!present || !pos || row_bytes <= 0 || row_bytes % 16) return hipErrorInvalidValue;
    const int vec = (int)(row_bytes / 16);
    const long threads = (long)B * vec;
    hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, present, pos,
                       reinterpret_cast<const f32x4 *>(src), static_cast<f32x4 *>(dst), vec, B);
    return hipGetLastError();
}

hipError_t launch_carry_absent(const uint8_t *present, const float *ctx_in, float *ctx_out, int C, float *probs, long ldp, int B,
                               hipStream_t s) {
    if (B <= 0 || !present) return hipSuccess;
    const int vec = C / 4;
    const long threads = (long)B * vec;
    hipLaunchKernelGGL(carry_absent_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, present, ctx_in, ctx_out, vec, probs, ldp, B);
    return hipGetLastError();
}

}  // namespace vad
