// kernel_scan.hip -- the hysteresis segmenter on the device: one lane scans one stream's probabilities
// (reference: the scan + padding pass of get_speech_timestamps, src/silero_vad/utils_vad.py:338-440; the code
// is scanner.hpp, the same source the host entry points in segmenter.cpp compile).
//
// Why on the device: after vad_forward_audio the probabilities already sit in HBM.  Scanning them there means
// only the segment lists (16 B per segment, a handful per minute of audio) cross PCIe instead of every
// probability (4 B per 32 ms chunk), and the host does no per-chunk work at all -- on a corpus run the host-side
// scan and the D2H copy of probs[B][T] are otherwise the largest non-GPU costs (bench.py --config corpus).
// Streams are independent and the scan is sequential in t, so the parallel axis is the stream: lane i walks row
// i.  Rows are read 16 B at a time; the state is O(1) registers (scanner.hpp).
#include <hip/hip_runtime.h>

#include "device_api.hpp"
#include "scanner.hpp"

namespace vad {
namespace {

__global__ void __launch_bounds__(64) scan_kernel(const float *probs, long ldp, const long *row_off, long n_streams, const long *n_chunks,
                                                  long n_chunks_all, const long *audio_len, vad_segment_params p,
                                                  vad_segment *out, long cap, long *counts) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams) return;
    const long n = n_chunks ? n_chunks[i] : n_chunks_all;
    Scanner sc(p, audio_len[i], out + i * cap, cap);
    const float *row = probs + (row_off ? row_off[i] : i * ldp);
    long t = 0;
    if ((((size_t)row) & 15) == 0) {
        for (; t + 4 <= n; t += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(row + t);
            sc.feed(t, v.x);
            sc.feed(t + 1, v.y);
            sc.feed(t + 2, v.z);
            sc.feed(t + 3, v.w);
        }
    }
    for (; t < n; ++t) sc.feed(t, row[t]);
    counts[i] = sc.finish();
}

}  // namespace

hipError_t launch_scan(const float *probs, long ldp, const long *row_off, long n_streams, const long *n_chunks, long n_chunks_all,
                       const long *audio_len, const vad_segment_params &p, vad_segment *out, long cap, long *counts,
                       hipStream_t s) {
    if (n_streams <= 0) return hipSuccess;
    hipLaunchKernelGGL(scan_kernel, dim3((unsigned)((n_streams + 63) / 64)), dim3(64), 0, s, probs, ldp, row_off, n_streams,
                       n_chunks, n_chunks_all, audio_len, p, out, cap, counts);
    return hipGetLastError();
}

}  // namespace vad
