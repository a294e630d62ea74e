// kernel_ingest.hip -- ragged recordings from PINNED host memory straight into a zero-padded [n][width] batch in HBM.
//
// A corpus run (BASELINE config 4; the reference fans one Python process out per file, examples/parallel_example.ipynb
// cells 5, 7, and pads each file's last chunk with zeros, src/silero_vad/utils_vad.py:326-327) is bounded by the host
// link, and the host side of the staged path (vad_stage_rows: every sample copied once more, pageable RAM -> pinned
// staging, by CPU threads the container has few of) was its second-largest cost in round 2.  When the recordings already
// sit in page-locked memory -- a decoder that writes into hipHostMalloc'ed / hipHostRegister'ed buffers -- nothing needs
// to touch them on the CPU: this kernel reads them over PCIe (host memory is mapped into the GPU's address space) and
// writes the device batch, padding included, in ONE launch for any number of rows.
//
//   * a persistent grid of 256 workgroups walks the (row, 16 KiB segment) pairs: 256 lanes x 16 B per request, MBs of reads
//     in flight -- far more than PCIe's bandwidth-delay product (~100 KB) -- from at most one small wave per SIMD (no LDS, a
//     few VGPRs: it fits beside the frontend's two 243-register waves and does not take a workgroup slot from them);
//   * rows whose source address is 16-byte aligned move as 16-byte vectors; others (a view that starts at an odd sample)
//     fall back to element-wise loads for that row (wave-uniform choice) -- correct for any alignment, fast for the usual;
//   * the row table (pointer, length) is itself read from pinned memory, so the host only fills a small table and launches.
#include <hip/hip_runtime.h>

#include "device_api.hpp"

namespace vad {
namespace {

constexpr int kSegBytes = 16384;

__global__ void __launch_bounds__(256) gather_rows_kernel(const RowDesc *rows, long n, long width_bytes, int esz, uint8_t *dst,
                                                          long segs_per_row) {
  const long items = n * segs_per_row;
  for (long item = blockIdx.x; item < items; item += gridDim.x) {     // persistent: a bounded footprint beside the compute kernels
    const long row = item / segs_per_row, seg = item % segs_per_row;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(rows[row].ptr);
    const long live = rows[row].len * esz;                         // bytes that exist; the rest of the row is zero
    uint8_t *d = dst + row * width_bytes;
    const long lo = seg * kSegBytes, hi = lo + kSegBytes < width_bytes ? lo + kSegBytes : width_bytes;
    using u32x4 = unsigned __attribute__((ext_vector_type(4)));
    if ((((size_t)src) & 15) == 0) {                               // wave-uniform
        // 16-byte vectors; width_bytes and lo are multiples of 16 (the engine checks), `live` need not be
        for (long o = lo + threadIdx.x * 16L; o < hi; o += 256 * 16L) {
            u32x4 v{0u, 0u, 0u, 0u};
            if (o + 16 <= live) {
                v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + o));
            } else if (o < live) {                                 // the vector that straddles the end of the recording
                unsigned char tmp[16] = {0};
                for (int k = 0; k < (int)(live - o); ++k) tmp[k] = src[o + k];
                v = *reinterpret_cast<u32x4 *>(tmp);
            }
            *reinterpret_cast<u32x4 *>(d + o) = v;
        }
    } else if (esz == 4) {
        for (long o = lo + threadIdx.x * 4L; o < hi; o += 256 * 4L)
            *reinterpret_cast<unsigned *>(d + o) = o < live ? *reinterpret_cast<const unsigned *>(src + o) : 0u;
    } else {
        for (long o = lo + threadIdx.x * 2L; o < hi; o += 256 * 2L)
            *reinterpret_cast<unsigned short *>(d + o) = o < live ? *reinterpret_cast<const unsigned short *>(src + o) : (unsigned short)0;
    }
  }
}

}  // namespace

hipError_t launch_gather_rows(const RowDesc *rows, long n, long width, int esz, void *dst, hipStream_t s) {
    if (n <= 0 || width <= 0) return hipSuccess;
    const long wb = width * esz, segs = (wb + kSegBytes - 1) / kSegBytes;
    const long items = n * segs;
    if (items > 0x7fffffffL) return hipErrorInvalidValue;
    // 256 workgroups x 256 lanes x 16 B x the loads the compiler keeps in flight: MBs outstanding, one wave per SIMD at most
    const unsigned grid = (unsigned)(items < 256 ? items : 256);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, s, rows, n, wb, esz, static_cast<uint8_t *>(dst), segs);
    return hipGetLastError();
}

}  // namespace vad
