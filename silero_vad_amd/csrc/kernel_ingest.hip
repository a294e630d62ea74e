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
//   * a persistent grid of VAD_GATHER_WAVES (96) ONE-WAVE workgroups walks the (row, 8 KiB segment) pairs, 8 x 16 B per lane in
//     flight: enough outstanding reads for the link, on 9 % of the chip's SIMDs (see the kernel's comment for why so few);
//   * rows whose source address is 16-byte aligned move as 16-byte vectors; others (a view that starts at an odd sample)
//     fall back to element-wise loads for that row (wave-uniform choice) -- correct for any alignment, fast for the usual;
//   * the row table (pointer, length) is itself read from pinned memory, so the host only fills a small table and launches.
#include <hip/hip_runtime.h>

#include "device_api.hpp"

namespace vad {
namespace {

constexpr int kSegBytes = 8192;        // one wave-iteration: 64 lanes x 16 B x 8 loads in flight
#ifndef VAD_GATHER_WAVES
#define VAD_GATHER_WAVES 96
#endif
constexpr int kGatherWaves = VAD_GATHER_WAVES;   // one-wave workgroups: the whole kernel occupies this many of the chip's 1024 SIMDs

// One wave per workgroup, a persistent grid of kGatherWaves: each iteration a wave moves one 8 KiB segment of one row with 8
// independent 16-byte loads per lane in flight (96 waves x 8 KiB = 768 KiB outstanding against PCIe's ~100-150 KB
// bandwidth-delay product; 48 waves reach the same rate alone and lose more to the compute kernels beside them).  The footprint is deliberate: a first version with 256 four-wave workgroups reached the same
// 53 GB/s but sat on every SIMD of the chip, where its registers kept the frontend (two 243-VGPR waves per SIMD) from
// placing its second wave: the compute kernels beside it ran 4x slower (tools/ingest_diag.py).  96 single waves touch 9 % of
// the SIMDs.
__global__ void __launch_bounds__(64) gather_rows_kernel(const RowDesc *rows, long n, long width_bytes, int esz, uint8_t *dst,
                                                         long segs_per_row) {
    using u32x4 = unsigned __attribute__((ext_vector_type(4)));
    const long items = n * segs_per_row;
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);          // few instructions, long waits: let them issue ahead of the compute waves' streams
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const long row = item / segs_per_row, seg = item % segs_per_row;
        const uint8_t *src = reinterpret_cast<const uint8_t *>(rows[row].ptr);
        const long live = rows[row].len * esz;                     // bytes that exist; the rest of the row is zero
        uint8_t *d = dst + row * width_bytes;
        const long lo = seg * kSegBytes, hi = lo + kSegBytes < width_bytes ? lo + kSegBytes : width_bytes;
        if ((((size_t)src) & 15) == 0 && lo + kSegBytes <= live && hi == lo + kSegBytes) {
            // the common case: a whole segment inside the recording, 16-byte aligned source -- 8 loads, then 8 stores
            u32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + lo + (k * 64 + lane) * 16L));
#pragma unroll
            for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<u32x4 *>(d + lo + (k * 64 + lane) * 16L));
        } else if ((((size_t)src) & 15) == 0) {
            // a segment that holds the end of the recording (or of the row): 16-byte vectors, the straddling one byte-wise
            for (long o = lo + lane * 16L; o < hi; o += 64 * 16L) {
                u32x4 v{0u, 0u, 0u, 0u};
                if (o + 16 <= live) {
                    v = *reinterpret_cast<const u32x4 *>(src + o);
                } else if (o < live) {
                    unsigned w[4] = {0u, 0u, 0u, 0u};
                    for (int k = 0; k < (int)(live - o); ++k) w[k >> 2] |= (unsigned)src[o + k] << (8 * (k & 3));
                    v = u32x4{w[0], w[1], w[2], w[3]};
                }
                *reinterpret_cast<u32x4 *>(d + o) = v;
            }
        } else if (esz == 4) {                                     // a source that starts at an odd address: element-wise
            for (long o = lo + lane * 4L; o < hi; o += 64 * 4L)
                *reinterpret_cast<unsigned *>(d + o) = o < live ? *reinterpret_cast<const unsigned *>(src + o) : 0u;
        } else {
            for (long o = lo + lane * 2L; o < hi; o += 64 * 2L)
                *reinterpret_cast<unsigned short *>(d + o) = o < live ? *reinterpret_cast<const unsigned short *>(src + o) : (unsigned short)0;
        }
    }
}

}  // namespace

hipError_t launch_gather_rows(const RowDesc *rows, long n, long width, int esz, void *dst, bool rows_on_device, hipStream_t s) {
    if (n <= 0 || width <= 0) return hipSuccess;
    const long wb = width * esz, segs = (wb + kSegBytes - 1) / kSegBytes;
    const long items = n * segs;
    if (items > 0x7fffffffL) return hipErrorInvalidValue;
    // sources in HBM (a packed window that one big DMA brought over): an HBM-to-HBM scatter, as wide as the chip -- it runs
    // for a fraction of a millisecond; sources in host memory: the narrow persistent grid described above
    const long waves = rows_on_device ? 4096 : kGatherWaves;
    const unsigned grid = (unsigned)(items < waves ? items : waves);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(64), 0, s, rows, n, wb, esz, static_cast<uint8_t *>(dst), segs);
    return hipGetLastError();
}

}  // namespace vad
