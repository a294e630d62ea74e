// staging.cpp -- host-side packing of ragged recordings into one zero-padded [n][width] batch, the
// layout vad_forward_audio takes (the reference pads each recording's last chunk with zeros and
// handles one file per worker process: src/silero_vad/utils_vad.py:326-327,
// examples/parallel_example.ipynb cells 5, 7).  Pure memcpy/memset work, split over the persistent host
// workers so that filling a pinned staging buffer keeps up with the PCIe link.  This is the path for
// PAGEABLE sources; recordings that already sit in pinned memory skip it (vad_upload_rows, engine.hip).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/silero_vad_hip.h"
#include "host_threads.hpp"

extern "C" int vad_stage_rows(const void *const *rows, const long *lens, long n, long width,
                              size_t elem_size, void *dst, int threads) {
    if (n < 0 || width < 0 || (elem_size != 2 && elem_size != 4)) return VAD_ERR_ARG;
    if (n == 0 || width == 0) return VAD_OK;
    if (!rows || !lens || !dst) return VAD_ERR_ARG;
    for (long i = 0; i < n; ++i)
        if (lens[i] < 0 || lens[i] > width || (lens[i] > 0 && !rows[i])) return VAD_ERR_ARG;
    int nt = threads > 0 ? threads : vad::default_host_threads(32);
    const size_t bytes = (size_t)n * width * elem_size;
    nt = (int)std::max<size_t>(1, std::min<size_t>(nt, bytes / (4u << 20) + 1));
    auto work = [&](long lo, long hi) {
        for (long i = lo; i < hi; ++i) {
            uint8_t *d = static_cast<uint8_t *>(dst) + (size_t)i * width * elem_size;
            const size_t live = (size_t)lens[i] * elem_size;
            if (live) std::memcpy(d, rows[i], live);
            std::memset(d + live, 0, (size_t)width * elem_size - live);
        }
    };
    if (nt == 1) {
        work(0, n);
        return VAD_OK;
    }
    // persistent workers (host_threads.hpp): no thread is created per bucket.  Rows are dealt out in 4 x nt blocks so
    // that a few long rows do not leave the other workers idle.
    const int blocks = (int)std::min<long>(n, 4L * nt);
    const long per = (n + blocks - 1) / blocks;
    vad::HostPool::get().run(nt, blocks, [&](int k) {
        const long lo = k * per, hi = std::min(n, lo + per);
        if (lo < hi) work(lo, hi);
    });
    return VAD_OK;
}
