// engine.hip -- the C ABI (include/silero_vad_hip.h): engine lifetime, scratch, kernel launches.
#include <hip/hip_runtime.h>

#include <chrono>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/silero_vad_hip.h"
#include "device_api.hpp"
#include "host_threads.hpp"
#include "layout.hpp"
#include "weights.hpp"

#ifndef VAD_AB
#define VAD_AB 0        // 1: the test build (libsilero_vad_hip_ab.so) that also carries the A/B forms of the fp32 frontend
#endif

// Read-only device data of one weight container: shared by an engine and its clones (vad_clone).
struct vad_images {
    int device = -1;
    uint8_t *d_blob = nullptr;                      // canonical container (impl=reference)
    vad::RefNet ref[2] = {};
    vad::RefNet *d_ref = nullptr;                   // the same two structs in device memory (exact_front.hpp reads them from the kernels)
    float *d_gx_silent = nullptr;                   // [2][512]: each net's gate pre-activations for a chunk of zeros (exact_front.hpp)
    float *d_front4[2] = {}, *d_whh[2] = {}, *d_whh_lat[2] = {}, *d_whh_rows[2] = {}, *d_tables[2] = {};
    uint16_t *d_whh_b9[2] = {}, *d_front_b9[2] = {};
#if VAD_AB
    float *d_front[2] = {}, *d_front_wino[2] = {};
#endif
    ~vad_images() {
        if (device < 0) return;
        (void)hipSetDevice(device);
        for (int ni = 0; ni < 2; ++ni) {
            if (d_front4[ni]) (void)hipFree(d_front4[ni]);
            if (d_whh[ni]) (void)hipFree(d_whh[ni]);
            if (d_whh_b9[ni]) (void)hipFree(d_whh_b9[ni]);
            if (d_whh_rows[ni]) (void)hipFree(d_whh_rows[ni]);
            if (d_front_b9[ni]) (void)hipFree(d_front_b9[ni]);
            if (d_whh_lat[ni]) (void)hipFree(d_whh_lat[ni]);
            if (d_tables[ni]) (void)hipFree(d_tables[ni]);
#if VAD_AB
            if (d_front[ni]) (void)hipFree(d_front[ni]);
            if (d_front_wino[ni]) (void)hipFree(d_front_wino[ni]);
#endif
        }
        if (d_blob) (void)hipFree(d_blob);
        if (d_ref) (void)hipFree(d_ref);
        if (d_gx_silent) (void)hipFree(d_gx_silent);
    }
};

struct vad_engine {
    std::shared_ptr<vad::Weights> weights;          // host: container + packed images (shared with clones)
    std::shared_ptr<vad_images> img;                // device: the packed images (shared with clones)
    bool host_only = false;
    int device = -1;
    std::string err;
    bool impl_reference = false;
    int enc0 = 2;                                   // fp32 frontend, encoder 0: 2 Winograd F(4,3) (the product); test builds: 0 direct, 1 F(2,3)
    bool fuse_step = true;                          // a ONE-step call small enough for the latency frontend runs the LSTM cell and the head in
                                                    // the same kernel (option "fuse_step")
    int rec_form = 0;                               // fp32 recurrence: 0 auto (VALU matrix-vector form for B <= 1024, same bits), 1 MFMA form always
    bool front_b9 = false;                          // frontend products: fp32 MFMA chain (default) | exact bf16 x 9 (option "front_mma")
    bool rec_b9 = false;                            // recurrence: fp32 MFMA chain (default) | exact bf16 x 9 products (option "rec")
    bool profile = false;
    bool fused_decimation = true;                   // 32 / 48 kHz: decimate inside the frontend's loads (option "fused_decimation")
    bool exact_all_silent = true;                   // ... and all-silent chunks take the net's constant (false: option exact_transitions=edges, study mode)
    bool exact_transitions = true;                  // chunks with an exactly silent frame beside a non-silent one are evaluated in double
                                                    // (option "exact_transitions"; csrc/exact_front.hpp)
    int one_max = 256;                              // a ONE-step call of at most this many streams takes the one-workgroup-per-stream kernel
                                                    // (kernel_step_one.hip; option "step_one": auto | 0 | <max streams>)
    long lat_tiles = 768;                           // launches of at most this many 16-chunk tiles take the latency form of the frontend
                                                    // (option "front": auto | throughput | latency -> 768 | 0 | LONG_MAX)
    long long *trace = nullptr;                     // bring-up: device buffer for VAD_TRACE builds

    // scratch (per engine: a clone has its own, so that an engine and its clones may be in flight on different streams)
    float *d_gx = nullptr;
    size_t gx_floats = 0;
    float *d_ctx_new = nullptr;
    size_t ctx_floats = 0;
    int *d_exact = nullptr;                         // [2 + chunks of a slab]: the throughput frontend's list of chunks for the fix-up pass
    size_t exact_ints = 0;
    void *d_tail = nullptr;                         // [B][N * dec] zero padded last chunk (L % N != 0)
    size_t tail_bytes = 0;
    void *d_realign = nullptr;                      // aligned copy of a misaligned input (rare)
    size_t realign_bytes = 0;
    void *d_decim = nullptr;                        // 16 kHz copy of a 64/80/... kHz input
    size_t decim_bytes = 0;
    unsigned long scratch_gen = 0;                  // bumped whenever a scratch buffer is reallocated: a hipGraph that
                                                    // captured calls of this engine holds the OLD addresses (vad_scratch_generation)
    long slab_steps = 0;                            // time steps per gx slab for the last reserve
    size_t gx_cap = 6ull << 30;                     // cap of the gx scratch; longer inputs are slabbed (option gx_cap_mib)

    // vad_upload_rows: a ring of pinned row tables (the gather kernel reads them over PCIe), one event per slot
    static constexpr int kTabSlots = 8;
    vad::RowDesc *h_tab[kTabSlots] = {};
    long tab_cap[kTabSlots] = {};
    hipEvent_t tab_ev[kTabSlots] = {};
    bool tab_busy[kTabSlots] = {};
    int tab_next = 0;

    // vad_upload_rows how = 0: the per-row DMAs are dealt round-robin to these side streams (several copy engines at once; one engine
    // spends ~30 us per copy whatever its size), forked from and joined to the caller's stream by events
    static constexpr int kDmaStreams = 4;
    hipStream_t dma_stream[kDmaStreams] = {};
    hipEvent_t dma_fork = nullptr, dma_join[kDmaStreams] = {};

    // vad_step_host: the device's view of the caller's page-locked buffers, remembered (a B = 1 caller hands over the same two buffers
    // 30 times a second for hours; resolving one costs a runtime call)
    const void *map_host[2] = {nullptr, nullptr};
    void *map_dev[2] = {nullptr, nullptr};
    unsigned long map_gen = 0;                       // ... valid while no page-locked range has been released since (g_unmap_gen)

    // profiling: 3 events per (call, slab), read back lazily by vad_kernel_times
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    long prof_calls = 0;
};

namespace {

int fail(vad_engine *e, int code, const std::string &msg) {
    if (e) e->err = msg;
    return code;
}
// bumped by vad_host_unregister: device views of host buffers that engines remember (vad_step_host) are dropped
std::atomic<unsigned long> g_unmap_gen{1};

int hip_fail(vad_engine *e, hipError_t rc, const char *what) {
    return fail(e, VAD_ERR_HIP, std::string(what) + ": " + hipGetErrorString(rc));
}
#define HIP_TRY(e, call)                                           \
    do {                                                           \
        hipError_t rc__ = (call);                                  \
        if (rc__ != hipSuccess) return hip_fail(e, rc__, #call);   \
    } while (0)

int net_index(int sr) { return sr == 16000 ? 0 : sr == 8000 ? 1 : -1; }

long slab_for(const vad_engine *e, int B, long T) {
    const long nst = (B + 15) / 16;
    const size_t per_step = (size_t)nst * 32 * 256 * sizeof(float);   // gx bytes per time step
    long s = (long)std::max<size_t>(1, e->gx_cap / per_step);
    return std::min(s, T);
}

int ensure_scratch(vad_engine *e, int sr, int B, long T, hipStream_t stream) {
    const long nst = (B + 15) / 16;
    const long slab = slab_for(e, B, T);
    const size_t need_gx = (size_t)nst * slab * 32 * 256;
    const size_t need_ctx = (size_t)B * (sr == 16000 ? 64 : 32);
    // worst case of the tail copy: fp32 samples at 48 kHz (dec = 3) -- sized here so that no forward call has to grow it
    // (a growth is a device synchronisation and cannot happen while the call is being captured into a hipGraph)
    const size_t need_tail = (size_t)B * (sr == 16000 ? 512 * 3 : 256) * sizeof(float);
    const size_t need_exact = 2 + (size_t)nst * 16 * slab;
    if (need_gx > e->gx_floats || need_ctx > e->ctx_floats || need_tail > e->tail_bytes || need_exact > e->exact_ints) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(e, VAD_ERR_CAPTURE, "scratch must grow during stream capture; call vad_reserve first");
        HIP_TRY(e, hipDeviceSynchronize());
        e->scratch_gen++;
        if (need_gx > e->gx_floats) {
            if (e->d_gx) (void)hipFree(e->d_gx);
            e->d_gx = nullptr;
            e->gx_floats = 0;
            if (hipMalloc((void **)&e->d_gx, need_gx * sizeof(float)) != hipSuccess)
                return fail(e, VAD_ERR_ALLOC, "cannot allocate gx scratch");
            e->gx_floats = need_gx;
        }
        if (need_ctx > e->ctx_floats) {
            if (e->d_ctx_new) (void)hipFree(e->d_ctx_new);
            e->d_ctx_new = nullptr;
            e->ctx_floats = 0;
            if (hipMalloc((void **)&e->d_ctx_new, need_ctx * sizeof(float)) != hipSuccess)
                return fail(e, VAD_ERR_ALLOC, "cannot allocate context scratch");
            e->ctx_floats = need_ctx;
        }
        if (need_exact > e->exact_ints) {
            if (e->d_exact) (void)hipFree(e->d_exact);
            e->d_exact = nullptr;
            e->exact_ints = 0;
            if (hipMalloc((void **)&e->d_exact, need_exact * sizeof(int)) != hipSuccess)
                return fail(e, VAD_ERR_ALLOC, "cannot allocate the exact-chunk list");
            HIP_TRY(e, hipMemset(e->d_exact, 0, 2 * sizeof(int)));      // (the fix-up pass leaves the counters at zero: kernel_exact.hip)
            e->exact_ints = need_exact;
        }
        if (need_tail > e->tail_bytes) {
            if (e->d_tail) (void)hipFree(e->d_tail);
            e->d_tail = nullptr;
            e->tail_bytes = 0;
            if (hipMalloc(&e->d_tail, need_tail) != hipSuccess)
                return fail(e, VAD_ERR_ALLOC, "cannot allocate tail scratch");
            e->tail_bytes = need_tail;
        }
    }
    e->slab_steps = slab;
    return VAD_OK;
}

// Grow-only scratch buffer (never while a stream is being captured); bumps the scratch generation.
int grow(vad_engine *e, void **buf, size_t *have, size_t need, hipStream_t stream, const char *what) {
    if (need <= *have) return VAD_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(e, VAD_ERR_CAPTURE, std::string(what) + " scratch must grow during stream capture");
    HIP_TRY(e, hipDeviceSynchronize());
    e->scratch_gen++;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    if (hipMalloc(buf, need) != hipSuccess) return fail(e, VAD_ERR_ALLOC, std::string("cannot allocate ") + what + " scratch");
    *have = need;
    return VAD_OK;
}

// L, ld: samples per row / row stride of `pcm` AS HANDED OVER (raw rate).  dec = 1: pcm is at the net's rate `sr`.
// dec = 2, 3: pcm is at dec x 16 kHz and the frontend reads every dec-th sample itself (sample-rate front door
// folded into the load; fp32 frontend).
template <typename PcmT>
int forward_core(vad_engine *e, int sr, int dec, int B, long L, const PcmT *pcm, long ld, float *ctx,
                 float *state, float *probs, long ldp, hipStream_t stream, float *ctx_next = nullptr, const uint8_t *present = nullptr) {
    const int ni = net_index(sr);
    const int N = sr == 16000 ? 512 : 256, C = N / 8;
    const long Ld = (L + dec - 1) / dec;                 // samples per row at the net's rate
    const long T = (Ld + N - 1) / N;
    if (ldp < T) return fail(e, VAD_ERR_ARG, "ldp < T");
    if (present && T != 1) return fail(e, VAD_ERR_ARG, "present[] is for one-step calls");
    if (((size_t)ctx & 15) || ((size_t)state & 15))
        return fail(e, VAD_ERR_ARG, "ctx and state must be 16-byte aligned");
    int rc = ensure_scratch(e, sr, B, T, stream);
    if (rc) return rc;
    // The kernels use 16-byte vector loads: rows must be 16-byte aligned.  A misaligned input (odd
    // row stride with B > 1, or an offset view) is copied once into aligned scratch.
    const size_t esz = sizeof(PcmT);
    if (((size_t)pcm & 15) || (B > 1 && (ld * esz) % 16)) {
        const long ldA = (L + 15) / 16 * 16;
        rc = grow(e, &e->d_realign, &e->realign_bytes, (size_t)B * ldA * esz, stream, "realign");
        if (rc) return rc;
        HIP_TRY(e, hipMemcpy2DAsync(e->d_realign, ldA * esz, pcm, ld * esz, L * esz, B,
                                    hipMemcpyDeviceToDevice, stream));
        pcm = reinterpret_cast<const PcmT *>(e->d_realign);
        ld = ldA;
    }
    // The last chunk is handed to the kernel as a zero-padded copy when it is partial -- or, with dec > 1, when its
    // vector loads (which cover whole groups of dec raw samples) would run past the end of the row.
    const void *tail = nullptr;
    if (Ld % N || Ld * dec > L) {
        // [B][N * dec], zero padded, RAW sample spacing (the kernel reads it with the same stride as the signal)
        const long rem = L - (T - 1) * N * dec;          // raw samples of the last chunk
        rc = grow(e, &e->d_tail, &e->tail_bytes, (size_t)B * N * dec * esz, stream, "tail");
        if (rc) return rc;
        HIP_TRY(e, hipMemsetAsync(e->d_tail, 0, (size_t)B * N * dec * esz, stream));
        HIP_TRY(e, hipMemcpy2DAsync(e->d_tail, (size_t)N * dec * esz, pcm + (T - 1) * N * dec, ld * esz, rem * esz, B,
                                    hipMemcpyDeviceToDevice, stream));
        tail = e->d_tail;
    }
    const long slab = e->slab_steps;
    const bool prof = e->profile;
    if (prof) e->prof_calls++;
    bool ctx_in_place = false;                           // the kernel wrote the next context into `ctx` itself (one-stream fused step)
    for (long t0 = 0; t0 < T; t0 += slab) {
        const long nt = std::min(slab, T - t0);
        vad::FrontArgs fa{};
#if VAD_AB
        fa.wfront = e->enc0 == 2 ? e->img->d_front4[ni] : e->enc0 == 1 ? e->img->d_front_wino[ni] : e->img->d_front[ni];
#else
        fa.wfront = e->img->d_front4[ni];
#endif
        fa.tables = e->img->d_tables[ni];
        fa.pcm = pcm;
        fa.tail = tail;
        fa.ld = ld; fa.L = Ld; fa.T = T; fa.t0 = t0; fa.nt = nt;
        fa.ctx_in = ctx;
        fa.ctx_out = ctx_next ? ctx_next : e->d_ctx_new;     // (the kernel may not write where other waves of the tile still read)
        fa.gx = e->d_gx;
        fa.B = B;
        fa.dec = dec;
        fa.trace = e->trace;
        // chunks with an exactly silent frame beside one that is not: double-precision gx (the product's fp32 frontend only)
        const bool exact = e->exact_transitions && e->enc0 == 2 && !e->front_b9;
        fa.exact_net = exact ? e->img->d_ref + ni : nullptr;
        fa.gx_silent = exact && e->exact_all_silent ? e->img->d_gx_silent + 512 * ni : nullptr;
        vad::RecArgs ra{};
        ra.whh = e->rec_b9 ? reinterpret_cast<const float *>(e->img->d_whh_b9[ni]) : e->img->d_whh[ni];
        ra.tables = e->img->d_tables[ni];
        ra.gx = e->d_gx;
        ra.state = state;
        ra.probs = probs;
        ra.ldp = ldp; ra.t0 = t0; ra.nt = nt; ra.B = B;
        ra.present = present;
        hipEvent_t *ev = nullptr;
        if (prof) {
            while (e->ev_pool.size() < e->ev_used + 3) {
                hipEvent_t x;
                HIP_TRY(e, hipEventCreate(&x));
                e->ev_pool.push_back(x);
            }
            ev = &e->ev_pool[e->ev_used];
            e->ev_used += 3;
            HIP_TRY(e, hipEventRecord(ev[0], stream));
        }
        const long tiles = (long)((B + 15) / 16) * nt;
        const bool one = T == 1 && dec == 1 && B <= e->one_max && !e->rec_b9 && !e->front_b9 && e->enc0 == 2;
        if (T == 1 && e->fuse_step && !e->rec_b9 && !e->front_b9 && e->enc0 == 2 && (tiles <= e->lat_tiles || one)) {
            // one step, few tiles (a stream pool's tick, a B = 1 call): frontend, LSTM cell and head in ONE kernel, no gx round trip
            vad::CellArgs ca{};
            ca.whh_lat = e->img->d_whh_lat[ni];
            ca.state = state;
            ca.probs = probs;
            ca.ldp = ldp;
            ca.present = present;
            // a handful of streams (the B = 1 call of every unmodified caller): one workgroup per stream, the same sums on the VALU --
            // and the context in place (one workgroup owns the stream: no second buffer, no copy operation behind the kernel)
            if (one) {
                if (!ctx_next) {
                    fa.ctx_out = ctx;
                    ctx_in_place = true;
                }
                HIP_TRY(e, vad::launch_step_one<PcmT>(sr, fa, ca, stream));
            } else HIP_TRY(e, vad::launch_step_lat<PcmT>(sr, fa, ca, stream));
            if (prof) {
                HIP_TRY(e, hipEventRecord(ev[1], stream));
                HIP_TRY(e, hipEventRecord(ev[2], stream));
            }
            continue;
        }
        bool front_stamped = false;
        if (e->front_b9) {
            fa.wfront = reinterpret_cast<const float *>(e->img->d_front_b9[ni]);
            HIP_TRY(e, vad::launch_front_b9<PcmT>(sr, fa, stream));
        } else
#if VAD_AB
        if (e->enc0 == 1) HIP_TRY(e, vad::launch_front_wino<PcmT>(sr, fa, stream));
        else if (e->enc0 == 0) HIP_TRY(e, vad::launch_front<PcmT>(sr, fa, stream));
        else
#endif
        if (one) HIP_TRY(e, vad::launch_front_one<PcmT>(sr, fa, stream));       // (fuse_step=0: the one-stream frontend + a recurrence kernel)
        else if (tiles <= e->lat_tiles) HIP_TRY(e, vad::launch_front_lat<PcmT>(sr, fa, stream));
        else {
            fa.exact_list = exact ? e->d_exact : nullptr;
            HIP_TRY(e, vad::launch_front_f43<PcmT>(sr, fa, stream));
            if (prof) HIP_TRY(e, hipEventRecord(ev[1], stream));         // (the frontend's own time ends here ...
            front_stamped = prof;
            HIP_TRY(e, vad::launch_exact_fix<PcmT>(sr, fa, stream));     //  ... the fix-up pass is part of the path, not of that kernel)
        }
        if (prof && !front_stamped) HIP_TRY(e, hipEventRecord(ev[1], stream));
        if (e->rec_b9) HIP_TRY(e, vad::launch_rec_b9(sr, ra, stream));
        else if (e->rec_form != 1 && B <= vad::kRecSmallMaxB) {
            // a file at a time, a bucket of a few hundred: W_hh h as matrix-vector products on the VALU, 1-4 streams per CU -- the same bits
            // at 1.2-2.7 us per step instead of 4.3
            ra.whh = e->img->d_whh_rows[ni];
            HIP_TRY(e, vad::launch_rec_small(sr, ra, stream));
        } else HIP_TRY(e, vad::launch_rec(sr, ra, stream));
        if (prof) HIP_TRY(e, hipEventRecord(ev[2], stream));
    }
    // rows without a chunk this tick: the step kernels left their (h, c) and probability alone; carry their context over and mark
    // their probability slot (kernel_present.hip)
    if (present) HIP_TRY(e, vad::launch_carry_absent(present, ctx, ctx_next ? ctx_next : ctx_in_place ? nullptr : e->d_ctx_new, C, probs, ldp, B, stream));
    if (!ctx_next && !ctx_in_place) HIP_TRY(e, hipMemcpyAsync(ctx, e->d_ctx_new, (size_t)B * C * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return VAD_OK;
}

template <typename PcmT>
int forward_impl(vad_engine *e, int sr, int B, long L, const PcmT *pcm, long ld, float *ctx,
                 float *state, float *probs, long ldp, void *stream_v) {
    if (!e) return VAD_ERR_ARG;
    if (e->host_only) return fail(e, VAD_ERR_NO_DEVICE, "host-only engine");
    if (B < 0 || L < 0 || (B > 0 && L > 0 && (!pcm || !ctx || !state || !probs)) || ld < L)
        return fail(e, VAD_ERR_ARG, "bad argument");
    hipStream_t stream = (hipStream_t)stream_v;
    if (sr > 16000 && sr % 16000 == 0) {
        // sample-rate front door: a multiple of 16 kHz is decimated to 16 kHz, x[:, ::sr/16000], exactly as the
        // reference does (vad_annotator.py:104-112), and takes the 16 kHz path.  For 32 and 48 kHz the fp32 frontend does
        // it while loading (no extra pass over HBM; 48 kHz: the product frontend only, the A/B forms have no stride-3
        // instantiation); higher multiples and impl=reference go through a decimated copy in engine scratch.
        if (B == 0 || L == 0) return VAD_OK;
        const int k = sr / 16000;
        HIP_TRY(e, hipSetDevice(e->device));
        if ((k == 2 || (k == 3 && e->enc0 == 2)) && !e->impl_reference && e->fused_decimation)
            return forward_core<PcmT>(e, 16000, k, B, L, pcm, ld, ctx, state, probs, ldp, stream);
        const long Ld = (L + k - 1) / k, ldd = (Ld + 15) / 16 * 16;
        int rc = grow(e, &e->d_decim, &e->decim_bytes, (size_t)B * ldd * sizeof(PcmT), stream, "decimation");
        if (rc) return rc;
        PcmT *dec = reinterpret_cast<PcmT *>(e->d_decim);
        HIP_TRY(e, vad::launch_decimate<PcmT>(pcm, ld, dec, ldd, B, Ld, k, stream));
        return forward_impl<PcmT>(e, 16000, B, Ld, dec, ldd, ctx, state, probs, ldp, stream_v);
    }
    const int ni = net_index(sr);
    if (ni < 0) return fail(e, VAD_ERR_SAMPLE_RATE, "Supported sampling rates: [8000, 16000] (or multiply of 16000)");
    if (B == 0 || L == 0) return VAD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    if (e->impl_reference) {
        const int N = sr == 16000 ? 512 : 256;
        if (ldp < (L + N - 1) / N) return fail(e, VAD_ERR_ARG, "ldp < T");
        HIP_TRY(e, vad::launch_ref_forward<PcmT>(e->img->ref[ni], sr, B, L, pcm, ld, ctx, state, probs, ldp, stream));
        return VAD_OK;
    }
    return forward_core<PcmT>(e, sr, 1, B, L, pcm, ld, ctx, state, probs, ldp, stream);
}

template <typename T>
int upload(vad_engine *e, T **dst, const std::vector<T> &src) {
    HIP_TRY(e, hipMalloc((void **)dst, src.size() * sizeof(T)));
    HIP_TRY(e, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return VAD_OK;
}

}  // namespace

extern "C" {

const char *vad_strerror(int status) {
    switch (status) {
        case VAD_OK: return "ok";
        case VAD_ERR_ARG: return "invalid argument";
        case VAD_ERR_SAMPLE_RATE: return "unsupported sampling rate (supported: 8000, 16000)";
        case VAD_ERR_WEIGHTS: return "malformed weight container";
        case VAD_ERR_NO_DEVICE: return "no usable gfx950 HIP device";
        case VAD_ERR_HIP: return "HIP runtime error";
        case VAD_ERR_ALLOC: return "device allocation failed";
        case VAD_ERR_CAPTURE: return "scratch growth during stream capture";
        case VAD_ERR_OPTION: return "unknown option";
        default: return "unknown status";
    }
}

const char *vad_last_error(const vad_engine *e) { return e ? e->err.c_str() : "null engine"; }
int vad_device(const vad_engine *e) { return e ? e->device : -1; }

int vad_geometry(int sr, int *chunk, int *context) {
    if (net_index(sr) < 0) return VAD_ERR_SAMPLE_RATE;
    if (chunk) *chunk = sr == 16000 ? 512 : 256;
    if (context) *context = sr == 16000 ? 64 : 32;
    return VAD_OK;
}

int vad_create_host_only(const void *weights, size_t nbytes, vad_engine **out) {
    if (!out) return VAD_ERR_ARG;
    *out = nullptr;
    vad_engine *e = new (std::nothrow) vad_engine();
    if (!e) return VAD_ERR_ALLOC;
    e->weights = std::make_shared<vad::Weights>();
    const std::string err = e->weights->load(weights, nbytes);
    if (!err.empty()) {
        delete e;
        return VAD_ERR_WEIGHTS;
    }
    e->host_only = true;
    *out = e;
    return VAD_OK;
}

int vad_create(const void *weights, size_t nbytes, int device, vad_engine **out) {
    if (!out) return VAD_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return VAD_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return VAD_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VAD_ERR_NO_DEVICE;
    vad_engine *e = nullptr;
    int rc = vad_create_host_only(weights, nbytes, &e);
    if (rc) return rc;
    e->host_only = false;
    e->device = device;
    auto bail = [&](int code) {
        vad_destroy(e);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(VAD_ERR_HIP);
    e->img = std::make_shared<vad_images>();
    vad_images &im = *e->img;
    im.device = device;
    for (int ni = 0; ni < 2; ++ni) {
        const vad::PackedNet &pk = e->weights->packed[ni];
        if (upload(e, &im.d_front4[ni], pk.front_wino4)) return bail(VAD_ERR_HIP);
        if (upload(e, &im.d_whh[ni], pk.whh)) return bail(VAD_ERR_HIP);
        if (upload(e, &im.d_whh_b9[ni], pk.whh_b9)) return bail(VAD_ERR_HIP);
        if (upload(e, &im.d_front_b9[ni], pk.front_b9)) return bail(VAD_ERR_HIP);
        if (upload(e, &im.d_whh_lat[ni], pk.whh_lat)) return bail(VAD_ERR_HIP);
        if (upload(e, &im.d_whh_rows[ni], pk.whh_rows)) return bail(VAD_ERR_HIP);
        if (upload(e, &im.d_tables[ni], pk.tables)) return bail(VAD_ERR_HIP);
#if VAD_AB
        if (upload(e, &im.d_front[ni], pk.front)) return bail(VAD_ERR_HIP);
        if (upload(e, &im.d_front_wino[ni], pk.front_wino)) return bail(VAD_ERR_HIP);
#endif
    }
    // canonical tensors for impl=reference
    const auto &blob = e->weights->blob;
    if (hipMalloc((void **)&im.d_blob, blob.size()) != hipSuccess) return bail(VAD_ERR_ALLOC);
    if (hipMemcpy(im.d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess)
        return bail(VAD_ERR_HIP);
    for (int ni = 0; ni < 2; ++ni) {
        const vad::NetTensors &t = e->weights->net[ni];
        auto dev = [&](const float *p) {
            return reinterpret_cast<const float *>(im.d_blob + ((const uint8_t *)p - blob.data()));
        };
        vad::RefNet &r = im.ref[ni];
        r.basis = dev(t.basis);
        for (int l = 0; l < 4; ++l) { r.ew[l] = dev(t.ew[l]); r.eb[l] = dev(t.eb[l]); }
        r.w_ih = dev(t.w_ih); r.w_hh = dev(t.w_hh); r.b_ih = dev(t.b_ih); r.b_hh = dev(t.b_hh);
        r.w_out = dev(t.w_out); r.b_out = dev(t.b_out);
    }
    if (hipMalloc((void **)&im.d_ref, sizeof(im.ref)) != hipSuccess) return bail(VAD_ERR_ALLOC);
    if (hipMemcpy(im.d_ref, im.ref, sizeof(im.ref), hipMemcpyHostToDevice) != hipSuccess) return bail(VAD_ERR_HIP);
    if (hipMalloc((void **)&im.d_gx_silent, 2 * 512 * sizeof(float)) != hipSuccess) return bail(VAD_ERR_ALLOC);
    for (int ni = 0; ni < 2; ++ni)
        if (vad::launch_exact_silent(ni == 0 ? 16000 : 8000, im.d_ref + ni, im.d_gx_silent + 512 * ni, nullptr) != hipSuccess) return bail(VAD_ERR_HIP);
    if (hipDeviceSynchronize() != hipSuccess) return bail(VAD_ERR_HIP);
    *out = e;
    return VAD_OK;
}

void vad_destroy(vad_engine *e) {
    if (e && e->dma_fork) {
        (void)hipSetDevice(e->device);
        for (int k = 0; k < vad_engine::kDmaStreams; ++k) {
            if (e->dma_stream[k]) { (void)hipStreamSynchronize(e->dma_stream[k]); (void)hipStreamDestroy(e->dma_stream[k]); }
            if (e->dma_join[k]) (void)hipEventDestroy(e->dma_join[k]);
        }
        (void)hipEventDestroy(e->dma_fork);
        e->dma_fork = nullptr;
    }
    if (!e) return;
    if (!e->host_only && e->device >= 0) {
        (void)hipSetDevice(e->device);
        (void)hipDeviceSynchronize();
        if (e->d_gx) (void)hipFree(e->d_gx);
        if (e->d_ctx_new) (void)hipFree(e->d_ctx_new);
        if (e->d_exact) (void)hipFree(e->d_exact);
        if (e->d_tail) (void)hipFree(e->d_tail);
        if (e->d_realign) (void)hipFree(e->d_realign);
        if (e->d_decim) (void)hipFree(e->d_decim);
        for (int i = 0; i < vad_engine::kTabSlots; ++i) {
            if (e->h_tab[i]) (void)hipHostFree(e->h_tab[i]);
            if (e->tab_ev[i]) (void)hipEventDestroy(e->tab_ev[i]);
        }
        for (auto &ev : e->ev_pool) (void)hipEventDestroy(ev);
    }
    delete e;                                        // the shared images go with their last owner (vad_images::~vad_images)
}

// A second handle on the same GPU that shares the (read-only) weight images and carries the same options, with its OWN
// scratch: what a caller needs to keep two calls in flight on two streams (streams.py issues the buckets of a corpus
// round-robin to such "lanes": one lane's latency-bound recurrence runs beside the other lane's frontend).
int vad_clone(const vad_engine *src, vad_engine **out) {
    if (!src || !out) return VAD_ERR_ARG;
    *out = nullptr;
    if (src->host_only) return VAD_ERR_NO_DEVICE;
    vad_engine *e = new (std::nothrow) vad_engine();
    if (!e) return VAD_ERR_ALLOC;
    e->weights = src->weights;
    e->img = src->img;
    e->device = src->device;
    e->impl_reference = src->impl_reference;
    e->enc0 = src->enc0;
    e->fused_decimation = src->fused_decimation;
    e->lat_tiles = src->lat_tiles;
    e->rec_b9 = src->rec_b9;
    e->rec_form = src->rec_form;
    e->front_b9 = src->front_b9;
    e->fuse_step = src->fuse_step;
    e->one_max = src->one_max;
    e->exact_transitions = src->exact_transitions;
    e->exact_all_silent = src->exact_all_silent;
    e->gx_cap = src->gx_cap;
    e->trace = src->trace;
    *out = e;
    return VAD_OK;
}

int vad_set_option(vad_engine *e, const char *name, const char *value) {
    if (!e || !name || !value) return VAD_ERR_ARG;
    const std::string n(name), v(value);
    if (n == "impl") {
        if (v == "mfma") e->impl_reference = false;
        else if (v == "reference") e->impl_reference = true;
        else return fail(e, VAD_ERR_OPTION, "impl must be mfma|reference");
        return VAD_OK;
    }
    if (n == "precision") {                          // one arithmetic: fp32 (the reference's); accepted so that bindings may state it
        if (v != "fp32") return fail(e, VAD_ERR_OPTION, "precision must be fp32 (the engine computes in fp32 only)");
        return VAD_OK;
    }
    if (n == "enc0") {                               // how the fp32 frontend evaluates encoder 0
        if (v == "winograd" || v == "winograd4") e->enc0 = 2;
#if VAD_AB
        else if (v == "winograd2") e->enc0 = 1;
        else if (v == "direct") e->enc0 = 0;
        else return fail(e, VAD_ERR_OPTION, "enc0 must be winograd|winograd4|winograd2|direct");
#else
        else return fail(e, VAD_ERR_OPTION, "enc0 must be winograd (the A/B forms winograd2|direct exist only in the test build, "
                                            "libsilero_vad_hip_ab.so)");
#endif
        return VAD_OK;
    }
    if (n == "rec") {                                // the recurrence's arithmetic: fp32 MFMA chain | exact bf16 x 9 piece products
        if (v == "fp32") e->rec_b9 = false;
        else if (v == "bf16x9") e->rec_b9 = true;
        else return fail(e, VAD_ERR_OPTION, "rec must be fp32|bf16x9");
        return VAD_OK;
    }
    if (n == "front_mma") {                          // the frontend's matrix products: fp32 MFMA chain | exact bf16 x 9 piece products
        if (v == "fp32") e->front_b9 = false;        // (bf16x9: every launch takes the throughput form, whatever its size -- the
        else if (v == "bf16x9") e->front_b9 = true;  //  arithmetic of a result must not depend on the batch it came in)
        else return fail(e, VAD_ERR_OPTION, "front_mma must be fp32|bf16x9");
        return VAD_OK;
    }
    if (n == "rec_form") {                           // which form of the fp32 recurrence a launch takes (A/B for tests; results are bit-identical)
        if (v == "auto" || v == "valu") e->rec_form = 0;
        else if (v == "mfma") e->rec_form = 1;
        else return fail(e, VAD_ERR_OPTION, "rec_form must be auto|mfma|valu");
        return VAD_OK;
    }
    if (n == "fuse_step") {                          // "0": a one-step call runs frontend and recurrence as two kernels (A/B for tests)
        e->fuse_step = (v != "0");
        return VAD_OK;
    }
    if (n == "front") {                              // which form of the frontend a launch takes (A/B for tests; results are bit-identical)
        if (v == "auto") e->lat_tiles = 768;
        else if (v == "throughput") e->lat_tiles = 0;
        else if (v == "latency") e->lat_tiles = 0x7fffffffL;
        else return fail(e, VAD_ERR_OPTION, "front must be auto|throughput|latency");
        return VAD_OK;
    }
    if (n == "step_one") {                           // one-step calls of at most N streams: one workgroup per stream ("0": never; A/B for tests)
        if (v == "auto") e->one_max = 256;           // one workgroup per CU: 25.0-26.5 us for 1..256 streams, the tile kernels 31.4-34.2 (512: 47.7 / 34.4)
        else {
            char *end = nullptr;
            const long k = std::strtol(v.c_str(), &end, 10);
            if (!end || *end || k < 0 || k > 4096) return fail(e, VAD_ERR_OPTION, "step_one: auto | 0 .. 4096");
            e->one_max = (int)k;
        }
        return VAD_OK;
    }
    if (n == "exact_transitions") {                  // "0": every chunk through the fp32 chains (A/B for tests and studies)
        if (v != "0" && v != "1" && v != "edges") return fail(e, VAD_ERR_OPTION, "exact_transitions: 0 | 1 (| edges: study mode)");
        e->exact_transitions = v != "0";
        e->exact_all_silent = v == "1";
        return VAD_OK;
    }
    if (n == "fused_decimation") {                   // "0": always decimate into scratch first (A/B for tests)
        e->fused_decimation = (v != "0");
        return VAD_OK;
    }
    if (n == "gx_cap_mib") {                         // scratch cap (MiB); inputs longer than it allows run in time slabs
        const long mib = std::strtol(value, nullptr, 0);
        if (mib < 1) return fail(e, VAD_ERR_OPTION, "gx_cap_mib must be >= 1");
        e->gx_cap = (size_t)mib << 20;
        return VAD_OK;
    }
    if (n == "trace_ptr") {                          // bring-up only; ignored by normal builds
        e->trace = reinterpret_cast<long long *>(std::strtoull(value, nullptr, 0));
        return VAD_OK;
    }
    if (n == "profile") {
        e->profile = (v == "1");
        e->ev_used = 0;
        e->prof_calls = 0;
        return VAD_OK;
    }
    return fail(e, VAD_ERR_OPTION, "unknown option " + n);
}

int vad_step(vad_engine *e, int sr, int B, const float *pcm, long ld, float *ctx, float *state,
             float *prob, void *stream) {
    const int N = (sr > 16000 && sr % 16000 == 0) ? 512 * (sr / 16000) : sr == 16000 ? 512 : 256;
    return forward_impl<float>(e, sr, B, N, pcm, ld, ctx, state, prob, 1, stream);
}

int vad_step_present(vad_engine *e, int sr, int B, const void *pcm, size_t elem_size, long ld, const float *ctx_in, float *ctx_out,
                     float *state, float *prob, const uint8_t *present, void *stream) {
    if (!e) return VAD_ERR_ARG;
    if (e->host_only) return fail(e, VAD_ERR_NO_DEVICE, "host-only engine");
    const int ni = net_index(sr);
    if (ni < 0) return fail(e, VAD_ERR_SAMPLE_RATE, "Supported sampling rates: [8000, 16000]");
    const int N = sr == 16000 ? 512 : 256;
    if (B < 0 || (elem_size != 2 && elem_size != 4) || ld < N || (B > 0 && (!pcm || !ctx_in || !state || !prob)))
        return fail(e, VAD_ERR_ARG, "bad argument");
    if (ctx_out == ctx_in) ctx_out = nullptr;                 // in place: through the engine's second buffer, like vad_step
    if ((size_t)ctx_out & 15) return fail(e, VAD_ERR_ARG, "ctx_out must be 16-byte aligned");
    if (e->impl_reference) return fail(e, VAD_ERR_OPTION, "impl=reference has no split-context / present[] form");
    if (B == 0) return VAD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    float *ci = const_cast<float *>(ctx_in);
    return elem_size == 2
        ? forward_core<int16_t>(e, sr, 1, B, N, static_cast<const int16_t *>(pcm), ld, ci, state, prob, 1, (hipStream_t)stream, ctx_out, present)
        : forward_core<float>(e, sr, 1, B, N, static_cast<const float *>(pcm), ld, ci, state, prob, 1, (hipStream_t)stream, ctx_out, present);
}

int vad_step_split(vad_engine *e, int sr, int B, const void *pcm, size_t elem_size, long ld, const float *ctx_in, float *ctx_out,
                   float *state, float *prob, void *stream) {
    if (e && !e->host_only && B > 0 && (!ctx_out || ctx_in == ctx_out))
        return fail(e, VAD_ERR_ARG, "vad_step_split: ctx_out must be a second, 16-byte aligned buffer");
    return vad_step_present(e, sr, B, pcm, elem_size, ld, ctx_in, ctx_out, state, prob, nullptr, stream);
}

int vad_step_host(vad_engine *e, int sr, int B, const void *host_pcm, size_t elem_size, void *dev_pcm, float *ctx, float *state,
                  float *dev_prob, float *host_prob, void *stream_v) {
    return vad_step_host_present(e, sr, B, host_pcm, elem_size, dev_pcm, ctx, state, dev_prob, host_prob, nullptr, nullptr, stream_v);
}

int vad_step_host_present(vad_engine *e, int sr, int B, const void *host_pcm, size_t elem_size, void *dev_pcm, float *ctx, float *state,
                          float *dev_prob, float *host_prob, const uint8_t *host_present, uint8_t *dev_present, void *stream_v) {
    if (!e) return VAD_ERR_ARG;
    if (e->host_only) return fail(e, VAD_ERR_NO_DEVICE, "host-only engine");
    const int ni = net_index(sr);
    if (ni < 0) return fail(e, VAD_ERR_SAMPLE_RATE, "Supported sampling rates: [8000, 16000]");
    if (B < 0 || (elem_size != 2 && elem_size != 4) || (B > 0 && (!host_pcm || !ctx || !state || !host_prob)) ||
        (host_present && !dev_present))
        return fail(e, VAD_ERR_ARG, "bad argument");
    if (host_present && e->impl_reference) return fail(e, VAD_ERR_OPTION, "impl=reference has no present[] form");
    if (B == 0) return VAD_OK;
    hipStream_t stream = (hipStream_t)stream_v;
    const long N = sr == 16000 ? 512 : 256;
    HIP_TRY(e, hipSetDevice(e->device));
    // dev_prob == NULL: the kernel stores the B probabilities straight into the page-locked host buffer (4 B per stream over the link,
    // visible to the host once the stream has passed the call) -- no device buffer, no second copy
    auto mapped = [&](int k, const void *hp) -> void * {
        const unsigned long gen = g_unmap_gen.load(std::memory_order_acquire);
        if (e->map_gen != gen) {                     // a range was unregistered since: what is remembered may point at nothing
            e->map_host[0] = e->map_host[1] = nullptr;
            e->map_gen = gen;
        }
        if (e->map_host[k] == hp && e->map_dev[k]) return e->map_dev[k];
        void *dv = nullptr;
        if (hipHostGetDevicePointer(&dv, const_cast<void *>(hp), 0) != hipSuccess || !dv) {
            (void)hipGetLastError();
            return nullptr;
        }
        e->map_host[k] = hp;
        e->map_dev[k] = dv;
        return dv;
    };
    float *out = dev_prob;
    if (!out) {
        out = static_cast<float *>(mapped(0, host_prob));
        if (!out) return fail(e, VAD_ERR_ARG, "vad_step_host: host_prob is not page-locked memory the runtime knows");
    }
    if (dev_pcm) {
        HIP_TRY(e, hipMemcpyAsync(dev_pcm, host_pcm, (size_t)B * N * elem_size, hipMemcpyHostToDevice, stream));
    } else {
        // dev_pcm == NULL: the kernel reads the chunks where they lie, through the device's view of the page-locked host buffer -- no copy
        // operation at all.  Right for a handful of streams (a B = 1 model call moves 2 KB: the copy engine's setup costs more than the
        // read), wrong for thousands (every lane's load would be a PCIe round trip).
        dev_pcm = mapped(1, host_pcm);
        if (!dev_pcm) return fail(e, VAD_ERR_ARG, "vad_step_host: host_pcm is not page-locked memory the runtime knows");
    }
    int rc;
    if (host_present) {
        HIP_TRY(e, hipMemcpyAsync(dev_present, host_present, (size_t)B, hipMemcpyHostToDevice, stream));
        rc = vad_step_present(e, sr, B, dev_pcm, elem_size, N, ctx, nullptr, state, out, dev_present, stream_v);
    } else {
        rc = elem_size == 2
            ? forward_impl<int16_t>(e, sr, B, N, static_cast<const int16_t *>(dev_pcm), N, ctx, state, out, 1, stream_v)
            : forward_impl<float>(e, sr, B, N, static_cast<const float *>(dev_pcm), N, ctx, state, out, 1, stream_v);
    }
    if (rc) return rc;
    if (dev_prob) HIP_TRY(e, hipMemcpyAsync(host_prob, dev_prob, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, stream));
    return VAD_OK;
}

// The blocking form: what `model(chunk, sr)` is to its caller.  The kernels store each stream's probability into the page-locked buffer
// as their LAST act (after the state and the context), so the host does not need the stream's completion signal -- the end-of-kernel
// cache release, the signal write and the runtime's handling of it are several microseconds of a 40 us call: the slots are filled with a
// bit pattern no kernel produces by itself, and the host reads them until every one has changed.  Bounded: a slot that has not changed
// after kSpinNs (a long batch, a debugger, or an input whose NaN payload is the pattern) is settled by hipStreamSynchronize.
namespace {
constexpr uint32_t kProbPending = 0xFFF0DEADu;       // a negative NaN with a payload: results are in [0, 1] or NaN computed on the device
constexpr long kSpinNs = 400 * 1000;
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}
}  // namespace

int vad_step_host_sync(vad_engine *e, int sr, int B, const void *host_pcm, size_t elem_size, float *ctx, float *state, float *host_prob,
                       void *stream_v) {
    if (!e) return VAD_ERR_ARG;
    if (B > 0 && host_prob) {
        volatile uint32_t *slot = reinterpret_cast<volatile uint32_t *>(host_prob);
        for (int b = 0; b < B; ++b) slot[b] = kProbPending;
        std::atomic_thread_fence(std::memory_order_seq_cst);     // the slots are marked before the launch is rung in
    }
    const int rc = vad_step_host_present(e, sr, B, host_pcm, elem_size, nullptr, ctx, state, nullptr, host_prob, nullptr, nullptr, stream_v);
    if (rc || B == 0) return rc;
    const volatile uint32_t *slot = reinterpret_cast<const volatile uint32_t *>(host_prob);
    const auto t0 = std::chrono::steady_clock::now();
    int b = 0;
    for (unsigned it = 0;; ++it) {
        while (b < B && slot[b] != kProbPending) ++b;
        if (b == B) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return VAD_OK;
        }
        cpu_relax();
        if ((it & 255) == 255 && std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > kSpinNs) break;
    }
    HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream_v));
    return VAD_OK;
}

int vad_forward_audio(vad_engine *e, int sr, int B, long L, const float *pcm, long ld, float *ctx,
                      float *state, float *probs, long ldp, void *stream) {
    return forward_impl<float>(e, sr, B, L, pcm, ld, ctx, state, probs, ldp, stream);
}

int vad_forward_audio_i16(vad_engine *e, int sr, int B, long L, const int16_t *pcm, long ld,
                          float *ctx, float *state, float *probs, long ldp, void *stream) {
    return forward_impl<int16_t>(e, sr, B, L, pcm, ld, ctx, state, probs, ldp, stream);
}

int vad_segment_probs_device(vad_engine *e, const float *probs, long ldp, const long *row_offsets, long n_streams,
                             const long *n_chunks,
                             long n_chunks_all, const long *audio_len, const vad_segment_params *p,
                             vad_segment *out, long cap_per_stream, long *counts, void *stream) {
    if (!e) return VAD_ERR_ARG;
    if (e->host_only) return fail(e, VAD_ERR_NO_DEVICE, "host-only engine");
    if (!p || n_streams < 0 || ldp < 0 || cap_per_stream < 0 || n_chunks_all < 0 ||
        (n_streams > 0 && (!probs || !audio_len || !counts || (cap_per_stream > 0 && !out))))
        return fail(e, VAD_ERR_ARG, "bad argument");
    if (p->sampling_rate != 8000 && p->sampling_rate != 16000)
        return fail(e, VAD_ERR_SAMPLE_RATE, "Currently silero VAD models support 8000 and 16000 (or multiply of 16000) sample rates");
    if (!n_chunks && !row_offsets && n_chunks_all > ldp) return fail(e, VAD_ERR_ARG, "n_chunks_all > ldp");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, vad::launch_scan(probs, ldp, row_offsets, n_streams, n_chunks, n_chunks_all, audio_len, *p, out, cap_per_stream,
                                counts, (hipStream_t)stream));
    return VAD_OK;
}

int vad_reserve(vad_engine *e, int sr, int B, long T) {
    if (!e) return VAD_ERR_ARG;
    if (e->host_only) return fail(e, VAD_ERR_NO_DEVICE, "host-only engine");
    if (sr > 16000 && sr % 16000 == 0) sr = 16000;       // a multiple of 16 kHz runs on the 16 kHz net (T counts 16 kHz chunks)
    if (net_index(sr) < 0) return fail(e, VAD_ERR_SAMPLE_RATE, "bad sr");
    if (B <= 0 || T <= 0) return fail(e, VAD_ERR_ARG, "bad argument");
    HIP_TRY(e, hipSetDevice(e->device));
    return ensure_scratch(e, sr, B, T, nullptr);
}

unsigned long vad_scratch_generation(const vad_engine *e) { return e ? e->scratch_gen : 0; }

size_t vad_scratch_bytes(const vad_engine *e) {
    return e ? (e->gx_floats + e->ctx_floats) * sizeof(float) : 0;
}

int vad_kernel_times(vad_engine *e, float *front_ms, float *rec_ms, long *calls) {
    if (!e) return VAD_ERR_ARG;
    if (e->ev_used == 0) return fail(e, VAD_ERR_OPTION, "no timings: set option profile=1 before the calls");
    HIP_TRY(e, hipSetDevice(e->device));
    float f = 0.f, r = 0.f;
    for (size_t i = 0; i + 2 < e->ev_used + 0 && i + 2 < e->ev_pool.size(); i += 3) {
        float a = 0.f, b = 0.f;
        HIP_TRY(e, hipEventSynchronize(e->ev_pool[i + 2]));
        HIP_TRY(e, hipEventElapsedTime(&a, e->ev_pool[i], e->ev_pool[i + 1]));
        HIP_TRY(e, hipEventElapsedTime(&b, e->ev_pool[i + 1], e->ev_pool[i + 2]));
        f += a;
        r += b;
    }
    if (front_ms) *front_ms = f;
    if (rec_ms) *rec_ms = r;
    if (calls) *calls = e->prof_calls;
    e->ev_used = 0;
    e->prof_calls = 0;
    return VAD_OK;
}

long vad_debug_packed_floats(const vad_engine *e, int sr, int which) {
    const int ni = net_index(sr);
    if (!e || ni < 0) return -1;
    const vad::PackedNet &p = e->weights->packed[ni];
    return which == 0 ? (long)p.front.size() : which == 1 ? (long)p.whh.size() : which == 2 ? (long)p.tables.size()
         : which == 5 ? (long)p.front_wino.size() : which == 6 ? (long)p.front_wino4.size()
         : which == 7 ? (long)p.whh_b9.size() / 2 : which == 8 ? (long)p.front_b9.size() / 2 : -1;
}

int vad_debug_packed_copy(const vad_engine *e, int sr, int which, float *dst, long n) {
    const int ni = net_index(sr);
    if (!e || ni < 0 || !dst) return VAD_ERR_ARG;
    const vad::PackedNet &p = e->weights->packed[ni];
    if (which == 7 || which == 8) {                 // three-piece bf16 images: raw 4-byte words holding two bf16 each
        const std::vector<uint16_t> &h = which == 7 ? p.whh_b9 : p.front_b9;
        if (n != (long)h.size() / 2) return VAD_ERR_ARG;
        std::memcpy(dst, h.data(), h.size() * sizeof(uint16_t));
        return VAD_OK;
    }
    const std::vector<float> *v = which == 0 ? &p.front : which == 1 ? &p.whh : which == 2 ? &p.tables
                                  : which == 5 ? &p.front_wino : which == 6 ? &p.front_wino4 : nullptr;
    if (!v || n != (long)v->size()) return VAD_ERR_ARG;
    std::memcpy(dst, v->data(), v->size() * sizeof(float));
    return VAD_OK;
}

int vad_debug_activation(vad_engine *e, int kind, const float *x, float *y, long n, void *stream) {
    if (!e || kind < 0 || kind > 1 || n < 0 || (n > 0 && (!x || !y))) return VAD_ERR_ARG;
    if (e->host_only) return fail(e, VAD_ERR_NO_DEVICE, "host-only engine");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, vad::launch_activation_probe(kind, x, y, n, (hipStream_t)stream));
    return VAD_OK;
}

int vad_streams_overlap(vad_engine *e, void *stream_a, void *stream_b) {
    if (!e) return -VAD_ERR_ARG;
    if (e->host_only) return -fail(e, VAD_ERR_NO_DEVICE, "host-only engine");
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    if (a == b) return 0;
    auto bad = [&](hipError_t rc, const char *what) { return -fail(e, VAD_ERR_HIP, std::string(what) + ": " + hipGetErrorString(rc)); };
    hipError_t rc;
    if ((rc = hipSetDevice(e->device)) != hipSuccess) return bad(rc, "hipSetDevice");
    hipEvent_t t0 = nullptr, ta = nullptr, tb = nullptr;
    if ((rc = hipEventCreate(&t0)) != hipSuccess || (rc = hipEventCreate(&ta)) != hipSuccess || (rc = hipEventCreate(&tb)) != hipSuccess) {
        if (t0) (void)hipEventDestroy(t0);
        if (ta) (void)hipEventDestroy(ta);
        return bad(rc, "hipEventCreate");
    }
    int verdict = -VAD_ERR_HIP;
    do {
        if ((rc = hipStreamSynchronize(a)) != hipSuccess || (rc = hipStreamSynchronize(b)) != hipSuccess) break;
        // TWO kernels of ~1 ms of dependent fp32 FMAs in ONE wave each on a (the second carries the in-order barrier of its stream, as
        // every kernel of a busy pipeline does); then one round of the same kernel on b.  If b has its own hardware queue its kernel is
        // done long before a's first; if the runtime put both streams on one queue, b's packet sits behind a's second one, which waits
        // for a's first: head-of-line blocking, exactly what makes an upload kernel alternate with a compute lane.
        if ((rc = hipEventRecord(t0, a)) != hipSuccess) break;
        if ((rc = vad::launch_foreign_spin(e->img->d_tables[0], 1, 60000, 1, a)) != hipSuccess) break;
        if ((rc = hipEventRecord(ta, a)) != hipSuccess) break;
        if ((rc = vad::launch_foreign_spin(e->img->d_tables[0], 1, 60000, 1, a)) != hipSuccess) break;
        if ((rc = vad::launch_foreign_spin(e->img->d_tables[0], 1, 1, 1, b)) != hipSuccess) break;
        if ((rc = hipEventRecord(tb, b)) != hipSuccess) break;
        if ((rc = hipEventSynchronize(ta)) != hipSuccess || (rc = hipEventSynchronize(tb)) != hipSuccess) break;
        float ms_a = 0.f, ms_b = 0.f;
        if ((rc = hipEventElapsedTime(&ms_a, t0, ta)) != hipSuccess || (rc = hipEventElapsedTime(&ms_b, t0, tb)) != hipSuccess) break;
        verdict = ms_b < 0.5f * ms_a ? 1 : 0;
    } while (false);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(ta);
    (void)hipEventDestroy(tb);
    if (verdict < 0) return bad(rc, "vad_streams_overlap");
    return verdict;
}

int vad_debug_foreign_load(vad_engine *e, int kind, int blocks, long iters, void *stream) {
    if (!e || kind < 0 || kind > 1 || blocks < 0 || iters < 0) return VAD_ERR_ARG;
    if (e->host_only) return fail(e, VAD_ERR_NO_DEVICE, "host-only engine");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, vad::launch_foreign_spin(e->img->d_tables[0], blocks, iters, kind, (hipStream_t)stream));
    return VAD_OK;
}

int vad_debug_frontend(vad_engine *e, int sr, int B, long L, const float *pcm, long ld,
                       const float *ctx, float *gx, void *stream_v) {
    if (!e) return VAD_ERR_ARG;
    if (e->host_only) return fail(e, VAD_ERR_NO_DEVICE, "host-only engine");
    const int ni = net_index(sr);
    if (ni < 0) return fail(e, VAD_ERR_SAMPLE_RATE, "bad sr");
    if (B <= 0 || L <= 0 || !pcm || !ctx || !gx) return fail(e, VAD_ERR_ARG, "bad argument");
    const int N = sr == 16000 ? 512 : 256;
    const long T = (L + N - 1) / N;
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(e, hipSetDevice(e->device));
    int rc = ensure_scratch(e, sr, B, T, stream);
    if (rc) return rc;
    if (e->slab_steps < T) return fail(e, VAD_ERR_ARG, "debug frontend: input too long");
    if (L % N || ((size_t)pcm & 15) || (ld * sizeof(float)) % 16 || ((size_t)ctx & 15))
        return fail(e, VAD_ERR_ARG, "debug frontend: whole chunks and 16-byte aligned rows only");
    vad::FrontArgs fa{};
#if VAD_AB
    fa.wfront = e->enc0 == 2 ? e->img->d_front4[ni] : e->enc0 == 1 ? e->img->d_front_wino[ni] : e->img->d_front[ni];
#else
    fa.wfront = e->img->d_front4[ni];
#endif
    fa.tables = e->img->d_tables[ni];
    fa.pcm = pcm;
    fa.ld = ld; fa.L = L; fa.T = T; fa.t0 = 0; fa.nt = T;
    fa.ctx_in = ctx;
    fa.ctx_out = nullptr;
    fa.gx = e->d_gx;
    fa.B = B;
    fa.trace = e->trace;
    const bool exact = e->exact_transitions && e->enc0 == 2 && !e->front_b9;
    fa.exact_net = exact ? e->img->d_ref + ni : nullptr;
    fa.gx_silent = exact && e->exact_all_silent ? e->img->d_gx_silent + 512 * ni : nullptr;
    if (e->front_b9) {
        fa.wfront = reinterpret_cast<const float *>(e->img->d_front_b9[ni]);
        HIP_TRY(e, vad::launch_front_b9<float>(sr, fa, stream));
    } else
#if VAD_AB
    if (e->enc0 == 1) HIP_TRY(e, vad::launch_front_wino<float>(sr, fa, stream));
    else if (e->enc0 == 0) HIP_TRY(e, vad::launch_front<float>(sr, fa, stream));
    else
#endif
    if (T == 1 && B <= e->one_max && e->enc0 == 2) HIP_TRY(e, vad::launch_front_one<float>(sr, fa, stream));
    else if ((long)((B + 15) / 16) * T <= e->lat_tiles) HIP_TRY(e, vad::launch_front_lat<float>(sr, fa, stream));
    else {
        fa.exact_list = exact ? e->d_exact : nullptr;
        HIP_TRY(e, vad::launch_front_f43<float>(sr, fa, stream));
        HIP_TRY(e, vad::launch_exact_fix<float>(sr, fa, stream));
    }
    HIP_TRY(e, vad::launch_unpack_gx(e->d_gx, gx, B, T, stream));
    return VAD_OK;
}


// ---- host-side ingest without a host-side copy ---------------------------------------------------------------------------
namespace {
struct Registered {
    size_t bytes;
    void *dev;
};
std::mutex g_reg_mutex;
std::map<const uint8_t *, Registered> g_registered;       // hipHostRegister'ed ranges and their device-side addresses

// device-visible address of a pinned host pointer, or nullptr if the runtime does not know it as page-locked memory: ranges
// registered through vad_host_register translate from the table; anything else (hipHostMalloc / torch pin_memory: identity-mapped;
// memory page-locked by somebody else's hipHostRegister: NOT necessarily) is resolved by the runtime -- the gather kernel must never
// dereference a host virtual address on faith.
const void *device_view(const void *p) {
    {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        if (!g_registered.empty()) {
            auto it = g_registered.upper_bound(static_cast<const uint8_t *>(p));
            if (it != g_registered.begin()) {
                --it;
                const size_t off = static_cast<const uint8_t *>(p) - it->first;
                if (off < it->second.bytes) return static_cast<const uint8_t *>(it->second.dev) + off;
            }
        }
    }
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, const_cast<void *>(p), 0) != hipSuccess || !dev) {
        (void)hipGetLastError();
        return nullptr;
    }
    return dev;
}
}  // namespace

int vad_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return VAD_ERR_ARG;
    if (hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        return VAD_ERR_HIP;
    }
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess || !dev) dev = p;
    std::lock_guard<std::mutex> g(g_reg_mutex);
    g_registered[static_cast<const uint8_t *>(p)] = Registered{bytes, dev};
    return VAD_OK;
}

int vad_host_unregister(void *p) {
    if (!p) return VAD_ERR_ARG;
    {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        g_registered.erase(static_cast<const uint8_t *>(p));
    }
    g_unmap_gen.fetch_add(1, std::memory_order_release);
    return hipHostUnregister(p) == hipSuccess ? VAD_OK : VAD_ERR_HIP;
}

int vad_upload_rows(vad_engine *e, const void *const *rows, const long *lens, long n, long width, size_t elem_size,
                    void *dst, int how, void *stream_v) {
    if (!e) return VAD_ERR_ARG;
    if (e->host_only) return fail(e, VAD_ERR_NO_DEVICE, "host-only engine");
    if (n < 0 || width < 0 || (elem_size != 2 && elem_size != 4) || how < 0 || how > 2)
        return fail(e, VAD_ERR_ARG, "bad argument");
    if (n == 0 || width == 0) return VAD_OK;
    if (!rows || !lens || !dst) return fail(e, VAD_ERR_ARG, "null pointer");
    if (((size_t)dst & 15) || (width * elem_size) % 16) return fail(e, VAD_ERR_ARG, "dst and its row pitch must be 16-byte aligned");
    for (long i = 0; i < n; ++i)
        if (lens[i] < 0 || lens[i] > width || (lens[i] > 0 && !rows[i])) return fail(e, VAD_ERR_ARG, "bad row");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(e, hipSetDevice(e->device));
    if (how == 0) {
        // copy engines: one H2D DMA per row (any alignment; no CU time), the padding by one fill of the batch first
        HIP_TRY(e, hipMemsetAsync(dst, 0, (size_t)n * width * elem_size, stream));
        // (one stream by default: dealing the copies to 2-4 streams reached 46 GB/s on one box and 29 on another, with the calls
        //  themselves blocking for milliseconds -- profiles/r04d_ingest_routes.md; SILERO_VAD_AMD_DMA_STREAMS=2..4 is the A/B)
        int ways = 1;
        if (const char *v = std::getenv("SILERO_VAD_AMD_DMA_STREAMS")) ways = std::max(1, std::min(vad_engine::kDmaStreams, std::atoi(v)));
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (n < 2 * ways || (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)) ways = 1;
        if (ways > 1) {
            if (!e->dma_fork) {
                // everything is created into locals and published only when complete (dma_fork, the "ready" flag, last): a failure
                // half way leaves the engine as it was -- the next call tries again -- instead of with null streams / events behind a
                // set flag.  (The side streams and events are per engine: one thread per engine, like every other call.)
                hipStream_t st[vad_engine::kDmaStreams] = {};
                hipEvent_t join[vad_engine::kDmaStreams] = {}, fork = nullptr;
                bool ok = hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess;
                for (int k = 0; ok && k < vad_engine::kDmaStreams; ++k)
                    ok = hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking) == hipSuccess &&
                         hipEventCreateWithFlags(&join[k], hipEventDisableTiming) == hipSuccess;
                if (!ok) {
                    for (int k = 0; k < vad_engine::kDmaStreams; ++k) {
                        if (st[k]) (void)hipStreamDestroy(st[k]);
                        if (join[k]) (void)hipEventDestroy(join[k]);
                    }
                    if (fork) (void)hipEventDestroy(fork);
                    (void)hipGetLastError();
                    return fail(e, VAD_ERR_HIP, "vad_upload_rows: cannot create the side streams of the per-row DMA route");
                }
                for (int k = 0; k < vad_engine::kDmaStreams; ++k) {
                    e->dma_stream[k] = st[k];
                    e->dma_join[k] = join[k];
                }
                e->dma_fork = fork;
            }
            HIP_TRY(e, hipEventRecord(e->dma_fork, stream));         // behind the fill (and whatever the caller queued before)
            for (int k = 0; k < ways; ++k) HIP_TRY(e, hipStreamWaitEvent(e->dma_stream[k], e->dma_fork, 0));
        }
        for (long i = 0; i < n; ++i)
            if (lens[i])
                HIP_TRY(e, hipMemcpyAsync(static_cast<uint8_t *>(dst) + (size_t)i * width * elem_size, rows[i],
                                          (size_t)lens[i] * elem_size, hipMemcpyHostToDevice,
                                          ways > 1 ? e->dma_stream[i % ways] : stream));
        if (ways > 1)
            for (int k = 0; k < ways; ++k) {
                HIP_TRY(e, hipEventRecord(e->dma_join[k], e->dma_stream[k]));
                HIP_TRY(e, hipStreamWaitEvent(stream, e->dma_join[k], 0));
            }
        return VAD_OK;
    }
    // gather kernel: the row table goes into a pinned slot the kernel reads itself; a slot is reused once its kernel is done
    const int slot = e->tab_next;
    e->tab_next = (slot + 1) % vad_engine::kTabSlots;
    if (e->tab_busy[slot]) {
        HIP_TRY(e, hipEventSynchronize(e->tab_ev[slot]));
        e->tab_busy[slot] = false;
    }
    if (e->tab_cap[slot] < n) {
        if (e->h_tab[slot]) (void)hipHostFree(e->h_tab[slot]);
        e->h_tab[slot] = nullptr;
        e->tab_cap[slot] = 0;
        // (hipHostFree waits for the device: a table that grows by a few rows per bucket would drain the whole pipeline every
        //  time -- capacities are powers of two from 4096 up, so a slot is reallocated a handful of times in a process' life)
        long cap = 4096;
        while (cap < n) cap *= 2;
        if (hipHostMalloc((void **)&e->h_tab[slot], (size_t)cap * sizeof(vad::RowDesc), hipHostMallocDefault) != hipSuccess)
            return fail(e, VAD_ERR_ALLOC, "cannot allocate the pinned row table");
        e->tab_cap[slot] = cap;
    }
    if (!e->tab_ev[slot]) HIP_TRY(e, hipEventCreateWithFlags(&e->tab_ev[slot], hipEventDisableTiming));
    if (how == 2) {                                    // rows[] are device addresses already
        for (long i = 0; i < n; ++i) e->h_tab[slot][i] = vad::RowDesc{lens[i] ? rows[i] : nullptr, lens[i]};
    } else {
        // Resolving a host address costs a runtime call (~1 us): thousands of short rows per slab, 92 slabs per 100 h of audio, made the
        // refill route's upload call 190 ms of its 245.  The rows of a corpus lie in a few page-locked allocations, and inside ONE
        // allocation the device view is the host address plus a constant: the allocation a row was resolved in is remembered (its
        // extent from hipMemGetAddressRange on the device view) and the rows that fall inside it are translated by arithmetic.
        const uint8_t *c_host = nullptr, *c_dev = nullptr;       // [c_host, c_host + c_bytes) -> c_dev + offset
        size_t c_bytes = 0;
        for (long i = 0; i < n; ++i) {
            const void *dv = nullptr;
            if (lens[i]) {
                const uint8_t *hp = static_cast<const uint8_t *>(rows[i]);
                if (c_bytes && hp >= c_host && hp + (size_t)lens[i] * elem_size <= c_host + c_bytes) {
                    dv = c_dev + (hp - c_host);
                } else {
                    dv = device_view(rows[i]);
                    if (!dv)
                        return fail(e, VAD_ERR_ARG, "vad_upload_rows: a row is not in page-locked memory the runtime knows "
                                                    "(hipHostMalloc / pin_memory / vad_host_register)");
                    hipDeviceptr_t base = nullptr;
                    size_t bytes = 0;
                    c_bytes = 0;
                    if (hipMemGetAddressRange(&base, &bytes, const_cast<void *>(dv)) == hipSuccess && base && bytes) {
                        const uint8_t *b = static_cast<const uint8_t *>(base), *d = static_cast<const uint8_t *>(dv);
                        if (d >= b && d < b + bytes) {
                            c_dev = b;
                            c_host = hp - (d - b);
                            c_bytes = bytes;
                        }
                    } else {
                        (void)hipGetLastError();
                    }
                }
            }
            e->h_tab[slot][i] = vad::RowDesc{dv, lens[i]};
        }
    }
    HIP_TRY(e, vad::launch_gather_rows(e->h_tab[slot], n, width, (int)elem_size, dst, how == 2, stream));
    HIP_TRY(e, hipEventRecord(e->tab_ev[slot], stream));
    e->tab_busy[slot] = true;
    return VAD_OK;
}

int vad_host_threads(void) { return vad::default_host_threads(256); }

int vad_bind_host_to_device(int device) {
    // CPUs of the GPU's NUMA node (sysfs), intersected with what the process may use; the calling thread is bound there,
    // threads created afterwards (the helper pool) inherit it, memory it allocates afterwards is first touched there
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), device) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    for (char *c = bdf; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
    char path[256];
    std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE *f = std::fopen(path, "r")) {
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    if (node < 0) return -1;
    std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = std::fopen(path, "r");
    if (!f) return -1;
    cpu_set_t cur, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(cur), &cur) != 0) {
        std::fclose(f);
        return -1;
    }
    int a = 0, b = 0, got = 0;
    while (std::fscanf(f, "%d", &a) == 1) {                   // "0-63,128-191"
        b = a;
        int c = std::fgetc(f);
        if (c == '-') {
            if (std::fscanf(f, "%d", &b) != 1) break;
            c = std::fgetc(f);
        }
        for (int k = a; k <= b && k < CPU_SETSIZE; ++k)
            if (CPU_ISSET(k, &cur)) {
                CPU_SET(k, &want);
                ++got;
            }
        if (c != ',') break;
    }
    std::fclose(f);
    if (got == 0) return -1;                                  // the node's CPUs are outside this process' mask: leave it alone
    if (sched_setaffinity(0, sizeof(want), &want) != 0) return -1;
    return node;
}

}  // extern "C"
