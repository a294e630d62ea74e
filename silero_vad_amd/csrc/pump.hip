// pump.hip -- vad_pump: BASELINE configs[4] as a native object.  `streams` live streams on one GPU, one tick = one 32 ms chunk of every
// stream: int16 audio lies in a page-locked ingest ring the audio sources write into, a tick is H2D -> the fused step kernel ->
// probabilities stored by the kernel straight into page-locked host memory -> VADIterator logic of every stream -> events.  No
// Python, no torch: what the reference's native streaming clients are around ONNX Runtime -- a tight loop of session.run per chunk
// with explicit state and the iterator logic inline (examples/cpp/silero-vad-onnx.cpp:335-390; Python twin
// src/silero_vad/utils_vad.py:507-549) -- for thousands of streams in lock step.
//
// Overlap is EXPLICIT, not left to which hardware queue the runtime hands a stream:
//   * three named HIP streams.  `copy[0]` / `copy[1]` carry nothing but the H2D copies of the even / odd ticks -- the link is the
//     scarce resource (8.4 MB per tick of 8 192 16 kHz streams, 146 us at 57 GB/s, against ~66 us of kernel).  `compute` carries
//     nothing but the step kernels, in tick order (the carried state demands that order anyway).  Two copy streams because a copy
//     engine leaves the link idle for ~17 us between two dependent copies of ONE stream (measured: profiles/r05_pump.md, 0.80 of the
//     link with two copies per tick on one stream); copies of consecutive ticks have no dependence on each other, and with two or more
//     ticks in flight the second engine's copy is already moving while the first one's successor is being set up;
//   * a tick may be cut into `parts` sub-batches of whole 16-stream tiles.  Part k's kernel waits for part k's copy BY EVENT
//     (h2d_done[buffer][k]); part k + 1's copy is already running beside it.  One part is the default: a second copy costs another
//     setup gap and buys ~15 us of latency;
//   * the device batch is double-buffered ([2][streams][N] int16): tick t + 2's copies wait BY EVENT for tick t's kernels
//     (batch_free[buffer]) before they overwrite what those read;
//   * the context is ping-ponged between two device buffers (vad_step_split: the kernel writes the next context beside the one it
//     reads), so a tick is exactly `parts` copies and `parts` kernels -- no D2D blit of the context, no D2H operation;
//   * a ring slot may be rewritten by its sources as soon as the tick that read it has been retired (vad_pump_poll), and is refused
//     (VAD_ERR_ARG) while that tick is in flight.
#include <hip/hip_runtime.h>
#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "../../include/silero_vad_hip.h"
#include "host_threads.hpp"

struct vad_pump {
    vad_engine *eng = nullptr;                   // a clone of the caller's engine: the pump's calls never touch the caller's scratch
    int device = 0, sr = 16000, N = 512, C = 64;
    int streams = 0, parts = 2, R = 4;
    std::vector<int> lo, hi;                     // part k = streams [lo[k], hi[k])
    double threshold = 0.5, min_silence = 1600, pad = 480;

    int16_t *h_pcm = nullptr;                    // [R][streams][N]   page-locked ingest ring
    float *h_prob = nullptr;                     // [R][streams]      page-locked, mapped: the kernels store here
    float *d_prob = nullptr;                     // device alias of h_prob
    int16_t *d_pcm = nullptr;                    // [2][streams][N]   device batch, double-buffered
    float *d_ctx[2] = {nullptr, nullptr};        // [streams][C]      ping-pong
    std::vector<float *> d_state;                // per part: [2][hi - lo][128]
    hipStream_t copy[2] = {nullptr, nullptr}, compute = nullptr;    // copy[t & 1]: the copies of tick t
    std::vector<hipEvent_t> h2d_done[2];         // [buffer][part]
    hipEvent_t batch_free[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> tick_done;           // [R]
    bool batch_used[2] = {false, false};

    long ticks = 0;                              // ticks submitted so far (tick t: batch buffer t & 1, context t & 1 -> (t + 1) & 1)
    std::deque<int> inflight;                    // ring slots of the submitted, not yet retired ticks, oldest first
    std::vector<uint8_t> slot_busy;              // [R]
    // VADIterator state of every stream (utils_vad.py:500-503)
    std::vector<uint8_t> active, triggered;
    std::vector<int64_t> temp_end, current;
    std::string err;
};

namespace {

int pfail(vad_pump *p, int code, const std::string &msg) {
    if (p) p->err = msg;
    return code;
}
#define PUMP_TRY(p, expr)                                                                               \
    do {                                                                                                \
        const hipError_t rc_ = (expr);                                                                  \
        if (rc_ != hipSuccess) return pfail(p, VAD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(rc_)); \
    } while (0)

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

void vad_pump_params_default(vad_pump_params *p, int sampling_rate, int streams) {
    if (!p) return;
    p->sampling_rate = sampling_rate;
    p->streams = streams;
    p->parts = 0;
    p->ring_slots = 0;
    p->threshold = 0.5;
    p->min_silence_duration_ms = 100;
    p->speech_pad_ms = 30;
}

const char *vad_pump_last_error(const vad_pump *p) { return p ? p->err.c_str() : "null pump"; }

void vad_pump_destroy(vad_pump *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (hipStream_t cs : p->copy)
        if (cs) (void)hipStreamSynchronize(cs);
    if (p->compute) (void)hipStreamSynchronize(p->compute);
    for (int b = 0; b < 2; ++b) {
        for (hipEvent_t ev : p->h2d_done[b]) (void)hipEventDestroy(ev);
        if (p->batch_free[b]) (void)hipEventDestroy(p->batch_free[b]);
        if (p->d_ctx[b]) (void)hipFree(p->d_ctx[b]);
    }
    for (hipEvent_t ev : p->tick_done) (void)hipEventDestroy(ev);
    for (float *s : p->d_state) (void)hipFree(s);
    if (p->d_pcm) (void)hipFree(p->d_pcm);
    if (p->h_pcm) (void)hipHostFree(p->h_pcm);
    if (p->h_prob) (void)hipHostFree(p->h_prob);
    for (hipStream_t cs : p->copy)
        if (cs) (void)hipStreamDestroy(cs);
    if (p->compute) (void)hipStreamDestroy(p->compute);
    if (p->eng) vad_destroy(p->eng);
    delete p;
}

int vad_pump_create(vad_engine *e, const vad_pump_params *prm, vad_pump **out) {
    if (!out) return VAD_ERR_ARG;
    *out = nullptr;
    if (!e || !prm || prm->streams <= 0) return VAD_ERR_ARG;
    int N = 0, C = 0;
    if (vad_geometry(prm->sampling_rate, &N, &C) != VAD_OK) return VAD_ERR_SAMPLE_RATE;
    vad_pump *p = new (std::nothrow) vad_pump();
    if (!p) return VAD_ERR_ALLOC;
    auto bail = [&](int code) {
        vad_pump_destroy(p);
        return code;
    };
    p->sr = prm->sampling_rate;
    p->N = N;
    p->C = C;
    p->streams = prm->streams;
    p->R = prm->ring_slots > 0 ? std::max(2, prm->ring_slots) : 4;
    p->threshold = prm->threshold;
    p->min_silence = (double)p->sr * prm->min_silence_duration_ms / 1000.0;     // Python floats (utils_vad.py:494-498)
    p->pad = (double)p->sr * prm->speech_pad_ms / 1000.0;
    p->device = vad_device(e);
    if (p->device < 0 || vad_clone(e, &p->eng) != VAD_OK) return bail(VAD_ERR_NO_DEVICE);
    // parts of whole 16-stream tiles (a tile is the kernels' unit; rows of a part start 16-byte aligned)
    const int tiles = (p->streams + 15) / 16;
    const int parts = std::max(1, std::min(prm->parts > 0 ? prm->parts : 1, tiles));
    for (int k = 0; k < parts; ++k) {
        const int a = std::min(p->streams, (int)((long)tiles * k / parts) * 16), b = std::min(p->streams, (int)((long)tiles * (k + 1) / parts) * 16);
        if (b > a) {
            p->lo.push_back(a);
            p->hi.push_back(b);
        }
    }
    p->parts = (int)p->lo.size();
    if (hipSetDevice(p->device) != hipSuccess) return bail(VAD_ERR_HIP);
    const size_t S = (size_t)p->streams;
    if (hipHostMalloc((void **)&p->h_pcm, (size_t)p->R * S * N * sizeof(int16_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&p->h_prob, (size_t)p->R * S * sizeof(float), hipHostMallocMapped) != hipSuccess)
        return bail(VAD_ERR_ALLOC);
    std::memset(p->h_pcm, 0, (size_t)p->R * S * N * sizeof(int16_t));
    std::memset(p->h_prob, 0, (size_t)p->R * S * sizeof(float));
    void *dv = nullptr;
    if (hipHostGetDevicePointer(&dv, p->h_prob, 0) != hipSuccess || !dv) return bail(VAD_ERR_HIP);
    p->d_prob = static_cast<float *>(dv);
    if (hipMalloc((void **)&p->d_pcm, 2 * S * N * sizeof(int16_t)) != hipSuccess) return bail(VAD_ERR_ALLOC);
    for (int b = 0; b < 2; ++b) {
        if (hipMalloc((void **)&p->d_ctx[b], S * C * sizeof(float)) != hipSuccess) return bail(VAD_ERR_ALLOC);
        if (hipMemset(p->d_ctx[b], 0, S * C * sizeof(float)) != hipSuccess) return bail(VAD_ERR_HIP);
    }
    int maxB = 0;
    for (int k = 0; k < p->parts; ++k) {
        const size_t bytes = (size_t)2 * (p->hi[k] - p->lo[k]) * 128 * sizeof(float);
        float *st = nullptr;
        if (hipMalloc((void **)&st, bytes) != hipSuccess) return bail(VAD_ERR_ALLOC);
        p->d_state.push_back(st);
        if (hipMemset(st, 0, bytes) != hipSuccess) return bail(VAD_ERR_HIP);
        maxB = std::max(maxB, p->hi[k] - p->lo[k]);
    }
    if (hipStreamCreateWithFlags(&p->copy[0], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&p->copy[1], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&p->compute, hipStreamNonBlocking) != hipSuccess)
        return bail(VAD_ERR_HIP);
    auto mk = [&](hipEvent_t *ev) { return hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess; };
    for (int b = 0; b < 2; ++b) {
        p->h2d_done[b].resize(p->parts);
        for (auto &ev : p->h2d_done[b])
            if (!mk(&ev)) return bail(VAD_ERR_HIP);
        if (!mk(&p->batch_free[b])) return bail(VAD_ERR_HIP);
    }
    p->tick_done.resize(p->R);
    for (auto &ev : p->tick_done)
        if (!mk(&ev)) return bail(VAD_ERR_HIP);
    if (vad_reserve(p->eng, p->sr, maxB, 1) != VAD_OK) return bail(VAD_ERR_ALLOC);
    if (hipDeviceSynchronize() != hipSuccess) return bail(VAD_ERR_HIP);
    p->slot_busy.assign(p->R, 0);
    p->active.assign(S, 1);
    p->triggered.assign(S, 0);
    p->temp_end.assign(S, 0);
    p->current.assign(S, 0);
    *out = p;
    return VAD_OK;
}

int vad_pump_geometry(const vad_pump *p, int *streams, int *chunk, int *ring_slots, int *parts) {
    if (!p) return VAD_ERR_ARG;
    if (streams) *streams = p->streams;
    if (chunk) *chunk = p->N;
    if (ring_slots) *ring_slots = p->R;
    if (parts) *parts = p->parts;
    return VAD_OK;
}

int16_t *vad_pump_slot(vad_pump *p, int r) {
    return (p && r >= 0 && r < p->R) ? p->h_pcm + (size_t)r * p->streams * p->N : nullptr;
}

const float *vad_pump_probs(const vad_pump *p, int r) {
    return (p && r >= 0 && r < p->R) ? p->h_prob + (size_t)r * p->streams : nullptr;
}

int vad_pump_submit(vad_pump *p, int r) {
    if (!p) return VAD_ERR_ARG;
    if (r < 0 || r >= p->R) return pfail(p, VAD_ERR_ARG, "vad_pump_submit: no such ring slot");
    if (p->slot_busy[r]) return pfail(p, VAD_ERR_ARG, "vad_pump_submit: the slot's previous tick has not been retired (vad_pump_poll)");
    PUMP_TRY(p, hipSetDevice(p->device));
    const int buf = (int)(p->ticks & 1);
    const size_t S = (size_t)p->streams, N = (size_t)p->N, C = (size_t)p->C;
    int16_t *batch = p->d_pcm + (size_t)buf * S * N;
    const int16_t *src = p->h_pcm + (size_t)r * S * N;
    const float *ctx_in = p->d_ctx[buf];
    float *ctx_out = p->d_ctx[buf ^ 1];
    hipStream_t copy = p->copy[buf];
    // the copies may not overwrite the batch buffer before the kernels of two ticks ago have read it
    if (p->batch_used[buf]) PUMP_TRY(p, hipStreamWaitEvent(copy, p->batch_free[buf], 0));
    for (int k = 0; k < p->parts; ++k) {
        const size_t a = (size_t)p->lo[k], n = (size_t)(p->hi[k] - p->lo[k]);
        PUMP_TRY(p, hipMemcpyAsync(batch + a * N, src + a * N, n * N * sizeof(int16_t), hipMemcpyHostToDevice, copy));
        PUMP_TRY(p, hipEventRecord(p->h2d_done[buf][k], copy));
    }
    for (int k = 0; k < p->parts; ++k) {
        const size_t a = (size_t)p->lo[k];
        const int n = p->hi[k] - p->lo[k];
        PUMP_TRY(p, hipStreamWaitEvent(p->compute, p->h2d_done[buf][k], 0));
        const int rc = vad_step_split(p->eng, p->sr, n, batch + a * N, sizeof(int16_t), (long)N, ctx_in + a * C, ctx_out + a * C, p->d_state[k],
                                      p->d_prob + (size_t)r * S + a, p->compute);
        if (rc != VAD_OK) return pfail(p, rc, std::string("vad_step_split: ") + vad_last_error(p->eng));
    }
    PUMP_TRY(p, hipEventRecord(p->batch_free[buf], p->compute));
    PUMP_TRY(p, hipEventRecord(p->tick_done[r], p->compute));
    p->batch_used[buf] = true;
    p->slot_busy[r] = 1;
    p->inflight.push_back(r);
    ++p->ticks;
    return VAD_OK;
}

long vad_pump_poll(vad_pump *p, int block, vad_iter_event *out, long cap, int *slot) {
    if (!p || cap < 0 || (cap > 0 && !out)) return VAD_PUMP_ERROR;
    if (p->inflight.empty()) return VAD_PUMP_IDLE;
    const int r = p->inflight.front();
    if (block) {
        if (hipEventSynchronize(p->tick_done[r]) != hipSuccess) {
            pfail(p, VAD_ERR_HIP, "hipEventSynchronize(tick_done)");
            return VAD_PUMP_ERROR;
        }
    } else {
        const hipError_t q = hipEventQuery(p->tick_done[r]);
        if (q == hipErrorNotReady) return VAD_PUMP_BUSY;
        if (q != hipSuccess) {
            pfail(p, VAD_ERR_HIP, "hipEventQuery(tick_done)");
            return VAD_PUMP_ERROR;
        }
    }
    p->inflight.pop_front();
    p->slot_busy[r] = 0;
    if (slot) *slot = r;
    return vad_iterator_feed(p->h_prob + (size_t)r * p->streams, p->active.data(), p->streams, p->N, p->threshold, p->min_silence, p->pad,
                             p->triggered.data(), p->temp_end.data(), p->current.data(), out, cap);
}

int vad_pump_open(vad_pump *p, int stream) {
    if (!p) return VAD_ERR_ARG;
    if (stream < 0 || stream >= p->streams) return pfail(p, VAD_ERR_ARG, "vad_pump_open: no such stream");
    PUMP_TRY(p, hipSetDevice(p->device));
    // zero (h, c) and the context the NEXT tick reads, ordered behind the ticks already submitted (the compute stream)
    int k = 0;
    while (stream >= p->hi[k]) ++k;
    const size_t n = (size_t)(p->hi[k] - p->lo[k]), row = (size_t)(stream - p->lo[k]);
    PUMP_TRY(p, hipMemsetAsync(p->d_state[k] + row * 128, 0, 128 * sizeof(float), p->compute));
    PUMP_TRY(p, hipMemsetAsync(p->d_state[k] + (n + row) * 128, 0, 128 * sizeof(float), p->compute));
    PUMP_TRY(p, hipMemsetAsync(p->d_ctx[p->ticks & 1] + (size_t)stream * p->C, 0, (size_t)p->C * sizeof(float), p->compute));
    p->active[stream] = 1;
    p->triggered[stream] = 0;
    p->temp_end[stream] = 0;
    p->current[stream] = 0;
    return VAD_OK;
}

int vad_pump_close(vad_pump *p, int stream) {
    if (!p) return VAD_ERR_ARG;
    if (stream < 0 || stream >= p->streams) return pfail(p, VAD_ERR_ARG, "vad_pump_close: no such stream");
    p->active[stream] = 0;                    // the slot is still computed (lock-step batch); it emits no events
    return VAD_OK;
}

int vad_pump_state(vad_pump *p, int stream, float *h, float *c, float *ctx) {
    if (!p) return VAD_ERR_ARG;
    if (stream < 0 || stream >= p->streams) return pfail(p, VAD_ERR_ARG, "vad_pump_state: no such stream");
    PUMP_TRY(p, hipSetDevice(p->device));
    PUMP_TRY(p, hipStreamSynchronize(p->compute));
    int k = 0;
    while (stream >= p->hi[k]) ++k;
    const size_t n = (size_t)(p->hi[k] - p->lo[k]), row = (size_t)(stream - p->lo[k]);
    if (h) PUMP_TRY(p, hipMemcpy(h, p->d_state[k] + row * 128, 128 * sizeof(float), hipMemcpyDeviceToHost));
    if (c) PUMP_TRY(p, hipMemcpy(c, p->d_state[k] + (n + row) * 128, 128 * sizeof(float), hipMemcpyDeviceToHost));
    if (ctx) PUMP_TRY(p, hipMemcpy(ctx, p->d_ctx[p->ticks & 1] + (size_t)stream * p->C, (size_t)p->C * sizeof(float), hipMemcpyDeviceToHost));
    return VAD_OK;
}

// The whole loop, natively.  SOURCE threads play the part of the audio sources: thread k owns a range of streams and, for every tick,
// WRITES their chunks into the tick's ring slot (rows -> slot; streaming stores: the data is bound for the DMA engine, not for this
// core's cache) as soon as the server has room for the tick -- the memory traffic an audio server's receive path causes.  The
// calling thread is the server loop: wait until the slot is completely written, submit the tick, and once `depth` ticks are in
// flight retire the oldest (wait, iterator logic, events).  depth 1: strictly one tick at a time -- the next chunks are written
// after the previous tick's events are out, so "written -> events" is the latency of ONE tick.  depth >= 2: the sources write the
// tick that comes next while `depth` ticks are in flight (they run depth + 1 ticks ahead of the retired ones).
long vad_pump_play(vad_pump *p, const int16_t *rows, long ld, long period, long first_tick, long n_ticks, int depth, int fill_threads,
                   vad_iter_event *out, long cap, vad_pump_stats *st) {
    if (!p) return VAD_PUMP_ERROR;
    const long N = p->N;
    if (!rows || ld < period || period < N || period % N || first_tick < 0 || n_ticks < 0 || cap < 0 || (cap > 0 && !out)) {
        pfail(p, VAD_ERR_ARG, "vad_pump_play: bad argument");
        return VAD_PUMP_ERROR;
    }
    if (!p->inflight.empty()) {
        pfail(p, VAD_ERR_ARG, "vad_pump_play: ticks in flight (retire them with vad_pump_poll first)");
        return VAD_PUMP_ERROR;
    }
    depth = std::max(1, std::min(depth, p->R - 1));
    const long ahead = depth == 1 ? 1 : depth + 1;                             // <= R: the slot's previous tick has been retired by then
    // waits spin when this process has CPUs to spare and yield when it does not (one of eight ranks under a 16-CPU quota has two: nine
    // spinning threads per rank would get the whole node throttled, host_threads.hpp)
    const bool polite = vad::default_host_threads(256) < 4;
    const bool silent = fill_threads < 0;        // diagnostic: the sources write nothing (the slots keep their content): device side only
    int nsrc = fill_threads > 0 ? fill_threads : std::max(1, std::min(8, vad::default_host_threads(32) - 2));
    nsrc = std::max(1, std::min(nsrc, (p->streams + 63) / 64));
    const long per = ((p->streams + nsrc - 1) / nsrc + 15) / 16 * 16;          // whole tiles per source thread
    const long last = first_tick + n_ticks;
    std::atomic<long> retired{first_tick};
    std::vector<std::atomic<int>> written(p->R);
    for (auto &w : written) w.store(0);
    std::atomic<bool> stop{false};
    std::atomic<long> fill_ns{0};
    auto source = [&](int k) {
        const long b0 = std::min<long>(p->streams, k * per), b1 = std::min<long>(p->streams, b0 + per);
        for (long t = first_tick; t < last; ++t) {
            // (spin, do not yield: a sched_yield hands the CPU to whatever else is runnable there for a scheduler period -- measured as
            //  single 6 ms stalls of a tick, 1-4 % of a 0.2-0.5 s run; a source thread of an audio server would block on its socket instead)
            while (t - retired.load(std::memory_order_acquire) >= ahead) {
                if (stop.load(std::memory_order_relaxed)) return;
                if (polite) std::this_thread::yield();
                else _mm_pause();
            }
            if (!silent && b1 > b0) {
                const double f0 = now_ms();
                int16_t *slot = p->h_pcm + (size_t)(t % p->R) * p->streams * N;
                const long off = (t * N) % period;
                for (long b = b0; b < b1; ++b) {
                    const __m128i *src = reinterpret_cast<const __m128i *>(rows + b * ld + off);
                    __m128i *dst = reinterpret_cast<__m128i *>(slot + b * N);
                    for (long i = 0; i < N / 8; ++i) _mm_stream_si128(dst + i, _mm_loadu_si128(src + i));
                }
                _mm_sfence();
                fill_ns.fetch_add((long)((now_ms() - f0) * 1e6), std::memory_order_relaxed);
            }
            written[t % p->R].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> sources;
    for (int k = 0; k < nsrc; ++k) sources.emplace_back(source, k);
    std::vector<double> t_written(p->R, 0.0), lat;
    lat.reserve((size_t)n_ticks);
    std::vector<vad_iter_event> scratch((size_t)p->streams);
    long n_events = 0;
    double wait_ms = 0.0, submit_ms = 0.0;
    bool ok = true;
    auto retire = [&]() -> bool {
        int r = -1;
        const double w0 = now_ms();
        const long m = vad_pump_poll(p, 1, scratch.data(), (long)scratch.size(), &r);
        const double w1 = now_ms();
        if (m < 0) return false;
        wait_ms += w1 - w0;
        lat.push_back(w1 - t_written[r]);
        for (long i = 0; i < m; ++i, ++n_events)
            if (n_events < cap) out[n_events] = scratch[i];
        retired.fetch_add(1, std::memory_order_release);
        return true;
    };
    const double t0 = now_ms();
    for (long t = first_tick; t < last && ok; ++t) {
        const int r = (int)(t % p->R);
        while (written[r].load(std::memory_order_acquire) < nsrc) {
            if (polite) std::this_thread::yield();
            else _mm_pause();
        }
        written[r].store(0, std::memory_order_relaxed);          // (the slot's next writers wait for this tick's retirement)
        const double s0 = now_ms();
        t_written[r] = s0;
        ok = vad_pump_submit(p, r) == VAD_OK;
        submit_ms += now_ms() - s0;
        if (ok && (int)p->inflight.size() >= depth) ok = retire();
    }
    while (ok && !p->inflight.empty()) ok = retire();
    const double t1 = now_ms();
    stop.store(true);
    for (auto &th : sources) th.join();
    if (!ok) {
        (void)hipStreamSynchronize(p->compute);                                  // leave nothing in flight behind an error
        while (!p->inflight.empty()) {
            p->slot_busy[p->inflight.front()] = 0;
            p->inflight.pop_front();
        }
        return VAD_PUMP_ERROR;
    }
    if (st) {
        std::sort(lat.begin(), lat.end());
        const size_t n = lat.size();
        st->ticks = n_ticks;
        st->events = n_events;
        st->wall_ms = t1 - t0;
        st->tick_ms_p50 = n ? lat[n / 2] : 0.0;
        st->tick_ms_p95 = n ? lat[std::min(n - 1, (size_t)(n * 0.95))] : 0.0;
        st->tick_ms_max = n ? lat[n - 1] : 0.0;
        st->fill_ms_mean = n_ticks ? (double)fill_ns.load() * 1e-6 / nsrc / n_ticks : 0.0;
        st->submit_ms_mean = n_ticks ? submit_ms / n_ticks : 0.0;
        st->wait_ms_mean = n_ticks ? wait_ms / n_ticks : 0.0;
        st->fill_threads = silent ? 0 : nsrc;
        st->depth = depth;
    }
    return n_events;
}

}  // extern "C"
