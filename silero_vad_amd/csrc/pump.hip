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
//   * the device batch has THREE buffers ([3][streams][N] int16): tick t + 3's copies wait BY EVENT for tick t's kernels
//     (batch_free[buffer]) before they overwrite what those read.  Three, not two: the link idles with only two ticks in flight (copy ->
//     kernel -> host -> next copy: 70 us in every 370, rocprofv3 copy trace), and with two buffers the third tick's copy would wait on the
//     device for an event that is still open when it is issued.  With three buffers and three ticks in flight no copy has an open
//     dependency: same-lease A/B +13 % at 8 kHz, +0.5-1.3 % at 16 kHz, never slower (profiles/r06_pump_three_buffers.md);
//   * the context is ping-ponged between two device buffers (vad_step_split: the kernel writes the next context beside the one it
//     reads), so a tick is exactly `parts` copies and `parts` kernels -- no D2D blit of the context, no D2H operation;
//   * a ring slot may be rewritten by its sources as soon as the tick that read it has been retired (vad_pump_poll), and is refused
//     (VAD_ERR_ARG) while that tick is in flight.
//
// Streams that have no chunk this tick (vad_pump_submit_present): in the reference a stream is stepped when ITS caller has a chunk
// (utils_vad.py:507-549: one model call per arrived chunk; silero-vad-onnx.cpp:335-390), so a live stream whose packet is late must
// come out of the tick untouched.  Every ring slot starts with a header of `streams` flag bytes (page-locked, in front of the audio,
// so that flags + audio are ONE H2D copy); a masked tick copies the header along, the step kernels skip the absent rows' (h, c) and
// probability, kernel_present.hip carries their contexts over (vad_step_present), and vad_pump_poll leaves their iterator counters
// alone.  An unmasked tick copies no header and runs exactly the kernels it always ran.
//
// COMPACT ticks (vad_pump_submit_compact): the absent streams' rows need not cross the link at all.  The sources write the chunks of
// the streams that deliver back to back at the start of the slot (row i = the i-th delivering stream, ascending); the tick copies
// [position table | flags | those rows] in ONE copy into a second pair of device buffers, and a row-expansion pass on the compute
// stream (kernel_present.hip expand_rows: HBM to HBM, ~8 MB at most, a few us) puts every row where the step kernels read it.  The
// link cost of a tick then falls with the delivery rate; everything behind the expansion is the masked tick, bit for bit.
//
// Waits block.  A source thread of a real server sleeps in its socket; the source threads of vad_pump_play, and its server loop, spin
// for at most 20 us on the counter they wait for and then sleep on it (futex), whatever the CPU budget: one of eight ranks under a
// 16-CPU quota has two CPUs for a server loop, a source thread and the HIP runtime's own threads, and a spinning (or yielding) thread
// there costs the tick it is waiting for (profiles/r06_pump_one_of_eight.md).
#include <hip/hip_runtime.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "../../include/silero_vad_hip.h"
#include "device_api.hpp"
#include "host_threads.hpp"

struct vad_pump {
    vad_engine *eng = nullptr;                   // a clone of the caller's engine: the pump's calls never touch the caller's scratch
    int device = 0, sr = 16000, N = 512, C = 64;
    int streams = 0, parts = 2, R = 4;
    std::vector<int> lo, hi;                     // part k = streams [lo[k], hi[k])
    double threshold = 0.5, min_silence = 1600, pad = 480;

    // a ring slot / a device batch buffer: [hpos bytes: int32 pos[streams], padded][hdr bytes: present[streams], padded][streams][N] int16
    // (the position table is written and copied by compact ticks only: it lies in FRONT of the flags so that a masked tick's one copy
    //  starts at the flags and a compact tick's one copy at the table)
    size_t hpos = 0, hdr = 0, slot_bytes = 0;
    uint8_t *h_ring = nullptr;                   // [R] slots, page-locked ingest ring
    float *h_prob = nullptr;                     // [R][streams]      page-locked, mapped: the kernels store here
    float *d_prob = nullptr;                     // device alias of h_prob
    static constexpr int NB = 3;                 // device batch buffers
    int nb = NB;                                 // ... in use (A/B knob SILERO_VAD_AMD_PUMP_BUFFERS=2: profiles/r06_pump_three_buffers.md)
    uint8_t *d_batch = nullptr;                  // [NB] device batch buffers (same layout as a ring slot)
    uint8_t *d_compact = nullptr;                // [NB] the same again: where a compact tick's copy lands
    float *d_ctx[2] = {nullptr, nullptr};        // [streams][C]      ping-pong
    std::vector<float *> d_state;                // per part: [2][hi - lo][128]
    hipStream_t copy[2] = {nullptr, nullptr}, compute = nullptr;    // copy[t & 1]: the copies of tick t
    std::vector<hipEvent_t> h2d_done[NB];        // [buffer][part]
    hipEvent_t batch_free[NB] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> tick_done;           // [R]
    bool batch_used[NB] = {false, false, false};

    long ticks = 0;                              // ticks submitted so far (tick t: batch buffer t % NB, copy stream t & 1, context t & 1 -> (t + 1) & 1)
    long retired = 0;                            // ticks retired so far (vad_pump_poll)
    struct Flight { int r; bool masked; };
    std::deque<Flight> inflight;                 // the submitted, not yet retired ticks, oldest first
    std::vector<uint8_t> slot_busy;              // [R]
    // VADIterator state of every stream (utils_vad.py:500-503)
    std::vector<uint8_t> active, triggered, feed_mask;
    std::vector<int64_t> temp_end, current;
    // open / close take effect on the HOST side (iterator reset, active flag) at the tick they were issued before: ticks submitted
    // earlier are still in flight and belong to the slot's previous occupant
    struct Op { long at_tick; int stream; bool open; };
    std::deque<Op> pending;
    std::vector<long> src_pos;                   // vad_pump_play with a presence pattern: chunks stream b has delivered so far
    bool poisoned = false;                       // a tick failed half-way: the carried state is no longer what any caller expects
    std::string err;

    int32_t *slot_pos(int r) const { return reinterpret_cast<int32_t *>(h_ring + (size_t)r * slot_bytes); }
    uint8_t *slot_present(int r) const { return h_ring + (size_t)r * slot_bytes + hpos; }
    int16_t *slot_pcm(int r) const { return reinterpret_cast<int16_t *>(h_ring + (size_t)r * slot_bytes + hpos + hdr); }
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

inline void cpu_relax() {
#if defined(__x86_64__)
    _mm_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

// dst <- src, `bytes` a multiple of 16, both 16-byte aligned on the destination side: streaming stores where the ISA has them (the
// data is bound for the DMA engine, not for this core's cache)
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void stream_copy_avx2(void *dst, const void *src, size_t bytes) {
    const __m256i *s = static_cast<const __m256i *>(src);
    __m256i *d = static_cast<__m256i *>(dst);
    for (size_t i = 0; i < bytes / 32; ++i) _mm256_stream_si256(d + i, _mm256_loadu_si256(s + i));
}
inline bool have_avx2() {
    static const bool v = __builtin_cpu_supports("avx2") && !std::getenv("SILERO_VAD_AMD_PUMP_SSE2");     // (A/B knob)
    return v;
}
inline bool want_prefetch() {
    static const bool v = !std::getenv("SILERO_VAD_AMD_PUMP_NO_PREFETCH");                                    // (A/B knob)
    return v;
}
#endif
// `next`: where the caller will read from next (the following stream's row, tens of KB away: no hardware prefetcher follows that) --
// its lines are requested while this row is being written
inline void stream_copy(void *dst, const void *src, size_t bytes, const void *next = nullptr) {
#if defined(__x86_64__)
    if (next && want_prefetch())
        for (size_t o = 0; o < bytes; o += 64) _mm_prefetch(static_cast<const char *>(next) + o, _MM_HINT_NTA);
    if (have_avx2() && bytes % 32 == 0 && (reinterpret_cast<size_t>(dst) & 31) == 0) return stream_copy_avx2(dst, src, bytes);
    const __m128i *s = static_cast<const __m128i *>(src);
    __m128i *d = static_cast<__m128i *>(dst);
    for (size_t i = 0; i < bytes / 16; ++i) _mm_stream_si128(d + i, _mm_loadu_si128(s + i));
#else
    (void)next;
    std::memcpy(dst, src, bytes);
#endif
}
inline void stream_fence() {
#if defined(__x86_64__)
    _mm_sfence();
#else
    std::atomic_thread_fence(std::memory_order_release);
#endif
}

// An event count: waiters spin for a bounded time on their own condition, then sleep in the kernel until somebody signals.  signal()
// is one atomic increment, plus a futex wake only if somebody sleeps.
struct Gate {
    std::atomic<uint32_t> seq{0};
    std::atomic<int> sleepers{0};
    static double spin_ms() {
        static const double v = [] {
            const char *s = std::getenv("SILERO_VAD_AMD_PUMP_SPIN_US");                                       // (A/B knob)
            return s ? std::atof(s) * 1e-3 : 0.020;
        }();
        return v;
    }

    template <class Cond>
    void wait(Cond cond) {
        const double t0 = now_ms(), limit = spin_ms();
        for (int i = 0;; ++i) {
            if (cond()) return;
            cpu_relax();
            if ((i & 63) == 63 && now_ms() - t0 > limit) break;
        }
        for (;;) {
            sleepers.fetch_add(1, std::memory_order_seq_cst);
            const uint32_t s = seq.load(std::memory_order_seq_cst);
            if (cond()) {
                sleepers.fetch_sub(1, std::memory_order_seq_cst);
                return;
            }
            // (a signal between the load of `seq` and here changes the word: the call returns at once)
            syscall(SYS_futex, reinterpret_cast<uint32_t *>(&seq), FUTEX_WAIT_PRIVATE, s, nullptr, nullptr, 0);
            sleepers.fetch_sub(1, std::memory_order_seq_cst);
        }
    }
    void signal() {
        seq.fetch_add(1, std::memory_order_seq_cst);
        if (sleepers.load(std::memory_order_seq_cst) > 0)
            syscall(SYS_futex, reinterpret_cast<uint32_t *>(&seq), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
    }
};

// host side of open / close: applied when the tick they were issued before is the next to retire (or at once if nothing is in flight)
void apply_ops(vad_pump *p) {
    while (!p->pending.empty() && p->pending.front().at_tick <= p->retired) {
        const vad_pump::Op op = p->pending.front();
        p->pending.pop_front();
        if (op.open) {
            p->active[op.stream] = 1;
            p->triggered[op.stream] = 0;
            p->temp_end[op.stream] = 0;
            p->current[op.stream] = 0;
            p->src_pos[op.stream] = 0;
        } else {
            p->active[op.stream] = 0;
        }
    }
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
    for (int b = 0; b < vad_pump::NB; ++b) {
        for (hipEvent_t ev : p->h2d_done[b]) (void)hipEventDestroy(ev);
        if (p->batch_free[b]) (void)hipEventDestroy(p->batch_free[b]);
    }
    for (int b = 0; b < 2; ++b)
        if (p->d_ctx[b]) (void)hipFree(p->d_ctx[b]);
    for (hipEvent_t ev : p->tick_done) (void)hipEventDestroy(ev);
    for (float *s : p->d_state) (void)hipFree(s);
    if (p->d_batch) (void)hipFree(p->d_batch);
    if (p->d_compact) (void)hipFree(p->d_compact);
    if (p->h_ring) (void)hipHostFree(p->h_ring);
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
    if (const char *v = std::getenv("SILERO_VAD_AMD_PUMP_BUFFERS")) p->nb = std::max(2, std::min(vad_pump::NB, std::atoi(v)));
    p->device = vad_device(e);
    if (p->device < 0 || vad_clone(e, &p->eng) != VAD_OK) return bail(VAD_ERR_NO_DEVICE);
    // the clone takes the caller's options with it; the pump steps through the product kernels whatever the caller's engine was set to
    // for its own A/B runs (impl=reference has no split-context step: every submit would fail)
    if (vad_set_option(p->eng, "impl", "mfma") != VAD_OK) return bail(VAD_ERR_OPTION);
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
    p->hdr = (S + 4095) / 4096 * 4096;
    p->hpos = (S * sizeof(int32_t) + 4095) / 4096 * 4096;
    p->slot_bytes = p->hpos + p->hdr + S * N * sizeof(int16_t);
    if (hipHostMalloc((void **)&p->h_ring, (size_t)p->R * p->slot_bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&p->h_prob, (size_t)p->R * S * sizeof(float), hipHostMallocMapped) != hipSuccess)
        return bail(VAD_ERR_ALLOC);
    std::memset(p->h_ring, 0, (size_t)p->R * p->slot_bytes);
    for (int r = 0; r < p->R; ++r) std::memset(p->slot_present(r), 1, S);
    std::memset(p->h_prob, 0, (size_t)p->R * S * sizeof(float));
    void *dv = nullptr;
    if (hipHostGetDevicePointer(&dv, p->h_prob, 0) != hipSuccess || !dv) return bail(VAD_ERR_HIP);
    p->d_prob = static_cast<float *>(dv);
    if (hipMalloc((void **)&p->d_batch, vad_pump::NB * p->slot_bytes) != hipSuccess ||
        hipMalloc((void **)&p->d_compact, vad_pump::NB * p->slot_bytes) != hipSuccess)
        return bail(VAD_ERR_ALLOC);
    // (compact ticks fill only the delivering streams' rows; rows no tick has filled yet are computed too: let them be silence)
    if (hipMemset(p->d_batch, 0, vad_pump::NB * p->slot_bytes) != hipSuccess) return bail(VAD_ERR_HIP);
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
    for (int b = 0; b < vad_pump::NB; ++b) {
        p->h2d_done[b].resize(p->parts);
        for (auto &ev : p->h2d_done[b])
            if (!mk(&ev)) return bail(VAD_ERR_HIP);
        if (!mk(&p->batch_free[b])) return bail(VAD_ERR_HIP);
    }
    p->tick_done.resize(p->R);
    // a process with CPUs to spare lets hipEventSynchronize spin on the tick's event (lowest latency); one with fewer than four (one of
    // eight ranks under a 16-CPU quota) sleeps on the interrupt instead: the CPU the spin would burn is the source thread's
    const bool blocking = vad::default_host_threads(256) < 4;
    for (auto &ev : p->tick_done)
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming | (blocking ? hipEventBlockingSync : 0)) != hipSuccess) return bail(VAD_ERR_HIP);
    if (vad_reserve(p->eng, p->sr, maxB, 1) != VAD_OK) return bail(VAD_ERR_ALLOC);
    if (hipDeviceSynchronize() != hipSuccess) return bail(VAD_ERR_HIP);
    p->slot_busy.assign(p->R, 0);
    p->active.assign(S, 1);
    p->feed_mask.assign(S, 1);
    p->triggered.assign(S, 0);
    p->temp_end.assign(S, 0);
    p->current.assign(S, 0);
    p->src_pos.assign(S, 0);
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

int16_t *vad_pump_slot(vad_pump *p, int r) { return (p && r >= 0 && r < p->R) ? p->slot_pcm(r) : nullptr; }

uint8_t *vad_pump_present(vad_pump *p, int r) { return (p && r >= 0 && r < p->R) ? p->slot_present(r) : nullptr; }

const float *vad_pump_probs(const vad_pump *p, int r) {
    return (p && r >= 0 && r < p->R) ? p->h_prob + (size_t)r * p->streams : nullptr;
}

}  // extern "C"

namespace {

// rows != nullptr: a compact tick whose rows lie in ARRIVAL order -- row i of the slot is the chunk of stream rows[i] (n_rows of them);
// flags and positions are built here.  rows == nullptr && compact: row i is the i-th stream (ascending) whose flag is set.
int submit_tick(vad_pump *p, int r, const uint8_t *present, bool compact, const int32_t *rows = nullptr, long n_rows = 0) {
    if (!p) return VAD_ERR_ARG;
    if (p->poisoned) return pfail(p, VAD_ERR_HIP, "the pump failed half-way through an earlier tick; destroy it (" + p->err + ")");
    if (r < 0 || r >= p->R) return pfail(p, VAD_ERR_ARG, "vad_pump_submit: no such ring slot");
    if (p->slot_busy[r]) return pfail(p, VAD_ERR_ARG, "vad_pump_submit: the slot's previous tick has not been retired (vad_pump_poll)");
    PUMP_TRY(p, hipSetDevice(p->device));
    const int buf = (int)(p->ticks % p->nb), pp = (int)(p->ticks & 1);
    const size_t S = (size_t)p->streams, N = (size_t)p->N, C = (size_t)p->C;
    if (rows != nullptr || n_rows != 0) {
        if (!rows || n_rows < 0 || n_rows > (long)S) return pfail(p, VAD_ERR_ARG, "vad_pump_submit_rows: bad row list");
        uint8_t *fl = p->slot_present(r);
        int32_t *pos = p->slot_pos(r);
        std::memset(fl, 0, S);
        for (long i = 0; i < n_rows; ++i) {
            const int32_t b = rows[i];
            if (b < 0 || (size_t)b >= S || fl[b]) {          // (validated before anything is queued: the slot's flags are scratch until then)
                std::memset(fl, 0, S);
                return pfail(p, VAD_ERR_ARG, "vad_pump_submit_rows: a stream out of range, or listed twice in one tick");
            }
            fl[b] = 1;
            pos[b] = (int32_t)i;
        }
        present = fl;
    }
    const bool masked = present != nullptr;
    if (compact && !masked) return pfail(p, VAD_ERR_ARG, "vad_pump_submit_compact: a compact tick needs its flags");
    if (masked && present != p->slot_present(r)) std::memcpy(p->slot_present(r), present, S);
    uint8_t *dbuf = p->d_batch + (size_t)buf * p->slot_bytes;
    int16_t *batch = reinterpret_cast<int16_t *>(dbuf + p->hpos + p->hdr);
    const uint8_t *d_present = masked ? dbuf + p->hpos : nullptr;
    uint8_t *cbuf = nullptr;                     // compact tick: where its one copy lands
    size_t n_present = 0;
    if (compact) {
        cbuf = p->d_compact + (size_t)buf * p->slot_bytes;
        d_present = cbuf + p->hpos;
        // the position table: row of the slot that holds stream b's chunk (the i-th delivering stream's chunk is row i)
        const uint8_t *fl = p->slot_present(r);
        int32_t *pos = p->slot_pos(r);
        if (rows != nullptr) {
            n_present = (size_t)n_rows;                  // (positions were written with the flags)
        } else {
            for (size_t b = 0; b < S; ++b) {
                pos[b] = (int32_t)n_present;
                n_present += fl[b] != 0;
            }
        }
    }
    const int16_t *src = p->slot_pcm(r);
    const float *ctx_in = p->d_ctx[pp];
    float *ctx_out = p->d_ctx[pp ^ 1];
    hipStream_t copy = p->copy[pp];
    // from the first queued operation on, a failure leaves the tick half-done: (h, c) of some parts advanced, the context ping-pong out
    // of step.  There is no retry that is right; the pump says so from then on.
    auto broken = [&](int code, const std::string &msg) {
        p->poisoned = true;
        return pfail(p, code, msg);
    };
#define TICK_TRY(expr)                                                                                  \
    do {                                                                                                \
        const hipError_t rc_ = (expr);                                                                  \
        if (rc_ != hipSuccess) return broken(VAD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(rc_)); \
    } while (0)
    // the copies may not overwrite the batch buffer before the kernels of two ticks ago have read it
    if (p->batch_used[buf]) TICK_TRY(hipStreamWaitEvent(copy, p->batch_free[buf], 0));
    if (compact) {
        // ONE copy whatever `parts` says: table + flags + the delivering streams' rows; the expansion pass on the compute stream puts
        // every row where the kernels read it (the batch buffer's previous readers are earlier on that stream)
        TICK_TRY(hipMemcpyAsync(cbuf, p->slot_pos(r), p->hpos + p->hdr + n_present * N * sizeof(int16_t), hipMemcpyHostToDevice, copy));
        TICK_TRY(hipEventRecord(p->h2d_done[buf][0], copy));
        TICK_TRY(hipStreamWaitEvent(p->compute, p->h2d_done[buf][0], 0));
        TICK_TRY(vad::launch_expand_rows(d_present, reinterpret_cast<const int32_t *>(cbuf), cbuf + p->hpos + p->hdr, batch,
                                         (long)(N * sizeof(int16_t)), p->streams, p->compute));
    }
    for (int k = 0; k < p->parts && !compact; ++k) {
        const size_t a = (size_t)p->lo[k], n = (size_t)(p->hi[k] - p->lo[k]);
        if (k == 0 && masked)        // the flags of ALL streams ride in front of part 0's audio: one copy (part 0 starts at stream 0)
            TICK_TRY(hipMemcpyAsync(dbuf + p->hpos, p->slot_present(r), p->hdr + n * N * sizeof(int16_t), hipMemcpyHostToDevice, copy));
        else
            TICK_TRY(hipMemcpyAsync(batch + a * N, src + a * N, n * N * sizeof(int16_t), hipMemcpyHostToDevice, copy));
        TICK_TRY(hipEventRecord(p->h2d_done[buf][k], copy));
    }
    for (int k = 0; k < p->parts; ++k) {
        const size_t a = (size_t)p->lo[k];
        const int n = p->hi[k] - p->lo[k];
        if (!compact) TICK_TRY(hipStreamWaitEvent(p->compute, p->h2d_done[buf][k], 0));
        if (k > 0 && masked && !compact) TICK_TRY(hipStreamWaitEvent(p->compute, p->h2d_done[buf][0], 0));     // (the flags came with part 0)
        const int rc = vad_step_present(p->eng, p->sr, n, batch + a * N, sizeof(int16_t), (long)N, ctx_in + a * C, ctx_out + a * C, p->d_state[k],
                                        p->d_prob + (size_t)r * S + a, masked ? d_present + a : nullptr, p->compute);
        if (rc != VAD_OK) return broken(rc, std::string("vad_step_present: ") + vad_last_error(p->eng));
    }
    TICK_TRY(hipEventRecord(p->batch_free[buf], p->compute));
    TICK_TRY(hipEventRecord(p->tick_done[r], p->compute));
#undef TICK_TRY
    p->batch_used[buf] = true;
    p->slot_busy[r] = 1;
    p->inflight.push_back(vad_pump::Flight{r, masked});
    ++p->ticks;
    return VAD_OK;
}

}  // namespace

extern "C" {

int vad_pump_submit_present(vad_pump *p, int r, const uint8_t *present) { return submit_tick(p, r, present, false); }

int vad_pump_submit_compact(vad_pump *p, int r, const uint8_t *present) { return submit_tick(p, r, present, true); }

int vad_pump_submit_rows(vad_pump *p, int r, const int32_t *stream_of_row, long n_rows) {
    if (!p) return VAD_ERR_ARG;
    if (!stream_of_row && n_rows != 0) return pfail(p, VAD_ERR_ARG, "vad_pump_submit_rows: bad row list");
    static const int32_t none = 0;
    return submit_tick(p, r, nullptr, true, stream_of_row ? stream_of_row : &none, n_rows);
}

int vad_pump_submit(vad_pump *p, int r) { return submit_tick(p, r, nullptr, false); }

long vad_pump_poll(vad_pump *p, int block, vad_iter_event *out, long cap, int *slot) {
    if (!p || cap < 0 || (cap > 0 && !out)) return VAD_PUMP_ERROR;
    if (p->inflight.empty()) return VAD_PUMP_IDLE;
    const vad_pump::Flight f = p->inflight.front();
    const int r = f.r;
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
    apply_ops(p);                                // opens / closes issued before this tick was submitted take effect with it
    const uint8_t *mask = p->active.data();
    if (f.masked) {                              // a stream without a chunk this tick: no model call, no iterator call (utils_vad.py:507-549)
        const uint8_t *pr = p->slot_present(r);
        for (int s = 0; s < p->streams; ++s) p->feed_mask[s] = p->active[s] & (pr[s] != 0);
        mask = p->feed_mask.data();
    }
    const long m = vad_iterator_feed(p->h_prob + (size_t)r * p->streams, mask, p->streams, p->N, p->threshold, p->min_silence, p->pad,
                                     p->triggered.data(), p->temp_end.data(), p->current.data(), out, cap);
    ++p->retired;
    if (p->inflight.empty()) {                   // nothing in flight: later opens / closes have nothing to wait for
        p->retired = p->ticks;
        apply_ops(p);
    }
    return m;
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
    // ... and the host side (iterator state, active flag) when those ticks have been retired: their probabilities belong to the slot's
    // previous occupant and must neither advance the new stream's sample counter nor open a segment for it
    p->pending.push_back(vad_pump::Op{p->ticks, stream, true});
    if (p->inflight.empty()) {
        p->retired = p->ticks;
        apply_ops(p);
    }
    return VAD_OK;
}

int vad_pump_close(vad_pump *p, int stream) {
    if (!p) return VAD_ERR_ARG;
    if (stream < 0 || stream >= p->streams) return pfail(p, VAD_ERR_ARG, "vad_pump_close: no such stream");
    // the slot is still computed (lock-step batch) but emits no events -- from the next tick submitted on: the ticks in flight carry
    // chunks the stream did deliver, their events are still its own
    p->pending.push_back(vad_pump::Op{p->ticks, stream, false});
    if (p->inflight.empty()) {
        p->retired = p->ticks;
        apply_ops(p);
    }
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
// Both sides BLOCK when they have to wait (Gate: 20 us of spinning, then a futex): the sources on the count of retired ticks, the
// server on the count of sources that have written the slot.
}  // extern "C"

namespace {

long play_loop(vad_pump *p, const int16_t *rows, long ld, long period, const uint8_t *pattern, long pattern_ticks, long first_tick,
               long n_ticks, int depth, int fill_threads, vad_iter_event *out, long cap, vad_pump_stats *st, bool compact) {
    if (!p) return VAD_PUMP_ERROR;
    const long N = p->N;
    if (!rows || ld < period || period < N || period % N || first_tick < 0 || n_ticks < 0 || cap < 0 || (cap > 0 && !out) ||
        (pattern && pattern_ticks <= 0)) {
        pfail(p, VAD_ERR_ARG, "vad_pump_play: bad argument");
        return VAD_PUMP_ERROR;
    }
    if (!p->inflight.empty()) {
        pfail(p, VAD_ERR_ARG, "vad_pump_play: ticks in flight (retire them with vad_pump_poll first)");
        return VAD_PUMP_ERROR;
    }
    depth = std::max(1, std::min(depth, p->R - 1));
    const long ahead = depth == 1 ? 1 : depth + 1;                             // <= R: the slot's previous tick has been retired by then
    const bool silent = fill_threads < 0;        // diagnostic: the sources write nothing (the slots keep their content): device side only
    int nsrc = fill_threads > 0 ? fill_threads : std::max(1, std::min(8, vad::default_host_threads(32) - 2));
    nsrc = std::max(1, std::min(nsrc, (p->streams + 63) / 64));
    const long per = ((p->streams + nsrc - 1) / nsrc + 15) / 16 * 16;          // whole tiles per source thread
    const long last = first_tick + n_ticks;
    std::atomic<long> retired{first_tick};
    std::vector<std::atomic<int>> written(p->R);
    for (auto &w : written) w.store(0);
    std::atomic<bool> stop{false};
    std::atomic<long> fill_ns{0}, chunks{0};
    Gate room, filled;                           // room: a tick was retired (sources wait); filled: a source finished a slot (server waits)
    auto source = [&](int k) {
        const long b0 = std::min<long>(p->streams, k * per), b1 = std::min<long>(p->streams, b0 + per);
        long mine = 0;
        for (long t = first_tick; t < last; ++t) {
            room.wait([&] { return t - retired.load(std::memory_order_acquire) < ahead || stop.load(std::memory_order_relaxed); });
            if (stop.load(std::memory_order_relaxed)) return;
            if (!silent && b1 > b0) {
                const double f0 = now_ms();
                const int r = (int)(t % p->R);
                int16_t *slot = p->slot_pcm(r);
                if (!pattern) {
                    const long off = (t * N) % period;
                    for (long b = b0; b < b1; ++b)
                        stream_copy(slot + b * N, rows + b * ld + off, (size_t)N * sizeof(int16_t), b + 1 < b1 ? rows + (b + 1) * ld + off : nullptr);
                    mine += b1 - b0;
                } else {
                    // stream b has a chunk this tick iff its flag says so; its audio advances only then (a late packet delays the
                    // stream's own next chunk, it does not skip audio)
                    const uint8_t *pat = pattern + (size_t)(t % pattern_ticks) * p->streams;
                    uint8_t *flags = p->slot_present(r);
                    // compact: the delivering streams' chunks lie back to back (vad_pump_submit_compact) -- this thread's first row
                    // is the number of streams before its range that deliver this tick
                    long row = 0;
                    if (compact)
                        for (long b = 0; b < b0; ++b) row += pat[b] != 0;
                    for (long b = b0; b < b1; ++b) {
                        flags[b] = pat[b];
                        if (!pat[b]) continue;
                        const long off = (p->src_pos[b]++ * N) % period;
                        stream_copy(slot + (compact ? row++ : b) * N, rows + b * ld + off, (size_t)N * sizeof(int16_t));
                        ++mine;
                    }
                }
                stream_fence();
                fill_ns.fetch_add((long)((now_ms() - f0) * 1e6), std::memory_order_relaxed);
            }
            written[t % p->R].fetch_add(1, std::memory_order_release);
            filled.signal();
        }
        chunks.fetch_add(mine, std::memory_order_relaxed);
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
        room.signal();
        return true;
    };
    const double t0 = now_ms();
    for (long t = first_tick; t < last && ok; ++t) {
        const int r = (int)(t % p->R);
        filled.wait([&] { return written[r].load(std::memory_order_acquire) >= nsrc; });
        written[r].store(0, std::memory_order_relaxed);          // (the slot's next writers wait for this tick's retirement)
        const double s0 = now_ms();
        t_written[r] = s0;
        ok = submit_tick(p, r, pattern && !silent ? p->slot_present(r) : nullptr, compact && pattern && !silent) == VAD_OK;
        submit_ms += now_ms() - s0;
        if (ok && (int)p->inflight.size() >= depth) ok = retire();
    }
    while (ok && !p->inflight.empty()) ok = retire();
    const double t1 = now_ms();
    stop.store(true);
    room.signal();
    for (auto &th : sources) th.join();
    if (!ok) {
        (void)hipStreamSynchronize(p->compute);                                  // leave nothing in flight behind an error
        while (!p->inflight.empty()) {
            p->slot_busy[p->inflight.front().r] = 0;
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
        st->chunks = silent ? (long)p->streams * n_ticks : chunks.load();
    }
    return n_events;
}

}  // namespace

extern "C" {

long vad_pump_play_gaps(vad_pump *p, const int16_t *rows, long ld, long period, const uint8_t *pattern, long pattern_ticks, long first_tick,
                        long n_ticks, int depth, int fill_threads, vad_iter_event *out, long cap, vad_pump_stats *st) {
    return play_loop(p, rows, ld, period, pattern, pattern_ticks, first_tick, n_ticks, depth, fill_threads, out, cap, st, false);
}

long vad_pump_play_compact(vad_pump *p, const int16_t *rows, long ld, long period, const uint8_t *pattern, long pattern_ticks, long first_tick,
                           long n_ticks, int depth, int fill_threads, vad_iter_event *out, long cap, vad_pump_stats *st) {
    return play_loop(p, rows, ld, period, pattern, pattern_ticks, first_tick, n_ticks, depth, fill_threads, out, cap, st, true);
}

long vad_pump_play(vad_pump *p, const int16_t *rows, long ld, long period, long first_tick, long n_ticks, int depth, int fill_threads,
                   vad_iter_event *out, long cap, vad_pump_stats *st) {
    return play_loop(p, rows, ld, period, nullptr, 0, first_tick, n_ticks, depth, fill_threads, out, cap, st, false);
}

}  // extern "C"
