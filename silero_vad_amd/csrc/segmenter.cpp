// segmenter.cpp -- host entry points of the hysteresis segmenter (scanner.hpp holds the scan itself, shared with
// the device kernel in kernel_scan.hip): speech probabilities -> speech segments, the post-processing half of
// get_speech_timestamps (reference src/silero_vad/utils_vad.py:338-450).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <queue>
#include <vector>

#include "../../include/silero_vad_hip.h"
#include "host_threads.hpp"
#include "scanner.hpp"

extern "C" void vad_segment_params_default(vad_segment_params *p, int sampling_rate) {
    if (!p) return;
    p->threshold = 0.5;
    p->neg_threshold = -1.0;
    p->sampling_rate = sampling_rate;
    p->min_speech_duration_ms = 250;
    p->max_speech_duration_s = std::numeric_limits<double>::infinity();
    p->min_silence_duration_ms = 100;
    p->speech_pad_ms = 30;
    p->min_silence_at_max_speech_ms = 98;
    p->use_max_poss_sil_at_max_speech = 1;
}

extern "C" long vad_segment_probs(const float *probs, long n, long audio_len,
                                  const vad_segment_params *p, vad_segment *out, long cap) {
    if (!p || n < 0 || audio_len < 0 || (n > 0 && !probs) || (cap > 0 && !out)) return -1;
    if (p->sampling_rate != 8000 && p->sampling_rate != 16000) return -2;
    vad::Scanner sc(*p, audio_len, out, cap);
    for (long i = 0; i < n; ++i) sc.feed(i, probs[i]);
    return sc.finish();
}

// Many streams at once: probs[i * ldp + t], t < n_chunks[i]; stream i's segments are written to
// out[i * cap_per_stream ...] and their number (may exceed cap_per_stream) to counts[i].  Streams
// are independent, so they are split over `threads` host threads (<= 0: hardware concurrency).
extern "C" long vad_segment_probs_batch(const float *probs, long ldp, long n_streams,
                                        const long *n_chunks, const long *audio_len,
                                        const vad_segment_params *p, vad_segment *out,
                                        long cap_per_stream, long *counts, int threads) {
    if (!p || n_streams < 0 || ldp < 0 || cap_per_stream < 0) return -1;
    if (n_streams > 0 && (!probs || !n_chunks || !audio_len || !counts)) return -1;
    if (cap_per_stream > 0 && !out) return -1;
    if (p->sampling_rate != 8000 && p->sampling_rate != 16000) return -2;
    for (long i = 0; i < n_streams; ++i)
        if (n_chunks[i] < 0 || n_chunks[i] > ldp || audio_len[i] < 0) return -1;
    int nt = threads > 0 ? threads : vad::default_host_threads(64);
    nt = (int)std::max<long>(1, std::min<long>(nt, n_streams / 64 + 1));
    auto work = [&](long lo, long hi) {
        for (long i = lo; i < hi; ++i)
            counts[i] = vad_segment_probs(probs + i * ldp, n_chunks[i], audio_len[i], p,
                                          out ? out + i * cap_per_stream : nullptr, cap_per_stream);
    };
    if (nt == 1) {
        work(0, n_streams);
    } else {
        const int blocks = (int)std::min<long>(n_streams, 4L * nt);
        const long per = (n_streams + blocks - 1) / blocks;
        vad::HostPool::get().run(nt, blocks, [&](int k) {
            const long lo = k * per, hi = std::min(n_streams, lo + per);
            if (lo < hi) work(lo, hi);
        });
    }
    long total = 0;
    for (long i = 0; i < n_streams; ++i) total += counts[i];
    return total;
}

// VADIterator (reference src/silero_vad/utils_vad.py:507-549) for every slot of a lock-step batch: one call advances all n streams
// by one chunk of `window` samples.  The reference's conventions: current_sample counts the END of the chunk, the entry test is
// p >= threshold, the exit threshold is fixed at threshold - 0.15 (in double, like Python), a pending end is dropped by the next
// loud chunk, positions are shifted back by one window and truncated like int().  Events come out in slot order.
extern "C" long vad_iterator_feed(const float *probs, const uint8_t *active, long n, int window, double threshold,
                                  double min_silence_samples, double speech_pad_samples, uint8_t *triggered,
                                  int64_t *temp_end, int64_t *current_sample, vad_iter_event *out, long cap) {
    if (n < 0 || window <= 0 || cap < 0 || (n > 0 && (!probs || !triggered || !temp_end || !current_sample)) || (cap > 0 && !out))
        return -1;
    const double exit_thr = threshold - 0.15;
    long m = 0;
    for (long s = 0; s < n; ++s) {
        if (active && !active[s]) continue;
        const double p = (double)probs[s];
        const int64_t cs = (current_sample[s] += window);
        const bool loud = p >= threshold;
        if (loud && temp_end[s]) temp_end[s] = 0;
        if (loud && !triggered[s]) {
            triggered[s] = 1;
            double start = (double)cs - speech_pad_samples - (double)window;
            if (start < 0) start = 0;
            if (m < cap) out[m] = vad_iter_event{(int32_t)s, 0, (int64_t)start};
            ++m;
            continue;
        }
        if (p < exit_thr && triggered[s]) {
            if (!temp_end[s]) temp_end[s] = cs;
            if ((double)(cs - temp_end[s]) < min_silence_samples) continue;
            const double end = (double)temp_end[s] + speech_pad_samples - (double)window;
            temp_end[s] = 0;
            triggered[s] = 0;
            if (m < cap) out[m] = vad_iter_event{(int32_t)s, 1, (int64_t)end};
            ++m;
        }
    }
    return m;
}

// The continuous-refill schedule (silero_vad_amd/streams.py RefillPlan): `n` recordings in admission order, recording q occupies a
// slot for need[q] slabs; at every slab boundary every free slot (lowest slot first) takes the next recording.  start[q] = the slab at
// which recording q is admitted, slot[q] = where.  (A heap of (slab at which the slot becomes free, slot): 150 000 recordings are a
// tenth of a second of Python and a millisecond here.)
extern "C" int vad_refill_schedule(const long *need, long n, long slots, long *start, long *slot) {
    if (n < 0 || slots <= 0 || (n > 0 && (!need || !start || !slot))) return VAD_ERR_ARG;
    std::priority_queue<std::pair<long, long>, std::vector<std::pair<long, long>>, std::greater<std::pair<long, long>>> free_at;
    for (long s = 0; s < slots; ++s) free_at.push({0, s});
    for (long q = 0; q < n; ++q) {
        if (need[q] < 0) return VAD_ERR_ARG;
        const std::pair<long, long> f = free_at.top();
        free_at.pop();
        start[q] = f.first;
        slot[q] = f.second;
        free_at.push({f.first + need[q], f.second});
    }
    return VAD_OK;
}

// The schedule as the table the stager walks: one row per (recording, slab it is active in) -- [slot, recording, first sample, samples,
// reset flag] -- ordered by slab, then by slot; cuts[k] .. cuts[k + 1] are slab k's rows.  Queue entry q is recording rec[q] of len[q]
// samples, admitted at start[q] into slot[q] for need[q] slabs of `width` samples (vad_refill_schedule).  A shard of 150 000 recordings
// has 1.3 M rows: built with numpy (repeat, stack, lexsort, gather) that was 60 ms in front of the first upload, here it is two counting
// passes -- slots in ascending order, a slot's recordings in admission order (which is their start order), so every slab's rows land
// in slot order by themselves.
extern "C" long vad_refill_table(const long *need, const long *start, const long *slot, const long *rec, const long *len, long n,
                                 long slots, long width, long n_slabs, long *rows, long *cuts) {
    if (n < 0 || slots <= 0 || width <= 0 || n_slabs < 0 || !cuts || (n > 0 && (!need || !start || !slot || !rec || !len || !rows))) return -VAD_ERR_ARG;
    std::vector<long> per_slab((size_t)n_slabs + 1, 0), first((size_t)slots + 1, 0), by_slot((size_t)n);
    for (long q = 0; q < n; ++q) {
        if (need[q] < 0 || start[q] < 0 || start[q] + need[q] > n_slabs || slot[q] < 0 || slot[q] >= slots) return -VAD_ERR_ARG;
        for (long j = 0; j < need[q]; ++j) ++per_slab[(size_t)(start[q] + j)];
        ++first[(size_t)slot[q] + 1];
    }
    long total = 0;
    for (long k = 0; k < n_slabs; ++k) {
        cuts[k] = total;
        total += per_slab[(size_t)k];
        per_slab[(size_t)k] = cuts[k];                       // from here: the next free row of slab k
    }
    cuts[n_slabs] = total;
    for (long s = 0; s < slots; ++s) first[(size_t)s + 1] += first[(size_t)s];
    for (long q = 0; q < n; ++q) by_slot[(size_t)first[(size_t)slot[q]]++] = q;      // stable: admission order inside a slot
    for (long i = 0; i < n; ++i) {
        const long q = by_slot[(size_t)i];
        for (long j = 0; j < need[q]; ++j) {
            long *r = rows + 5 * per_slab[(size_t)(start[q] + j)]++;
            const long at = j * width;
            r[0] = slot[q];
            r[1] = rec[q];
            r[2] = at;
            r[3] = std::min(width, len[q] - at);
            r[4] = j == 0;
        }
    }
    return total;
}
