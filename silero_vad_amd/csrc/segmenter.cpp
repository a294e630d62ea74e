// segmenter.cpp -- host-side hysteresis segmenter: speech probabilities -> speech segments.
//
// Semantics are those of the post-processing half of get_speech_timestamps
// (reference src/silero_vad/utils_vad.py:315-319 derived sample counts, :338-422 scan,
// :424-426 trailing segment, :428-440 padding pass; C++ twin of the scan in
// examples/cpp/silero-vad-onnx.cpp:196-331).  "Identical segments" is a graded parity criterion,
// so the arithmetic types follow the Python: derived sample counts are doubles, sample positions
// are integers, comparisons are strict exactly where the reference's are.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/silero_vad_hip.h"

namespace {

struct Candidate { int64_t at; int64_t len; };   // a silence inside the current speech run

inline int64_t floor_half(int64_t v) {            // Python's v // 2
    return (v >= 0) ? v / 2 : -((-v + 1) / 2);
}

class Scanner {
public:
    Scanner(const vad_segment_params &p, int64_t audio_len) : audio_len_(audio_len) {
        sr_ = p.sampling_rate;
        win_ = sr_ == 16000 ? 512 : 256;
        enter_ = p.threshold;
        leave_ = p.neg_threshold >= 0.0 ? p.neg_threshold : std::max(p.threshold - 0.15, 0.01);
        min_speech_ = (double)sr_ * p.min_speech_duration_ms / 1000.0;
        pad_ = (double)sr_ * p.speech_pad_ms / 1000.0;
        max_speech_ = (double)sr_ * p.max_speech_duration_s - win_ - 2.0 * pad_;
        min_sil_ = (double)sr_ * p.min_silence_duration_ms / 1000.0;
        min_sil_at_max_ = (double)sr_ * p.min_silence_at_max_speech_ms / 1000.0;
        longest_silence_cut_ = p.use_max_poss_sil_at_max_speech != 0;
    }

    void feed(int64_t index, float prob) {
        const int64_t pos = win_ * index;             // chunk START (utils_vad.py:349)
        const bool hot = (double)prob >= enter_;

        if (hot && pending_end_) {                      // speech resumed after a tentative end
            const int64_t gap = pos - pending_end_;
            if ((double)gap > min_sil_at_max_) cuts_.push_back({pending_end_, gap});
            pending_end_ = 0;
            if (resume_at_ < last_cut_) resume_at_ = pos;
        }
        if (hot && !in_speech_) {                       // a run starts; nothing else this chunk
            in_speech_ = true;
            has_start_ = true;
            start_ = pos;
            return;
        }
        if (in_speech_ && (double)(pos - start_) > max_speech_) {
            if (split_overlong(pos)) return;
        }
        if ((double)prob < leave_ && in_speech_) {
            if (!pending_end_) pending_end_ = pos;
            const int64_t quiet = pos - pending_end_;
            if (!longest_silence_cut_ && (double)quiet > min_sil_at_max_) last_cut_ = pending_end_;
            if ((double)quiet < min_sil_) return;       // not silent for long enough yet
            if ((double)(pending_end_ - start_) > min_speech_) out_.push_back({start_, pending_end_});
            clear_run();
            in_speech_ = false;
        }
    }

    std::vector<vad_segment> finish() {
        if (has_start_ && (double)(audio_len_ - start_) > min_speech_)
            out_.push_back({start_, audio_len_});
        pad_segments();
        return out_;
    }

private:
    // the run exceeded max_speech_duration_s; returns true if the chunk is fully handled
    bool split_overlong(int64_t pos) {
        if (longest_silence_cut_ && !cuts_.empty()) {
            // first longest candidate (Python max() keeps the earliest of equals)
            const Candidate *best = &cuts_[0];
            for (const auto &c : cuts_)
                if (c.len > best->len) best = &c;
            const int64_t cut = best->at, len = best->len;
            out_.push_back({start_, cut});
            const int64_t restart = cut + len;
            if (restart < cut + pos) {                  // utils_vad.py:377 (as written there)
                start_ = restart;
            } else {
                in_speech_ = false;
                has_start_ = false;
            }
            last_cut_ = resume_at_ = pending_end_ = 0;
            cuts_.clear();
            return false;
        }
        if (last_cut_) {
            out_.push_back({start_, last_cut_});
            if (resume_at_ < last_cut_) {
                in_speech_ = false;
                has_start_ = false;
            } else {
                start_ = resume_at_;
            }
            last_cut_ = resume_at_ = pending_end_ = 0;
            cuts_.clear();
            return false;
        }
        out_.push_back({start_, pos});                  // hard cut at the current chunk
        clear_run();
        in_speech_ = false;
        return true;
    }

    void clear_run() {
        has_start_ = false;
        last_cut_ = resume_at_ = pending_end_ = 0;
        cuts_.clear();
    }

    void pad_segments() {
        const size_t n = out_.size();
        for (size_t i = 0; i < n; ++i) {
            vad_segment &s = out_[i];
            if (i == 0) s.start = (int64_t)std::max(0.0, (double)s.start - pad_);
            if (i + 1 < n) {
                vad_segment &nx = out_[i + 1];
                const int64_t gap = nx.start - s.end;
                if ((double)gap < 2.0 * pad_) {         // share a short gap at its midpoint
                    s.end += floor_half(gap);
                    nx.start = std::max<int64_t>(0, nx.start - floor_half(gap));
                } else {
                    s.end = (int64_t)std::min((double)audio_len_, (double)s.end + pad_);
                    nx.start = (int64_t)std::max(0.0, (double)nx.start - pad_);
                }
            } else {
                s.end = (int64_t)std::min((double)audio_len_, (double)s.end + pad_);
            }
        }
    }

    int sr_ = 16000;
    int64_t win_ = 512, audio_len_ = 0;
    double enter_ = 0.5, leave_ = 0.35;
    double min_speech_ = 0, pad_ = 0, max_speech_ = 0, min_sil_ = 0, min_sil_at_max_ = 0;
    bool longest_silence_cut_ = true;

    bool in_speech_ = false, has_start_ = false;
    int64_t start_ = 0;
    int64_t pending_end_ = 0;   // 0 doubles as "none", as in the reference
    int64_t last_cut_ = 0, resume_at_ = 0;
    std::vector<Candidate> cuts_;
    std::vector<vad_segment> out_;
};

}  // namespace

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
    Scanner sc(*p, audio_len);
    for (long i = 0; i < n; ++i) sc.feed(i, probs[i]);
    const std::vector<vad_segment> segs = sc.finish();
    const long m = (long)segs.size();
    for (long i = 0; i < std::min(m, cap); ++i) out[i] = segs[(size_t)i];
    return m;
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
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nt = (int)std::max<long>(1, std::min<long>(nt, n_streams / 64 + 1));
    auto work = [&](long lo, long hi) {
        for (long i = lo; i < hi; ++i)
            counts[i] = vad_segment_probs(probs + i * ldp, n_chunks[i], audio_len[i], p,
                                          out ? out + i * cap_per_stream : nullptr, cap_per_stream);
    };
    if (nt == 1) {
        work(0, n_streams);
    } else {
        std::vector<std::thread> pool;
        const long per = (n_streams + nt - 1) / nt;
        for (int k = 0; k < nt; ++k) {
            const long lo = k * per, hi = std::min(n_streams, lo + per);
            if (lo < hi) pool.emplace_back(work, lo, hi);
        }
        for (auto &th : pool) th.join();
    }
    long total = 0;
    for (long i = 0; i < n_streams; ++i) total += counts[i];
    return total;
}
