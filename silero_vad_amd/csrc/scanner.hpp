// scanner.hpp -- the hysteresis segmenter: speech probabilities -> speech segments, written once for the host
// (segmenter.cpp, one stream per host thread slice) and for the device (kernel_scan.hip, one stream per lane).
//
// Semantics are those of the post-processing half of get_speech_timestamps
// (reference src/silero_vad/utils_vad.py:315-319 derived sample counts, :338-422 scan,
// :424-426 trailing segment, :428-440 padding pass; C++ twin of the scan in
// examples/cpp/silero-vad-onnx.cpp:196-331).  "Identical segments" is a graded parity criterion,
// so the arithmetic types follow the Python: derived sample counts are doubles, sample positions
// are integers, comparisons are strict exactly where the reference's are.
//
// The state is O(1): the reference keeps a list of candidate silences (`possible_ends`) but only ever asks for
// its first longest member (`max(possible_ends, key=...)`, utils_vad.py:369) and only ever clears it wholesale,
// so the running first-longest candidate is all that has to be remembered.  Segments go to a caller-provided
// buffer of `cap` entries; the scan keeps counting past `cap` (the caller grows the buffer and rescans).
#pragma once
#include <cstdint>

#include "../../include/silero_vad_hip.h"

#if defined(__HIPCC__)
#define VAD_HD __host__ __device__
#else
#define VAD_HD
#endif

namespace vad {

class Scanner {
public:
    VAD_HD Scanner(const vad_segment_params &p, int64_t audio_len, vad_segment *out, long cap)
        : audio_len_(audio_len), out_(out), cap_(cap) {
        sr_ = p.sampling_rate;
        win_ = sr_ == 16000 ? 512 : 256;
        enter_ = p.threshold;
        const double lo = p.threshold - 0.15;
        leave_ = p.neg_threshold >= 0.0 ? p.neg_threshold : (lo > 0.01 ? lo : 0.01);
        min_speech_ = (double)sr_ * p.min_speech_duration_ms / 1000.0;
        pad_ = (double)sr_ * p.speech_pad_ms / 1000.0;
        max_speech_ = (double)sr_ * p.max_speech_duration_s - win_ - 2.0 * pad_;
        min_sil_ = (double)sr_ * p.min_silence_duration_ms / 1000.0;
        min_sil_at_max_ = (double)sr_ * p.min_silence_at_max_speech_ms / 1000.0;
        longest_silence_cut_ = p.use_max_poss_sil_at_max_speech != 0;
    }

    VAD_HD void feed(int64_t index, float prob) {
        const int64_t pos = win_ * index;             // chunk START (utils_vad.py:349)
        const bool hot = (double)prob >= enter_;

        if (hot && pending_end_) {                      // speech resumed after a tentative end
            const int64_t gap = pos - pending_end_;
            if ((double)gap > min_sil_at_max_ && (!have_cut_ || gap > best_len_)) {
                have_cut_ = true;                       // first longest candidate silence of this run
                best_at_ = pending_end_;
                best_len_ = gap;
            }
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
            if ((double)(pending_end_ - start_) > min_speech_) emit(start_, pending_end_);
            clear_run();
            in_speech_ = false;
        }
    }

    // closes the trailing segment, runs the padding pass over the stored segments, returns the number found
    VAD_HD long finish() {
        if (has_start_ && (double)(audio_len_ - start_) > min_speech_) emit(start_, audio_len_);
        pad_segments();
        return n_;
    }

private:
    VAD_HD void emit(int64_t a, int64_t b) {
        if (n_ < cap_) {
            out_[n_].start = a;
            out_[n_].end = b;
        }
        ++n_;
    }

    // the run exceeded max_speech_duration_s; returns true if the chunk is fully handled
    VAD_HD bool split_overlong(int64_t pos) {
        if (longest_silence_cut_ && have_cut_) {
            const int64_t cut = best_at_, len = best_len_;
            emit(start_, cut);
            const int64_t restart = cut + len;
            if (restart < cut + pos) {                  // utils_vad.py:377 (as written there)
                start_ = restart;
            } else {
                in_speech_ = false;
                has_start_ = false;
            }
            last_cut_ = resume_at_ = pending_end_ = 0;
            have_cut_ = false;
            return false;
        }
        if (last_cut_) {
            emit(start_, last_cut_);
            if (resume_at_ < last_cut_) {
                in_speech_ = false;
                has_start_ = false;
            } else {
                start_ = resume_at_;
            }
            last_cut_ = resume_at_ = pending_end_ = 0;
            have_cut_ = false;
            return false;
        }
        emit(start_, pos);                              // hard cut at the current chunk
        clear_run();
        in_speech_ = false;
        return true;
    }

    VAD_HD void clear_run() {
        has_start_ = false;
        last_cut_ = resume_at_ = pending_end_ = 0;
        have_cut_ = false;
    }

    static VAD_HD int64_t floor_half(int64_t v) {     // Python's v // 2
        return (v >= 0) ? v / 2 : -((-v + 1) / 2);
    }

    VAD_HD void pad_segments() {
        const long n = n_ < cap_ ? n_ : cap_;
        for (long i = 0; i < n; ++i) {
            vad_segment &s = out_[i];
            if (i == 0) {
                const double v = (double)s.start - pad_;
                s.start = (int64_t)(v > 0.0 ? v : 0.0);
            }
            if (i + 1 < n) {
                vad_segment &nx = out_[i + 1];
                const int64_t gap = nx.start - s.end;
                if ((double)gap < 2.0 * pad_) {         // share a short gap at its midpoint
                    s.end += floor_half(gap);
                    const int64_t ns = nx.start - floor_half(gap);
                    nx.start = ns > 0 ? ns : 0;
                } else {
                    const double e = (double)s.end + pad_, b = (double)nx.start - pad_;
                    s.end = (int64_t)(e < (double)audio_len_ ? e : (double)audio_len_);
                    nx.start = (int64_t)(b > 0.0 ? b : 0.0);
                }
            } else {
                const double e = (double)s.end + pad_;
                s.end = (int64_t)(e < (double)audio_len_ ? e : (double)audio_len_);
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
    bool have_cut_ = false;     // running first-longest member of the reference's `possible_ends`
    int64_t best_at_ = 0, best_len_ = 0;

    vad_segment *out_;
    long cap_, n_ = 0;
};

}  // namespace vad
