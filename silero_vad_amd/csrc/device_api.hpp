// device_api.hpp -- launcher declarations shared by engine.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/silero_vad_hip.h"

namespace vad {

// device pointers to the canonical (un-packed) tensors of one net, for impl="reference"
struct RefNet {
    const float *basis;
    const float *ew[4], *eb[4];
    const float *w_ih, *w_hh, *b_ih, *b_hh, *w_out, *b_out;
};

__device__ __forceinline__ float load_pcm(const float *p) { return *p; }
__device__ __forceinline__ float load_pcm(const int16_t *p) { return (float)(*p) * (1.0f / 32768.0f); }

template <typename PcmT>
hipError_t launch_ref_forward(const RefNet &w, int sr, int B, long L, const PcmT *pcm, long ld,
                              float *ctx, float *state, float *probs, long ldp, hipStream_t s);

// dst[b][i] = src[b][i * k] (the reference's x[:, ::step] for 32/48/... kHz input)
template <typename PcmT>
hipError_t launch_decimate(const PcmT *src, long lds, PcmT *dst, long ldd, int B, long Ld, int k, hipStream_t s);

// ---- product path --------------------------------------------------------------------------------
// Frontend: PCM -> in-wave FFT magnitude -> 4 conv blocks -> W_ih GEMM, fp32 MFMA.
//   one wave = 16 chunks (16 streams x one time step); gx is written in MFMA D-fragment order
//   gx[stream_tile][t][mblock 32][lane 64][4]   (stream_tile = 16 consecutive streams)
// Also writes ctx_out[b][C] = last C samples of the (zero padded) input of every stream.
struct FrontArgs {
    const float *wfront;     // packed GEMM stream (layout.hpp)
    const float *tables;     // biases, window, twiddles
    const void *pcm;         // [B][L] float or int16, rows 16-byte aligned
    const void *tail;        // [B][N] zero padded copy of the last chunk when L % N != 0, else null
    long ld, L, T;           // row stride (elements), samples per stream, chunks per stream
    long t0, nt;             // time slab [t0, t0 + nt) processed by this launch
    const float *ctx_in;     // [B][C] context of chunk 0 (read when t == 0)
    float *ctx_out;          // [B][C] written by the waves that own t == T-1 (may alias ctx_in)
    float *gx;               // scratch, tile-major, indexed with slab-relative t
    int B;
    int dec;                 // 0/1: pcm is at the net's rate; 2, 3: pcm is at 32 / 48 kHz and ld, pcm, tail index RAW samples --
                             // the kernel reads every dec-th one (fp32 frontends; 3: kernel_front_f43.hip only); L, T, t0, nt stay
                             // in 16 kHz terms
    long long *trace;        // bring-up only (VAD_TRACE builds): 16 slots per workgroup, else null
    // chunks that hold an exactly silent STFT frame beside one that is not (exact_front.hpp) get their gx from a double-precision
    // evaluation.  exact_net: the canonical fp32 tensors (device copy of RefNet), or null = the feature is off.  The latency frontend
    // computes such chunks itself; the throughput frontend appends their ids ((st * nt + tl) * 16 + j) to exact_list[1 + *exact_list],
    // and launch_exact_fix overwrites their columns of gx.  exact_list: [0] count, [1] workgroups done (both zero between launches),
    // [2 ...] ids.
    const RefNet *exact_net;
    int *exact_list;
    const float *gx_silent;  // [512] gx of a chunk of zeros with zero context (a constant of the net), or null: all-silent chunks keep the
                             // fp32 chains' value (option exact_transitions=edges; study mode)
};
// gx_silent of one net: exact_front.hpp's double-precision evaluation of a chunk of zeros, 512 floats to `out` (device).
hipError_t launch_exact_silent(int sr, const RefNet *net_dev, float *out, hipStream_t s);
// Behind a throughput-frontend launch with a.exact_list set: recompute the listed chunks (exact_front.hpp) into a.gx; resets the list.
template <typename PcmT>
hipError_t launch_exact_fix(int sr, const FrontArgs &a, hipStream_t s);
// (A/B form, test builds only -- VAD_AB, libsilero_vad_hip_ab.so: encoder 0 tap by tap, straight-line code, kernel_front.hip)
template <typename PcmT>
hipError_t launch_front(int sr, const FrontArgs &a, hipStream_t s);
// Same function, encoder 0 as one Winograd F(4,3) tile over the 4 frames, loop-structured code (kernel_front_f43.hip);
// `wfront` points to the F(4,3) image (layout.hpp w4_* units).  The product's fp32 frontend.
template <typename PcmT>
hipError_t launch_front_f43(int sr, const FrontArgs &a, hipStream_t s);
// The LATENCY form of the same kernel (kernel_front_lat.hip): one workgroup = 4 waves = ONE tile, every layer's output rows
// split over the waves; bit-identical results.  For launches of a few hundred tiles (one step of a stream pool, a B = 1 call).
template <typename PcmT>
hipError_t launch_front_lat(int sr, const FrontArgs &a, hipStream_t s);
// ONE step (nt == 1) with the LSTM cell and the head fused behind the latency frontend: no gx round trip, no second kernel.  The
// workgroup of a tile applies W_hh (image `whh_lat`: gate by gate in W_ih's block order) to h_{t-1}, exchanges the four gates through
// LDS, updates (h, c) in `state` in place and writes the tile's 16 probabilities.  Bit-identical to front + rec_kernel.
struct CellArgs {
    const float *whh_lat;    // [gate 4][kg 8][row block 8][lane 64][4]
    float *state;            // [2][B][128] in/out
    float *probs;            // [B][ldp], this step's column is t0
    long ldp;
    const uint8_t *present;  // [B] or null (= all): rows with present[b] == 0 have NO chunk this tick -- their (h, c) and their
                             // probability slot are not written (the carry pass of kernel_present.hip finishes the row)
};
template <typename PcmT>
hipError_t launch_step_lat(int sr, const FrontArgs &a, const CellArgs &c, hipStream_t s);
// The same ONE step with one workgroup per STREAM and every sum formed on the VALU in the MFMA program's order (kernel_step_one.hip):
// identical bits, a third of the time for a handful of streams (a B = 1 model call).  launch_front_one: the frontend half alone (gx in
// the fragment layout), for the bit-identity tests.
template <typename PcmT>
hipError_t launch_step_one(int sr, const FrontArgs &a, const CellArgs &c, hipStream_t s);
template <typename PcmT>
hipError_t launch_front_one(int sr, const FrontArgs &a, hipStream_t s);
// Same function, encoder 0 as two Winograd F(2,3) tiles over the frame pairs, straight-line code (kernel_front_wino.hip);
// `wfront` points to the F(2,3) image (layout.hpp w_* units).  A/B form, test builds only (VAD_AB; option enc0=winograd2).
template <typename PcmT>
hipError_t launch_front_wino(int sr, const FrontArgs &a, hipStream_t s);

// Recurrence: persistent over the slab's time steps; W_hh pinned in VGPRs, h exchanged through
// LDS, LSTM pointwise + head fused.  One workgroup (8 waves) per 16 streams.
struct RecArgs {
    const float *whh;        // packed recurrent image
    const float *tables;
    const float *gx;
    float *state;            // [2][B][128] in/out
    float *probs;            // [B][ldp]
    long ldp, t0, nt;
    int B;
    const uint8_t *present;  // [B] or null (= all); one-step calls only: an absent row's (h, c) and probability are not written
};
hipError_t launch_rec(int sr, const RecArgs &a, hipStream_t s);
// Live streams that have no chunk this tick (vad_step_present): behind the step kernels, for every row with present[b] == 0 the
// context is carried over unchanged, ctx_out[b] = ctx_in[b] (the frontend wrote the tail of whatever the row's PCM slot held), and
// the probability slot gets the sentinel VAD_PROB_ABSENT.  The step kernels themselves left the row's (h, c) unwritten.
hipError_t launch_carry_absent(const uint8_t *present, const float *ctx_in, float *ctx_out, int C, float *probs, long ldp, int B,
                               hipStream_t s);
// A compact tick of the pump (vad_pump_submit_compact): row pos[b] of `src` (the delivering streams' chunks back to back, row_bytes each,
// a multiple of 16) -> row b of `dst` for every b with present[b] != 0; rows of absent streams are not touched.  HBM to HBM.
hipError_t launch_expand_rows(const uint8_t *present, const int32_t *pos, const uint8_t *src, void *dst, long row_bytes, int B, hipStream_t s);
// The same recurrence, bit for bit, for at most kRecSmallMaxB streams (1, 2 or 4 per workgroup): W_hh * h as matrix-vector products on the VALU (kernel_rec_small.hip);
// `whh` points at the row image (layout.hpp "whh_rows").
constexpr int kRecSmallMaxB = 1024;
hipError_t launch_rec_small(int sr, const RecArgs &a, hipStream_t s);
// The throughput frontend with every matrix product as exact bf16 x 9 piece products on the bf16 matrix pipe (kernel_front_b9.hip);
// `a.wfront` points at the three-piece image (layout.hpp "bf16 x 9 frontend image").  Same gx layout as launch_front_f43.
template <typename PcmT>
hipError_t launch_front_b9(int sr, const FrontArgs &a, hipStream_t s);
// The same recurrence with W_hh * h as exact bf16 x 9 piece products on the bf16 matrix pipe (kernel_rec_b9.hip); `whh` points
// to the three-piece image (layout.hpp "bf16 x 9 recurrent image").  Option "rec" = "bf16x9".
hipError_t launch_rec_b9(int sr, const RecArgs &a, hipStream_t s);

// Ingest (kernel_ingest.hip): rows[i].len elements (esz bytes each) at rows[i].ptr -- PINNED host memory, or device memory --
// -> dst[i][0 .. width), zero padded.  `rows` itself is read by the kernel (pinned or device memory).
struct RowDesc {
    const void *ptr;
    long len;
};
hipError_t launch_gather_rows(const RowDesc *rows, long n, long width, int esz, void *dst, bool rows_on_device, hipStream_t s);

// Segmenter on the device (kernel_scan.hip): lane i scans probs[i * ldp ...] (or probs[row_off[i] ...] if row_off is
// not null), n_chunks[i] entries (or n_chunks_all if n_chunks is null), writes its segments to out[i * cap ...] and
// their number (may exceed cap) to counts[i].
hipError_t launch_scan(const float *probs, long ldp, const long *row_off, long n_streams, const long *n_chunks, long n_chunks_all,
                       const long *audio_len, const vad_segment_params &p, vad_segment *out, long cap, long *counts,
                       hipStream_t s);

// test hook: y[i] = sigmoid (kind 0) / tanh (kind 1) of x[i] exactly as the recurrent kernels evaluate them (activations.hpp)
hipError_t launch_activation_probe(int kind, const float *x, float *y, long n, hipStream_t s);

// test hook: VALU-only spinner (kind 0 packed fp32, 1 scalar fp32), `blocks` one-wave workgroups
hipError_t launch_foreign_spin(float *sink, int blocks, long iters, int kind, hipStream_t s);

// gx (fragment order) -> row-major [B][T][512] for vad_debug_frontend
hipError_t launch_unpack_gx(const float *gx, float *out, int B, long T, hipStream_t s);

}  // namespace vad
