/*
 * silero_vad_hip.h -- C ABI of the MI355X (gfx950) Silero-VAD v6 inference engine.
 *
 * The reference has no C ABI for this path: its boundary is a duck-typed Python model object
 * (TorchScript VADRNNJITMerge, or OnnxWrapper over ONNX Runtime).  The entry points below are
 * what a binding for that object has to call; each cites the reference interface it replaces
 * (paths relative to the reference repo; "JIT!/" = TorchScript source inside
 * src/silero_vad/data/silero_vad.jit).  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer marked "dev" is a DEVICE pointer on the
 *     engine's GPU (HBM-resident), everything else is host memory;
 *   - the caller owns all I/O buffers; the engine owns only its packed weights and scratch;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous with respect to the host unless stated otherwise;
 *   - one engine per GPU; an engine is not thread-safe (the reference model object is not
 *     either: src/silero_vad/utils_vad.py:51-92 mutates _state/_context per call);
 *   - the arithmetic is fp32 throughout, as in the reference: every contraction is a chain
 *     of v_mfma_f32_16x16x4_f32 (an fmaf chain, one rounding per product); what differs from the reference is WHICH
 *     fp32 sums are formed -- the STFT is a real FFT instead of the dense DFT-basis convolution, encoder 0 is evaluated in
 *     Winograd F(4,3) form (transformed weights, formed in double and rounded once) -- not their precision.  Agreement
 *     with the reference is therefore by tolerance, not bitwise: max |dp| 2e-6 on the reference's fixtures, <= 2e-5
 *     asserted on everything the GPU suite runs (the contract is 1e-4), final (h, c) within 1e-4;
 *   - sample rates 16000 (chunk N=512, context C=64) and 8000 (N=256, C=32); multiples of 16000 are decimated.
 */
#ifndef SILERO_VAD_HIP_H
#define SILERO_VAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vad_engine vad_engine;

enum vad_status {
    VAD_OK = 0,
    VAD_ERR_ARG = 1,        /* null pointer / negative size                                    */
    VAD_ERR_SAMPLE_RATE = 2,/* sr not in {8000, 16000}   (vad_annotator.py:95)                 */
    VAD_ERR_WEIGHTS = 3,    /* malformed weight container / missing tensor                     */
    VAD_ERR_NO_DEVICE = 4,  /* no HIP device, or not a gfx950 part                             */
    VAD_ERR_HIP = 5,        /* a HIP runtime call failed; see vad_last_error()                  */
    VAD_ERR_ALLOC = 6,      /* scratch allocation failed                                        */
    VAD_ERR_CAPTURE = 7,    /* scratch would have to grow while the stream is being captured    */
    VAD_ERR_OPTION = 8      /* unknown option name/value                                        */
};

/* ---- lifetime --------------------------------------------------------------------------------
 * Replaces load_silero_vad()/init_jit_model (src/silero_vad/model.py:6-36,
 * src/silero_vad/utils_vad.py:194-198): parses the SVADW001 weight container
 * (tools/export_weights.py; state_dict of silero_vad.jit), re-packs it into MFMA fragment order
 * and uploads it to `device`.                                                                  */
int  vad_create(const void *weights, size_t nbytes, int device, vad_engine **out);
void vad_destroy(vad_engine *e);
/* A second handle on the same GPU: shares the read-only weight images (no second copy) and starts with the same options,
 * owns its scratch.  An engine and its clones may have calls in flight on different streams at the same time (one engine
 * may not: its scratch is per call).  The reference gets concurrency by one model object per process
 * (examples/parallel_example.ipynb cell 5); inside one process on one GPU this is the equivalent.  Destroy every handle. */
int  vad_clone(const vad_engine *e, vad_engine **out);
const char *vad_strerror(int status);
const char *vad_last_error(const vad_engine *e);   /* detail text of the last failing call */
int  vad_device(const vad_engine *e);

/* Geometry of a sample rate (returns VAD_ERR_SAMPLE_RATE otherwise).
 * chunk: 512|256 (utils_vad.py:60), context: 64|32 (utils_vad.py:66).                           */
int  vad_geometry(int sr, int *chunk, int *context);

/* Options (strings so that bindings need no enum mirror):
 *   "impl"      = "mfma" (default, the product path) | "reference" (slow all-VALU kernels kept
 *                 as an on-device A/B for tests; never the default)
 *   "precision" = "fp32": the only arithmetic (accepted so that a binding may state it; anything else is VAD_ERR_OPTION)
 *   "gx_cap_mib"= cap, in MiB, of the engine's scratch for the LSTM input-gate pre-activations (default 6144);
 *                 a call whose B x T needs more is processed in time slabs, transparently
 *   "enc0"      = "winograd" (default; "winograd4" is a synonym): encoder 0 -- more than half of the frontend's matrix
 *                 work -- as ONE Winograd F(4,3) tile over the chunk's 4 STFT frames, 6 GEMMs instead of 10, all in fp32
 *                 (csrc/kernel_front_f43.hip).  The test build libsilero_vad_hip_ab.so (__graft_entry__.build, -DVAD_AB=1)
 *                 also accepts "winograd2" (two F(2,3) tiles, csrc/kernel_front_wino.hip) and "direct" (tap by tap,
 *                 csrc/kernel_front.hip: bitwise the plain fmaf chain over the taps) as A/B forms for the parity tests
 *   "rec"       = "fp32" (default) | "bf16x9": how the recurrence evaluates W_hh * h -- as the fp32 MFMA chain, or as the nine
 *                 exact products of three bf16 pieces per operand (x = p0 + p1 + p2 exactly) on the bf16 matrix pipe with fp32
 *                 accumulation (csrc/kernel_rec_b9.hip).  Not narrower than fp32 (no operand bit is dropped, every product is
 *                 exact), but a different summation: opt-in; measured against a float64 recurrence by the GPU suite
 *   "rec_form"  = "auto" (default) | "mfma": the fp32 recurrence has two forms with bit-identical results -- 16 streams per CU on the
 *                 matrix pipe (csrc/kernel_rec.hip: 4.3 us per step whatever the batch) and, for B <= 1024, W_hh h as fmaf chains on the
 *                 VALU in the MFMA chain's summation order, 1-4 streams per CU (csrc/kernel_rec_small.hip: 1.0-2.6 us per step; a file
 *                 at a time).  "mfma" forces the first (tests; callers that overlap several calls and want a call on few CUs)
 *   "front_mma" = "fp32" (default) | "bf16x9": the same for the frontend's matrix products (encoders 0-3 and W_ih): the fp32 MFMA
 *                 chain, or exact bf16 x 9 piece products (csrc/kernel_front_b9.hip; FFT, Winograd transforms, Nyquist update,
 *                 biases and ReLU stay the fp32 VALU code).  With "bf16x9" EVERY launch takes that kernel, whatever its size
 *                 (no latency form, no fused step: the arithmetic of a result must not depend on the batch it came in).
 *                 Opt-in; 10-20 % faster than the fp32 frontend, not more
 *   "front"     = "auto" (default) | "throughput" | "latency": the frontend has two forms with bit-identical results --
 *                 one wave per 16-chunk tile (csrc/kernel_front_f43.hip: tens of thousands of tiles per launch) and one
 *                 4-wave workgroup per tile (csrc/kernel_front_lat.hip: a stream pool's step, a B = 1 call); "auto" takes
 *                 the latency form for launches of at most 768 tiles.  The other two values force one form (tests)
 *   "fuse_step" = "1" (default) | "0": a ONE-step call that takes the latency form (vad_step of a stream pool, a B = 1 call)
 *                 runs the LSTM cell and the head inside the frontend's kernel -- no second launch, no trip of the gate
 *                 pre-activations through HBM; bit-identical to the two-kernel path ("0": force that, A/B for tests)
 *   "exact_transitions" = "1" (default) | "0": a chunk that holds an EXACTLY silent STFT frame (every sample zero: a muted source, a DTX
 *                 gap, the zero padding behind a recording's end) beside a frame that is not silent gets its gate pre-activations from a
 *                 double-precision evaluation of the frontend (csrc/exact_front.hpp: the reference's definition, its own fp32 DFT basis,
 *                 every sum in double, one rounding), because every one-accumulator fp32 summation is ill-conditioned exactly there
 *                 (carried (h, c) up to 1.2e-4 from float64 against <= 2e-5 on continuous audio; with the option 1e-5).  A pure function
 *                 of the chunk's own samples, identical bits on every route.  A chunk whose FOUR frames are silent takes the net's constant
 *                 for a chunk of zeros, evaluated the same way once (the chains make the same rounding error in every silent chunk and the
 *                 cell state integrates it: 4.6e-5 one chunk into a silence, 1e-5 with the constant).  "0": the fp32 chains everywhere,
 *                 "edges": without the constant (A/B for tests and studies).
 *                 fp32 frontend only (not with front_mma = bf16x9)
 *   "step_one"  = "auto" (default: 256, the number of CUs) | "0".."4096": a ONE-step call of at most this many streams (the B = 1
 *                 `model(chunk, sr)` of every unmodified caller; a tick of a small stream pool) takes one workgroup per STREAM
 *                 (csrc/kernel_step_one.hip: every sum of the step as an fmaf chain on the VALU in the MFMA program's summation order,
 *                 read from the same packed weight images -- bit-identical to the tile kernels; 25.0-26.5 us for 1..256 streams against
 *                 31.4-34.2 us, because no matrix instruction computes 16 columns for one stream and every stream has a CU to itself;
 *                 beyond one workgroup per CU the tile kernels win); "0": the 16-stream tile kernels for every batch (A/B for tests)
 *   "fused_decimation" = "1" (default) | "0": for sr = 32000 and 48000 the fp32 frontend reads every 2nd / 3rd sample
 *                 itself; "0" forces the separate decimation pass that the higher multiples of 16000 use (A/B for tests)
 *   "profile"   = "0" | "1"   record hipEvents around each kernel (vad_kernel_times)
 *   "trace_ptr" = device address (bring-up only): builds compiled with -DVAD_TRACE=1 write 16
 *                 int64 phase timestamps per frontend workgroup there; the one-stream step kernel leaves the shader clock of its
 *                 phase boundaries there in every build (tools/b1_phase_trace.py); "0" = off (default)  */
int  vad_set_option(vad_engine *e, const char *name, const char *value);

/* ---- the hot path ----------------------------------------------------------------------------
 * One step for B independent streams: the functional form of VADRNNJITMerge.forward
 * (JIT!/vad/model/vad_annotator.py:14-90) == the ONNX graph I/O used by OnnxWrapper.__call__
 * (src/silero_vad/utils_vad.py:57-92, session.run at :80-82):
 *     x1 = cat(ctx, pcm);  prob, state' = net(x1, state);  ctx' = x1[:, -C:]
 *   pcm    dev [B][N]   fp32 in [-1,1], row stride `ld` floats
 *   ctx    dev [B][C]   in/out  (zeros after a reset)
 *   state  dev [2][B][128] in/out  (h stacked on c; zeros after a reset)
 *   prob   dev [B]      out                                                                     */
int  vad_step(vad_engine *e, int sr, int B, const float *pcm, long ld, float *ctx, float *state,
              float *prob, void *stream);

/* The same step from HOST audio to HOST probabilities, the shape of the reference's streaming callers (host chunk in, probability out:
 * src/silero_vad/utils_vad.py:507-549 VADIterator; examples/cpp/silero-vad-onnx.cpp:103-142 feeds int16-derived chunks the same way):
 * one copy of the B chunks host -> dev_pcm, the step, one copy of the B probabilities dev_prob -> host_prob, all asynchronous on
 * `stream` (record an event behind the call and wait for it before reading host_prob).  May be issued inside a stream capture: the
 * three operations then become one hipGraph (what StreamPool replays per tick).
 *   host_pcm   host [B][N], PAGE-LOCKED; elem_size 2 = int16 (scaled by 1/32768 in the kernel's loads), 4 = fp32
 *   dev_pcm    dev  [B][N] of the same element type: staging the caller owns (16-byte aligned); or NULL: no copy -- the kernel reads
 *              host_pcm through the device's view of it (for a handful of streams: a B = 1 call is 2 KB)
 *   dev_prob   dev  [B], or NULL: the kernel then stores the probabilities straight into host_prob (no device copy, no D2H operation)
 *   host_prob  host [B], page-locked                                                                                            */
int  vad_step_host(vad_engine *e, int sr, int B, const void *host_pcm, size_t elem_size, void *dev_pcm, float *ctx, float *state,
                   float *dev_prob, float *host_prob, void *stream);

/* vad_step_host(dev_pcm = NULL, dev_prob = NULL) that RETURNS WHEN THE B PROBABILITIES ARE IN host_prob: the blocking call every
 * unmodified caller of the reference makes, `model(chunk, sr).item()` once per 32 ms (src/silero_vad/utils_vad.py:324-336
 * get_speech_timestamps, :528 VADIterator.__call__; examples/cpp/silero-vad-onnx.cpp:103-142 around session.Run).  The kernels store a
 * stream's probability as their last act, so the call waits for the B slots themselves to change (a bounded spin on the page-locked
 * memory; hipStreamSynchronize if they have not after 0.4 ms) instead of for the stream's completion signal.  ctx / state are device
 * buffers as in vad_step; later work on `stream` is ordered behind the step as usual.  A blocking call: not inside a stream capture
 * (capture vad_step_host).                                                                                                      */
int  vad_step_host_sync(vad_engine *e, int sr, int B, const void *host_pcm, size_t elem_size, float *ctx, float *state, float *host_prob,
                        void *stream);

/* vad_step with the two contexts apart and either PCM type, all on the device: reads ctx_in, writes the next context to ctx_out
 * (a second buffer: the kernel may not write where other waves still read), so that a caller that alternates two context buffers
 * pays no device-to-device copy per step.  The functional form once more -- the ONNX graph's inputs and outputs are distinct
 * tensors too (src/silero_vad/utils_vad.py:80-83: `state` in, `stateN` out).
 *   pcm  dev [B][N], elem_size 2 = int16 (scaled by 1/32768 in the kernel's loads) | 4 = fp32, row stride `ld` elements      */
int  vad_step_split(vad_engine *e, int sr, int B, const void *pcm, size_t elem_size, long ld, const float *ctx_in, float *ctx_out,
                    float *state, float *prob, void *stream);

/* The step for LIVE streams that do not all have a chunk this tick.  In the reference a stream's (h, c) and context change only when
 * that stream's own caller calls the model -- one call per chunk that arrived (src/silero_vad/utils_vad.py:507-549 VADIterator.__call__;
 * the model object replaces _state / _context inside that call and nowhere else, JIT!/vad/model/vad_annotator.py:72,86-87; the native
 * loop around session.run: examples/cpp/silero-vad-onnx.cpp:335-390) -- so a stream whose packet is late simply is not stepped.  In a
 * lock-step batch that is a per-row flag:
 *   present  dev [B] bytes (device memory, or page-locked host memory's device alias), or NULL = every row has a chunk.
 *            present[b] == 0: row b comes out of the call exactly as it went in -- (h, c) not written, its context carried over
 *            bit for bit (ctx_out[b] = ctx_in[b]), whatever its PCM row holds is ignored -- and prob[b] = VAD_PROB_ABSENT.
 *            The other rows' results do not depend on the flags (bit-identical to the call without them).
 *   ctx_out  a second buffer (as vad_step_split), or NULL / == ctx_in: in place (as vad_step)
 *   pcm      dev [B][N], elem_size 2 = int16 | 4 = fp32, row stride `ld` elements
 * With present == NULL this IS vad_step_split / vad_step: same kernels, same cost.  With flags it is one more small launch (the carry
 * of the absent rows' contexts, csrc/kernel_present.hip).                                                                       */
#define VAD_PROB_ABSENT (-1.0f)
int  vad_step_present(vad_engine *e, int sr, int B, const void *pcm, size_t elem_size, long ld, const float *ctx_in, float *ctx_out,
                      float *state, float *prob, const uint8_t *present, void *stream);
/* vad_step_host with the flags: host_present host [B] page-locked (NULL = all present: exactly vad_step_host), dev_present dev [B]
 * staging the caller owns; one more small H2D copy on `stream`.                                                                  */
int  vad_step_host_present(vad_engine *e, int sr, int B, const void *host_pcm, size_t elem_size, void *dev_pcm, float *ctx, float *state,
                           float *dev_prob, float *host_prob, const uint8_t *host_present, uint8_t *dev_present, void *stream);

/* T = ceil(L / N) lock-step steps for B streams: VADRNNJITMerge.audio_forward
 * (JIT!/vad/model/vad_annotator.py:128-156; ONNX twin utils_vad.py:94-110) with the carried
 * state made explicit (pass zeroed ctx/state for the reference's reset-then-run behaviour).
 * A last partial chunk is right-padded with zeros, as the reference does (:141-148).
 * `sr` may also be a multiple of 16000 (32000, 48000, ...): the input is then decimated to 16 kHz,
 * x[:, ::sr/16000] (no filter, first sample kept -- vad_annotator.py:104-112, utils_vad.py:39-42), on
 * the device -- for 32 and 48 kHz inside the frontend's own loads, without an extra pass over HBM -- and L / T
 * refer to the input / to ceil(ceil(L / k) / 512) chunks.
 *   pcm    dev [B][L]   row stride `ld`
 *   probs  dev [B][T]   row stride `ldp`                                                        */
int  vad_forward_audio(vad_engine *e, int sr, int B, long L, const float *pcm, long ld,
                       float *ctx, float *state, float *probs, long ldp, void *stream);

/* Same, 16-bit PCM in HBM; samples are scaled by 1/32768 on load, the convention of the
 * reference's non-Python clients (examples/cpp/wav.h:113-118, examples/onnx_sequence/run.py:119). */
int  vad_forward_audio_i16(vad_engine *e, int sr, int B, long L, const int16_t *pcm, long ld,
                           float *ctx, float *state, float *probs, long ldp, void *stream);

/* Pre-size the engine-owned scratch for (sr, B, T) so that later calls allocate nothing (needed
 * before capturing vad_step/vad_forward_audio into a hipGraph; T counts chunks at the net's rate, sr may be a
 * multiple of 16000 -- the worst-case tail copy of a 48 kHz fp32 input is included).  Synchronous.                 */
int  vad_reserve(vad_engine *e, int sr, int B, long T);
size_t vad_scratch_bytes(const vad_engine *e);
/* The engine's scratch only ever grows; when a later call (or vad_reserve) needs more than it has, the
 * buffers are freed and reallocated (synchronously, and never while a stream is being captured:
 * VAD_ERR_CAPTURE).  A hipGraph that captured vad_step / vad_forward_audio holds the OLD device addresses,
 * so it MUST be re-captured after such a growth: this counter changes exactly when that happened.
 * (silero_vad_amd/streams.py StreamPool checks it before every replay.)                               */
unsigned long vad_scratch_generation(const vad_engine *e);

/* With option profile=1 the engine brackets its kernels with hipEvents on the caller's stream
 * (no host synchronisation at record time).  This call waits for them and returns the GPU time in
 * ms SUMMED over all vad_step/vad_forward_audio calls since the previous query (`calls` of them):
 * front = framing + STFT + encoder + input-gate GEMM kernel, rec = LSTM recurrence + head kernel. */
int  vad_kernel_times(vad_engine *e, float *front_ms, float *rec_ms, long *calls);

/* ---- post-processing on the host ----------------------------------------------------------------
 * The hysteresis segmenter of get_speech_timestamps (src/silero_vad/utils_vad.py:338-450) as a
 * native routine (C++ twin in the reference: examples/cpp/silero-vad-onnx.cpp:196-389).
 * Field names and defaults are those of the Python signature (utils_vad.py:212-227).             */
typedef struct vad_segment_params {
    double threshold;                    /* 0.5   (double: Python compares in double)       */
    double neg_threshold;                /* < 0 => max(threshold - 0.15, 0.01)  (:342-343)  */
    int    sampling_rate;                /* 8000 | 16000 (already decimated)                */
    int    min_speech_duration_ms;       /* 250                                             */
    double max_speech_duration_s;        /* +inf                                            */
    int    min_silence_duration_ms;      /* 100                                             */
    int    speech_pad_ms;                /* 30                                              */
    int    min_silence_at_max_speech_ms; /* 98                                              */
    int    use_max_poss_sil_at_max_speech; /* 1                                             */
} vad_segment_params;

typedef struct vad_segment { int64_t start, end; } vad_segment;   /* sample indices */

void vad_segment_params_default(vad_segment_params *p, int sampling_rate);

/* probs[n] -> speech segments of an audio of `audio_len` samples.  Writes at most `cap`
 * segments to `out`; returns the number of segments found (may exceed cap), <0 on bad args.      */
long vad_segment_probs(const float *probs, long n, long audio_len, const vad_segment_params *p,
                       vad_segment *out, long cap);

/* The same scan for many independent streams (one row of probs per stream, row stride `ldp`,
 * n_chunks[i] valid entries, audio of audio_len[i] samples): what a corpus run does after
 * vad_forward_audio (the reference fans this out as one Python process per file,
 * examples/parallel_example.ipynb cells 5, 7).  Stream i's segments go to
 * out[i * cap_per_stream ...], their number (may exceed cap_per_stream) to counts[i].  Streams are
 * split over `threads` persistent host threads (<= 0: vad_host_threads()).  Returns the total number of
 * segments, <0 on bad arguments.                                                                */
long vad_segment_probs_batch(const float *probs, long ldp, long n_streams, const long *n_chunks,
                             const long *audio_len, const vad_segment_params *p, vad_segment *out,
                             long cap_per_stream, long *counts, int threads);

/* The same scan ON THE DEVICE, for probabilities that vad_forward_audio just left in HBM: one GPU lane per
 * stream, so that only the segment lists cross PCIe (16 B per segment) instead of every probability, and the host
 * does no per-chunk work (the reference does it in a Python loop per file: utils_vad.py:348-440).  All pointers
 * are DEVICE pointers except `p`.  n_chunks may be NULL: every stream then has n_chunks_all (<= ldp) entries.
 * row_offsets may be NULL: stream i's probabilities then start at probs[i * ldp]; otherwise at probs[row_offsets[i]]
 * (ragged rows packed back to back, as the continuous-refill scheduler leaves them).
 * Stream i's segments go to out[i * cap_per_stream ...], their number (may exceed cap_per_stream: grow and call
 * again) to counts[i].  Asynchronous on `stream`.  Same results as vad_segment_probs_batch, bit for bit (one
 * source, csrc/scanner.hpp).                                                                        */
int  vad_segment_probs_device(vad_engine *e, const float *probs, long ldp, const long *row_offsets, long n_streams,
                              const long *n_chunks, long n_chunks_all, const long *audio_len, const vad_segment_params *p,
                              vad_segment *out, long cap_per_stream, long *counts, void *stream);

/* The streaming caller, VADIterator.__call__ (src/silero_vad/utils_vad.py:507-549), for every slot of a lock-step batch: probs[n]
 * = this tick's probabilities (what vad_step left, copied to the host), active (may be NULL = all) marks the slots that carry a
 * live stream.  triggered / temp_end / current_sample are the iterator's per-stream state (zero after a reset, :500-503), updated
 * in place.  Writes at most `cap` events (slot order; kind 0 = {'start': sample}, 1 = {'end': sample}, the numbers the reference
 * iterator returns for that stream) and returns how many there were (may exceed cap; at most n), <0 on bad arguments.
 * threshold, min_silence_samples = sr * min_silence_duration_ms / 1000 and speech_pad_samples = sr * speech_pad_ms / 1000 are
 * doubles because the reference compares Python floats (:494-498).                                                          */
typedef struct vad_iter_event { int32_t slot; int32_t kind; int64_t sample; } vad_iter_event;
long vad_iterator_feed(const float *probs, const uint8_t *active, long n, int window, double threshold,
                       double min_silence_samples, double speech_pad_samples, uint8_t *triggered,
                       int64_t *temp_end, int64_t *current_sample, vad_iter_event *out, long cap);

/* ---- live streams: the pump --------------------------------------------------------------------------------------------
 * BASELINE configs[4] as one native object: `streams` live streams on one GPU advance in lock step, one tick = one 32 ms chunk of
 * every stream, host int16 audio in, VADIterator events out -- the loop the reference's native streaming clients run around
 * their runtime, one session.run per chunk with explicit state and the iterator logic inline
 * (examples/cpp/silero-vad-onnx.cpp:335-390; Python: src/silero_vad/utils_vad.py:507-549 VADIterator.__call__), for thousands of
 * streams at once.  The pump owns: a page-locked ingest ring [ring_slots][streams][N] int16 that the audio sources write
 * into, the device batch (three buffers), the carried (h, c) and context of every stream in HBM, the iterator state of every
 * stream, three HIP streams (copies of even / odd ticks, kernels) and the events that order them (csrc/pump.hip has the schedule:
 * tick t + 1's H2D runs beside tick t's kernel BY EVENT, nothing left to hardware-queue assignment).
 * One caller thread drives submit / poll; any thread may write a ring slot that is not in flight.                          */
typedef struct vad_pump vad_pump;
typedef struct vad_pump_params {
    int    sampling_rate;            /* 8000 | 16000                                                                         */
    int    streams;                  /* live streams (slots) on this GPU                                                     */
    int    parts;                    /* sub-batches per tick, each its own copy and kernel (<= 0: 1)                         */
    int    ring_slots;               /* ticks of audio the ingest ring holds (<= 0: 4; at least 2)                           */
    double threshold;                /* VADIterator arguments and defaults (utils_vad.py:477-498): 0.5                       */
    int    min_silence_duration_ms;  /* 100                                                                                  */
    int    speech_pad_ms;            /* 30                                                                                   */
} vad_pump_params;
typedef struct vad_pump_stats {
    long   ticks, events;
    double wall_ms;
    double tick_ms_p50, tick_ms_p95, tick_ms_max;     /* slot written -> its events on the host, per tick                    */
    double fill_ms_mean, submit_ms_mean, wait_ms_mean;/* host time per tick: writing the slot, issuing the tick, blocked     */
    int    fill_threads, depth;
    long   chunks;                                    /* chunks that were stepped (= ticks x streams unless streams were absent)  */
} vad_pump_stats;
enum { VAD_PUMP_IDLE = -1,   /* vad_pump_poll: nothing submitted                                                             */
       VAD_PUMP_BUSY = -2,   /* vad_pump_poll(block = 0): the oldest tick has not finished                                   */
       VAD_PUMP_ERROR = -3 };/* see vad_pump_last_error                                                                      */

void vad_pump_params_default(vad_pump_params *p, int sampling_rate, int streams);
/* The pump works on a clone of `e` (vad_clone): the caller's engine stays free for other calls.  Every stream starts open, from
 * zero state (VADIterator.reset_states, utils_vad.py:500-505).                                                             */
int  vad_pump_create(vad_engine *e, const vad_pump_params *p, vad_pump **out);
void vad_pump_destroy(vad_pump *p);
const char *vad_pump_last_error(const vad_pump *p);
int  vad_pump_geometry(const vad_pump *p, int *streams, int *chunk, int *ring_slots, int *parts);
/* Ring slot r: [streams][N] int16, page-locked; the sources write stream b's next chunk at slot + b * N.                   */
int16_t *vad_pump_slot(vad_pump *p, int r);
/* Start one tick over ring slot r: asynchronous (copies and kernels are queued; returns at once).  VAD_ERR_ARG while the slot's
 * previous tick is in flight.  Ticks execute in submission order.                                                          */
int  vad_pump_submit(vad_pump *p, int r);
/* The tick for live streams that do not all have a chunk (see vad_step_present): present = host [streams] bytes, 0 = stream b has no
 * chunk this tick -- its (h, c), context and iterator counters stay exactly as they are, its slot of vad_pump_probs holds
 * VAD_PROB_ABSENT, whatever its part of the ring slot holds is ignored; NULL = every stream has one (== vad_pump_submit, same copies,
 * same kernels).  Every ring slot has its own page-locked flag row, vad_pump_present(p, r), in front of its audio (flags and audio
 * cross the link in ONE copy): write the flags there and pass that pointer, or pass any host array (copied there).  A failure after
 * the tick's first operation was queued poisons the pump (every later call fails): the carried state is half-advanced.          */
uint8_t *vad_pump_present(vad_pump *p, int r);
int  vad_pump_submit_present(vad_pump *p, int r, const uint8_t *present);
/* The same tick from a COMPACT slot: the sources wrote only the chunks of the streams that deliver, back to back -- row i of
 * vad_pump_slot(p, r) is the chunk of the i-th stream (ascending) whose flag is set -- and only the flags, a position table and those
 * rows cross the link (one copy); a row-expansion pass on the device puts every chunk where the step kernels read it.  Results are
 * those of vad_pump_submit_present with the same flags, bit for bit; the link cost of a tick falls with the delivery rate.  present
 * must not be NULL.  Compact, masked and full ticks may be mixed freely.                                                         */
int  vad_pump_submit_compact(vad_pump *p, int r, const uint8_t *present);
/* A compact tick in ARRIVAL order -- what a receive path can fill without knowing who else will deliver: row i of vad_pump_slot(p, r)
 * is the chunk of stream stream_of_row[i], i < n_rows (receive threads append with one atomic row counter per slot); every other stream
 * is absent this tick.  A stream listed twice (two packets in one tick: keep the second for the next tick) or out of range is
 * VAD_ERR_ARG and nothing is queued.  n_rows == 0: a tick in which nobody delivers.  Results as vad_pump_submit_present with the same
 * set of streams, bit for bit.                                                                                                   */
int  vad_pump_submit_rows(vad_pump *p, int r, const int32_t *stream_of_row, long n_rows);
/* Retire the OLDEST submitted tick: wait for it (block != 0) or return VAD_PUMP_BUSY, run the iterator logic of every open
 * stream over its probabilities and write the tick's events (stream order; at most `cap`, the return value is how many there
 * were, <= streams).  *slot = the ring slot that is free again.  The probabilities stay readable in vad_pump_probs(p, slot)
 * until that slot's next tick.                                                                                             */
long vad_pump_poll(vad_pump *p, int block, vad_iter_event *out, long cap, int *slot);
const float *vad_pump_probs(const vad_pump *p, int r);   /* [streams] */
/* A new stream takes slot `stream`: zero (h, c), context and iterator state, ordered behind the ticks already submitted
 * (reset_states, JIT!/vad/model/vad_annotator.py:157-162 + utils_vad.py:500-505).  close: the slot is still computed (lock-step
 * batch) but emits no events.  Both take effect BEHIND the ticks that are in flight when they are called, on the device (stream
 * order) and on the host (the iterator state is reset / muted when those ticks have been retired: their probabilities and events
 * belong to the slot's previous occupant).                                                                                 */
int  vad_pump_open(vad_pump *p, int stream);
int  vad_pump_close(vad_pump *p, int stream);
/* Test / migration hook: copy stream's carried h[128], c[128], ctx[C] to the host (any may be NULL).  Synchronises.        */
int  vad_pump_state(vad_pump *p, int stream, float *h, float *c, float *ctx);
/* The whole loop in one native call -- for tests, benchmarks and file-fed servers: stream b plays rows[b * ld ...] circularly with
 * period `period` samples (a multiple of N): at tick t its chunk is rows[b * ld + (t * N) % period ...].  `fill_threads` SOURCE
 * threads (0: min(8, vad_host_threads() - 2); < 0: the sources are silent, the slots keep their content -- isolates the device
 * side) each own a range of streams and WRITE their chunks into ring slot t % ring_slots for t = first_tick ... first_tick +
 * n_ticks - 1; the calling thread submits a tick once it is completely written and, once `depth` ticks are in flight (clamped to
 * 1 ... ring_slots - 1), retires the oldest.  depth 1 = strictly one tick at a time (the next chunks are written after the
 * previous tick's events are out): tick_ms_* is then the latency of one tick, slot written -> events on the host; depth >= 2: the
 * sources write the next tick while `depth` ticks are in flight.  All events are appended to `out` (at most
 * cap; the return value is their number) and `st` (may be NULL) is filled in (fill_ms_mean: per source thread and tick).          */
long vad_pump_play(vad_pump *p, const int16_t *rows, long ld, long period, long first_tick, long n_ticks, int depth, int fill_threads,
                   vad_iter_event *out, long cap, vad_pump_stats *st);
/* The same loop with streams that miss ticks: pattern = [pattern_ticks][streams] bytes, stream b has a chunk at tick t iff
 * pattern[(t % pattern_ticks) * streams + b] != 0.  A stream's audio advances only when it delivers (its k-th delivered chunk is
 * rows[b * ld + (k * N) % period ...], k counted since the stream was opened): a late packet delays the stream, it does not skip
 * audio.  The sources write the flags into the slot's flag row and every tick is a vad_pump_submit_present.  pattern == NULL:
 * vad_pump_play.                                                                                                            */
long vad_pump_play_gaps(vad_pump *p, const int16_t *rows, long ld, long period, const uint8_t *pattern, long pattern_ticks, long first_tick,
                        long n_ticks, int depth, int fill_threads, vad_iter_event *out, long cap, vad_pump_stats *st);
/* vad_pump_play_gaps with compact slots: the sources write the delivering streams' chunks back to back (a source thread's first row is
 * the number of delivering streams in front of its range) and every tick is a vad_pump_submit_compact.  Same events, same state.    */
long vad_pump_play_compact(vad_pump *p, const int16_t *rows, long ld, long period, const uint8_t *pattern, long pattern_ticks, long first_tick,
                           long n_ticks, int depth, int fill_threads, vad_iter_event *out, long cap, vad_pump_stats *st);

/* ---- host-side ingest ---------------------------------------------------------------------------------
 * Pack n recordings of different lengths (lens[i] samples of elem_size 2 = int16 or 4 = float32 at
 * rows[i]) into one zero-padded row-major [n][width] batch at dst (typically pinned host memory that
 * is then copied to the GPU and handed to vad_forward_audio[_i16]).  Zero padding on the right is what
 * the reference does to a recording's last chunk (src/silero_vad/utils_vad.py:326-327); the network
 * is causal, so padding further does not change the recording's own probabilities.  The copy is
 * split over `threads` persistent host threads (<= 0: vad_host_threads(), at most 32).              */
int  vad_stage_rows(const void *const *rows, const long *lens, long n, long width, size_t elem_size,
                    void *dst, int threads);

/* The same packing WITHOUT a host-side copy, for recordings that already sit in page-locked host memory (hipHostMalloc,
 * torch pin_memory, or vad_host_register below): rows[i] / lens[i] as above, dst = DEVICE [n][width] (16-byte aligned,
 * width * elem_size a multiple of 16).  Asynchronous on `stream`; rows must stay valid until the stream has passed the call.
 *   how = 0  copy engines: one H2D DMA per row (any alignment, no CU time) behind one fill of the batch
 *   how = 1  one gather kernel that reads the rows over PCIe and writes the padded batch (kernel_ingest.hip): a single
 *            launch for any number of rows -- what the continuous-refill scheduler needs (thousands of short rows per slab)
 *   how = 2  rows[] are DEVICE addresses: the recordings were brought over packed, by one large DMA of the arena range
 *            that holds them, and are scattered into the padded batch at HBM speed (the route streams.py takes for a
 *            PackedRecordings whose recordings lie back to back: PCIe carries exactly the live bytes, in big copies)
 * For pageable sources use vad_stage_rows + one copy instead.                                                          */
int  vad_upload_rows(vad_engine *e, const void *const *rows, const long *lens, long n, long width, size_t elem_size,
                     void *dst, int how, void *stream);
/* Page-lock / unlock a host range so that it can be a vad_upload_rows source (hipHostRegister; a decoder's output
 * buffers, a memory-mapped corpus shard).  Process-wide.                                                              */
int  vad_host_register(void *p, size_t bytes);
int  vad_host_unregister(void *p);

/* The continuous-refill schedule of silero_vad_amd/streams.py RefillPlan (the reference's padded lock-step batch, tuning/utils.py:146-160,
 * with rows retired and re-admitted): recording q (in admission order) occupies a stream slot for need[q] time slabs; at every slab
 * boundary every free slot, lowest first, takes the next recording.  Writes the slab at which q is admitted and its slot.  Host only. */
int  vad_refill_schedule(const long *need, long n, long slots, long *start, long *slot);
/* ... and the schedule as the table the stager walks: one row of five longs per (recording, slab it is active in) -- slot, recording,
 * first sample, samples, reset flag -- ordered by slab, then by slot; cuts[k] .. cuts[k + 1] (n_slabs + 1 entries) are slab k's rows.
 * Queue entry q is recording rec[q] of len[q] samples; width = samples per slab; rows has room for sum(need) rows.  Returns the number
 * of rows, or -VAD_ERR_ARG.  Host only.                                                                                           */
long vad_refill_table(const long *need, const long *start, const long *slot, const long *rec, const long *len, long n, long slots,
                      long width, long n_slabs, long *rows, long *cuts);

/* Do two streams of the engine's device run BESIDE each other?  The HIP runtime maps streams onto a handful of hardware queues
 * (GPU_MAX_HW_QUEUES, 4 by default; which stream lands where depends on the order in which the process' streams were first
 * used), and two streams on one queue execute their kernels one after the other whatever the events say.  A pipeline that wants
 * its upload kernel beside its compute kernels (vad_upload_rows how = 1 on one stream, vad_forward_audio on another) asks before
 * it commits to a pair of streams, and takes another stream if the answer is 0 (silero_vad_amd/streams.py does).  Returns 1: a
 * short kernel on b finished while a ~1 ms kernel on a was still running; 0: it waited for it; < 0: -vad_status.  Synchronises
 * both streams; costs ~1 ms.                                                                                                */
int  vad_streams_overlap(vad_engine *e, void *stream_a, void *stream_b);

/* Host threads the native helpers (vad_stage_rows, vad_segment_probs_batch) use by default in THIS process:
 * min(CPU affinity, cgroup CPU quota) / LOCAL_WORLD_SIZE, i.e. the node's CPU budget divided among the one-process-per-GPU
 * ranks torchrun started (SILERO_VAD_AMD_HOST_THREADS overrides).                                                     */
int  vad_host_threads(void);
/* Narrow the calling thread's CPU affinity to the CPUs of `device`'s NUMA node (threads and pinned buffers created
 * afterwards follow).  Returns the node, or -1 if it is unknown or outside the process' CPU mask (nothing changed).   */
int  vad_bind_host_to_device(int device);

/* ---- test / bring-up hooks (not part of the drop-in surface) -------------------------------------
 * Host-only: size and contents of the packed weight images the kernels consume, so CPU tests can
 * check the fragment packing without a GPU.  which: 0 = frontend GEMM stream (enc0 "direct"), 1 = recurrent
 * W_hh image, 2 = small tables (biases, head, window, twiddles), 5 = frontend stream in Winograd F(2,3) form,
 * 6 = frontend stream in Winograd F(4,3) form (the product's), 7 = recurrent image as three bf16 pieces per weight
 * (option rec=bf16x9), 8 = the F(4,3) frontend program as three bf16 pieces per weight (option front_mma=bf16x9); 7 and 8 are
 * returned as raw 4-byte words holding two bf16 each.                                                              */
long vad_debug_packed_floats(const vad_engine *e, int sr, int which);
int  vad_debug_packed_copy(const vad_engine *e, int sr, int which, float *dst, long n);
/* Host-only engine for the hooks above (no device needed).                                       */
int  vad_create_host_only(const void *weights, size_t nbytes, vad_engine **out);
/* Device: run the frontend only and return the LSTM input-gate pre-activations
 * gx[B][T][512] = W_ih * enc(stft(x)) + b_ih + b_hh  (row-major, gate order i,f,g,o).            */
int  vad_debug_frontend(vad_engine *e, int sr, int B, long L, const float *pcm, long ld,
                        const float *ctx, float *gx, void *stream);

/* Device: y[i] = sigmoid(x[i]) (kind 0) or tanh(x[i]) (kind 1) exactly as the recurrent kernels evaluate them
 * (v_exp_f32 / v_rcp_f32 based, csrc/activations.hpp); x, y device pointers.  For the accuracy test.        */
int  vad_debug_activation(vad_engine *e, int kind, const float *x, float *y, long n, void *stream);

/* Device: launch a "foreign tenant" on `stream`: `blocks` one-wave workgroups that execute nothing but fp32
 * VALU FMAs for `iters` rounds (kind 0: packed v_pk_fma_f32, kind 1: scalar v_fma_f32).  The GPU tests run
 * it beside the engine's kernels to check their results under a co-resident foreign kernel.  Asynchronous. */
int  vad_debug_foreign_load(vad_engine *e, int kind, int blocks, long iters, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SILERO_VAD_HIP_H */
