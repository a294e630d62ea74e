"""Many streams at once: the two callers BASELINE.json's configs name on top of the engine.

* ``ragged_probs`` / ``RaggedPlan`` -- offline corpora (configs[3]).  ``audio_forward`` assumes
  equal-length rows (JIT!/vad/model/vad_annotator.py:141-148); real corpora are ragged.  The
  network is causal (a chunk's probability depends only on samples up to the end of that chunk,
  and the reference right-pads the last partial chunk with zeros, :141-148), so a recording that
  is right-padded with zeros to its bucket's length yields bit-identical probabilities for its
  own ceil(len / N) chunks.  The plan sorts recordings by length and cuts buckets so that padding
  waste and bytes per GPU call stay bounded; each bucket is one lock-step ``vad_forward_audio``
  call.  Recordings that already sit in PINNED host memory are DMA'd / gathered straight into the device batch
  (vad_upload_rows: no host-side copy at all); pageable ones go through a threaded copy into pinned staging first
  (vad_stage_rows).  int16 halves the PCIe bytes.

* ``StreamPool`` -- live streams (configs[4]).  Per-stream LSTM state and context stay resident
  in HBM (the resumable unit of the reference model object, vad_annotator.py:7-8,72,87); one
  hipGraph-captured ``vad_step`` serves every open stream per 32 ms tick; streams are opened and
  closed by slot, a fresh slot starting from zero state like ``reset_states()`` (:157-162).

* ``BatchVADIterator`` -- the reference's streaming event logic (``VADIterator``,
  src/silero_vad/utils_vad.py:507-549) for all slots of a pool at once, vectorised on the host.
"""
import collections
import contextlib
import ctypes
import os
import time
from typing import List, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import lib


# Cumulative counters of the corpus path since the last STATS.clear() (bench.py --config corpus reads them):
# stage_s host packing into pinned memory, h2d_bytes / h2d_s the H2D copies themselves (hipEvents on the copy
# stream), scan_s the native segmenter, buckets, padded / real samples staged.
STATS = collections.defaultdict(float)
TRACE = None             # bring-up: set to a list to collect (slab, t before on_slab, t after, t after stage) from the refill loop


def chunk_size(sr: int) -> int:
    if sr not in (8000, 16000):
        raise ValueError("Supported sampling rates: [8000, 16000] (or multiply of 16000)")
    return 512 if sr == 16000 else 256


def _rates(sr: int):
    """(rate of the net that serves `sr`, decimation step k, INPUT samples per chunk): a multiple of 16 kHz runs on the 16 kHz net
    over every k-th sample (src/silero_vad/utils_vad.py:301-307, JIT!/vad/model/vad_annotator.py:104-112).  The corpus schedulers
    keep such recordings at their raw rate all the way into HBM -- lengths, buckets, slabs and row pitches count RAW samples, a chunk
    is 512 k of them -- and the frontend's loads take every k-th sample (csrc/fft_wave.hpp load_vec<SL, DEC>): no decimated copy on
    the host, none on the device."""
    if sr > 16000 and sr % 16000 == 0:
        return 16000, sr // 16000, 512 * (sr // 16000)
    return sr, 1, chunk_size(sr)


# ---- offline: ragged corpora ------------------------------------------------------------------------
def _bucket_cuts(L: np.ndarray, max_waste: float, max_bytes: int, itemsize: int):
    """The greedy bucketing of RaggedPlan on arrays: -> (order, cuts) with order = the non-empty recordings by descending length (ties in
    index order) and bucket k = order[cuts[k] : cuts[k + 1]].  A bucket takes recordings while the zero padding to its first (longest)
    member wastes at most max_waste of its samples and its padded size stays within max_bytes; the waste grows with every (shorter)
    member, so the first violation ends the bucket -- found block-wise on prefix sums instead of one Python iteration per recording
    (151 552 recordings of a corpus shard: 85 ms of planning in front of the first upload before, ~10 after)."""
    live = np.flatnonzero(L > 0)
    order = live[np.argsort(-L[live], kind="stable")]
    Ls = L[order]
    cs = np.zeros(len(order) + 1, dtype=np.int64)
    np.cumsum(Ls, out=cs[1:])
    cuts = [0]
    s, n = 0, len(order)
    while s < n:
        mx = int(Ls[s])
        hi = min(n - s, max(1, max_bytes // (mx * itemsize)))     # the first member is always taken
        cnt, k0, blk = hi, 2, 256
        while k0 <= hi:
            k = np.arange(k0, min(hi, k0 + blk - 1) + 1, dtype=np.int64)          # members the bucket would have
            waste = 1.0 - (cs[s + k] - cs[s]) / (mx * k)
            bad = np.flatnonzero(waste > max_waste)
            if len(bad):
                cnt = int(k[bad[0]]) - 1
                break
            k0 += blk
            blk *= 2
        s += cnt
        cuts.append(s)
    return order, cuts


class RaggedPlan:
    """Buckets of recording indices, each processed as one lock-step batch.

    ``lengths`` in samples.  A bucket holds recordings whose zero-padding to the bucket's longest
    member wastes at most ``max_waste`` of the bucket's samples, and at most ``max_bytes`` of
    staged PCM (``itemsize`` bytes per sample)."""

    def __init__(self, lengths: Sequence[int], max_waste: float = 0.15, max_bytes: int = 1 << 30,
                 itemsize: int = 4):
        L = np.asarray(lengths, dtype=np.int64).reshape(-1)
        self.lengths = L.tolist()
        self.empty = np.flatnonzero(L <= 0).tolist()
        order, cuts = _bucket_cuts(L, max_waste, max_bytes, itemsize)
        self.buckets: List[List[int]] = [order[cuts[k]:cuts[k + 1]].tolist() for k in range(len(cuts) - 1)]

    def padded_samples(self) -> int:
        return sum(self.lengths[b[0]] * len(b) for b in self.buckets)

    def real_samples(self) -> int:
        return sum(n for n in self.lengths if n > 0)


class _StagePool:
    """Pinned host buffers + device buffers reused for every bucket (pinned allocation is expensive): one more slot
    than there are compute lanes, so that staging + H2D of the next bucket overlap the kernels of the buckets in flight."""

    def __init__(self, device, slots=2):
        self.device = device
        self.slots = slots
        self.host = [None] * slots
        self.dev = [None] * slots
        self.done = [None] * slots          # event: H2D of this slot finished (host buffer reusable)
        self.consumed = [None] * slots      # event: the kernels that read this slot's device buffer finished
        self.meta = [None] * slots          # pinned int64 scratch per slot (refill: scatter indices, resets)
        self.stream = torch.cuda.Stream(device)

    def meta_buffer(self, i, n):
        """Pinned int64[n] owned by slot i (valid until the slot is handed out again): `tensor.pin_memory()` per slab
        costs a pinned allocation each time -- 5.6 ms for 1 MB on the MI355X hosts, more than the slab's kernels."""
        if self.meta[i] is None or self.meta[i].numel() < n:
            self.meta[i] = torch.empty(max(n, 1024), dtype=torch.int64, pin_memory=True)
        return self.meta[i][:n]

    def get(self, k, nbytes, dev_bytes=None):
        """Slot for bucket k with `nbytes` of pinned staging (0: the source is pinned already, no staging) and
        `dev_bytes` (default: nbytes) of device buffer."""
        i = k % self.slots
        dev_bytes = nbytes if dev_bytes is None else dev_bytes
        t0 = time.perf_counter()
        if self.done[i] is not None:
            self.done[i].synchronize()
        STATS["slot_wait_s"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        if nbytes and (self.host[i] is None or self.host[i].numel() < nbytes):
            self.host[i] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
        if self.dev[i] is None or self.dev[i].numel() < dev_bytes:
            cap = max(dev_bytes, 1 << 20)
            # The device buffer is written on self.stream first: allocate it there, so that a block the caching
            # allocator recycles is ordered after its previous use on that stream, and make self.stream wait for
            # whatever the consumer stream still has in flight (the block may come from its pool, e.g. the
            # previous call's probs during their D2H copy).  Consumers call record_stream() on their views.
            cur = torch.cuda.current_stream(self.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.dev[i] = torch.empty(cap, dtype=torch.uint8, device=self.device)
            self.consumed[i] = None
            STATS["slot_allocs"] += 1
        STATS["slot_alloc_s"] += time.perf_counter() - t0
        return i


def _distinct_queue_stream(engine, device, others, tries=12, priority=0):
    """A torch stream whose kernels demonstrably run BESIDE those of every stream in `others` (vad_streams_overlap).  torch hands out
    streams from a pool, the HIP runtime maps them onto ~4 hardware queues in the order of their first use, and two streams on one
    queue serialise: an upload kernel that lands on a compute lane's queue alternates with the lane instead of overlapping it -- the
    scattered-pinned routes then read half the link (profiles/r05_ingest_queues.md).  torch's pool hands its streams out round-robin
    whatever is still alive, so the search simply asks for the next one; every candidate, the last one included, is probed, and when
    none of `tries` candidates runs beside all of `others` the best one found is returned and STATS["stream_collisions"] says so (the
    caller's pipeline is then correct but partly serial).  The probe synchronises the streams involved (~1 ms per pair): it is skipped
    -- a plain new stream is returned -- while the current stream is being captured into a graph."""
    st = torch.cuda.Stream(device, priority=priority)
    if engine is None or not hasattr(engine, "streams_overlap") or torch.cuda.is_current_stream_capturing():
        return st
    best, best_hits = st, -1
    for k in range(tries):
        hits = sum(1 for o in others if engine.streams_overlap(o, st))
        if hits > best_hits:
            best, best_hits = st, hits
        if hits == len(others):
            break
        STATS["stream_retries"] += 1
        if k + 1 < tries:
            st = torch.cuda.Stream(device, priority=priority)
    if best_hits < len(others):
        STATS["stream_collisions"] += 1
    return best


def _copy_stream(pool, model, device, others):
    """The staging pool's stream for the arena windows' DMAs, on a hardware queue that carries none of `others` (compute lanes, the
    cut / upload stream): the runtime orders a copy inside its stream's queue, so a 37 ms window DMA on a lane's queue stands in front
    of that lane's kernels (the raw 48 kHz corpus leg read 0.71 instead of 0.96 of the link when an unprobed stream landed there
    after other legs had created streams of their own).  Made once per pool."""
    st = getattr(pool, "copy_stream", None)
    if st is None:
        st = pool.copy_stream = _distinct_queue_stream(getattr(model, "engine", None), device, list(others))
    return st


def _compute_lanes(model, want):
    """[(model, stream)]: lane 0 is the caller's model on the caller's stream; further lanes are CLONES of its engine
    (vad_clone: the same weight images and options, their own scratch -- no second weight copy, no option drift) on
    their own streams.  A bucket's recurrence is latency-bound -- 4.4 us per time step on as few CUs as the bucket has
    stream tiles -- so buckets on different lanes overlap: one lane's recurrence runs beside the other lane's frontend
    instead of in front of it.  Memory: each lane holds its own gx scratch (up to the engine's gx_cap_mib)."""
    cur = torch.cuda.current_stream(model.device)
    lanes = [(model, cur)]
    eng = getattr(model, "engine", None)
    if want > 1 and eng is not None and hasattr(eng, "clone"):
        sibs = getattr(model, "_lane_siblings", None)
        if sibs is None or getattr(model, "_lane_options", None) != dict(eng.options):
            sibs = model._lane_siblings = []             # the primary's options changed: fresh clones carry the new ones
            model._lane_options = dict(eng.options)
        while len(sibs) < want - 1:             # (each lane on a hardware queue of its own: _distinct_queue_stream)
            sibs.append((type(model)(engine=eng.clone()), _distinct_queue_stream(eng, model.device, [cur] + [s for _, s in sibs])))
        lanes.extend(sibs[: want - 1])
    return lanes


class PackedRecordings:
    """Many recordings inside ONE host tensor -- a decoder's output arena, a memory-mapped shard: recording i is
    `base[offsets[i] : offsets[i] + lengths[i]]`.  Accepted wherever the corpus functions take a list of recordings; it
    spares the per-recording Python work (150 000 tensor views cost more host time than their audio costs GPU time) and,
    when `base` is pinned, the whole set is one page-locked source for `vad_upload_rows`."""

    def __init__(self, base: torch.Tensor, offsets, lengths):
        if base.dim() != 1 or base.is_cuda or not base.is_contiguous() or base.dtype not in (torch.int16, torch.float32):
            raise ValueError("base must be a contiguous 1-D CPU tensor of int16 or float32 samples")
        self.base = base
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.lengths = np.ascontiguousarray(lengths, dtype=np.int64)
        if self.offsets.shape != self.lengths.shape or self.offsets.ndim != 1:
            raise ValueError("offsets and lengths need one entry per recording")
        if len(self.offsets) and (self.offsets.min() < 0 or self.lengths.min() < 0
                                  or int((self.offsets + self.lengths).max()) > base.numel()):
            raise ValueError("a recording lies outside base")

    def __len__(self):
        return len(self.offsets)

    def __getitem__(self, i):
        o, m = int(self.offsets[i]), int(self.lengths[i])
        return self.base[o:o + m]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def windows(self, max_bytes: int):
        """Index ranges [lo, hi) of recordings that lie in arena order without overlap and whose span (first sample of the
        first .. last sample of the last) is at most `max_bytes`: each can be brought to the GPU by ONE copy."""
        esz = self.base.element_size()
        out, n = [], len(self)
        lo = 0
        end = self.offsets + self.lengths
        while lo < n:
            hi = lo + 1
            while hi < n and self.offsets[hi] >= end[hi - 1] and (end[hi] - self.offsets[lo]) * esz <= max_bytes:
                hi += 1
            out.append((lo, hi))
            lo = hi
        return out


def _arena_windows(offs: np.ndarray, lens: np.ndarray, limit: int, lead_limit: int = 0):
    """The arena windows of a PackedRecordings (WindowedPlan's rules; also the refill route's window feed): (order, bounds, o, e) --
    `order` = the live recordings in the order they are walked (by offset when nothing overlaps, else as handed over), `bounds` =
    [(lo, hi)] index ranges of `order`, one per window: runs in which the offsets do not turn back, cut greedily where the span would
    exceed `limit` samples (a window always takes its first recording); o / e = first / one-past-last sample of order's recordings.
    On arrays: one searchsorted per window instead of one iteration per recording.  lead_limit: the FIRST windows are cut at lead_limit,
    2 lead_limit, 4 lead_limit ... samples until `limit` is reached (the refill route's first slabs read little and should not wait
    for a whole window)."""
    live = np.flatnonzero(lens > 0)
    by_offset = live[np.argsort(offs[live], kind="stable")]
    ends = offs[by_offset] + lens[by_offset]
    overlap_free = bool(np.all(offs[by_offset][1:] >= ends[:-1])) if len(by_offset) > 1 else True
    order = by_offset if overlap_free else live
    o, e = offs[order], offs[order] + lens[order]
    turns = np.flatnonzero(o[1:] < e[:-1]) + 1 if len(order) > 1 else np.zeros(0, dtype=np.int64)
    starts = np.concatenate([[0], turns, [len(order)]]).astype(np.int64) if len(order) else np.zeros(1, dtype=np.int64)
    bounds = []
    for r in range(len(starts) - 1):
        lo, hi = int(starts[r]), int(starts[r + 1])
        while lo < hi:
            lim = min(limit, lead_limit << min(len(bounds), 40)) if lead_limit else limit
            nxt = lo + max(1, int(np.searchsorted(e[lo:hi], o[lo] + lim, side="right")))     # (e - a0) > limit ends the window
            bounds.append((lo, nxt))
            lo = nxt
    return order, bounds, o, e


class WindowedPlan:
    """RaggedPlan per arena window.  A window is a set of recordings whose span -- first sample of the first .. last sample
    of the last -- is at most `window_bytes` and which lie in it in arena order without overlap, so that ONE DMA of the span
    brings exactly their bytes to the GPU; the recordings of a window are bucketed among themselves.
      * recordings that do not overlap anywhere are walked in arena order (sorted by offset), whatever order they were handed
        over in;
      * a set with overlaps -- a ring buffer that was refilled: the same offsets mean DIFFERENT audio at different times -- is
        walked in the order it was handed over, and a window ends where the order turns back (the wrap of the ring): a byte is
        never copied once for two recordings.
    `density` = live bytes / copied bytes: a sparse set (recordings scattered over a large arena) is better served by the
    gather kernel (streams.ragged_buckets)."""

    def __init__(self, rec: PackedRecordings, max_waste, max_bytes, itemsize, window_bytes, on_windows=None):
        """on_windows(plan): called as soon as the windows, their spans and the density are known and BEFORE the buckets are planned --
        the hook for starting the first windows' DMA while the rest of the planning runs."""
        self.lengths = rec.lengths.tolist()
        offs, lens = rec.offsets, rec.lengths
        self.empty = np.flatnonzero(lens <= 0).tolist()
        order, bounds, o, e = _arena_windows(offs, lens, window_bytes // itemsize)
        live = order
        self.windows = [order[a:b].tolist() for a, b in bounds]      # recording indices of each window (arena order)
        self.span = [(int(o[a]), int(e[b - 1])) for a, b in bounds]  # (first, last + 1) sample
        copied = sum(b - a for a, b in self.span)
        self.density = float(lens[live].sum()) / copied if copied else 0.0
        self.buckets, self.window_of = [], []
        if on_windows is not None:
            on_windows(self)
        for w, (a, b) in enumerate(bounds):
            idx = order[a:b]
            sub_order, cuts = _bucket_cuts(lens[idx], max_waste, max_bytes, itemsize)
            glob = idx[sub_order]
            for k in range(len(cuts) - 1):
                self.buckets.append(glob[cuts[k]:cuts[k + 1]].tolist())
                self.window_of.append(w)

    def mean_window_fill(self):
        return float(np.mean([len(w) for w in self.windows])) if self.windows else 0.0


def _as_packed(audios):
    """A list of recordings that are all contiguous views of ONE host storage (a decoder's output buffer, sliced) as a
    PackedRecordings over that storage; None if they are not."""
    if isinstance(audios, PackedRecordings) or len(audios) < 2:
        return None
    first = audios[0]
    if not torch.is_tensor(first) or first.is_cuda or first.dtype not in (torch.int16, torch.float32):
        return None
    st = first.untyped_storage()
    key, dtype, esz = st.data_ptr(), first.dtype, first.element_size()
    offs = np.empty(len(audios), dtype=np.int64)
    lens = np.empty(len(audios), dtype=np.int64)
    for i, a in enumerate(audios):
        if not torch.is_tensor(a) or a.dtype != dtype or a.dim() != 1 or a.is_cuda or (a.numel() > 1 and a.stride(0) != 1) \
                or a.untyped_storage().data_ptr() != key:
            return None
        offs[i] = a.storage_offset()
        lens[i] = a.shape[0]
    base = torch.empty(0, dtype=dtype).set_(st, 0, (st.nbytes() // esz,))
    return PackedRecordings(base, offs, lens)


def _describe(audios):
    """(is int16, lengths as a list of ints) of a list of recordings or a PackedRecordings."""
    if isinstance(audios, PackedRecordings):
        return audios.base.dtype == torch.int16, audios.lengths.tolist()
    as_i16 = len(audios) > 0 and all(torch.is_tensor(a) and a.dtype == torch.int16 for a in audios)
    return as_i16, [int(a.shape[0]) if hasattr(a, "shape") else len(a) for a in audios]


class _Sources:
    """The recordings of one call as flat arrays: host address and length of each, made contiguous / of one dtype once,
    and whether ALL of them sit in page-locked memory (then no host-side copy is needed at all)."""

    def __init__(self, audios, dtype, check_pinned):
        n = len(audios)
        if isinstance(audios, PackedRecordings):
            if audios.base.dtype != dtype:
                raise TypeError("mixed int16 / float recordings in one call")
            self.keep = [audios.base]
            self.len = audios.lengths
            self.ptr = (audios.base.data_ptr() + audios.offsets * audios.base.element_size()).astype(np.uint64)
            self.ptr[self.len == 0] = 0
            self.pinned = bool(check_pinned and n > 0 and audios.base.is_pinned())
            return
        self.keep = []
        self.ptr = np.zeros(n, dtype=np.uint64)
        self.len = np.zeros(n, dtype=np.int64)
        pinned_storage = {}
        all_pinned = check_pinned and n > 0
        for i, a in enumerate(audios):
            a = a if torch.is_tensor(a) else torch.as_tensor(a)
            if a.dim() != 1:
                raise ValueError("More than one dimension in audio. Are you trying to process audio with 2 channels?")
            if (a.dtype == torch.int16) != (dtype == torch.int16):
                # (an int16 recording among float ones would be read as floats in [-32768, 32767]: never silently)
                raise TypeError("mixed int16 / float recordings in one call")
            if a.dtype != dtype or not a.is_contiguous() or a.is_cuda:
                a = a.to("cpu", dtype).contiguous()      # (a strided view of int16 PCM lands here: same dtype, new layout)
            self.keep.append(a)
            self.ptr[i] = a.data_ptr() if a.numel() else 0
            self.len[i] = a.shape[0]
            if all_pinned and a.numel():
                key = a.untyped_storage().data_ptr()
                if key not in pinned_storage:
                    pinned_storage[key] = bool(a.is_pinned())
                all_pinned = pinned_storage[key]
        self.pinned = bool(all_pinned)

    def tables(self, idxs):
        rows = np.ascontiguousarray(self.ptr[idxs])
        lens = np.ascontiguousarray(self.len[idxs])
        return (rows, lens, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p)),
                lens.ctypes.data_as(ctypes.POINTER(ctypes.c_long)))


def _upload_mode():
    """How pinned recordings reach the GPU: "window" (a PackedRecordings whose recordings lie back to back: one DMA per arena
    window, batches cut on the device; the default where it applies), "gather" (one kernel that reads host memory; the default
    otherwise), "dma" (copy engines, one copy per row), "stage" (force the pageable path).  SILERO_VAD_AMD_UPLOAD overrides."""
    return os.environ.get("SILERO_VAD_AMD_UPLOAD", "")


def _stage_into(src: "_Sources", idxs, width, dst: torch.Tensor):
    """Pack recordings `idxs` into dst[len(idxs), width] (zero padded) with the native threaded copy."""
    _, _, rows_p, lens_p = tabs = src.tables(idxs)
    rc = lib().vad_stage_rows(rows_p, lens_p, len(idxs), width, dst.element_size(), dst.data_ptr(), 0)
    del tabs
    if rc:
        raise _lib.VadError(rc, "vad_stage_rows")


def ragged_buckets(audios: Sequence, model, sampling_rate: int = 16000, max_waste: float = 0.15,
                   max_bytes: int = 256 << 20, plan: RaggedPlan = None, post=None, meta=None, lanes: int = 2, prepare_only: bool = False):
    """Generator over the plan's buckets: yields (indices, probs[len(indices), T_bucket] on the CPU).
    Recording i of a bucket owns the first ceil(len_i / N) entries of its row.

    `post` (GPU models only): a function (probs_dev, indices, meta_dev) -> list of device tensors that is enqueued
    right after the bucket's kernels; the generator then yields (indices, [those tensors on the CPU], probs_dev) and
    the probabilities themselves never leave the GPU (ragged_speech_segments scans them there).  `meta` (indices ->
    small CPU int64 tensor) rides to the GPU with the bucket's PCM on the copy stream (pinned, asynchronous), so that
    nothing in the loop blocks the host on the compute stream.  `lanes`: buckets are issued round-robin to this many
    engines on their own streams (_compute_lanes); results are yielded in bucket order.  `prepare_only`: plan the run, create the lanes
    and their streams, size every lane's scratch and the staging slots for the plan's largest bucket -- everything that allocates --
    and stop (ragged_reserve): a run over the same recordings then allocates nothing.

    Ingest: recordings in pinned host memory go straight to the device batch (vad_upload_rows: no host copy);
    pageable ones are packed into pinned staging by the native threaded copy and copied from there."""
    t_setup = time.perf_counter()
    n = _rates(sampling_rate)[2]                          # input samples per chunk (512 k for a multiple of 16 kHz)
    as_i16, lengths = _describe(audios)
    dtype = torch.int16 if as_i16 else torch.float32
    esz = 2 if as_i16 else 4
    fast = model.audio_forward_device
    dev = getattr(model, "device", None)
    on_gpu = dev is not None and torch.device(dev).type == "cuda"
    mode = _upload_mode()
    lane_list = pool = cur = None
    if on_gpu:                                            # (before the plan: the first windows' DMA starts while the buckets are planned)
        lane_list = _compute_lanes(model, max(1, int(lanes)))
        pool = getattr(model, "_stage_pool", None)
        if pool is None or pool.slots != len(lane_list) + 1:
            pool = model._stage_pool = _StagePool(dev, len(lane_list) + 1)
            # the upload stream beside every compute lane, not in front of one of them
            pool.stream = _distinct_queue_stream(getattr(model, "engine", None), dev, [st for _, st in lane_list])
        cur = torch.cuda.current_stream(dev)
    copies = []                                           # (start event, end event) of every H2D copy, for STATS
    # One large H2D DMA per arena window (copy engines: no CU time, no host time, exactly the live bytes), two windows
    # ahead of the kernels; the padded [rows][pitch] batches are then cut out of the window's device copy at HBM speed.
    win = {"buf": [None, None, None], "free": [None, None, None], "ev": {}, "next": 0, "span": [], "base": None, "stream": None}

    def ensure_window(w):
        while win["next"] <= w and win["next"] < len(win["span"]):
            v = win["next"]
            a, b = win["span"][v]
            j = v % 3
            nb = (b - a) * esz
            if win["buf"][j] is None or win["buf"][j].numel() < nb:
                # (the old block goes back to the caching allocator: its cuts on pool.stream were recorded against it
                #  -- record_stream below -- and the copy stream waits for the last of them before anything else)
                if win["free"][j] is not None:
                    win["stream"].wait_event(win["free"][j])
                with torch.cuda.stream(win["stream"]):
                    win["buf"][j] = torch.empty(max(nb, 1 << 20), dtype=torch.uint8, device=dev)
                win["buf"][j].record_stream(pool.stream)  # allocated on the copy stream, read by the cut kernels on pool.stream
                win["free"][j] = None
            if win["free"][j] is not None:                # every bucket cut from the buffer's previous window is done
                win["stream"].wait_event(win["free"][j])
            with torch.cuda.stream(win["stream"]):
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(win["stream"])
                win["buf"][j][:nb].view(dtype).copy_(win["base"][a:b], non_blocking=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(win["stream"])
            copies.append((e0, e1))
            STATS["h2d_bytes"] += nb
            win["ev"][v] = e1
            win["next"] += 1

    if plan is None and on_gpu and mode in ("", "window") and hasattr(getattr(model, "engine", None), "upload_rows"):
        # recordings that lie in ONE pinned host buffer (a PackedRecordings, or a list of views of one pinned tensor) and cover
        # most of the range they span: one DMA per arena window, batches cut on the device
        packed = audios if isinstance(audios, PackedRecordings) else _as_packed(audios)
        if packed is not None and packed.base.is_pinned():
            wbytes = int(os.environ.get("SILERO_VAD_AMD_WINDOW_BYTES", 0)) or max(2 * max_bytes, 1 << 30)

            def takes_windows(wp_):
                return mode == "window" or (wp_.density >= 0.6 and wp_.mean_window_fill() >= 16)

            def first_windows(wp_):
                # the windows are known, the buckets are not yet: the first three windows' copies start NOW, the bucket planning (tens
                # of ms for a corpus shard) runs beside them
                if prepare_only or not takes_windows(wp_):
                    return
                win["span"], win["base"] = wp_.span, packed.base
                win["stream"] = _copy_stream(pool, model, dev, [st for _, st in lane_list] + [pool.stream])
                win["stream"].wait_stream(cur)            # (the arena was written before this call in stream order, if at all)
                ensure_window(2)

            wp = WindowedPlan(packed, max_waste, max_bytes, esz, window_bytes=wbytes, on_windows=first_windows)
            if takes_windows(wp):
                plan, audios = wp, packed
    plan = plan or RaggedPlan(lengths, max_waste, max_bytes, esz)
    src = _Sources(audios, dtype, check_pinned=on_gpu and mode != "stage" and hasattr(getattr(model, "engine", None), "upload_rows"))
    if not on_gpu:                                        # CPU stand-in models (tests)
        if prepare_only:
            return
        for idxs in plan.buckets:
            width = max(plan.lengths[idxs[0]], n)
            host = torch.empty((len(idxs), width), dtype=dtype)
            _stage_into(src, idxs, width, host)
            yield idxs, fast(host, sampling_rate).cpu()
        return
    direct = src.pinned
    how = 0 if mode == "dma" else 1
    windowed = direct and isinstance(plan, WindowedPlan)
    if windowed:
        if win["stream"] is None:                          # (a plan handed in by the caller, or prepare_only: nothing was started early)
            win["span"], win["base"] = plan.span, audios.base
            win["stream"] = _copy_stream(pool, model, dev, [st for _, st in lane_list] + [pool.stream])
        last_bucket_of = {}
        for k, w in enumerate(plan.window_of):
            last_bucket_of[w] = k
    align = 16 // esz                                      # device rows are 16-byte aligned: the kernels' vector loads
    # every lane's scratch is sized up front for the largest bucket it can meet: a growth inside the loop would
    # synchronise the device and stall all lanes (vad_reserve)
    if plan.buckets and hasattr(getattr(model, "engine", None), "reserve"):
        shapes = [(len(b), (max(plan.lengths[b[0]], n) + n - 1) // n) for b in plan.buckets]
        big = max(shapes, key=lambda bt: ((bt[0] + 15) // 16) * bt[1])
        wide = max(shapes, key=lambda bt: bt[0])
        t_res = time.perf_counter()
        for lane_model, _ in lane_list:
            for bt in {big, wide}:
                lane_model.engine.reserve(sampling_rate, bt[0], bt[1])
        STATS["reserve_s"] += time.perf_counter() - t_res
    if prepare_only:
        if plan.buckets:
            big_b = max(plan.buckets, key=lambda b: len(b) * ((max(plan.lengths[b[0]], n) + align - 1) // align * align))
            nb = len(big_b) * ((max(plan.lengths[big_b[0]], n) + align - 1) // align * align) * esz
            for k in range(pool.slots):
                pool.get(k, nb if not direct else 0, nb)
            if windowed:                                   # the three window buffers: warm torch's allocator with blocks of the largest window
                wb = max((b - a) * esz for a, b in plan.span)
                with torch.cuda.stream(win["stream"]):
                    warm = [torch.empty(max(wb, 1 << 20), dtype=torch.uint8, device=dev) for _ in range(3)]
                del warm
        torch.cuda.synchronize(dev)
        return
    for _, st in lane_list[1:]:
        st.wait_stream(cur)                               # sibling lanes start behind whatever the caller has queued
    # Buckets on different lanes overlap because a bucket's recurrence sits on the few CUs its stream tiles need (16 streams per CU,
    # kernel_rec.hip) beside the other lane's frontend.  The matrix-vector form of the recurrence (kernel_rec_small.hip) finishes a
    # bucket of a few hundred recordings 2-3 x sooner but spreads it over the whole chip, with nothing left to overlap with: the
    # pipelined corpus runs 4 % (window route) to 30 % (gather kernel) slower with it.  Lanes therefore pin the MFMA form.
    import os as _os
    lane_form = _os.environ.get("SILERO_VAD_AMD_LANE_REC", "mfma")     # (A/B: "auto" lets the lanes take the matrix-vector form)
    lane_engines = []
    if len(lane_list) > 1 and lane_form != "auto":
        lane_engines = [e for e in (getattr(m, "engine", None) for m, _ in lane_list) if e is not None and hasattr(e, "set_transient")]
    pinned = {}                                            # engine -> the value it had when the pin was applied

    def pin_lanes(on):
        """The pin holds only while this generator runs: it is taken back before every `yield` (the caller's own engine is lane 0;
        a consumer that never exhausts the generator must not be left with the slower form) and re-applied on resumption.  The
        value to restore is sampled EVERY time the pin goes on (the consumer may have changed the option between two yields), and
        it is restored only if the option still holds the pinned value (a change made while pinned is the consumer's and stays)."""
        for eng_l in lane_engines:
            if on:
                pinned[id(eng_l)] = eng_l.options.get("rec_form", "auto")
                eng_l.set_transient("rec_form", lane_form)
            elif id(eng_l) in pinned:
                before = pinned.pop(id(eng_l))
                if eng_l.options.get("rec_form", "auto") == before:      # (set_transient does not touch .options)
                    eng_l.set_transient("rec_form", before)

    pin_lanes(True)
    import os as _os
    stage_sync = _os.environ.get("SILERO_VAD_AMD_STAGE_SYNC", "1") != "0"      # (0: A/B -- the staged copies wait on the device)

    def stage(k):
        idxs = plan.buckets[k]
        # at least one full window: audio_forward rejects shorter inputs (vad_annotator.py:124)
        L = max(plan.lengths[idxs[0]], n)
        width = (L + align - 1) // align * align          # row pitch
        nbytes = len(idxs) * width * esz
        i = pool.get(k, nbytes if not direct else 0, nbytes)
        if not windowed:
            STATS["h2d_bytes"] += nbytes
        STATS["buckets"] += 1
        STATS["padded"] += len(idxs) * L
        STATS["real"] += sum(plan.lengths[j] for j in idxs)
        d = pool.dev[i][:nbytes].view(dtype).view(len(idxs), width)
        late_wait = None
        if pool.consumed[i] is not None:                  # the device buffer's previous reader is done
            if windowed or (direct and how != 0) or not stage_sync:
                pool.stream.wait_event(pool.consumed[i])
            else:
                late_wait = pool.consumed[i]              # (a DMA: waited for on the HOST, behind the staging -- see below)
        m_host = None
        if meta is not None:                              # in the slot's own pinned scratch (meta_buffer: why)
            mt = meta(idxs)
            m_host = pool.meta_buffer(i, mt.numel()).view(mt.shape)
            m_host.copy_(mt)
        t0 = time.perf_counter()
        if not direct:
            host = pool.host[i][:nbytes].view(dtype).view(len(idxs), width)
            _stage_into(src, idxs, width, host)
            STATS["stage_s"] += time.perf_counter() - t0
            if late_wait is not None:
                # the H2D copy below is a DMA: issued behind an OPEN device-side wait it leaves the copy engines' fast path (the refill
                # window feed's finding, profiles/r06_refill_window_feed.md).  The reader of this slot's previous bucket started three
                # buckets ago and is normally done by the time this one is staged: wait for it here, issue the copy without a dependency
                late_wait.synchronize()
        if windowed:
            w = plan.window_of[k]
            ensure_window(w + 2)                          # this bucket's window and the two after it are on their way
            pool.stream.wait_event(win["ev"][w])
            a0 = plan.span[w][0]
            rows = np.ascontiguousarray(win["buf"][w % 3].data_ptr() + (audios.offsets[idxs] - a0) * esz, dtype=np.uint64)
            lens = np.ascontiguousarray(src.len[idxs])
        with torch.cuda.stream(pool.stream):
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(pool.stream)
            if windowed:
                model.engine.upload_rows(rows.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p)),
                                         lens.ctypes.data_as(ctypes.POINTER(ctypes.c_long)), len(idxs), width, esz, d, 2)
                if last_bucket_of[w] == k:                # the window's buffer may take another window once this cut is done
                    win["free"][w % 3] = torch.cuda.Event()
                    win["free"][w % 3].record(pool.stream)
                STATS["upload_call_s"] += time.perf_counter() - t0
            elif direct:
                if late_wait is not None:                 # (the per-row DMA route: the same rule)
                    late_wait.synchronize()
                tabs = src.tables(idxs)
                model.engine.upload_rows(tabs[2], tabs[3], len(idxs), width, esz, d, how)
                STATS["upload_call_s"] += time.perf_counter() - t0
            else:
                d.copy_(host, non_blocking=True)
            m_dev = m_host.to(dev, non_blocking=True) if m_host is not None else None
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(pool.stream)
        if not windowed:
            copies.append((ev0, ev))
        pool.done[i] = ev
        return d[:, :L], ev, i, m_dev, m_host

    def finish(done_bucket):
        idxs, outs, _, probs_dev = done_bucket
        if post is None:
            return idxs, outs[0]
        return idxs, outs, probs_dev

    STATS["setup_s"] += time.perf_counter() - t_setup
    try:
        staged = stage(0) if plan.buckets else None
        inflight = []                                     # buckets enqueued and not yet yielded, oldest first
        for k, idxs in enumerate(plan.buckets):
            x, ev, slot, m_dev, _keep = staged
            lane_model, lane_stream = lane_list[k % len(lane_list)]
            lane_stream.wait_event(ev)
            with torch.cuda.stream(lane_stream):
                x.record_stream(lane_stream)              # allocated on pool.stream, read on the lane's stream
                if m_dev is not None:
                    m_dev.record_stream(lane_stream)
                probs = lane_model.audio_forward_device(x, sampling_rate)   # asynchronous
                pool.consumed[slot] = torch.cuda.Event()
                pool.consumed[slot].record(lane_stream)
                back = [probs] if post is None else post(probs, idxs, m_dev)
                outs = []
                for o in back:                            # pinned blocks come from torch's caching host allocator
                    h = torch.empty(o.shape, dtype=o.dtype, pin_memory=True)
                    h.copy_(o, non_blocking=True)
                    STATS["d2h_bytes"] += h.numel() * h.element_size()
                    outs.append(h)
                done = torch.cuda.Event()
                done.record(lane_stream)
            staged = stage(k + 1) if k + 1 < len(plan.buckets) else None   # the next bucket is on its way meanwhile
            inflight.append((idxs, outs, done, probs))
            while len(inflight) > len(lane_list):
                first = inflight.pop(0)
                t_w = time.perf_counter()
                first[2].synchronize()
                STATS["result_wait_s"] += time.perf_counter() - t_w
                pin_lanes(False)
                yield finish(first)
                pin_lanes(True)
        for first in inflight:
            t_w = time.perf_counter()
            first[2].synchronize()
            STATS["result_wait_s"] += time.perf_counter() - t_w
            pin_lanes(False)
            yield finish(first)
            pin_lanes(True)
    finally:
        pin_lanes(False)
        # also when the consumer stops early or an exception propagates: the sibling lanes' work is ordered before
        # whatever the caller enqueues next, and the copy events are drained
        for _, st in lane_list[1:]:
            cur.wait_stream(st)
        cur.wait_stream(pool.stream)
        if windowed:                                      # cuts still pending on pool.stream read the window buffers: the copy
            win["stream"].wait_stream(pool.stream)        # stream (where they were allocated and will be freed) falls in behind them
            cur.wait_stream(win["stream"])
        for a, b in copies:
            b.synchronize()
            STATS["h2d_s"] += a.elapsed_time(b) / 1e3


def ragged_reserve(audios: Sequence, model, sampling_rate: int = 16000, max_waste: float = 0.15, max_bytes: int = 256 << 20,
                   lanes: int = 2):
    """Everything a `ragged_probs` / `ragged_speech_segments` run over these recordings would allocate -- the compute lanes and their
    streams, every lane's scratch (vad_reserve: a later growth synchronises the device and, on some boxes, costs 0.1-0.2 s of
    hipFree / hipMalloc for the multi-GB gx scratch: profiles/r05_ingest_routes.md), the staging slots -- sized for the plan's
    largest bucket, up front.  Call it once before a corpus run (or a timed region); the run itself then allocates nothing."""
    for _ in ragged_buckets(audios, model, sampling_rate, max_waste, max_bytes, lanes=lanes, prepare_only=True):
        pass


def ragged_probs(audios: Sequence, model, sampling_rate: int = 16000, max_waste: float = 0.15,
                 max_bytes: int = 256 << 20, plan: RaggedPlan = None) -> List[torch.Tensor]:
    """Speech probabilities of many recordings of different lengths.

    Returns one 1-D CPU float tensor per recording (ceil(len / N) entries), bit-identical to
    ``model.audio_forward(audio[None], sr)[0]`` on that recording alone.  ``audios`` may be float
    tensors in [-1, 1] or int16 PCM (all of one kind).  `model` needs ``audio_forward_device``
    (HipSileroVAD); staging + H2D of bucket k+1 overlap the kernels of bucket k.  `sampling_rate` may be a multiple of 16000: the
    recordings then stay at their raw rate all the way into HBM (_rates)."""
    n = _rates(sampling_rate)[2]
    lengths = _describe(audios)[1]
    out: List[torch.Tensor] = [torch.empty(0)] * len(audios)
    for idxs, probs in ragged_buckets(audios, model, sampling_rate, max_waste, max_bytes, plan):
        for row, i in enumerate(idxs):
            out[i] = probs[row, : (lengths[i] + n - 1) // n].clone()
    return out


def ragged_speech_segments(audios: Sequence, model, sampling_rate: int = 16000, max_waste: float = 0.15,
                           max_bytes: int = 256 << 20, threads: int = 0, device_scan: bool = None,
                           as_arrays: bool = False, **scan_kw) -> List[list]:
    """Speech segments (sample indices) of many recordings: bucketed GPU batches, then the segmenter.
    scan_kw: the threshold/duration arguments of get_speech_timestamps.  `audios`: a list of 1-D tensors or a
    PackedRecordings.  as_arrays: return (counts int64[n], segments int64[sum(counts), 2]) -- recording i owns rows
    cumsum(counts)[i-1] .. -- instead of a list of lists of dicts (for corpora of 10^5+ recordings the dicts cost more
    host time than the GPU work; the arrays are also what a host-side gather to rank 0 wants).

    device_scan (default: on for a GPU model backed by the native engine): the scan runs on the GPU right behind the
    kernels of its bucket (vad_segment_probs_device, one lane per recording) and only counts + segment lists come
    back over PCIe; otherwise the probabilities are copied to the host and scanned by the native threaded scanner.
    Both give the same segments (one source, csrc/scanner.hpp).  `sampling_rate` may be a multiple of 16000 (raw recordings, _rates):
    the segments are then in samples of the 16 kHz signal x[::k], as the reference's scan sees it (utils_vad.py:301-307)."""
    net_sr, dec, n = _rates(sampling_rate)
    lengths = _describe(audios)[1]
    dev = getattr(model, "device", None)
    on_gpu = dev is not None and torch.device(dev).type == "cuda"
    if device_scan is None:
        device_scan = on_gpu and hasattr(getattr(model, "engine", None), "_h")
    counts_all = np.zeros(len(lengths), dtype=np.int64)
    parts = []                                                 # (indices, counts, segs[rows, cap, 2]) per bucket
    if device_scan:
        params = _segment_params(net_sr, **scan_kw)
        cap0 = 24                                              # segments per recording copied back optimistically
        lens_t = torch.as_tensor(lengths, dtype=torch.int64)

        def meta(idxs):                                        # [2, n]: chunks and (16 kHz) samples of each recording of the bucket
            lens = lens_t[idxs]
            return torch.stack([(lens + n - 1) // n, (lens + dec - 1) // dec])

        def post(probs_dev, idxs, both):
            counts, segs = _device_scan(model.engine, probs_dev, both[0], both[1], params, cap0)
            return [counts, segs]

        for idxs, (counts, segs), probs_dev in ragged_buckets(audios, model, sampling_rate, max_waste, max_bytes,
                                                              post=post, meta=meta):
            t0 = time.perf_counter()
            cnt = counts.numpy()
            if len(cnt) and int(cnt.max()) > cap0:             # rare: rescan this bucket with room for all
                both = meta(idxs).to(probs_dev.device)
                c2, s2 = _device_scan(model.engine, probs_dev, both[0], both[1], params, int(cnt.max()))
                cnt, segs = c2.cpu().numpy(), s2.cpu()
            parts.append((np.asarray(idxs, dtype=np.int64), cnt.copy(), segs.numpy()))
            STATS["scan_s"] += time.perf_counter() - t0
    else:
        for idxs, probs in ragged_buckets(audios, model, sampling_rate, max_waste, max_bytes):
            lens = [lengths[i] for i in idxs]
            t0 = time.perf_counter()
            segs = segment_probs_batch(probs, [(m + n - 1) // n for m in lens], [(m + dec - 1) // dec for m in lens], net_sr,
                                       threads=threads, **scan_kw)
            cnt = np.asarray([len(sg) for sg in segs], dtype=np.int64)
            arr = np.zeros((len(segs), max(1, int(cnt.max()) if len(cnt) else 1), 2), dtype=np.int64)
            for r, sg in enumerate(segs):
                for k, d in enumerate(sg):
                    arr[r, k] = (d["start"], d["end"])
            parts.append((np.asarray(idxs, dtype=np.int64), cnt, arr))
            STATS["scan_s"] += time.perf_counter() - t0
    t0 = time.perf_counter()
    for idxs, cnt, _ in parts:
        counts_all[idxs] = cnt
    first = np.concatenate([[0], np.cumsum(counts_all)])
    flat = np.zeros((int(first[-1]), 2), dtype=np.int64)
    for idxs, cnt, sg in parts:                                # scatter every bucket's segments to their recording's rows
        if not len(idxs) or not cnt.any():
            continue
        k = np.arange(sg.shape[1])[None, :]
        mask = k < cnt[:, None]
        flat[(first[idxs][:, None] + k)[mask]] = sg[mask]
    STATS["scan_s"] += time.perf_counter() - t0
    if as_arrays:
        return counts_all, flat
    fl = flat.tolist()
    return [[{"start": a, "end": b} for a, b in fl[first[i]:first[i + 1]]] for i in range(len(lengths))]


def _segment_params(sampling_rate=16000, threshold=0.5, neg_threshold=None, min_speech_duration_ms=250,
                    max_speech_duration_s=float("inf"), min_silence_duration_ms=100, speech_pad_ms=30,
                    min_silence_at_max_speech=98, use_max_poss_sil_at_max_speech=True):
    p = _lib.SegmentParams()
    lib().vad_segment_params_default(ctypes.byref(p), int(sampling_rate))
    p.threshold = float(threshold)
    p.neg_threshold = -1.0 if neg_threshold is None else float(neg_threshold)
    p.min_speech_duration_ms = int(min_speech_duration_ms)
    p.max_speech_duration_s = float(max_speech_duration_s)
    p.min_silence_duration_ms = int(min_silence_duration_ms)
    p.speech_pad_ms = int(speech_pad_ms)
    p.min_silence_at_max_speech_ms = int(min_silence_at_max_speech)
    p.use_max_poss_sil_at_max_speech = 1 if use_max_poss_sil_at_max_speech else 0
    return p


def _device_scan(engine, probs_dev, n_chunks_dev, audio_len_dev, params, cap, row_offsets=None):
    """Enqueue the GPU scan of probs_dev[B, T] on the current stream -> (counts[B], segs[B, cap, 2]) int64, device.
    With `row_offsets` (device int64[B]) stream i's probabilities start at probs_dev.flatten()[row_offsets[i]]."""
    B, T = probs_dev.shape
    if row_offsets is not None:
        B = row_offsets.shape[0]
    counts = torch.empty((B,), dtype=torch.int64, device=probs_dev.device)
    segs = torch.empty((B, max(cap, 1), 2), dtype=torch.int64, device=probs_dev.device)
    _lib.check(engine._h, lib().vad_segment_probs_device(
        engine._h, probs_dev.data_ptr(), probs_dev.stride(0) if probs_dev.shape[0] > 1 else T,
        row_offsets.data_ptr() if row_offsets is not None else None, B,
        n_chunks_dev.data_ptr() if n_chunks_dev is not None else None, T, audio_len_dev.data_ptr(),
        ctypes.byref(params), segs.data_ptr(), segs.shape[1], counts.data_ptr(),
        ctypes.c_void_p(torch.cuda.current_stream(probs_dev.device).cuda_stream)))
    return counts, segs


def segment_probs_batch_device(engine, probs_dev: torch.Tensor, n_chunks, audio_lengths, sampling_rate=16000,
                               **scan_kw) -> List[list]:
    """`segment_probs_batch` for probabilities that are already on the GPU (probs_dev[B, T], CUDA): the scan runs
    there (vad_segment_probs_device) and only the segment lists are copied back."""
    B, T = probs_dev.shape
    params = _segment_params(sampling_rate, **scan_kw)
    both = torch.stack([torch.as_tensor(n_chunks, dtype=torch.int64), torch.as_tensor(audio_lengths, dtype=torch.int64)])
    if both.shape != (2, B):
        raise ValueError("n_chunks and audio_lengths need one entry per row of probs")
    if int(both[0].max() if B else 0) > T or int(both.min() if B else 0) < 0:
        raise ValueError("n_chunks must lie in [0, T] and audio lengths must not be negative")
    both = both.to(probs_dev.device)
    cap = 24
    while True:
        counts, segs = _device_scan(engine, probs_dev.contiguous(), both[0], both[1], params, cap)
        cnt = counts.cpu().numpy()
        if B == 0 or int(cnt.max()) <= cap:
            break
        cap = int(cnt.max())
    m = int(cnt.max()) if B else 0
    sg = segs[:, :max(m, 1)].cpu().numpy()
    return [[{"start": int(a), "end": int(b)} for a, b in sg[i, : cnt[i]]] for i in range(B)]


def segment_probs_batch(probs: torch.Tensor, n_chunks, audio_lengths, sampling_rate=16000, threshold=0.5,
                        neg_threshold=None, min_speech_duration_ms=250,
                        max_speech_duration_s=float("inf"), min_silence_duration_ms=100,
                        speech_pad_ms=30, min_silence_at_max_speech=98,
                        use_max_poss_sil_at_max_speech=True, threads=0) -> List[list]:
    """`timestamps.segment_probs` for every row of probs[B, T] in one native call (host threads)."""
    probs = torch.as_tensor(probs, dtype=torch.float32).contiguous().cpu()
    B, T = probs.shape
    p = _segment_params(sampling_rate, threshold, neg_threshold, min_speech_duration_ms, max_speech_duration_s,
                        min_silence_duration_ms, speech_pad_ms, min_silence_at_max_speech,
                        use_max_poss_sil_at_max_speech)
    nck = np.ascontiguousarray(n_chunks, dtype=np.int64)
    alen = np.ascontiguousarray(audio_lengths, dtype=np.int64)
    if nck.shape != (B,) or alen.shape != (B,):
        raise ValueError("n_chunks and audio_lengths need one entry per row of probs")
    cap = T // 2 + 2
    lp = ctypes.POINTER(ctypes.c_long)
    while True:     # counts[] may exceed cap (e.g. min_silence 0 with a small max_speech): grow and rescan
        segs = np.zeros((B, cap, 2), dtype=np.int64)
        counts = np.zeros(B, dtype=np.int64)
        rc = lib().vad_segment_probs_batch(
            ctypes.cast(probs.data_ptr(), _lib.f32p) if B * T else None, T, B,
            nck.ctypes.data_as(lp), alen.ctypes.data_as(lp), ctypes.byref(p),
            ctypes.cast(segs.ctypes.data, ctypes.POINTER(_lib.Segment)), cap, counts.ctypes.data_as(lp),
            int(threads))
        if rc < 0:
            from .timestamps import _raise_scan_error
            _raise_scan_error(rc)
        if B == 0 or int(counts.max()) <= cap:
            break
        cap = int(counts.max())
    return [[{"start": int(s), "end": int(e)} for s, e in segs[i, : counts[i]]] for i in range(B)]


# ---- offline: continuous refill ------------------------------------------------------------------------------------
class RefillPlan:
    """A fixed number of stream slots, processed in time slabs of `slab_chunks` chunks; a slot whose recording ends
    inside a slab is handed the next recording at the following slab boundary (state and context reset) instead of
    idling until the longest member of a bucket is done.  This is the reference's padded lock-step batch
    (`SileroVadPadder`, tuning/utils.py:146-160; bounded blocks, examples/onnx_sequence/run.py:34-56) with rows
    retired and re-admitted: the waste per recording is below one slab, whatever the spread of the lengths.

    The schedule depends on the lengths only, so it is computed up front: `slabs` is a list of
    (slot, recording, first_sample, n_samples, reset) tuples per slab.  Recordings are admitted longest first, behind a few of the
    shortest (so that results start to flow at once)."""

    def __init__(self, lengths: Sequence[int], slots: int, slab_chunks: int, chunk: int, order=None, ramp: int = 1):
        """order: the admission order (recording indices; empty recordings are skipped) instead of longest-first -- the window feed
        admits recordings in ARENA order, so that the bytes the slots need next are the bytes the next window DMA brings.
        ramp (with order): the slots START over `ramp` slabs, a 1 / ramp of them per slab, lowest slots first -- the first slab then
        needs the audio of slots / ramp recordings instead of all slots' (2 GiB for 2 048 slots of 30 s recordings: 37 ms of link
        time in front of the first kernel), and the first recordings retire that much earlier."""
        lens = self.lengths = np.ascontiguousarray(lengths, dtype=np.int64).reshape(-1)     # (an array: a shard has 10^5 recordings)
        self.slots, self.slab_chunks, self.chunk = int(slots), int(slab_chunks), int(chunk)
        if self.slots < 1 or self.slab_chunks < 1:
            raise ValueError("slots and slab_chunks must be positive")
        width = self.slab_chunks * self.chunk
        live = np.flatnonzero(lens > 0)
        if order is not None:
            queue = np.ascontiguousarray(order, dtype=np.int64).reshape(-1)
            queue = queue[lens[queue] > 0]
            if len(queue) != len(live) or len(np.unique(queue)) != len(queue):
                raise ValueError("order must name every non-empty recording once")
        else:
            queue = live[np.argsort(-lens[live], kind="stable")]  # longest first, ties in input order
        # ... except that a sixteenth of the slots START with the shortest recordings: with the longest in every slot nothing retires
        # before the longest recording's last slab (a shard of 30 s recordings: 13 of 580 slabs, 5 % of the run, before the first result
        # reaches the host); the shortest retire after their own few slabs and the results flow from then on.  They are taken from the end
        # of the queue, where they would have filled the last slabs' gaps: a 1 / 16 of the slots' worth of the shortest does not move the
        # makespan of a shard.
        early = min(self.slots // 16, len(queue) - self.slots) if len(queue) > self.slots else 0
        if early > 0 and order is None:
            queue = np.concatenate([queue[len(queue) - early:][::-1], queue[:len(queue) - early]])
        self.empty = np.flatnonzero(lens <= 0).tolist()
        need = np.ascontiguousarray((lens[queue] + width - 1) // width)   # slabs each recording occupies its slot for
        # event-driven form of "at every slab boundary, every free slot (in slot order) takes the next recording": a heap of (slab at
        # which the slot becomes free, slot) -- native (vad_refill_schedule: a corpus shard has 10^5 recordings)
        lp = ctypes.POINTER(ctypes.c_long)
        ramp = int(ramp) if order is not None else 1
        g = self.slots // ramp if ramp > 1 else 0
        if ramp > 1 and g >= 1 and len(queue) > self.slots:
            # the staggered start, expressed in the scheduler's own terms: behind the first g recordings (slots 0 .. g - 1 at slab 0)
            # the queue holds one PLACEHOLDER per remaining slot that keeps it busy for its delay (slot s starts at slab s // g, the
            # last slots at ramp - 1); the recordings behind them take the slots as the placeholders end, in queue order
            delay = np.minimum(np.arange(g, self.slots, dtype=np.int64) // g, ramp - 1)
            need_q = np.ascontiguousarray(np.concatenate([need[:g], delay, need[g:]]))
            start_q = np.zeros(len(need_q), dtype=np.int64)
            slot_q = np.zeros(len(need_q), dtype=np.int64)
            if lib().vad_refill_schedule(need_q.ctypes.data_as(lp), len(need_q), self.slots, start_q.ctypes.data_as(lp), slot_q.ctypes.data_as(lp)):
                raise ValueError("vad_refill_schedule: bad arguments")
            real = np.ones(len(need_q), dtype=bool)
            real[g:g + len(delay)] = False
            start, slot = np.ascontiguousarray(start_q[real]), np.ascontiguousarray(slot_q[real])
        else:
            start = np.zeros(len(queue), dtype=np.int64)
            slot = np.zeros(len(queue), dtype=np.int64)
            if lib().vad_refill_schedule(need.ctypes.data_as(lp), len(queue), self.slots, start.ctypes.data_as(lp), slot.ctypes.data_as(lp)):
                raise ValueError("vad_refill_schedule: bad arguments")
        # one row per (recording, slab it is active in): [slot, recording, first sample, samples, reset], by slab, then by slot (native:
        # 1.3 M rows for a shard -- vad_refill_table)
        n_slabs = int((start + need).max()) if len(queue) else 0
        rows = np.empty((int(need.sum()), 5), dtype=np.int64)
        cuts = np.zeros(n_slabs + 1, dtype=np.int64)
        rec = np.ascontiguousarray(queue, dtype=np.int64)
        qlen = np.ascontiguousarray(lens[queue])
        got = lib().vad_refill_table(need.ctypes.data_as(lp), start.ctypes.data_as(lp), slot.ctypes.data_as(lp), rec.ctypes.data_as(lp),
                                     qlen.ctypes.data_as(lp), len(queue), self.slots, width, n_slabs, rows.ctypes.data_as(lp),
                                     cuts.ctypes.data_as(lp))
        if got != len(rows):
            raise ValueError("vad_refill_table: bad arguments")
        # the schedule as arrays (slot, recording, first sample, samples, reset flag) per slab, for the vectorised stager
        self.slab_arrays = [rows[cuts[k]:cuts[k + 1]] for k in range(n_slabs)]
        # first / last slab each recording is active in (-1: empty recording): what the window feed plans its device buffers on
        self.first_slab = np.full(len(lens), -1, dtype=np.int64)
        self.last_slab = np.full(len(lens), -1, dtype=np.int64)
        self.first_slab[queue] = start
        self.last_slab[queue] = start + need - 1

    @property
    def slabs(self) -> List[list]:
        """The schedule as lists of (slot, recording, first sample, samples, reset) tuples per slab."""
        return [[(int(a), int(b), int(c), int(d), bool(e)) for a, b, c, d, e in arr] for arr in self.slab_arrays]

    def n_chunks(self, i: int) -> int:
        return (int(self.lengths[i]) + self.chunk - 1) // self.chunk

    def padded_chunks(self) -> int:
        return len(self.slab_arrays) * self.slots * self.slab_chunks

    def real_chunks(self) -> int:
        live = self.lengths[self.lengths > 0]
        return int(((live + self.chunk - 1) // self.chunk).sum())


def _assign_window_buffers(first: np.ndarray, last: np.ndarray, ahead: int, slack: int = 1):
    """Device buffers for the refill route's arena windows, planned up front (the schedule is static).  Window w is first read by
    the gather of slab first[w], last by the gather of slab last[w]; its DMA is ISSUED while slab issue[w] = max(0, first[w] - ahead)
    is being staged, before that slab's gather.  A buffer may take window w if the window it held was last read by a slab BEFORE
    issue[w] (its release event exists by then).  Windows are issued in index order (first[] is non-decreasing in arena order).
    `slack`: the previous window must have been last read at least `slack` slabs before the issue slab -- the refill loop knows, when it
    stages slab k, that the uploads of slabs <= k - 2 are COMPLETE (it waited for that staging slot), so with slack 2 a window's DMA never
    carries an open device-side dependency (a copy that does is taken off the copy engines' fast path: 35.6 instead of 19 ms per GiB for
    hundreds of ms, profiles/r06_refill_window_feed.md).
    Returns (buffer index per window, number of buffers, issue slab per window)."""
    first = np.asarray(first, dtype=np.int64)
    last = np.asarray(last, dtype=np.int64)
    issue = np.maximum(first - int(ahead), 0)
    issue = np.maximum.accumulate(issue) if len(issue) else issue      # in-order issue: a window is never issued before its predecessor
    buf = np.zeros(len(first), dtype=np.int64)
    import heapq
    busy = []                                                           # (last slab of the window held, buffer)
    free = []
    n_buf = 0
    for w in range(len(first)):
        while busy and busy[0][0] <= issue[w] - slack:
            heapq.heappush(free, heapq.heappop(busy)[1])
        if free:
            j = heapq.heappop(free)
        else:
            j, n_buf = n_buf, n_buf + 1
        buf[w] = j
        heapq.heappush(busy, (int(last[w]), j))
    return buf, n_buf, issue


def _refill_iter(audios: Sequence, model, sampling_rate: int, slots: int, slab_chunks: int, plan: "RefillPlan" = None, on_slab=None,
                 prepare_only: bool = False):
    """The continuous-refill loop (RefillPlan) as a generator: stages slab k + 1 while the kernels of slab k run, scatters every slab's
    probabilities into one flat device tensor (recording i owns out_flat[base[i] : base[i + 1]]) and yields k once slab k has been
    ENQUEUED.  `on_slab(k, finished, out_flat, base)` is called right before that, with the recordings whose last chunk lies in slab k
    (np.int64 array, may be empty): the hook for work that follows a recording's retirement in stream order.  The generator's return
    value is (out_flat, base, plan).  `prepare_only`: plan the run and make everything that allocates -- the window buffers, the staging
    slots, the engine's scratch, the flat probability tensor -- then stop (refill_reserve): a run over the same recordings allocates
    nothing (7 GiB of window buffers for a 1 263 h shard are 60 ms when they come fresh from the driver, 0.5 s when an outgrown block
    has to go back first)."""
    t_setup = time.perf_counter()
    net_sr, _, n = _rates(sampling_rate)
    eng = model.engine
    dev = torch.device(getattr(eng, "torch_device", None) or torch.device("cuda", eng.device))
    on_gpu = dev.type == "cuda"
    as_i16, lengths = _describe(audios)
    dtype, esz = (torch.int16, 2) if as_i16 else (torch.float32, 4)
    lens_np = np.asarray(lengths, dtype=np.int64).reshape(-1)
    slots = max(1, min(int(slots), int((lens_np > 0).sum())))
    mode = _upload_mode()
    # The window feed (recordings that lie in ONE pinned arena, GPU engine, no plan handed in): recordings are admitted in ARENA order,
    # the arena goes to the GPU by one DMA per window (copy engines: no CU time, exactly the live bytes) a few slabs ahead of the
    # slots that read it, and a slab's rows are cut out of the windows' device copies at HBM speed -- instead of a gather kernel that
    # reads 128 KB row pieces over PCIe beside the frontend (0.90 of the link).  The device buffers are planned up front
    # (_assign_window_buffers: the schedule is static); a corpus whose long recordings would pin more than the budget of window
    # buffers (SILERO_VAD_AMD_REFILL_WINDOW_BUDGET, 16 GiB of the 288) keeps the gather route.
    wf = None
    if plan is None and on_gpu and mode in ("", "window") and hasattr(eng, "upload_rows"):
        packed = audios if isinstance(audios, PackedRecordings) else _as_packed(audios)
        if packed is not None and packed.base.is_pinned() and int((lens_np > 0).sum()):
            slab_bytes = slots * slab_chunks * n * esz
            wbytes = int(os.environ.get("SILERO_VAD_AMD_REFILL_WINDOW", 0)) or (1 << 30)        # (256 MiB windows: 0.92 of the link, 1 GiB: 0.95)
            # (the slots start over `ramp` slabs and the first windows are small -- 1/8, 1/4, 1/2 of a window: the first kernel runs
            #  after 7 ms of link time instead of 37, the first recordings retire that much earlier)
            ramp = max(1, int(os.environ.get("SILERO_VAD_AMD_REFILL_RAMP", "8")))
            order, bounds, o_, e_ = _arena_windows(packed.offsets, packed.lengths, wbytes // esz, lead_limit=(wbytes // esz) // 8 if int(os.environ.get("SILERO_VAD_AMD_REFILL_LEAD", ramp > 1)) else 0)
            spans = np.asarray([(o_[a], e_[b - 1]) for a, b in bounds], dtype=np.int64).reshape(-1, 2)
            copied = int((spans[:, 1] - spans[:, 0]).sum())
            dense = mode == "window" or float(lens_np[lens_np > 0].sum()) >= 0.6 * copied
            if dense:
                wplan = RefillPlan(lens_np, slots, slab_chunks, n, order=order, ramp=ramp)
                win_of = np.full(len(lens_np), -1, dtype=np.int64)
                for w, (a, b) in enumerate(bounds):
                    win_of[order[a:b]] = w
                w_first = np.asarray([wplan.first_slab[order[a:b]].min() for a, b in bounds], dtype=np.int64)
                w_last = np.asarray([wplan.last_slab[order[a:b]].max() for a, b in bounds], dtype=np.int64)
                wmax = int((spans[:, 1] - spans[:, 0]).max()) * esz
                wmax = (wmax + 255) // 256 * 256
                ahead = max(2, -(-2 * wmax // max(slab_bytes, 1))) + 1          # two windows' worth of slabs in front of the reader
                slack = int(os.environ.get("SILERO_VAD_AMD_REFILL_SLACK", "2"))      # (1: A/B -- the DMAs may then wait on the device)
                buf_of, n_buf, issue = _assign_window_buffers(w_first, w_last, ahead, slack=slack)
                budget = int(os.environ.get("SILERO_VAD_AMD_REFILL_WINDOW_BUDGET", 0))
                if not budget:                                               # 16 GiB of the 288, less on a device that others fill
                    held = getattr(getattr(model, "_stage_pool", None), "refill_windows", None)
                    free = torch.cuda.mem_get_info(dev)[0] + (held.numel() if held is not None else 0)
                    budget = min(16 << 30, free // 4)
                if n_buf * wmax <= budget:
                    plan = wplan
                    wf = {"packed": packed, "spans": spans, "win_of": win_of, "first": w_first, "last": w_last, "buf_of": buf_of,
                          "n_buf": n_buf, "wmax": wmax, "issue": issue, "next": 0, "ev": {}, "release": {}, "holder": {}, "slack": slack}
                    STATS["refill_window_feed"] += 1
                    STATS["refill_window_buffers"] = max(STATS["refill_window_buffers"], n_buf)
    plan = plan or RefillPlan(lens_np, slots, slab_chunks, n)
    B, S, width = plan.slots, plan.slab_chunks, plan.slab_chunks * n
    src = _Sources(audios, dtype, check_pinned=on_gpu and mode != "stage" and hasattr(eng, "upload_rows"))
    direct = src.pinned                                # pinned recordings: one gather kernel per slab, no host copy
    how = 0 if mode == "dma" else 1
    base = np.zeros(len(audios) + 1, dtype=np.int64)   # recording i owns out_flat[base[i] : base[i] + n_chunks(i)]
    np.cumsum(np.where(lens_np > 0, (lens_np + n - 1) // n, 0), out=base[1:])
    total = int(base[-1])
    out_flat = torch.zeros(total + 1, dtype=torch.float32, device=dev)      # [+1]: sink for the padding chunks
    ctx = torch.zeros((B, chunk_size(net_sr) // 8), dtype=torch.float32, device=dev)
    state = torch.zeros((2, B, 128), dtype=torch.float32, device=dev)
    done = np.zeros(len(audios), dtype=np.int64)       # chunks of each recording already produced
    ctxm = contextlib.nullcontext() if not on_gpu else torch.cuda.device(dev)
    with ctxm:
        if on_gpu:
            pool = getattr(model, "_stage_pool", None)
            cur = torch.cuda.current_stream(dev)
            if pool is None:
                pool = model._stage_pool = _StagePool(dev)
                pool.stream = _distinct_queue_stream(getattr(model, "engine", None), dev, [cur],    # the upload beside the slab's kernels
                                                     priority=int(os.environ.get("SILERO_VAD_AMD_UPLOAD_PRIORITY", "0")))
        STATS["padded"] += plan.padded_chunks() * n
        STATS["real"] += int(lens_np[lens_np > 0].sum())

        ptr0 = src.ptr
        if on_gpu and hasattr(eng, "reserve"):
            eng.reserve(sampling_rate, B, S)
        if wf is not None:
            # the window buffers: one device block, kept on the staging pool from call to call; written by the copy stream, read by
            # the cut kernels on pool.stream
            need_b = wf["n_buf"] * wf["wmax"]
            blk = getattr(pool, "refill_windows", None)
            if blk is None or blk.numel() < need_b:
                if blk is not None:                        # a larger plan than the last call's: the old block goes back to the DRIVER
                    blk = pool.refill_windows = None       # first (GiBs that torch's cache would keep beside the new one)
                    torch.cuda.synchronize(dev)
                    torch.cuda.empty_cache()
                blk = pool.refill_windows = torch.empty(need_b, dtype=torch.uint8, device=dev)
                STATS["slot_allocs"] += 1
            # (the DMAs on a hardware queue that carries neither the slabs' kernels nor the cuts: a copy queued behind a kernel of
            #  another stream on the same queue waits for it)
            wf["stream"] = _copy_stream(pool, model, dev, [cur, pool.stream])
            blk.record_stream(wf["stream"])
            blk.record_stream(pool.stream)
            wf["stream"].wait_stream(cur)                  # (the block's previous use, the arena's writer: both in front of this call)
            wf["stream"].wait_stream(pool.stream)
            wf["blk"] = blk
            # every recording's device address, once: its window's buffer + its place in the window's span
            w_ = wf["win_of"]
            livem = w_ >= 0
            dptr = np.zeros(len(lens_np), dtype=np.uint64)
            dptr[livem] = (blk.data_ptr() + wf["buf_of"][w_[livem]] * wf["wmax"]
                           + (wf["packed"].offsets[livem] - wf["spans"][w_[livem], 0]) * esz).astype(np.uint64)
            ptr0 = dptr

        def issue_windows(upto):
            """start the DMA of every window whose issue slab is <= upto (in order; a buffer's previous window was released by an
            earlier slab's cuts -- _assign_window_buffers)"""
            while wf["next"] < len(wf["first"]) and wf["issue"][wf["next"]] <= upto:
                v = wf["next"]
                j = int(wf["buf_of"][v])
                a, b = (int(x) for x in wf["spans"][v])
                nb = (b - a) * esz
                prev = wf["holder"].get(j)
                if prev is not None:
                    # recorded behind the last cut that read window `prev`, at least two slabs ago: complete by now (stage() has waited
                    # for the upload of slab k - 2), so this returns at once and the DMA below is issued WITHOUT a device-side dependency
                    rel = wf["release"].pop(prev)
                    if wf["slack"] >= 2:
                        rel.synchronize()
                    else:
                        wf["stream"].wait_event(rel)
                with torch.cuda.stream(wf["stream"]):
                    wf["blk"][j * wf["wmax"]: j * wf["wmax"] + nb].view(dtype).copy_(wf["packed"].base[a:b], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(wf["stream"])
                wf["ev"][v] = ev
                wf["holder"][j] = v
                wf["by_first"].setdefault(int(wf["first"][v]), []).append(v)
                STATS["h2d_bytes"] += nb
                wf["next"] += 1

        if wf is not None:
            wf["by_first"] = {}
            wf["by_last"] = {}
            for v, l in enumerate(wf["last"]):
                wf["by_last"].setdefault(int(l), []).append(v)

        # The scatter indices and reset lists of G consecutive slabs cross the link in ONE copy (the schedule is static: a group's are
        # known when its first slab is staged).  A small copy per slab costs the large transfers beside it far more than its bytes:
        # the refill route's window DMAs ran at 0.907 of the link with a 2 MB copy per slab and the scans' three small ones, 0.949
        # without them (profiles/r06_refill_window_feed.md).
        import os as _os
        G = max(1, int(_os.environ.get("SILERO_VAD_AMD_REFILL_GROUP", "8")))
        n_slabs_all = len(plan.slab_arrays)
        grp = {"g": -1, "host": [None, None], "ev": [None, None], "dev": None, "roff": {}, "next": 0}

        def prepare(j):
            """slab j's scatter description and reset list into its group's page-locked buffer: int64 [per slab: first output index of
            every slot (B) | chunks every slot produces (B)] [resets of every slab] -- 48 KB per slab instead of the B S indices
            themselves (2 MB: 0.75 % of the link's bytes on the gather route); the indices are formed on the device (scatter_index).
            One slab per call: the host work stays spread over the slabs."""
            g = j // G
            lo, cnt = g * G, min(G, n_slabs_all - g * G)
            q, b = j - lo, g % 2
            cap = cnt * 3 * B
            if q == 0:
                if on_gpu and grp["ev"][b] is not None:
                    grp["ev"][b].synchronize()                                 # the copy that last read this buffer (two groups ago)
                if grp["host"][b] is None or grp["host"][b].numel() < cap:
                    grp["host"][b] = torch.empty(max(cap, 1024), dtype=torch.int64, pin_memory=on_gpu)
                grp["roff"][g] = [cnt * 2 * B]
            e = plan.slab_arrays[j]                                        # [entries, 5], vectorised bookkeeping
            sl, rec, take = e[:, 0], e[:, 1], e[:, 3]
            nck = (take + n - 1) // n
            hb = grp["host"][b].numpy()
            v = hb[q * 2 * B:(q + 1) * 2 * B]
            v[:] = 0                                                       # a slot without a recording produces nothing
            v[sl] = base[rec] + done[rec]
            v[B + sl] = nck
            done[rec] += nck
            rs = sl[e[:, 4] != 0]
            r0 = grp["roff"][g][-1]
            hb[r0:r0 + len(rs)] = rs
            grp["roff"][g].append(r0 + len(rs))

        def slab_meta(k):
            """(first index per slot, chunks per slot, resets) of slab k as tensors where the kernels run; a group's ONE copy is issued
            (on pool.stream) with its first slab"""
            while grp["next"] < n_slabs_all and grp["next"] <= k + G:    # this slab's group is complete, the next one grows by a slab
                prepare(grp["next"])
                grp["next"] += 1
            g = k // G
            if g != grp["g"]:
                b, roff = g % 2, grp["roff"][g]
                grp["g"] = g
                grp["roff"].pop(g - 1, None)
                if not on_gpu:
                    grp["dev"] = grp["host"][b][: roff[-1]].clone()
                else:
                    with torch.cuda.stream(pool.stream):
                        grp["dev"] = grp["host"][b][: roff[-1]].to(dev, non_blocking=True)
                        grp["ev"][b] = torch.cuda.Event()
                        grp["ev"][b].record(pool.stream)
            q, roff = k - g * G, grp["roff"][g]
            return grp["dev"][q * 2 * B:(q + 1) * 2 * B], grp["dev"][roff[q]:roff[q + 1]]

        cols_d = torch.arange(S, dtype=torch.int64, device=dev)[None, :]

        def scatter_index(sn):
            """[B S] output positions of a slab: slot b's chunk c goes to first[b] + c while c < chunks[b], to the sink otherwise"""
            first, count = sn[:B, None], sn[B:, None]
            return torch.where(cols_d < count, first + cols_d, total).reshape(-1)

        def stage(k):
            e = plan.slab_arrays[k]                                        # [entries, 5], vectorised bookkeeping
            sl, rec, at, take = e[:, 0], e[:, 1], e[:, 2], e[:, 3]
            rows = np.zeros(B, dtype=np.uint64)
            lens = np.zeros(B, dtype=np.int64)
            rows[sl] = ptr0[rec] + (at * esz).astype(np.uint64)
            lens[sl] = take
            rows_p = rows.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p))
            lens_p = lens.ctypes.data_as(ctypes.POINTER(ctypes.c_long))
            nbytes = B * width * esz
            t0 = time.perf_counter()
            if on_gpu:
                i = pool.get(k, 0 if direct else nbytes, nbytes)
                host = None if direct else pool.host[i][:nbytes].view(dtype).view(B, width)
            else:
                i, host = 0, torch.empty((B, width), dtype=dtype)
            if host is not None:
                rc = lib().vad_stage_rows(rows_p, lens_p, B, width, esz, host.data_ptr(), 0)
                if rc:
                    raise _lib.VadError(rc, "vad_stage_rows")
                STATS["stage_s"] += time.perf_counter() - t0
            STATS["buckets"] += 1
            if not on_gpu:
                idx_k, rs_k = slab_meta(k)
                return host, None, i, idx_k, rs_k
            if wf is None:
                STATS["h2d_bytes"] += nbytes
            else:
                issue_windows(k)
                for v in wf["by_first"].pop(k, ()):                         # the windows this slab is the first to read
                    pool.stream.wait_event(wf["ev"].pop(v))
            d = pool.dev[i][:nbytes].view(dtype).view(B, width)
            idx_k, rs_k = slab_meta(k)                                      # (the group's one copy goes out with its first slab)
            if pool.consumed[i] is not None:
                # (a DMA -- the staged copy, the per-row copies -- is issued behind a COMPLETE event, waited for on the host: behind an
                #  open device-side wait it leaves the copy engines' fast path, as in the bucket routes' stage(); the upload kernels wait
                #  on the device)
                if wf is None and (not direct or how == 0) and os.environ.get("SILERO_VAD_AMD_STAGE_SYNC", "1") != "0":
                    pool.consumed[i].synchronize()
                else:
                    pool.stream.wait_event(pool.consumed[i])
            with torch.cuda.stream(pool.stream):
                if wf is not None:
                    eng.upload_rows(rows_p, lens_p, B, width, esz, d, 2)      # (2: the rows are device addresses)
                    STATS["upload_call_s"] += time.perf_counter() - t0
                    for v in wf["by_last"].pop(k, ()):                         # the windows this slab is the last to read: released
                        rel = torch.cuda.Event()
                        rel.record(pool.stream)
                        wf["release"][v] = rel
                elif direct:
                    eng.upload_rows(rows_p, lens_p, B, width, esz, d, how)
                    STATS["upload_call_s"] += time.perf_counter() - t0
                else:
                    d.copy_(host, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(pool.stream)
            pool.done[i] = ev
            return d, ev, i, idx_k, rs_k

        n_slabs = len(plan.slab_arrays)
        STATS["setup_s"] += time.perf_counter() - t_setup
        if prepare_only:
            if on_gpu:
                for k in range(pool.slots):
                    pool.get(k, 0 if direct else B * width * esz, B * width * esz)
                torch.cuda.synchronize(dev)
            return None, base, plan
        staged = stage(0) if n_slabs else None
        for k in range(n_slabs):
            x, ev, slot, idx, rs = staged
            if on_gpu:
                cur.wait_event(ev)
                for t in (x, idx, rs):
                    t.record_stream(cur)
            if rs.numel():                                                # re-admitted slots start like reset_states()
                ctx.index_fill_(0, rs, 0.0)
                state.index_fill_(1, rs, 0.0)
            probs = eng.forward_audio(x, sampling_rate, ctx, state)     # [B, S], carried ctx/state updated in place
            out_flat.index_put_((scatter_index(idx),), probs.reshape(-1))
            if on_gpu:
                pool.consumed[slot] = torch.cuda.Event()
                pool.consumed[slot].record(cur)
            t_a = time.perf_counter()
            if on_slab is not None:
                e = plan.slab_arrays[k]
                on_slab(k, e[e[:, 2] + e[:, 3] >= lens_np[e[:, 1]], 1], out_flat, base)
            t_b = time.perf_counter()
            staged = stage(k + 1) if k + 1 < n_slabs else None          # CPU packs k+1 meanwhile
            if TRACE is not None:
                TRACE.append((k, t_a, t_b, time.perf_counter()))
            yield k
    return out_flat, base, plan


def _refill_run(audios, model, sampling_rate, slots, slab_chunks, plan=None):
    it = _refill_iter(audios, model, sampling_rate, slots, slab_chunks, plan)
    while True:
        try:
            next(it)
        except StopIteration as stop:
            return stop.value


def refill_reserve(audios: Sequence, model, sampling_rate: int = 16000, slots: int = 1024, slab_chunks: int = 32):
    """Everything a refill run over these recordings allocates, allocated now (`ragged_reserve`'s twin): the arena windows' device
    buffers, the staging slots, the engine's scratch, the block the flat probability tensor will take.  Returns the plan."""
    it = _refill_iter(audios, model, sampling_rate, slots, slab_chunks, prepare_only=True)
    while True:
        try:
            next(it)
        except StopIteration as stop:
            return stop.value[2]


def refill_probs(audios: Sequence, model, sampling_rate: int = 16000, slots: int = 1024, slab_chunks: int = 32,
                 plan: RefillPlan = None, _keep_on_device: bool = False):
    """Speech probabilities of many recordings of different lengths through `slots` persistent stream slots
    (RefillPlan).  Returns one 1-D CPU float tensor per recording, bit-identical to
    ``model.audio_forward(audio[None], sr)[0]``: the state and context a slot carries from slab to slab are exactly
    what a single call carries from chunk to chunk, and a re-admitted slot starts from zeros like `reset_states()`.
    Staging of slab k+1 (native threaded copy into pinned memory + H2D on a side stream) overlaps the kernels of
    slab k.  `audios`: float tensors in [-1, 1] or int16 PCM (all of one kind); `sampling_rate` may be a multiple of 16000 (raw
    recordings, _rates: a slab is slab_chunks x 512 k raw samples, the carried context is the 16 kHz net's)."""
    out_flat, base, plan = _refill_run(audios, model, sampling_rate, slots, slab_chunks, plan)
    if _keep_on_device:
        return out_flat, base, plan
    flat = out_flat.cpu()
    return [flat[base[i]:base[i + 1]].clone() for i in range(len(audios))]


def refill_segments_stream(audios: Sequence, model, sampling_rate: int = 16000, slots: int = 1024, slab_chunks: int = 32, **scan_kw):
    """The continuous-refill scheduler with RESULTS AS RECORDINGS RETIRE -- what the reference's worker pool does, each file's
    timestamps handed back when that file is done (examples/parallel_example.ipynb cell 7).  Generator of
    (recording indices int64[m], counts int64[m], segments int64[m, cap, 2]) batches: behind the slab in which a recording's last
    chunk was computed, ONE device scan (vad_segment_probs_device, a lane per recording, over the recording's own row of the flat
    probability tensor) turns the retired recordings into segment lists, which ride back over PCIe asynchronously; a batch is
    yielded as soon as its copy has landed (a slab or two behind the kernels; nothing here blocks the pipeline).  Probabilities
    never leave the GPU.  Segments are in samples of the net's rate (x[::k] for raw 32 / 48 kHz recordings)."""
    net_sr, dec, _ = _rates(sampling_rate)
    lengths = (np.asarray(_describe(audios)[1], dtype=np.int64).reshape(-1) + dec - 1) // dec     # samples at the net's rate: the scan's unit
    eng = model.engine
    on_gpu = hasattr(eng, "_h")
    if not on_gpu:                                                        # CPU stand-in engines (tests): scan at the end, on the host
        flat, base, _ = _refill_run(audios, model, sampling_rate, slots, slab_chunks)
        from .timestamps import segment_probs
        for i in range(len(lengths)):
            sg = segment_probs(flat[base[i]:base[i + 1]], int(lengths[i]), net_sr, **scan_kw) if lengths[i] > 0 else []
            arr = np.asarray([[d["start"], d["end"]] for d in sg], dtype=np.int64).reshape(1, -1, 2)
            yield np.asarray([i], dtype=np.int64), np.asarray([len(sg)], dtype=np.int64), arr
        return
    params = _segment_params(net_sr, **scan_kw)
    cap0 = 32
    pending = collections.deque()                                         # (indices, counts pinned, segs pinned, event, device handles)
    ready = []

    def collect(block):
        while pending and (block or pending[0][3].query()):
            idx, bufs, m, ev, flat_ref, meta_d = pending.popleft()
            ev.synchronize()
            cnt = bufs[1][:m].numpy().copy()
            segs = bufs[2][:m].numpy().copy()
            side["free"].append(bufs)
            if len(cnt) and int(cnt.max()) > cap0:                          # rare: a recording with more segments than were copied back
                with torch.cuda.stream(side["stream"]):
                    c2, s2 = _device_scan(eng, flat_ref[None], meta_d[0], meta_d[1], params, int(cnt.max()), row_offsets=meta_d[2])
                    cnt, segs = c2.cpu().numpy(), s2.cpu().numpy()
            STATS["d2h_bytes"] += segs.nbytes + cnt.nbytes
            ready.append((idx, cnt.copy(), segs))

    side = {"free": []}                                                  # page-locked result buffers, recycled (a pinned allocation per
                                                                          # slab costs milliseconds and can wait for the device)

    def pinned_set(m):
        """(meta int64[3, m], counts int64[m], segs int64[m, cap0, 2]) in page-locked memory, from the free list when one is big enough"""
        for i, bufs in enumerate(side["free"]):
            if bufs[0].shape[1] >= m:
                side["free"].pop(i)
                break
        else:
            cap = max(64, 1 << int(np.ceil(np.log2(max(m, 1)))))
            bufs = (torch.empty((3, cap), dtype=torch.int64, pin_memory=True), torch.empty((cap,), dtype=torch.int64, pin_memory=True),
                    torch.empty((cap, cap0, 2), dtype=torch.int64, pin_memory=True))
        return bufs

    import os as _os
    G = max(1, int(_os.environ.get("SILERO_VAD_AMD_REFILL_GROUP", "8")))
    acc = {"lists": [], "since": 0, "first": True, "args": None}

    def on_slab(k, finished, out_flat, base):
        # the recordings that retire are scanned G slabs at a time (the very first ones at once: results start to flow with the first
        # retirement): a scan is three small copies, and small copies cost the large transfers beside them (see _refill_iter)
        acc["args"] = (out_flat, base)
        if len(finished):
            acc["lists"].append(finished)
        acc["since"] += 1
        if acc["lists"] and (acc["first"] or acc["since"] >= G):
            flush(k)
        collect(False)

    def flush(k):
        out_flat, base = acc["args"]
        finished = np.concatenate(acc["lists"]) if len(acc["lists"]) > 1 else acc["lists"][0]
        acc["lists"], acc["since"], acc["first"] = [], 0, False
        if len(finished):
            # The scan of the recordings that retired in this slab runs on a SIDE stream, behind the slab's kernels by event: one lane per
            # recording walks ~1 000 probabilities one after the other (2-5 ms for the few dozen recordings of a slab) -- on the
            # compute stream that would stand between this slab's kernels and the next slab's (measured: the leg became compute-bound,
            # 0.81 of the link); beside them it is one wave.
            t0 = time.perf_counter()
            dev = out_flat.device
            cur = torch.cuda.current_stream(dev)
            if "stream" not in side:
                # (on a hardware queue of its own: a scan that lands on the upload's or the compute stream's queue serialises with it
                #  -- measured as 3.4 ms holes in the upload stream every time a wave of recordings retires, 0.79 instead of 0.90 of
                #  the link, depending on which streams the process happened to create before)
                pool = getattr(model, "_stage_pool", None)
                others = [cur] + ([pool.stream] if pool is not None else [])
                if getattr(pool, "copy_stream", None) is not None:            # (the window feed's DMA queue)
                    others.append(pool.copy_stream)
                side["stream"] = _distinct_queue_stream(eng, dev, others)
            done_k = torch.cuda.Event()
            done_k.record(cur)
            side["stream"].wait_event(done_k)
            m = len(finished)
            bufs = pinned_set(m)
            bufs[0][0, :m] = torch.from_numpy(base[finished + 1] - base[finished])
            bufs[0][1, :m] = torch.from_numpy(lengths[finished])
            bufs[0][2, :m] = torch.from_numpy(base[finished])
            with torch.cuda.stream(side["stream"]):
                meta = bufs[0][:, :m].to(dev, non_blocking=True)
                counts, segs = _device_scan(eng, out_flat[None], meta[0], meta[1], params, cap0, row_offsets=meta[2])
                bufs[1][:m].copy_(counts, non_blocking=True)
                bufs[2][:m].copy_(segs, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side["stream"])
            pending.append((finished.copy(), bufs, m, ev, out_flat, meta))
            STATS["scan_s"] += time.perf_counter() - t0

    for _ in _refill_iter(audios, model, sampling_rate, slots, slab_chunks, None, on_slab):
        while ready:
            yield ready.pop(0)
    if acc["lists"]:
        flush(-1)
    collect(True)
    while ready:
        yield ready.pop(0)
    if "stream" in side:                                                  # later work on the caller's stream is ordered behind the scans
        torch.cuda.current_stream().wait_stream(side["stream"])
    empty = np.flatnonzero(lengths <= 0)
    if len(empty):
        yield empty.astype(np.int64), np.zeros(len(empty), dtype=np.int64), np.zeros((len(empty), 1, 2), dtype=np.int64)


def refill_speech_segments(audios: Sequence, model, sampling_rate: int = 16000, slots: int = 1024,
                           slab_chunks: int = 32, as_arrays: bool = False, **scan_kw) -> List[list]:
    """`ragged_speech_segments` over the continuous-refill scheduler (refill_segments_stream: every recording is scanned on the GPU
    behind the slab it retires in, its segment list crosses PCIe while later slabs run).  as_arrays: (counts int64[n], segments
    int64[sum(counts), 2]) like ragged_speech_segments, instead of a list of lists of dicts."""
    n_rec = len(audios)
    counts_all = np.zeros(n_rec, dtype=np.int64)
    parts = []
    for idx, cnt, segs in refill_segments_stream(audios, model, sampling_rate, slots, slab_chunks, **scan_kw):
        counts_all[idx] = cnt
        parts.append((idx, cnt, segs))
    t0 = time.perf_counter()
    first = np.concatenate([[0], np.cumsum(counts_all)])
    flat = np.zeros((int(first[-1]), 2), dtype=np.int64)
    for idx, cnt, sg in parts:
        if not len(idx) or not cnt.any():
            continue
        k = np.arange(sg.shape[1])[None, :]
        mask = k < cnt[:, None]
        flat[(first[idx][:, None] + k)[mask]] = sg[mask]
    STATS["scan_s"] += time.perf_counter() - t0
    if as_arrays:
        return counts_all, flat
    fl = flat.tolist()
    return [[{"start": a, "end": b} for a, b in fl[first[i]:first[i + 1]]] for i in range(n_rec)]


# ---- live streams --------------------------------------------------------------------------------------
# Live streams: StreamPool drives the engine directly (no host synchronisation per tick), fp32 like everything else.
class StreamPool:
    """`capacity` concurrently live streams on one GPU, one `tick` per 32 ms chunk.

    tick(chunks[capacity, N]) -> probs[capacity] (a CUDA tensor that is overwritten by the next
    tick).  Rows of closed slots are computed too (lock-step batch) and ignored.  With
    `graph=True` the step is captured once into a hipGraph and replayed; the input is then copied
    into a fixed staging buffer first.

    `tick(chunks, present=flags)`: live streams do not arrive in lock step.  A row whose flag is 0 has NO chunk this tick and is
    not stepped: its (h, c) and context stay bit for bit what they were and its probability slot holds -1.0 (VAD_PROB_ABSENT) --
    what the reference's per-stream callers do by simply not calling the model (src/silero_vad/utils_vad.py:507-549, one
    `__call__` per chunk that arrived).  The other rows' results do not depend on the flags.  Without flags the tick is the
    same launch it always was (a second hipGraph holds the flagged form; it is captured on first use).

    `dtype=torch.int16`: the chunks are 16-bit PCM, scaled by 1/32768 inside the kernel's loads -- what every
    non-Python client of the reference feeds (examples/onnx_sequence/run.py:115-119, examples/cpp/wav.h:113-118) and
    half the PCIe bytes of fp32.

    `host_slots=R` (R >= 1) makes the pool a HOST-TO-HOST server, the shape of the reference's streaming caller (host chunk
    in, probability out: src/silero_vad/utils_vad.py:507-549): a page-locked ingest ring `host_pcm[R, capacity, N]` that the
    audio sources write into directly (no staging copy on the host) and `host_prob[R, capacity]`; slot r has ONE hipGraph =
    H2D of host_pcm[r] -> the fused step -> D2H of the probabilities into host_prob[r], replayed on the pool's own stream.
    `submit(r)` starts a tick and returns, `wait(r)` blocks until its probabilities are readable on the host;
    `tick_host(r)` = both.  Ticks of one pool execute in submission order (the carried state demands it); with R >= 2 the
    next tick may be submitted while the host still reads the previous one, and several pools (each on a clone of the
    engine) overlap one pool's H2D with another's kernel."""

    def __init__(self, engine, sampling_rate: int = 16000, capacity: int = 8192, graph: bool = True,
                 dtype: torch.dtype = torch.float32, host_slots: int = 0):
        if dtype not in (torch.float32, torch.int16):
            raise TypeError(f"StreamPool dtype must be float32 or int16, got {dtype}")
        self.engine = engine
        self.sr = int(sampling_rate)
        self.n = chunk_size(self.sr)
        self.capacity = int(capacity)
        self.dtype = dtype
        self.device = torch.device("cuda", engine.device)
        with torch.cuda.device(self.device):
            self.ctx = torch.zeros((self.capacity, self.n // 8), device=self.device)
            self.state = torch.zeros((2, self.capacity, 128), device=self.device)
            self.pcm = torch.zeros((self.capacity, self.n), dtype=dtype, device=self.device)
            self.prob = torch.zeros((self.capacity,), device=self.device)
            self.host_slots = int(host_slots)
            self.host_pcm = self.host_prob = self.stream = None
            if self.host_slots:
                self.host_pcm = torch.zeros((self.host_slots, self.capacity, self.n), dtype=dtype, pin_memory=True)
                self.host_prob = torch.zeros((self.host_slots, self.capacity), dtype=torch.float32, pin_memory=True)
                self.stream = torch.cuda.Stream(self.device)
                self._done = [torch.cuda.Event() for _ in range(self.host_slots)]
                self.host_present = torch.ones((self.host_slots, self.capacity), dtype=torch.uint8, pin_memory=True)
            self.present = torch.ones((self.capacity,), dtype=torch.uint8, device=self.device)     # flags of the tick (device staging)
        self._graph_present = None
        self._host_graphs_present = None
        self.open_mask = np.zeros(self.capacity, dtype=bool)
        self._free = list(range(self.capacity - 1, -1, -1))
        self._graph = None
        self._host_graphs = None
        self._graph_gen = None
        self._use_graph = bool(graph)
        engine.reserve(self.sr, self.capacity, 1)
        if graph:
            self._capture()

    def _launch(self, masked=False):
        if masked:  # rows with self.present[b] == 0 are not stepped (vad_step_present)
            self.engine.step_present(self.pcm, self.sr, self.ctx, self.state, self.prob, self.present)
        elif self.dtype == torch.float32:
            self.engine.step(self.pcm, self.sr, self.ctx, self.state, self.prob)
        else:       # one chunk of int16 PCM per stream: a ONE-step vad_forward_audio_i16 call (the same fused kernel)
            self.engine.forward_audio(self.pcm, self.sr, self.ctx, self.state, self.prob.view(self.capacity, 1))

    def _host_tick(self, r, stream=None, masked=False):
        """What slot r's graph holds: ingest ring -> HBM, the step, probabilities -> host (one native call: vad_step_host)."""
        # (the kernel stores the probabilities straight into the page-locked host_prob[r]: no device buffer, no D2H operation)
        if masked:
            self.engine.step_host(self.host_pcm[r], self.pcm, self.sr, self.ctx, self.state, None, self.host_prob[r], stream,
                                  host_present=self.host_present[r], dev_present=self.present)
        else:
            self.engine.step_host(self.host_pcm[r], self.pcm, self.sr, self.ctx, self.state, None, self.host_prob[r], stream)

    def _capture(self):
        """Capture one step into a hipGraph.  The graph bakes in the engine's scratch addresses, so it is tied to
        the engine's scratch generation (include/silero_vad_hip.h, vad_scratch_generation) and is captured again,
        with the carried state preserved, when another user of the same engine made the scratch grow."""
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            keep_ctx, keep_state = self.ctx.clone(), self.state.clone()
            self.engine.reserve(self.sr, self.capacity, 1)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self._launch()                             # warm-up outside capture (lazy module load)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._launch()
            self._graph = g
            if self._graph_present is not None:            # (only pools that have ticked with flags hold the flagged graphs)
                with torch.cuda.stream(side):
                    self._launch(masked=True)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch(masked=True)
                self._graph_present = g
            if self.host_slots:
                self._host_graphs = []
                for r in range(self.host_slots):
                    hg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(hg, stream=self.stream):
                        self._host_tick(r)
                    self._host_graphs.append(hg)
                if self._host_graphs_present is not None:
                    self._host_graphs_present = []
                    for r in range(self.host_slots):
                        hg = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(hg, stream=self.stream):
                            self._host_tick(r, masked=True)
                        self._host_graphs_present.append(hg)
            torch.cuda.synchronize(self.device)
            self.ctx.copy_(keep_ctx)                       # the warm-up step advanced them
            self.state.copy_(keep_state)
            torch.cuda.synchronize(self.device)            # (a host-mode replay on self.stream may follow at once)
            self._graph_gen = self.engine.scratch_generation()

    # -- slots ---------------------------------------------------------------------------------------
    def open(self) -> int:
        """Admit a stream: returns its slot; the slot starts from zero state/context."""
        if not self._free:
            raise RuntimeError("StreamPool is full")
        s = self._free.pop()
        with self._state_stream():
            self.ctx[s].zero_()
            self.state[:, s].zero_()
        self.open_mask[s] = True
        return s

    def open_all(self):
        """Admit `capacity` streams at once (every slot from zero state)."""
        with self._state_stream():
            self.ctx.zero_()
            self.state.zero_()
        self.open_mask[:] = True
        self._free = []

    def close(self, slot: int):
        if not self.open_mask[slot]:
            raise ValueError(f"slot {slot} is not open")
        self.open_mask[slot] = False
        self._free.append(int(slot))

    def reset(self, slot: int):
        with self._state_stream():
            self.ctx[slot].zero_()
            self.state[:, slot].zero_()

    def _state_stream(self):
        """Where writes to the carried state are issued.  A host-mode pool (host_slots > 0) runs its ticks on its own
        non-blocking stream: the zeroing of a slot must be ordered on THAT stream, or a tick submitted right behind
        open()/reset() could read the state before the zeroing has landed.  Device-mode pools tick on torch's current
        stream, where the writes already are.  (tick() and submit() must not be mixed on one pool.)"""
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    # -- the tick --------------------------------------------------------------------------------------
    def tick(self, chunks: torch.Tensor, present=None) -> torch.Tensor:
        if chunks.shape != (self.capacity, self.n):
            raise ValueError(f"expected chunks of shape {(self.capacity, self.n)}, got {tuple(chunks.shape)}")
        if (chunks.dtype == torch.int16) != (self.dtype == torch.int16):
            raise TypeError(f"this pool takes {self.dtype} chunks, got {chunks.dtype}")
        with torch.cuda.device(self.device):
            self.pcm.copy_(chunks, non_blocking=True)
            if present is not None:
                self.present.copy_(self._flags(present), non_blocking=True)
            self.tick_staged(masked=present is not None)
        return self.prob

    def _flags(self, present) -> torch.Tensor:
        f = torch.as_tensor(present)
        if f.shape != (self.capacity,):
            raise ValueError(f"expected {self.capacity} present flags, got shape {tuple(f.shape)}")
        return (f != 0).to(torch.uint8)

    def tick_staged(self, masked: bool = False):
        """One step over whatever `self.pcm` holds (for callers that write the staging buffer
        themselves, e.g. a device-side audio source); masked: only the rows flagged in `self.present`."""
        if self._graph is not None:
            if masked and self._graph_present is None:
                self._graph_present = False                # (asks _capture for the flagged graphs too)
                self._capture()
            elif self.engine.scratch_generation() != self._graph_gen:
                self._capture()                            # the engine's scratch moved: the old graph is stale
            (self._graph_present if masked else self._graph).replay()
        else:
            self._launch(masked)
        return self.prob

    # -- host to host ------------------------------------------------------------------------------------
    def submit(self, r: int = 0, present=None):
        """Start one tick over `host_pcm[r]` on the pool's stream; returns at once (see the class docstring).  present: flags of
        the streams that have a chunk in the slot (None = all; or True = `host_present[r]` as the sources left it)."""
        if not self.host_slots:
            raise RuntimeError("StreamPool was created without host_slots")
        masked = present is not None
        if masked and present is not True:
            self.host_present[r].copy_(self._flags(present))
        if self._use_graph:
            if masked and self._host_graphs_present is None:
                self._host_graphs_present = []
                self._capture()
            elif self.engine.scratch_generation() != self._graph_gen:
                self._capture()
            with torch.cuda.stream(self.stream):             # (replay launches on torch's current stream)
                (self._host_graphs_present if masked else self._host_graphs)[r].replay()
        else:       # one native call on the pool's stream: no stream switch on the host side
            self._host_tick(r, self.stream.cuda_stream, masked)
        self._done[r].record(self.stream)

    def wait(self, r: int = 0) -> torch.Tensor:
        """Block until the tick submitted for slot r has delivered; returns `host_prob[r]` (page-locked, overwritten by the
        slot's next tick)."""
        self._done[r].synchronize()
        return self.host_prob[r]

    def tick_host(self, r: int = 0) -> torch.Tensor:
        self.submit(r)
        return self.wait(r)


class BatchVADIterator:
    """`VADIterator` (src/silero_vad/utils_vad.py:458-549) for every slot of a lock-step batch.

    feed(probs[B]) advances all streams by one chunk and returns the list of events of this tick as
    (slot, {'start': n}) / (slot, {'end': n}) -- the same numbers the reference iterator emits for
    that stream (`current_sample` counts chunk ENDS, exit threshold fixed at threshold - 0.15,
    positions shifted back by one window, :526-547)."""

    def __init__(self, batch: int, threshold: float = 0.5, sampling_rate: int = 16000,
                 min_silence_duration_ms: int = 100, speech_pad_ms: int = 30):
        if sampling_rate not in (8000, 16000):
            raise ValueError("VADIterator does not support sampling rates other than [8000, 16000]")
        self.threshold = threshold
        self.sampling_rate = sampling_rate
        self.window = chunk_size(sampling_rate)
        self.min_silence_samples = sampling_rate * min_silence_duration_ms / 1000
        self.speech_pad_samples = sampling_rate * speech_pad_ms / 1000
        self.triggered = np.zeros(batch, dtype=bool)
        self.temp_end = np.zeros(batch, dtype=np.int64)
        self.current_sample = np.zeros(batch, dtype=np.int64)
        self._events = None

    def reset(self, slot=None):
        sl = slice(None) if slot is None else slot
        self.triggered[sl] = False
        self.temp_end[sl] = 0
        self.current_sample[sl] = 0

    def feed(self, probs, active=None):
        """One tick: probs[B] (host or device tensor / array) -> [(slot, {'start': n} | {'end': n}), ...] in slot order.  The
        per-stream logic runs in the native library (vad_iterator_feed, csrc/segmenter.cpp): one pass over the batch, no
        per-stream Python."""
        if torch.is_tensor(probs):
            probs = probs.detach().cpu().numpy()
        p = np.ascontiguousarray(probs, dtype=np.float32).reshape(-1)
        B = self.triggered.shape[0]
        if p.shape[0] != B:
            raise ValueError(f"expected {B} probabilities, got {p.shape[0]}")
        act = None if active is None else np.ascontiguousarray(active, dtype=np.bool_)
        if act is not None and act.shape[0] != B:
            raise ValueError(f"expected {B} active flags, got {act.shape[0]}")
        if self._events is None or len(self._events) < B:
            self._events = (_lib.IterEvent * B)()
        m = lib().vad_iterator_feed(p.ctypes.data, None if act is None else act.ctypes.data, B, self.window,
                                    float(self.threshold), float(self.min_silence_samples), float(self.speech_pad_samples),
                                    self.triggered.ctypes.data, self.temp_end.ctypes.data, self.current_sample.ctypes.data,
                                    self._events, B)
        if m < 0:
            raise ValueError("vad_iterator_feed: bad arguments")
        ev = self._events
        return [(ev[i].slot, {"end" if ev[i].kind else "start": ev[i].sample}) for i in range(m)]


class StreamPump:
    """BASELINE configs[4] through the native pump (include/silero_vad_hip.h "live streams: the pump", csrc/pump.hip): `streams`
    live streams on one GPU, host int16 chunks in, VADIterator events out, with no Python and no torch on the per-tick path
    (the reference's native streaming loop, examples/cpp/silero-vad-onnx.cpp:335-390, for thousands of streams in lock step).

        pump = StreamPump(engine, 16000, streams=8192)
        pump.slot(r)[b] = next int16 chunk of stream b          # numpy view of the page-locked ingest ring, [streams, N]
        pump.submit(r)                                           # asynchronous: H2D -> step kernels -> probabilities on the host
        events, r = pump.poll()                                  # retire the oldest tick: [(stream, {'start' | 'end': sample}), ...]
        pump.probs(r)                                            # its probabilities, [streams] float32 (page-locked)

    `play(rows, ...)` runs the whole loop natively over memory-resident recordings (tests, benchmarks, file-fed servers)."""

    def __init__(self, engine, sampling_rate: int = 16000, streams: int = 8192, parts: int = 0, ring_slots: int = 0,
                 threshold: float = 0.5, min_silence_duration_ms: int = 100, speech_pad_ms: int = 30):
        if sampling_rate not in (8000, 16000):
            raise ValueError("VADIterator does not support sampling rates other than [8000, 16000]")
        self._L = engine._L
        prm = _lib.PumpParams()
        self._L.vad_pump_params_default(ctypes.byref(prm), int(sampling_rate), int(streams))
        prm.parts, prm.ring_slots = int(parts), int(ring_slots)
        prm.threshold, prm.min_silence_duration_ms, prm.speech_pad_ms = float(threshold), int(min_silence_duration_ms), int(speech_pad_ms)
        h = ctypes.c_void_p()
        rc = self._L.vad_pump_create(engine._h, ctypes.byref(prm), ctypes.byref(h))
        if rc:
            raise _lib.VadError(rc, "vad_pump_create")
        self._h = h
        g = [ctypes.c_int() for _ in range(4)]
        self._L.vad_pump_geometry(h, *[ctypes.byref(x) for x in g])
        self.streams, self.n, self.ring_slots, self.parts = (x.value for x in g)
        self.sr = int(sampling_rate)
        self._events = (_lib.IterEvent * self.streams)()
        self._slots = [np.ctypeslib.as_array(ctypes.cast(self._L.vad_pump_slot(h, r), ctypes.POINTER(ctypes.c_int16)),
                                             shape=(self.streams, self.n)) for r in range(self.ring_slots)]
        self._probs = [np.ctypeslib.as_array(ctypes.cast(self._L.vad_pump_probs(h, r), ctypes.POINTER(ctypes.c_float)),
                                             shape=(self.streams,)) for r in range(self.ring_slots)]
        self._present = [np.ctypeslib.as_array(ctypes.cast(self._L.vad_pump_present(h, r), ctypes.POINTER(ctypes.c_uint8)),
                                               shape=(self.streams,)) for r in range(self.ring_slots)]

    def close(self):
        if getattr(self, "_h", None):
            self._slots = self._probs = self._present = None
            self._L.vad_pump_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc:
            raise _lib.VadError(rc, self._L.vad_pump_last_error(self._h).decode())

    def slot(self, r: int) -> np.ndarray:
        return self._slots[r]

    def probs(self, r: int) -> np.ndarray:
        return self._probs[r]

    def present(self, r: int) -> np.ndarray:
        """Slot r's flag row, [streams] uint8 (page-locked, in front of the slot's audio): 0 = the stream has no chunk this tick."""
        return self._present[r]

    def submit(self, r: int, present=None, compact: bool = False):
        """Start the tick over ring slot r.  present: None = every stream has a chunk; True = the flags the sources wrote into
        `present(r)`; an array of `streams` flags = copied there.  A stream whose flag is 0 is not stepped: its (h, c), context and
        iterator counters stay as they are and `probs(r)[b]` reads -1.0 (vad_pump_submit_present).
        compact=True (vad_pump_submit_compact): `slot(r)` holds only the delivering streams' chunks, back to back -- row i is the chunk
        of the i-th stream whose flag is set -- and only those rows cross the link; same results, bit for bit."""
        if compact and present is None:
            raise ValueError("a compact tick needs its present flags")
        if present is None:
            self._check(self._L.vad_pump_submit(self._h, int(r)))
            return
        if present is not True:
            f = np.asarray(present)
            if f.shape != (self.streams,):
                raise ValueError(f"expected {self.streams} present flags, got shape {f.shape}")
            self._present[r][:] = f != 0
        fn = self._L.vad_pump_submit_compact if compact else self._L.vad_pump_submit_present
        self._check(fn(self._h, int(r), self._present[r].ctypes.data))

    def submit_rows(self, r: int, stream_of_row):
        """A compact tick in ARRIVAL order (vad_pump_submit_rows): row i of `slot(r)` holds the chunk of stream `stream_of_row[i]`; every
        other stream is absent this tick.  A stream listed twice, or out of range, raises and queues nothing."""
        rows = np.ascontiguousarray(stream_of_row, dtype=np.int32).reshape(-1)
        self._check(self._L.vad_pump_submit_rows(self._h, int(r), rows.ctypes.data if len(rows) else None, len(rows)))

    def poll(self, block: bool = True):
        """-> (events, ring slot) of the oldest submitted tick; (None, None) if nothing is in flight or (block=False) it has not finished."""
        r = ctypes.c_int(-1)
        m = self._L.vad_pump_poll(self._h, 1 if block else 0, self._events, self.streams, ctypes.byref(r))
        if m in (-1, -2):
            return None, None
        if m < 0:
            raise RuntimeError("vad_pump_poll: " + self._L.vad_pump_last_error(self._h).decode())
        ev = self._events
        return [(ev[i].slot, {"end" if ev[i].kind else "start": ev[i].sample}) for i in range(m)], r.value

    def open_stream(self, stream: int):
        self._check(self._L.vad_pump_open(self._h, int(stream)))

    def close_stream(self, stream: int):
        self._check(self._L.vad_pump_close(self._h, int(stream)))

    def state(self, stream: int):
        """(h[128], c[128], ctx[C]) of one stream (synchronises; tests)."""
        h, c, x = np.empty(128, np.float32), np.empty(128, np.float32), np.empty(self.n // 8, np.float32)
        self._check(self._L.vad_pump_state(self._h, int(stream), h.ctypes.data, c.ctypes.data, x.ctypes.data))
        return h, c, x

    def play(self, rows: np.ndarray, n_ticks: int, first_tick: int = 0, depth: int = 2, fill_threads: int = 0, max_events: int = 0,
             pattern: np.ndarray = None, compact: bool = False):
        """vad_pump_play: stream b plays rows[b] (int16, length a multiple of the chunk) circularly, chunk by chunk.
        pattern [pattern_ticks, streams] uint8 (vad_pump_play_gaps): stream b delivers a chunk at tick t iff pattern[t % pattern_ticks, b];
        its audio advances only when it delivers.  compact=True (vad_pump_play_compact): the sources write compact slots, only the
        delivering streams' rows cross the link.
        -> (events [(stream, {...}), ...] (the first max_events of them), stats dict)."""
        if rows.dtype != np.int16 or rows.ndim != 2 or rows.shape[0] != self.streams or not rows.flags.c_contiguous:
            raise ValueError(f"rows must be C-contiguous int16 [{self.streams}, period]")
        if pattern is not None and (pattern.dtype != np.uint8 or pattern.ndim != 2 or pattern.shape[1] != self.streams or not pattern.flags.c_contiguous):
            raise ValueError(f"pattern must be C-contiguous uint8 [pattern_ticks, {self.streams}]")
        cap = int(max_events)
        buf = (_lib.IterEvent * max(cap, 1))()
        st = _lib.PumpStats()
        fn = self._L.vad_pump_play_compact if compact and pattern is not None else self._L.vad_pump_play_gaps
        m = fn(self._h, rows.ctypes.data, rows.shape[1], rows.shape[1], None if pattern is None else pattern.ctypes.data,
                                       0 if pattern is None else pattern.shape[0], int(first_tick), int(n_ticks), int(depth),
                                       int(fill_threads), buf if cap else None, cap, ctypes.byref(st))
        if m < 0:
            raise RuntimeError("vad_pump_play: " + self._L.vad_pump_last_error(self._h).decode())
        ev = [(buf[i].slot, {"end" if buf[i].kind else "start": buf[i].sample}) for i in range(min(m, cap))]
        return ev, {k: getattr(st, k) for k, _ in _lib.PumpStats._fields_}
