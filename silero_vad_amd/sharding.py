"""Multi-GPU use of the path: streams are independent, so they are sharded across ranks with NO
data-path collective -- one process per GPU, each with its own engine and a
replica of the 1.2 MB weights, the 7 xGMI links idle by design.  This mirrors the reference's only
parallel pattern, process-per-worker over files (examples/parallel_example.ipynb cells 5, 7).  The
only communication is a host-side gather of the (tiny) results to rank 0.
"""
import os
from typing import List, Sequence

import torch


def world():
    """(rank, world_size, local_rank) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def shard_range(n_items: int, world_size: int, rank: int) -> range:
    """Contiguous, balanced block of item indices owned by `rank` (first n % W ranks get one more)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return range(lo, lo + q + (1 if rank < r else 0))


def shard_by_duration(lengths: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Indices owned by `rank` when items are dealt out by total duration instead of by count (SURVEY.md 8e:
    "contiguous blocks balanced by total duration"): longest first, each to the rank with the least work so far
    (ties: lowest rank).  Deterministic, so every rank computes the same partition without communicating; the
    returned indices are in input order."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    import heapq
    load = [(0, r) for r in range(world_size)]
    mine = []
    for i in sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i)):
        total, r = heapq.heappop(load)
        if r == rank:
            mine.append(i)
        heapq.heappush(load, (total + int(lengths[i]), r))
    return sorted(mine)


def gather_to_rank0(obj, group=None):
    """Host-side gather of a picklable per-rank result; returns the list on rank 0, None elsewhere.
    Works without an initialised process group (single process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    out = [None] * ws if rk == 0 else None
    dist.gather_object(obj, out, dst=0, group=group)
    return out


def batch_speech_timestamps(audios: Sequence[torch.Tensor], model, sampling_rate: int = 16000,
                            rank: int = 0, world_size: int = 1, balance: str = "duration",
                            scheduler: str = "buckets", **kwargs) -> List[list]:
    """`get_speech_timestamps` over many recordings (the reference's pattern is one worker process
    per file, examples/parallel_example.ipynb cells 5, 7): this rank processes its shard --
    `balance="duration"` (default) deals recordings out by total audio length, `"count"` by contiguous
    index blocks; recordings of any lengths go through `scheduler="buckets"` (length-sorted lock-step
    batches, `streams.ragged_speech_segments`) or `"refill"` (persistent slots, continuous refill,
    `streams.refill_speech_segments`), both scanned on the GPU -- and rank 0 receives every result in
    input order (other ranks get None).  kwargs are those of get_speech_timestamps."""
    import warnings

    from .streams import ragged_speech_segments, refill_speech_segments
    from .timestamps import get_speech_timestamps

    if balance not in ("duration", "count") or scheduler not in ("buckets", "refill"):
        raise ValueError("balance must be duration|count and scheduler buckets|refill")
    if balance == "duration":
        mine = shard_by_duration([int(a.shape[-1]) if hasattr(a, "shape") else len(a) for a in audios], world_size, rank)
    else:
        mine = list(shard_range(len(audios), world_size, rank))
    results = {}
    scan_kw = {k: kwargs[k] for k in kwargs if k not in ("return_seconds", "time_resolution",
                                                         "visualize_probs", "progress_tracking_callback",
                                                         "window_size_samples")}
    fast = getattr(model, "audio_forward_device", None)
    if fast is None or kwargs.get("visualize_probs") or kwargs.get("progress_tracking_callback"):
        for i in mine:
            results[i] = get_speech_timestamps(audios[i], model, sampling_rate=sampling_rate, **kwargs)
    elif mine:
        step, sr = 1, sampling_rate
        if sr > 16000 and sr % 16000 == 0:                 # utils_vad.py:301-307
            step, sr = sr // 16000, 16000
            warnings.warn('Sampling rate is a multiply of 16000, casting to 16000 manually!')
        local = []
        for i in mine:
            a = audios[i] if torch.is_tensor(audios[i]) else torch.as_tensor(audios[i])
            while a.dim() > 1 and a.shape[0] == 1:
                a = a.squeeze(0)
            local.append(a)
        # The recordings go to the scheduler AS THEY ARE, with the raw rate: every zero-copy route stays open (pinned 48 kHz
        # recordings are read by the DMA / the gather kernel where they lie) and the frontend's loads take every step-th sample
        # (streams._rates) -- the reference's x[::step] (utils_vad.py:301-307) without the copy.  Segments come back in samples of
        # the 16 kHz signal, like the reference's before its final `* step`.
        lens = [(int(a.shape[0]) + step - 1) // step for a in local]
        segs = (ragged_speech_segments if scheduler == "buckets" else refill_speech_segments)(local, model, sampling_rate, **scan_kw)
        seconds, res = kwargs.get("return_seconds", False), kwargs.get("time_resolution", 1)
        for r, i in enumerate(mine):
            out = segs[r]
            if seconds:                                    # utils_vad.py:442-446
                total = lens[r] / sr
                for seg in out:
                    seg["start"] = max(round(seg["start"] / sr, res), 0)
                    seg["end"] = min(round(seg["end"] / sr, res), total)
            elif step > 1:                                 # utils_vad.py:447-450
                for seg in out:
                    seg["start"] *= step
                    seg["end"] *= step
            results[i] = out
    gathered = gather_to_rank0(results)
    if gathered is None:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    return [merged[i] for i in range(len(audios))]
