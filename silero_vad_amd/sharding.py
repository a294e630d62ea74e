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
                            rank: int = 0, world_size: int = 1, **kwargs) -> List[list]:
    """`get_speech_timestamps` over many recordings: this rank processes its shard, equal-length
    recordings going through the GPU as one lock-step batch; rank 0 receives every result in input
    order (other ranks get None).  kwargs are those of get_speech_timestamps."""
    from .timestamps import get_speech_timestamps, segment_probs

    mine = shard_range(len(audios), world_size, rank)
    results = {}
    by_len = {}
    for i in mine:
        by_len.setdefault(len(audios[i]), []).append(i)
    fast = getattr(model, "audio_forward_device", None)
    plain = {k: kwargs[k] for k in kwargs if k not in ("return_seconds", "time_resolution",
                                                       "visualize_probs", "progress_tracking_callback",
                                                       "window_size_samples")}
    for length, idxs in by_len.items():
        if fast is None or len(idxs) == 1 or kwargs.get("return_seconds") or sampling_rate > 16000:
            for i in idxs:
                results[i] = get_speech_timestamps(audios[i], model, sampling_rate=sampling_rate, **kwargs)
            continue
        batch = torch.stack([torch.as_tensor(audios[i], dtype=torch.float32) for i in idxs])
        probs = fast(batch, sampling_rate).cpu()
        for row, i in enumerate(idxs):
            results[i] = segment_probs(probs[row], length, sampling_rate, **plain)
    gathered = gather_to_rank0(results)
    if gathered is None:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    return [merged[i] for i in range(len(audios))]
