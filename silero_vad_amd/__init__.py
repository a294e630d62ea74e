"""MI355X-native Silero-VAD v6 inference engine: hand-written HIP kernels for gfx950 behind a
C ABI (include/silero_vad_hip.h), with the reference package's Python surface on top.

    from silero_vad_amd import load_silero_vad, get_speech_timestamps, VADIterator
"""
from .engine import Engine, HipSileroVAD, load_silero_vad  # noqa: F401
from .timestamps import (VADIterator, collect_chunks, drop_chunks, get_speech_timestamps,  # noqa: F401
                         read_audio, save_audio, segment_probs)
from .streams import (BatchVADIterator, PackedRecordings, RaggedPlan, RefillPlan, StreamPool, StreamPump, refill_probs, refill_reserve, refill_segments_stream, refill_speech_segments, ragged_buckets, ragged_probs, ragged_reserve,  # noqa: F401
                      ragged_speech_segments, segment_probs_batch, segment_probs_batch_device)
from .sharding import shard_range, shard_by_duration, gather_to_rank0, batch_speech_timestamps  # noqa: F401

__version__ = "0.1.0"
