#!/usr/bin/env python3
"""Where the host time of the corpus path goes (GPU box): cProfile of ragged_speech_segments / refill_speech_segments."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import load_silero_vad, ragged_speech_segments, refill_speech_segments  # noqa: E402

sr, n, R = 16000, 512, 1024
model = load_silero_vad(device=0)
rng = np.random.default_rng(101)
base = torch.from_numpy((rng.standard_normal(8 << 20) * 3000).astype(np.int16))
lens = rng.integers(20 * sr, 40 * sr, size=R)
offs = rng.integers(0, (8 << 20) - 40 * sr, size=R)
audios = [base[o:o + m] for o, m in zip(offs, lens)]
chunks = int(sum((m + n - 1) // n for m in lens))
for name, fn in (("buckets256M", lambda: ragged_speech_segments(audios, model, sr, max_waste=0.1, max_bytes=256 << 20)),
                 ("buckets1G", lambda: ragged_speech_segments(audios, model, sr, max_waste=0.1, max_bytes=1 << 30)),
                 ("buckets1G_waste25", lambda: ragged_speech_segments(audios, model, sr, max_waste=0.25, max_bytes=1 << 30)),
                 ("refill256x32", lambda: refill_speech_segments(audios, model, sr, slots=256, slab_chunks=32)),
                 ("refill512x32", lambda: refill_speech_segments(audios, model, sr, slots=512, slab_chunks=32)),
                 ("refill512x64", lambda: refill_speech_segments(audios, model, sr, slots=512, slab_chunks=64)),
                 ("refill1024x32", lambda: refill_speech_segments(audios, model, sr, slots=1024, slab_chunks=32))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"== {name}: {dt * 1e3:.1f} ms, {chunks / dt / 1e6:.1f} M chunks/s")
    pr = cProfile.Profile()
    pr.enable()
    fn()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(4)
