#!/usr/bin/env python3
"""Bring-up (GPU box): cProfile of one refill_speech_segments pass.  python tools/refill_profile.py [recordings]"""
import sys, time, cProfile, pstats
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import load_silero_vad, refill_speech_segments
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sr = 16000
model = load_silero_vad(device=0)
rng = np.random.default_rng(101)
base_len = 8 << 20
base_i = torch.from_numpy((rng.standard_normal(base_len) * 1000).astype(np.int16))
lens = rng.integers(20 * sr, 40 * sr, size=R)
offs = rng.integers(0, base_len - 40 * sr, size=R)
audios = [base_i[o:o + m] for o, m in zip(offs, lens)]
run = lambda: refill_speech_segments(audios, model, sr, slots=max(64, R // 2), slab_chunks=64)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); print("pass s", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
