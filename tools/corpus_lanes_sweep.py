#!/usr/bin/env python3
"""Bring-up measurement (GPU box): corpus bucket path, int16, for sample sizes and compute-lane counts.
    python tools/corpus_lanes_sweep.py"""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import load_silero_vad
from silero_vad_amd import streams as S

sr, n = 16000, 512
model = load_silero_vad(device=0)
rng = np.random.default_rng(101)
base_len = 8 << 20
tt = np.arange(base_len, dtype=np.float32) / sr
base = (0.03 * rng.standard_normal(base_len).astype(np.float32) + 0.2 * np.sin(2 * np.pi * 170.0 * tt) * (np.sin(2 * np.pi * 0.7 * tt) > 0))
base_i = torch.from_numpy((base * 32767.0).clip(-32768, 32767).astype(np.int16))
for R in (1024, 2048, 4096):
    lens = rng.integers(20 * sr, 40 * sr, size=R)
    offs = rng.integers(0, base_len - 40 * sr, size=R)
    audios = [base_i[o:o + m] for o, m in zip(offs, lens)]
    chunks = int(sum((m + n - 1) // n for m in lens))
    for lanes in (1, 2, 3):
        def step():
            params = S._segment_params(sr)
            lengths = [int(a.shape[0]) for a in audios]
            def meta(idxs):
                l = torch.tensor([lengths[i] for i in idxs], dtype=torch.int64)
                return torch.stack([(l + n - 1) // n, l])
            def post(probs_dev, idxs, both):
                c, s = S._device_scan(model.engine, probs_dev, both[0], both[1], params, 24)
                return [c, s]
            tot = 0
            for idxs, (counts, segs), _ in S.ragged_buckets(audios, model, sr, 0.1, 1 << 30, post=post, meta=meta, lanes=lanes):
                tot += int(counts.sum())
            return tot
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        print(f"R={R} lanes={lanes}: {chunks / dt / 1e6:.1f} M chunks/s, {dt * 1e3:.1f} ms per pass", flush=True)
