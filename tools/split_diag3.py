#!/usr/bin/env python3
"""Bring-up: run-to-run agreement of an intermediate stage dumped by a VAD_SPLIT_DUMP build."""
import json, sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import Engine
dev = torch.device("cuda", 0)
eng = Engine(0)
wav = torch.from_numpy(np.load(ROOT / "tests/golden/audio_16k.npz")["pcm"].astype(np.float32) / 32768.0).to(dev)
sr, n, B, T = 16000, 512, 4096, 24
idx = (torch.arange(B, device=dev)[:, None] * 7919 + torch.arange(T * n, device=dev)[None]) % len(wav)
x = wav[idx].contiguous()
ctx = torch.zeros((B, 64), device=dev)
eng.set_option("precision_front", "f16x3")
runs = []
for r in range(5):
    g = eng.debug_frontend(x, sr, ctx)
    torch.cuda.synchronize()
    runs.append(g.view(torch.int32).clone())
# majority reference = elementwise mode approximated by run 0 vs others
bad_tiles = []
for r in range(1, 5):
    diff = (runs[r] != runs[0]).view(B // 16, 16, T, 512).any(dim=3).any(dim=1)   # [st][t]
    bad_tiles.append(int(diff.sum()))
print(json.dumps({"lib": sys.argv[1] if len(sys.argv) > 1 else "", "tiles_differing_vs_run0": bad_tiles, "tiles": B // 16 * T}))
