#!/usr/bin/env python3
"""Kernel time of a B = 1 (and B = 2, 4, 8) step through the one-stream kernel and through the tile kernel (engine hipEvents)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from silero_vad_amd import load_silero_vad
m = load_silero_vad(device=0)
eng = m.engine
for sr, n in ((16000, 512), (8000, 256)):
    for B in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
        x = torch.randn((B, n), device=m.device) * 0.1
        ctx = torch.zeros((B, n // 8), device=m.device); st = torch.zeros((2, B, 128), device=m.device); p = torch.empty((B,), device=m.device)
        res = {}
        for one in ("0", "4096"):
            eng.set_option("step_one", one)
            for _ in range(50): eng.step(x, sr, ctx, st, p)
            torch.cuda.synchronize()
            eng.set_option("profile", "1")
            for _ in range(300):
                eng.step(x, sr, ctx, st, p); torch.cuda.synchronize()
            f, r, c = eng.kernel_times()
            eng.set_option("profile", "0")
            res[one] = round(f / c * 1e3, 2)
        eng.set_option("step_one", "auto")
        print(sr, "B", B, "kernel us: tile", res["0"], "one-stream", res["4096"])
