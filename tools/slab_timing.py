#!/usr/bin/env python3
"""Does keeping a time slab's gx in the 256 MiB Infinity Cache pay?  Wall time of one C2 call (4096 x 256
chunks) for several scratch caps (engine option gx_cap_mib -> slab length)."""
import json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import Engine
dev = torch.device("cuda", 0); eng = Engine(0)
sr, n, B, T = 16000, 512, 4096, 256
x = torch.randn((B, T * n), device=dev) * 0.05
ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev)
for cap in (6144, 1024, 512, 256, 128, 64, 32):
    eng.set_option("gx_cap_mib", cap)
    for _ in range(2):
        eng.forward_audio(x, sr, ctx, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        eng.forward_audio(x, sr, ctx, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    eng.set_option("profile", "1")
    eng.forward_audio(x, sr, ctx, st)
    f, r, c = eng.kernel_times()
    eng.set_option("profile", "0")
    print(json.dumps({"gx_cap_mib": cap, "slab_steps": min(T, cap * 2**20 // (B * 2048)), "ms_per_call": round(dt * 1e3, 3),
                      "front_ms_sum": round(f, 3), "rec_ms_sum": round(r, 3), "Mchunks_s": round(B * T / dt / 1e6, 1)}), flush=True)
