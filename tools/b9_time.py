#!/usr/bin/env python3
"""Time the bf16 x 9 frontend of the library named by SILERO_VAD_AMD_LIB at the C2 shape (GPU box; no parity check: ablation
variants compute wrong results by construction).  Prints one line per library."""
import os
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import Engine
eng = Engine(0)
dev = torch.device("cuda", 0)
out = []
for sr in ([int(os.environ['VAD_B9_TIME_SR'])] if os.environ.get('VAD_B9_TIME_SR') else [16000, 8000]):
    B, T = 4096, 256
    n = 512 if sr == 16000 else 256
    x = 0.1 * torch.randn((B, T * n), device=dev)
    for mma in (sys.argv[1:] or ["bf16x9"]):
        eng.set_option("front_mma", mma)
        eng.set_option("rec", os.environ.get("VAD_B9_TIME_REC", "fp32"))
        st = torch.zeros((2, B, 128), device=dev)
        ctx = torch.zeros((B, n // 8), device=dev)
        for _ in range(12):
            eng.forward_audio(x, sr, ctx, st)
        torch.cuda.synchronize()
        eng.set_option("profile", "1")
        for _ in range(10):
            eng.forward_audio(x, sr, ctx, st)
        torch.cuda.synchronize()
        f, r, c = eng.kernel_times()
        eng.set_option("profile", "0")
        out.append(f"{sr // 1000}k {mma} front {f / c:.3f} rec {r / c:.3f}")
print(os.path.basename(os.environ.get("SILERO_VAD_AMD_LIB", "product")), " | ".join(out), flush=True)
