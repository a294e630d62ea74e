#!/usr/bin/env python3
"""Bring-up: determinism (gx of 6144 tiles, 6 runs), agreement with the fp32 kernels and C2 timing of the
split frontend in the library selected by SILERO_VAD_AMD_LIB."""
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
eng.set_option("precision_front", "fp32")
ref = eng.debug_frontend(x, sr, ctx).clone()
eng.set_option("precision_front", "f16x3")
runs = []
for r in range(6):
    g = eng.debug_frontend(x, sr, ctx)
    torch.cuda.synchronize()
    runs.append(g.clone())
diff = [int((runs[r].view(torch.int32) != runs[0].view(torch.int32)).view(B // 16, 16, T, 512).any(dim=3).any(dim=1).sum())
        for r in range(1, 6)]
err = max(float((g - ref).abs().max()) for g in runs)
out = {"lib": sys.argv[1] if len(sys.argv) > 1 else "", "tiles_differing_vs_run0": diff,
       "max_gx_err_vs_fp32": err, "gx_scale": float(ref.abs().max())}
# timing at C2
del runs, ref, x
Tc = 256
pcm = torch.randn((B, Tc * n), device=dev) * 0.05
ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev)
eng.set_option("precision", "f16x3")
eng.reserve(sr, B, Tc)
for _ in range(2):
    eng.forward_audio(pcm, sr, ctx, st)
eng.set_option("profile", "1")
for _ in range(5):
    eng.forward_audio(pcm, sr, ctx, st)
f, r, c = eng.kernel_times()
out["front_ms"] = round(f / c, 4); out["rec_ms"] = round(r / c, 4)
print(json.dumps(out), flush=True)
