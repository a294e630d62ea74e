#!/usr/bin/env python3
"""Is kernel time data dependent?  Times both frontends on several 4096 x 256-chunk inputs in one process."""
import json, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import Engine
sys.path.insert(0, str(ROOT))
import bench
dev = torch.device("cuda", 0); eng = Engine(0)
sr, n, B, T = 16000, 512, 4096, 256
wav = torch.from_numpy(np.load(ROOT / "tests/golden/audio_16k.npz")["pcm"].astype(np.float32) / 32768.0).to(dev)
idx = (torch.arange(B, device=dev)[:, None] * 7919 + torch.arange(T * n, device=dev)[None]) % len(wav)
data = {"speech": wav[idx].contiguous(), "bench_synth": bench.synth_pcm(B, T * n, sr, dev, 17 + sr),
        "randn0.05": torch.randn((B, T * n), device=dev) * 0.05, "zeros": torch.zeros((B, T * n), device=dev),
        "randn0.5": torch.randn((B, T * n), device=dev) * 0.5}
del idx
eng.reserve(sr, B, T)
for name, x in data.items():
    row = {"data": name}
    for prec in ("f16x3", "fp32"):
        eng.set_precision(prec)
        ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev)
        for _ in range(2):
            eng.forward_audio(x, sr, ctx, st)
        eng.set_option("profile", "1")
        for _ in range(4):
            ctx.zero_(); st.zero_()
            eng.forward_audio(x, sr, ctx, st)
        f, r, c = eng.kernel_times()
        eng.set_option("profile", "0")
        row[prec] = {"front_ms": round(f / c, 3), "rec_ms": round(r / c, 3)}
    print(json.dumps(row), flush=True)
