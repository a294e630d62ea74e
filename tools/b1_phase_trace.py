#!/usr/bin/env python3
"""Phase boundaries of the one-stream step kernel (kernel_step_one.hip VAD_STAMP): shader-clock stamps of thread 0, B = 1."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from silero_vad_amd import load_silero_vad
m = load_silero_vad(device=0); eng = m.engine
names = ["start", "slice+tables in LDS", "STFT done", "route known", "tin done", "enc0 m1..m4", "enc0 done", "enc1..3 done", "W_ih done", "W_hh done", "end"]
for sr, n in ((16000, 512), (8000, 256)):
    x = torch.randn((1, n), device=m.device) * 0.1
    ctx = torch.zeros((1, n // 8), device=m.device); st = torch.zeros((2, 1, 128), device=m.device); p = torch.empty((1,), device=m.device)
    tr = torch.zeros(16, dtype=torch.int64, device=m.device)
    for _ in range(200): eng.step(x, sr, ctx, st, p)
    eng.set_option("trace_ptr", str(tr.data_ptr()))
    rows = []
    for _ in range(50):
        eng.step(x, sr, ctx, st, p); torch.cuda.synchronize(); rows.append(tr.cpu().numpy().copy())
    eng.set_option("trace_ptr", "0")
    d = np.median(np.diff(np.stack(rows)[:, :11], axis=1), axis=0)
    print(sr, "cycles per phase:", {names[i + 1]: int(d[i]) for i in range(10)}, "total", int(d.sum()))
