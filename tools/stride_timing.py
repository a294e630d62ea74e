#!/usr/bin/env python3
"""Does the PCM row stride matter (4096 rows x 512 KiB is a power-of-two stride)?  Times the kernels for several row
paddings, alternating.  Round-1 answer: no -- what differs between the first and the later configurations of a run is
the GPU's clock ramp (+4 %; bench.py therefore runs untimed ramp steps first), not the stride."""
import json, sys, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
from silero_vad_amd import Engine
dev = torch.device("cuda", 0); eng = Engine(0)
sr, n, B, T = 16000, 512, 4096, 256
L = T * n
for pad in (64, 0, 64, 0, 16, 0):
    buf = torch.randn((B, L + pad), device=dev) * 0.05
    x = buf[:, :L]
    ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev)
    for _ in range(2): eng.forward_audio(x, sr, ctx, st)
    eng.set_option("profile", "1")
    for _ in range(5): eng.forward_audio(x, sr, ctx, st)
    f, r, c = eng.kernel_times(); eng.set_option("profile", "0")
    print(json.dumps({"pad_floats": pad, "front_ms": round(f / c, 3), "rec_ms": round(r / c, 3)}), flush=True)
    del buf, x
