#!/usr/bin/env python3
"""Bring-up: where do the two forms of the frontend differ?  (GPU box)"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import Engine
eng = Engine(0)
dev = torch.device("cuda", 0)
for sr in (16000, 8000):
    n = 512 if sr == 16000 else 256
    rng = np.random.default_rng(0)
    for name, x in (("noise", 0.1 * rng.standard_normal((16, n)).astype(np.float32)),
                    ("zeros", np.zeros((16, n), np.float32)),
                    ("impulse", np.eye(16, n, 300 % n, dtype=np.float32))):
        xt = torch.from_numpy(x).to(dev)
        out = {}
        for form in ("throughput", "latency"):
            eng.set_option("front", form)
            out[form] = eng.debug_frontend(xt, sr, torch.zeros((16, n // 8), device=dev)).cpu().numpy()[:, 0]   # [16, 512]
        eng.set_option("front", "auto")
        d = out["throughput"] != out["latency"]
        ad = np.abs(out["throughput"] - out["latency"])
        print(sr, name, "differing", int(d.sum()), "of", d.size, "max", float(ad.max()), "rel", float((ad / (np.abs(out["throughput"]) + 1e-30)).max()),
              "per gate", [int(d[:, 128 * q:128 * (q + 1)].sum()) for q in range(4)], "streams with diffs", int(d.any(1).sum()))
