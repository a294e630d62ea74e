#!/usr/bin/env python3
"""Stress: N full-size launches (4096 streams x 256 chunks = 65 536 tiles each) of the f16x3 path must be
bit-identical to each other and agree with the fp32 path.  Prints one JSON line (also timing)."""
import json, sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import Engine
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
SR = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
dev = torch.device("cuda", 0)
eng = Engine(0)
tag = "16k" if SR == 16000 else "8k"
wav = torch.from_numpy(np.load(ROOT / f"tests/golden/audio_{tag}.npz")["pcm"].astype(np.float32) / 32768.0).to(dev)
sr, n, B, T = SR, (512 if SR == 16000 else 256), 4096, 256
idx = (torch.arange(B, device=dev)[:, None] * 7919 + torch.arange(T * n, device=dev)[None]) % len(wav)
x = wav[idx].contiguous()
def run(prec_f, prec_r):
    eng.set_option("precision_front", prec_f); eng.set_option("precision_rec", prec_r)
    ctx = torch.zeros((B, n // 8), device=dev); st = torch.zeros((2, B, 128), device=dev)
    p = eng.forward_audio(x, sr, ctx, st)
    torch.cuda.synchronize()
    return p.clone(), st
ref, _ = run("fp32", "fp32")
out = {"lib": sys.argv[1] if len(sys.argv) > 1 else "", "launches": N, "sr": sr}
for name, pf, pr in (("front", "f16x3", "fp32"), ("rec", "fp32", "f16x3"), ("both", "f16x3", "f16x3")):
    p0, s0 = run(pf, pr)
    bad_streams, worst = 0, 0.0
    for i in range(N):
        p, s = run(pf, pr)
        bad_streams += int((p.view(torch.int32) != p0.view(torch.int32)).any(dim=1).sum())
        worst = max(worst, float((p - ref).abs().max()))
    out[name] = {"streams_differing_total": bad_streams, "max_dp_vs_fp32": worst}
eng.set_option("precision", "f16x3")
eng.set_option("profile", "1")
ctx = torch.zeros((B, n // 8), device=dev); st = torch.zeros((2, B, 128), device=dev)
for _ in range(5):
    eng.forward_audio(x, sr, ctx, st)
f, r, c = eng.kernel_times()
out["front_ms"] = round(f / c, 4); out["rec_ms"] = round(r / c, 4)
print(json.dumps(out), flush=True)
