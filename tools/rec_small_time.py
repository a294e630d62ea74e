#!/usr/bin/env python3
"""Per-step time of the small-batch recurrence (kernel_rec_small.hip) for 1, 2 and 4 streams per workgroup, against rec_kernel (rec_form=mfma)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from silero_vad_amd import load_silero_vad
m = load_silero_vad(device=0); eng = m.engine
T = 512
for B in (1, 256, 512, 1024):
    x = torch.randn((B, T * 512), device=m.device) * 0.1
    ctx = torch.zeros((B, 64), device=m.device); st = torch.zeros((2, B, 128), device=m.device)
    res = {}
    for form in ("auto", "mfma"):
        eng.set_option("rec_form", form)
        for _ in range(3): eng.forward_audio(x, 16000, ctx, st)
        torch.cuda.synchronize()
        eng.set_option("profile", "1")
        for _ in range(5):
            eng.forward_audio(x, 16000, ctx, st); torch.cuda.synchronize()
        f, r, c = eng.kernel_times()
        eng.set_option("profile", "0")
        res[form] = r / c / T * 1e3
    eng.set_option("rec_form", "auto")
    print("B", B, "us per step: small-batch form %.3f, matrix form %.3f" % (res["auto"], res["mfma"]))
