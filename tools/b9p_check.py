#!/usr/bin/env python3
"""Bring-up (GPU box): the wide bf16 x 9 frontend against the narrow one -- gate pre-activations and the whole path, bit for bit --
and both against the fp32 frontend; then kernel times at the C2 / C3 shape."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import Engine
eng = Engine(0)
dev = torch.device("cuda", 0)
wav = {16000: np.load("tests/golden/audio_16k.npz")["pcm"], 8000: np.load("tests/golden/audio_8k.npz")["pcm"]}
for sr in (16000, 8000):
    n = 512 if sr == 16000 else 256
    for B, T, dt in ((37, 9, torch.float32), (64, 40, torch.int16), (1, 3, torch.float32), (400, 5, torch.float32)):
        rows = np.stack([np.roll(wav[sr], -b * 7919)[:T * n] for b in range(B)])
        x = torch.from_numpy(rows if dt == torch.int16 else rows.astype(np.float32) / 32768.0).to(dev)
        res = {}
        for mma in ("fp32", "bf16x9_narrow", "bf16x9_pair"):
            eng.set_option("front_mma", mma)
            ctx = torch.zeros((B, n // 8), device=dev); st = torch.zeros((2, B, 128), device=dev)
            p = eng.forward_audio(x, sr, ctx, st).clone()
            gx = eng.debug_frontend(x.float() / (32768.0 if dt == torch.int16 else 1.0), sr, torch.zeros((B, n // 8), device=dev)).clone()
            res[mma] = (p, st.clone(), ctx.clone(), gx)
        eng.set_option("front_mma", "fp32")
        same = all(torch.equal(a, b) for a, b in zip(res["bf16x9_narrow"], res["bf16x9_pair"]))
        d32 = float((res["bf16x9_pair"][0] - res["fp32"][0]).abs().max())
        dg = float((res["bf16x9_pair"][3] - res["bf16x9_narrow"][3]).abs().max())
        print(f"sr {sr} B {B} T {T} {dt}: pair == narrow (probs, state, ctx, gx): {same}  max|dgx| {dg:.3e}  |dp| pair vs fp32 {d32:.3e}", flush=True)
# times
for sr in (16000, 8000):
    B, T = 4096, 256
    n = 512 if sr == 16000 else 256
    x = 0.1 * torch.randn((B, T * n), device=dev)
    out = []
    for mma in ("fp32", "bf16x9_narrow", "bf16x9_pair"):
        for rec in ("fp32", "bf16x9"):
            if mma == "fp32" and rec != "fp32":
                continue
            eng.set_option("front_mma", mma); eng.set_option("rec", rec)
            st = torch.zeros((2, B, 128), device=dev); ctx = torch.zeros((B, n // 8), device=dev)
            for _ in range(40):
                eng.forward_audio(x, sr, ctx, st)
            torch.cuda.synchronize()
            eng.set_option("profile", "1")
            for _ in range(20):
                eng.forward_audio(x, sr, ctx, st)
            torch.cuda.synchronize()
            f, r, c = eng.kernel_times()
            eng.set_option("profile", "0")
            out.append(f"{mma}/{rec}: front {f / c:.3f} rec {r / c:.3f}")
    eng.set_option("front_mma", "fp32"); eng.set_option("rec", "fp32")
    print(f"{sr // 1000}k  " + " | ".join(out), flush=True)
