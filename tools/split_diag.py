#!/usr/bin/env python3
"""Bring-up on the GPU box: determinism and cross-precision agreement of the split kernels, per kernel.
    [SILERO_VAD_AMD_LIB=build/variants/lib_<v>.so] python tools/split_diag.py
Prints one JSON line per (workload, front precision, rec precision)."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import Engine  # noqa: E402

dev = torch.device("cuda", 0)
eng = Engine(0)
wav = torch.from_numpy(np.load(ROOT / "tests/golden/audio_16k.npz")["pcm"].astype(np.float32) / 32768.0).to(dev)
sr, n = 16000, 512


def rows(B, T):
    idx = (torch.arange(B, device=dev)[:, None] * 7919 + torch.arange(T * n, device=dev)[None]) % len(wav)
    return wav[idx].contiguous()


def run(x, fp, rp):
    eng.set_option("precision_front", fp)
    eng.set_option("precision_rec", rp)
    B = x.shape[0]
    ctx = torch.zeros((B, 64), device=dev)
    st = torch.zeros((2, B, 128), device=dev)
    p = eng.forward_audio(x, sr, ctx, st)
    torch.cuda.synchronize()
    return p.clone()


def gx(x, fp):
    eng.set_option("precision_front", fp)
    ctx = torch.zeros((x.shape[0], 64), device=dev)
    g = eng.debug_frontend(x, sr, ctx)
    torch.cuda.synchronize()
    return g.clone()


import os
SHAPES = ((1, 1875), (4096, 24)) if os.environ.get("DIAG_SHORT") else ((1, 1875), (16, 64), (64, 24), (1024, 8), (4096, 24))
for B, T in SHAPES:
    x = rows(B, T)
    ref = run(x, "fp32", "fp32")
    g32 = gx(x, "fp32") if B * T <= 1 << 16 else None
    if g32 is not None:
        a, b = gx(x, "f16x3"), gx(x, "f16x3")
        d = (a - g32).abs()
        bad_tiles = (d.amax(dim=2) > 1e-3 * g32.abs().amax()).nonzero()
        print(json.dumps({"B": B, "T": T, "what": "gx split", "deterministic": bool(torch.equal(a, b)),
                          "max_diff_vs_fp32": float(d.max()), "scale": float(g32.abs().max()),
                          "n_bad_chunks": int(len(bad_tiles)), "first_bad": bad_tiles[:6].tolist(),
                          "run_to_run": float((a - b).abs().max())}), flush=True)
    for fp, rp in (("f16x3", "fp32"), ("fp32", "f16x3"), ("f16x3", "f16x3")):
        a, b = run(x, fp, rp), run(x, fp, rp)
        print(json.dumps({"B": B, "T": T, "front": fp, "rec": rp, "deterministic": bool(torch.equal(a, b)),
                          "max_dp_vs_fp32": float((a - ref).abs().max()),
                          "run_to_run": float((a - b).abs().max())}), flush=True)
