#!/usr/bin/env python3
"""Bring-up of the bf16 x 9 frontend (GPU box): gx against the fp32 frontend per gate / per layer-sized block, then the whole
path against the oracle, then timing at the C2 shape."""
import json
import sys
import time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import Engine
from oracle import Oracle
eng = Engine(0)
dev = torch.device("cuda", 0)
res = {}
for sr in (16000, 8000):
    n = 512 if sr == 16000 else 256
    rng = np.random.default_rng(0)
    T = 3
    t = np.arange(40 * T * n, dtype=np.float64).reshape(40, T * n)
    speechy = (0.2 * np.sin(2 * np.pi * 180.0 * t / sr) * (1 + np.sin(2 * np.pi * 3.0 * t / sr)) + 0.05 * rng.standard_normal(t.shape)).astype(np.float32)
    for name, x in (("noise", 0.1 * rng.standard_normal((40, T * n)).astype(np.float32)), ("speechy", speechy),
                    ("impulse", np.eye(40, T * n, 300, dtype=np.float32))):
        xt = torch.from_numpy(x).to(dev)
        out = {}
        for mma in ("fp32", "bf16x9"):
            eng.set_option("front_mma", mma)
            eng.set_option("front", "throughput")
            out[mma] = eng.debug_frontend(xt, sr, torch.zeros((40, n // 8), device=dev)).cpu().numpy()
        eng.set_option("front_mma", "fp32")
        eng.set_option("front", "auto")
        ad = np.abs(out["fp32"] - out["bf16x9"])
        sc = np.abs(out["fp32"]).max()
        print(sr, name, "max|d gx|", float(ad.max()), "scale", float(sc), "per gate", [float(ad[..., 128 * q:128 * (q + 1)].max()) for q in range(4)],
              "per step", [float(ad[:, k].max()) for k in range(T)], "worst streams", np.argsort(-ad.reshape(40, -1).max(1))[:4].tolist(), flush=True)
        res[f"gx_{sr}_{name}"] = [float(ad.max()), float(sc)]
# whole path against the oracle
o = Oracle()
for sr in (16000, 8000):
    n = 512 if sr == 16000 else 256
    rng = np.random.default_rng(1)
    B, T = 24, 40
    t = np.arange(B * T * n, dtype=np.float64).reshape(B, T * n)
    pcm = (0.2 * np.sin(2 * np.pi * 180.0 * t / sr) * (1 + np.sin(2 * np.pi * 3.0 * t / sr)) + 0.05 * rng.standard_normal(t.shape)).astype(np.float32)
    o.reset_states()
    want = o.audio_forward(pcm, sr)
    for mma, rec in (("fp32", "fp32"), ("bf16x9", "fp32"), ("bf16x9", "bf16x9")):
        eng.set_option("front_mma", mma)
        eng.set_option("rec", rec)
        st = torch.zeros((2, B, 128), device=dev)
        ctx = torch.zeros((B, n // 8), device=dev)
        got = eng.forward_audio(torch.from_numpy(pcm).to(dev), sr, ctx, st).cpu().numpy()
        print(sr, mma, rec, "max|dp| vs oracle", float(np.abs(got - want).max()), "state", float(np.abs(st.cpu().numpy() - np.asarray(o._state)).max()), flush=True)
        res[f"path_{sr}_{mma}_{rec}"] = float(np.abs(got - want).max())
eng.set_option("rec", "fp32")
# timing at C2
for sr, B, T in ((16000, 4096, 256), (8000, 4096, 256)):
    n = 512 if sr == 16000 else 256
    x = (0.1 * torch.randn((B, T * n), device=dev))
    for mma in ("fp32", "bf16x9", "fp32", "bf16x9"):
        eng.set_option("front_mma", mma)
        st = torch.zeros((2, B, 128), device=dev)
        ctx = torch.zeros((B, n // 8), device=dev)
        for _ in range(3):
            eng.forward_audio(x, sr, ctx, st)
        torch.cuda.synchronize()
        eng.set_option("profile", "1")
        t0 = time.perf_counter()
        for _ in range(10):
            eng.forward_audio(x, sr, ctx, st)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 10
        f, r, c = eng.kernel_times()
        eng.set_option("profile", "0")
        print(sr, mma, "front ms", f / c, "rec ms", r / c, "wall ms", wall * 1e3, flush=True)
        res[f"time_{sr}_{mma}"] = [f / c, r / c]
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/b9_diag.json").write_text(json.dumps(res, indent=1))
