import sys, json
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import Engine
eng = Engine(0); dev = torch.device("cuda", 0)
for B in (8192, 4096, 16):
    x = torch.randn((B, 512), device=dev) * 0.05
    ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev); out = torch.zeros((B, 1), device=dev)
    res = {}
    for form, fuse in (("latency", "1"), ("latency", "0"), ("throughput", "0")):
        eng.set_option("front", form); eng.set_option("fuse_step", fuse)
        for _ in range(200): eng.step(x, 16000, ctx, st, out)
        eng.set_option("profile", "1")
        for _ in range(50): eng.step(x, 16000, ctx, st, out)
        f, r, c = eng.kernel_times(); eng.set_option("profile", "0")
        res[form + ("+cell" if fuse == "1" else "")] = (round(f / c * 1e3, 1), round(r / c * 1e3, 1))
    print("B", B, "(front, rec) us", res)
