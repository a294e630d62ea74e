#!/usr/bin/env python3
"""Bring-up: where does the time of vad_upload_rows go?  (run on the GPU box)
   call time vs completion time of the DMA-per-row and the gather-kernel routes for one ~1 GB bucket of int16 rows in pinned
   memory, against one plain pinned -> device copy; and the compute kernels' time with / without a gather beside them."""
import ctypes, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import Engine

dev = torch.device("cuda", 0)
eng = Engine(0)
n, width = 1024, 480_000                       # 1024 x 30 s x int16 = 0.98 GB
base = torch.randint(-3000, 3000, (64 << 20,), dtype=torch.int16).pin_memory()
rng = np.random.default_rng(0)
offs = rng.integers(0, base.numel() - width, n) // 8 * 8
lens = rng.integers(width * 2 // 3, width + 1, n)
rows = (ctypes.c_void_p * n)(*[base.data_ptr() + 2 * int(o) for o in offs])
clens = (ctypes.c_long * n)(*[int(v) for v in lens])
dst = torch.empty((n, width), dtype=torch.int16, device=dev)
flat = torch.empty(n * width, dtype=torch.int16).pin_memory()
side = torch.cuda.Stream(dev)
nbytes = float(lens.sum()) * 2

def t_upload(how, reps=4):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            t0 = time.perf_counter()
            eng.upload_rows(rows, clens, n, width, 2, dst, how)
            t1 = time.perf_counter()
            side.synchronize()
            t2 = time.perf_counter()
        out.append((t1 - t0, t2 - t0))
    return out[-1]

for how, name in ((0, "dma per row"), (1, "gather kernel")):
    c, tot = t_upload(how)
    print(f"{name:14s}: call {c*1e3:8.2f} ms, complete {tot*1e3:8.2f} ms, {nbytes/tot/1e9:6.1f} GB/s of live bytes")
torch.cuda.synchronize(); t0 = time.perf_counter(); dst.view(-1).copy_(flat, non_blocking=True); torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"plain copy    : {t*1e3:8.2f} ms, {n*width*2/t/1e9:6.1f} GB/s")

# compute beside a gather
B, T = 4096, 64
pcm = torch.randn((B, T * 512), device=dev) * 0.05
ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev); probs = torch.empty((B, T), device=dev)
eng.reserve(16000, B, T)
def compute(k=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        eng.forward_audio(pcm, 16000, ctx, st, probs)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / k
for _ in range(3): compute()
alone = compute()
for how, name in ((1, "gather"), (0, "dma")):
    with torch.cuda.stream(side):
        for _ in range(3):
            eng.upload_rows(rows, clens, n, width, 2, dst, how)
    beside = compute()
    torch.cuda.synchronize()
    print(f"forward_audio 4096x64: alone {alone:.3f} ms, beside {name} {beside:.3f} ms")

# does the upload CALL block when other streams have work queued?
lane = torch.cuda.Stream(dev)
for how, name in ((1, "gather"), (0, "dma")):
    torch.cuda.synchronize()
    with torch.cuda.stream(lane):
        for _ in range(20):
            eng.forward_audio(pcm, 16000, ctx, st, probs)        # ~28 ms of kernels queued on another stream
    with torch.cuda.stream(side):
        t0 = time.perf_counter()
        eng.upload_rows(rows, clens, n, width, 2, dst, how)
        t1 = time.perf_counter()
        eng.upload_rows(rows, clens, n, width, 2, dst, how)
        t2 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name}: upload call with 28 ms of kernels queued on another stream: first {1e3*(t1-t0):.2f} ms, second (behind the first upload) {1e3*(t2-t1):.2f} ms")
