#!/usr/bin/env python3
"""vad_upload_rows (gather kernel over PCIe) alone, at the refill scheduler's shape (2 048 rows x 128 KB) and the bucket scheduler's
(256 rows x 1 MB): GB/s of live bytes."""
import ctypes, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from silero_vad_amd import Engine
eng = Engine(device=0)
dev = torch.device("cuda", 0)
arena = torch.empty(1 << 30, dtype=torch.int16, pin_memory=True)
arena.view(-1, 1 << 20)[:] = torch.arange(1 << 20, dtype=torch.int16)
src = torch.empty(256 << 20, dtype=torch.uint8, pin_memory=True); dst0 = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dst0.copy_(src, non_blocking=True); torch.cuda.synchronize(); link = (256 << 20) / (time.perf_counter() - t0) / 1e9
print("plain copy GB/s %.2f" % link)
rng = np.random.default_rng(0)
for n, width, fill in ((2048, 65536, 1.0), (2048, 65536, 0.93), (256, 524288, 1.0), (4096, 32768, 1.0), (1024, 131072, 1.0)):
    dst = torch.empty((n, width), dtype=torch.int16, device=dev)
    offs = (rng.permutation(arena.numel() // width - 1)[:n].astype(np.int64) * width) // 8 * 8
    lens = np.full(n, width, dtype=np.int64)
    if fill < 1.0:
        k = int(n * (1 - fill) * 2)
        lens[:k] = rng.integers(0, width, size=k)
    rows = (arena.data_ptr() + offs * 2).astype(np.uint64)
    rp = rows.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p)); lp = lens.ctypes.data_as(ctypes.POINTER(ctypes.c_long))
    best = 0
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.upload_rows(rp, lp, n, width, 2, dst, 1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = max(best, float(lens.sum()) * 2 / dt / 1e9)
    print(f"rows {n} x {width * 2 // 1024} KB, fill {fill}: live GB/s {best:.2f}  ({best / link:.3f} of the plain copy)")
