#!/usr/bin/env python3
"""Bring-up: per-workgroup phase timeline of front_b9_kernel (needs a -DVAD_TRACE=1 build: tools/variants.py trace).  GPU box:
    SILERO_VAD_AMD_LIB=build/variants/lib_trace.so python tools/trace_b9.py
Slots per workgroup: 0 start, 1 tables + first units in LDS, 2 after the 4 FFTs, 3 after encoder 0/1, 4 after encoder 3, 5 end
(100 MHz wall clock); 8, 9 shader-cycle counter at start / end; 10 HW_ID, 11 XCC_ID."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import Engine  # noqa: E402

sr = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
n = 512 if sr == 16000 else 256
B, T = 4096, 256
dev = torch.device("cuda", 0)
eng = Engine(0)
eng.set_option("front_mma", "bf16x9")
pcm = torch.randn((B, T * n), device=dev) * 0.03
ctx = torch.zeros((B, n // 8), device=dev)
st = torch.zeros((2, B, 128), device=dev)
nwg = B // 16 * T // 4
trace = torch.zeros((nwg, 16), dtype=torch.int64, device=dev)
for _ in range(10):
    eng.forward_audio(pcm, sr, ctx, st)
torch.cuda.synchronize()
eng.set_option("trace_ptr", hex(trace.data_ptr()))
eng.forward_audio(pcm, sr, ctx, st)
torch.cuda.synchronize()
eng.set_option("trace_ptr", "0")
t = trace.cpu().numpy()
ok = t[:, 5] > 0
print("workgroups traced:", int(ok.sum()), "of", nwg, "kernel span ms:", (t[ok, 5].max() - t[ok, 0].min()) / 1e5)
life = (t[ok, 5] - t[ok, 0]) / 100.0                    # us
cyc = (t[ok, 9] - t[ok, 8]).astype(np.float64)
print("workgroup lifetime us: median %.1f  p10 %.1f  p90 %.1f ;  shader cycles per lifetime median %.0f => clock %.3f GHz"
      % (np.median(life), np.percentile(life, 10), np.percentile(life, 90), np.median(cyc), np.median(cyc / (life * 1e3))))
names = ["prologue (tables, units 0-1)", "4 x load + FFT", "encoder 0 + 1 (%d parts)" % (4 if sr == 16000 else 2), "encoder 2 + 3", "W_ih (4 gates)"]
for i, nm in enumerate(names):
    d = (t[ok, i + 1] - t[ok, i]) / 100.0
    print("  %-32s median %7.2f us   p10 %7.2f   p90 %7.2f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
# concurrency: workgroups alive per CU at the middle of the kernel
hw = t[ok, 10]
xcc = t[ok, 11] & 15
cu = (xcc << 16) | (hw & 0xFF00) | ((hw >> 12) & 0xF) << 20     # cu 11:8, sh 12, se 15:13
mid = (t[ok, 0].min() + t[ok, 5].max()) // 2
alive = (t[ok, 0] <= mid) & (t[ok, 5] >= mid)
ids, counts = np.unique(cu[alive], return_counts=True)
print("CUs seen:", len(np.unique(cu)), " workgroups alive at mid-kernel:", int(alive.sum()), " per CU:", dict(zip(*np.unique(counts, return_counts=True))))
