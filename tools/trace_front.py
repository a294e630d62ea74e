#!/usr/bin/env python3
"""Bring-up: per-workgroup phase timeline of front_kernel (needs a -DVAD_TRACE=1 build, see
tools/variants.py).  Run on the GPU box:
    SILERO_VAD_AMD_LIB=build/variants/lib_trace.so python tools/trace_front.py gpurun_out/trace.npy
Slots per workgroup: 0 start, 1 tables ready, 2..4 after FFT 0..2, 5 before FFT 3, 6 after FFT 3,
7 enc1 done, 8 enc3 done, 9 end (100 MHz wall clock), 10 HW_ID, 11 XCC_ID."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import Engine  # noqa: E402

out = sys.argv[1]
B, T, sr = 4096, 256, 16000
dev = torch.device("cuda", 0)
eng = Engine(0)
pcm = torch.randn((B, T * 512), device=dev) * 0.03
ctx = torch.zeros((B, 64), device=dev)
st = torch.zeros((2, B, 128), device=dev)
nwg = B // 16 * T // 4
trace = torch.zeros((nwg, 16), dtype=torch.int64, device=dev)
eng.reserve(sr, B, T)
eng.forward_audio(pcm, sr, ctx, st)                      # warm-up without tracing
torch.cuda.synchronize()
eng.set_option("trace_ptr", hex(trace.data_ptr()))
eng.forward_audio(pcm, sr, ctx, st)
torch.cuda.synchronize()
eng.set_option("trace_ptr", "0")
np.save(out, trace.cpu().numpy())
t = trace.cpu().numpy()
print("workgroups traced:", int((t[:, 9] > 0).sum()), "span ms:", (t[:, 9].max() - t[:, 0].min()) / 1e5)
