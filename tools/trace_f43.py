#!/usr/bin/env python3
"""Bring-up: per-WAVE phase timeline of the fp32 throughput frontend front_f43_kernel (needs the -DVAD_TRACE=1 build:
`VAD_VARIANT_UNITS=kernel_front_f43.hip python tools/variants.py build trace`).  GPU box:
    SILERO_VAD_AMD_LIB=build/variants/lib_trace.so python tools/trace_f43.py [16000|8000] [out.json]
Slots per wave (shader clock, s_memtime): 0 start, 1 tables + units 0, 1 in LDS, 2 + 2 v samples of frame v arrived, 3 + 2 v its FFT done,
10 encoders 0 + 1 done, 11 encoders 2 + 3 done, 12 end; 13 HW_ID, 14 XCC_ID, 15 100 MHz wall clock at the end.

Reading: a wave shares its SIMD with one other wave of the same kernel at a random phase, and on this part the fp32 MFMA and the VALU of
all waves of a SIMD serialise on one issue pipe (profiles/r03a_issue_pipes2.md).  With D = a tile's own issue demand (MFMA x 32 + VALU
cycles) and L = a wave's lifetime, the partner takes D / L of every cycle on average; a phase of length L_p and own demand D_p therefore
holds   stall_p = L_p (1 - D / L) - D_p   cycles in which NEITHER wave issued."""
import json
import os
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import Engine  # noqa: E402

sr = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
n = 512 if sr == 16000 else 256
B, T = int(os.environ.get("TRACE_B", 4096)), int(os.environ.get("TRACE_T", 256))
dev = torch.device("cuda", 0)
eng = Engine(0)
pcm = torch.randn((B, T * n), device=dev) * 0.03
ctx = torch.zeros((B, n // 8), device=dev)
st = torch.zeros((2, B, 128), device=dev)
nw = B // 16 * T
trace = torch.zeros((nw, 32), dtype=torch.int64, device=dev)
for _ in range(40):
    eng.forward_audio(pcm, sr, ctx, st)
torch.cuda.synchronize()
eng.set_option("trace_ptr", hex(trace.data_ptr()))
eng.set_option("profile", "1")
eng.forward_audio(pcm, sr, ctx, st)
torch.cuda.synchronize()
front_ms = eng.kernel_times()[0]
eng.set_option("profile", "0")
eng.set_option("trace_ptr", "0")
t = trace.cpu().numpy().astype(np.float64)
ok = t[:, 12] > 0
t = t[ok]
life = t[:, 12] - t[:, 0]
L = float(np.median(life))
# own issue demand of a tile per phase (cycles): MFMAs x 32 + VALU cycles (bench.py WORK; profiles/r03f_fp32_summary.md)
if sr == 16000:
    mf = {"enc01": (1536 + 640) * 32, "enc23": 256 * 32, "wih": 1024 * 32}
    valu = {"fft": 16000 / 4, "enc01": 3000 + 2000 + 1000, "enc23": 100, "wih": 200}
else:
    mf = {"enc01": (768 + 640) * 32, "enc23": 256 * 32, "wih": 1024 * 32}
    valu = {"fft": 7400 / 4, "enc01": 1700 + 1500 + 600, "enc23": 100, "wih": 200}
D = sum(mf.values()) + 4 * valu["fft"] + valu["enc01"] + valu["enc23"] + valu["wih"]
share = D / L
phases = [("prologue (tables, units 0-1, barrier)", 0, 1, 0.0)]
for v in range(4):
    phases.append((f"frame {v}: shift-register moves", (1 if v == 0 else 3 + 2 * (v - 1)), 16 + v, 33.0 * 4 * max(0, v)))
    phases.append((f"frame {v}: loads issued -> samples arrived", 16 + v, 2 + 2 * v, 80.0))
    phases.append((f"frame {v}: FFT", 2 + 2 * v, 3 + 2 * v, valu["fft"]))
phases += [("encoder 0 + 1", 9, 10, mf["enc01"] + valu["enc01"]), ("encoder 2 + 3", 10, 11, mf["enc23"] + valu["enc23"]),
           ("W_ih (4 gates) + gx stores", 11, 12, mf["wih"] + valu["wih"])]
wall = t[:, 15]
span_ms = (wall.max() - wall.min()) / 1e5
clock = float(np.median(life) / 1.0)         # cycles; GHz from the wall clock below
out = {"sr": sr, "waves": int(ok.sum()), "front_ms_hipevents": front_ms, "lifetime_cycles_median": L, "own_demand_cycles": D,
       "own_share_of_simd": share, "phases": []}
print(f"sr {sr} B {B} T {T}: {int(ok.sum())} waves traced, front {front_ms:.3f} ms; wave lifetime median {L:.0f} cycles (p10 {np.percentile(life, 10):.0f}, "
      f"p90 {np.percentile(life, 90):.0f}); own issue demand {D:.0f} = {share:.3f} of it (two waves per SIMD: {2 * share:.3f} of the pipe)")
tot_stall = 0.0
for name, a, b, dp in phases:
    d = t[:, b] - t[:, a]
    med = float(np.median(d))
    stall = med * (1 - share) - dp
    tot_stall += stall
    out["phases"].append({"phase": name, "median_cycles": med, "p10": float(np.percentile(d, 10)), "p90": float(np.percentile(d, 90)),
                          "own_demand": dp, "stall_cycles": stall})
    print(f"  {name:40s} median {med:9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}   own demand {dp:8.0f}   "
          f"neither wave issues: {stall:8.0f} ({stall / L * 100:5.1f} % of the lifetime)")
print(f"  sum of the stall estimates {tot_stall:.0f} cycles = {tot_stall / L * 100:.1f} % of a wave's lifetime")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
