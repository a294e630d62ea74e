#!/usr/bin/env python3
"""Bring-up (GPU box): do the waves of ANOTHER kernel (pure fp32 VALU, one wave per SIMD) run beside the wide bf16 x 9 frontend (one
wave per SIMD, matrix pipe + exposed latencies)?  front alone, spinner alone, both on two streams."""
import ctypes, os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import Engine
eng = Engine(0)
dev = torch.device("cuda", 0)
sr, B, T, n = 16000, 4096, 256, 512
x = 0.1 * torch.randn((B, T * n), device=dev)
mma = sys.argv[1] if len(sys.argv) > 1 else "bf16x9"
eng.set_option("front_mma", mma)
ctx = torch.zeros((B, n // 8), device=dev)
gx = None
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
L = eng._L

def front():
    L.vad_debug_frontend(eng._h, sr, 64, n * 4, x.data_ptr(), x.stride(0), ctx.data_ptr(), gxs.data_ptr(), ctypes.c_void_p(s1.cuda_stream))

# the frontend alone through forward_audio's front kernel: use profile times
st = torch.zeros((2, B, 128), device=dev)
def fwd(stream):
    with torch.cuda.stream(stream):
        eng.forward_audio(x, sr, ctx, st)
for _ in range(30): fwd(s1)
torch.cuda.synchronize()
def wall(fn, reps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
t_fwd = wall(lambda: fwd(s1))
for blocks, iters in ((1024, 130000), (1024, 260000), (2048, 130000)):
    def spin():
        L.vad_debug_foreign_load(eng._h, 1, blocks, iters, ctypes.c_void_p(s2.cuda_stream))
    spin(); torch.cuda.synchronize()
    t_spin = wall(spin)
    def both():
        spin(); fwd(s1)
    t_both = wall(both)
    print(f"{mma}: forward (front + rec) alone {t_fwd:.3f} ms, spinner {blocks} x {iters} alone {t_spin:.3f} ms, both {t_both:.3f} ms "
          f"(sum {t_fwd + t_spin:.3f}, max {max(t_fwd, t_spin):.3f})", flush=True)
