#!/usr/bin/env python3
"""Stress of the blocking B = 1 call (vad_step_host_sync behind model(chunk, sr)): N calls over a long signal with silences, clicks and a NaN
stretch; the per-call probabilities must equal ONE audio_forward of the same signal bit for bit (same kernels' arithmetic, same state)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from silero_vad_amd import load_silero_vad
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
m = load_silero_vad(device=0)
rng = np.random.default_rng(3)
wav = np.load(Path(__file__).resolve().parents[1] / "tests/golden/audio_16k.npz")["pcm"].astype(np.float32) / 32768.0
sig = np.tile(wav, N * 512 // len(wav) + 1)[: N * 512].copy()
for _ in range(200):                                     # digital silences of 0.3 .. 3 chunks at random places
    a = int(rng.integers(0, len(sig) - 2000)); sig[a: a + int(rng.integers(150, 1600))] = 0.0
x = torch.from_numpy(sig)
m.reset_states()
t0 = time.perf_counter()
got = np.empty(N, dtype=np.float32)
for t in range(N):
    got[t] = m(x[t * 512:(t + 1) * 512], 16000).item()
dt = time.perf_counter() - t0
m.reset_states()
want = m.audio_forward(x.unsqueeze(0), 16000)[0].numpy()
same = got.tobytes() == want[:N].tobytes()
print(f"{N} calls in {dt:.2f} s = {dt / N * 1e6:.2f} us per call; identical to one audio_forward: {same}; max |d| {np.abs(got - want[:N]).max():.3g}")
sys.exit(0 if same else 1)
