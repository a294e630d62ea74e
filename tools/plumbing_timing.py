#!/usr/bin/env python3
"""BASELINE configs[0] (plumbing): get_speech_timestamps on the 60 s fixture through the per-chunk model protocol,
B = 1, for each precision policy of the wrapper.  Prints chunks/s and the segment count (reference: 19)."""
import json, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import load_silero_vad, get_speech_timestamps
wav = torch.from_numpy(np.load(ROOT / "tests/golden/audio_16k.npz")["pcm"].astype(np.float32) / 32768.0)
for prec in ("auto", "f16x3", "fp32"):
    m = load_silero_vad(device=0, precision=prec)
    get_speech_timestamps(wav[:16000 * 5], m)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ts = get_speech_timestamps(wav, m)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"precision": prec, "segments": len(ts), "chunks_per_s": round(1875 / dt, 1), "ms_per_chunk": round(dt / 1875 * 1e3, 4)}))
