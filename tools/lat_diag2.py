import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from silero_vad_amd import Engine
from oracle import Oracle
orc = Oracle()
eng = Engine(0); dev = torch.device("cuda", 0)
wav = np.load(ROOT / "tests/golden/audio_16k.npz")["pcm"].astype(np.float32) / 32768.0
for k in (2, 3):
    for B, odd in ((5, True), (5, False), (2, True), (16, True)):
        L16 = 3 * 512 + 77
        L = L16 * k - (k - 1 if odd else 0)
        raw = np.zeros((B, L), np.float32)
        rows = np.stack([np.roll(wav, -b * 313)[:L16] for b in range(B)])
        raw[:, ::k] = rows
        want = orc.audio_forward(rows, 16000)
        for form in ("throughput", "latency"):
            eng.set_option("front", form)
            x = torch.from_numpy(raw).to(dev)
            ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev)
            p = eng.forward_audio(x, 16000 * k, ctx, st).cpu().numpy()
            err = np.abs(p - want).max(1)
            print(k, B, "odd" if odd else "even", form, "max err per stream", np.round(err, 6))
        eng.set_option("front", "auto")
