import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from silero_vad_amd import Engine, _lib
from oracle import Oracle
orc = Oracle()
eng = Engine(0, library=_lib.lib_ab()); dev = torch.device("cuda", 0)
for sr, f in ((16000, "audio_16k"), (8000, "audio_8k")):
    n = 512 if sr == 16000 else 256
    wav = np.load(ROOT / f"tests/golden/{f}.npz")["pcm"].astype(np.float32) / 32768.0
    for scale in (1.0, 1e-1, 1e-2, 1e-3, 1e-4):
        rows = np.stack([np.roll(wav, -b * 4001)[:256 * n] for b in range(32)]) * np.float32(scale)
        want, _, wst = orc.forward_audio(rows, sr)
        out = {}
        for algo in ("winograd", "winograd2", "direct"):
            eng.set_option("enc0", algo)
            x = torch.from_numpy(rows).to(dev)
            ctx = torch.zeros((32, n // 8), device=dev); st = torch.zeros((2, 32, 128), device=dev)
            p = eng.forward_audio(x, sr, ctx, st).cpu().numpy()
            s = st.cpu().numpy()
            out[algo] = (float(np.abs(p - want).max()), float((np.abs(s - wst) / np.maximum(1, np.abs(wst))).max()))
        eng.set_option("enc0", "winograd")
        print(sr, "scale", scale, {k: (round(v[0] * 1e6, 2), round(v[1] * 1e6, 2)) for k, v in out.items()}, "(max|dp|, state err) x 1e-6 vs the oracle; p max", float(want.max()))
