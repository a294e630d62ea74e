import json, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import Engine
dev = torch.device("cuda", 0); eng = Engine(0)
wav = np.load(ROOT / "tests/golden/audio_16k.npz")["pcm"].astype(np.float32) / 32768.0
sr, n, B = 16000, 512, 7
rows = np.stack([np.roll(wav, -b * 6007)[:8 * n] for b in range(B)])
x = torch.from_numpy(rows).to(dev)
z = torch.zeros((B, 64), device=dev)
gw = eng.debug_frontend(x, sr, z).clone()
c1 = x[:, 4 * n - 64:4 * n].contiguous()
g2 = eng.debug_frontend(x[:, 4 * n:].contiguous(), sr, c1).clone()
g1 = eng.debug_frontend(x[:, :4 * n].contiguous(), sr, z).clone()
print("gx first half equal", bool(torch.equal(gw[:, :4], g1)), "second half equal", bool(torch.equal(gw[:, 4:], g2)),
      float((gw[:, 4:] - g2).abs().max()))
def fw(xx, ctx, st):
    p = eng.forward_audio(xx.contiguous(), sr, ctx, st); torch.cuda.synchronize(); return p.clone()
ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev)
pw = fw(x, ctx, st); stw = st.clone()
ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev)
pa = fw(x[:, :4 * n], ctx, st); sta = st.clone()
pb = fw(x[:, 4 * n:], ctx, st)
print("probs first equal", bool(torch.equal(pw[:, :4], pa)), "second equal", bool(torch.equal(pw[:, 4:], pb)),
      "state equal", bool(torch.equal(stw, st)), float((pw[:, 4:] - pb).abs().max()))
# rec only: fp32 front
eng.set_option("precision_front", "fp32")
ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev)
pw = fw(x, ctx, st)
ctx = torch.zeros((B, 64), device=dev); st = torch.zeros((2, B, 128), device=dev)
pa = fw(x[:, :4 * n], ctx, st); pb = fw(x[:, 4 * n:], ctx, st)
print("fp32 front + split rec: second equal", bool(torch.equal(pw[:, 4:], pb)))
