import sys, numpy as np, torch
sys.path.insert(0, '.')
from silero_vad_amd import Engine
eng = Engine(0)
dev = torch.device("cuda", 0)
wav = np.load("tests/golden/audio_16k.npz")["pcm"].astype(np.float32) / 32768.0
for sr, n in ((16000, 512), (8000, 256)):
    B, T = 64, 8
    rows = np.stack([np.roll(wav, -b * 7919)[:T * n] for b in range(B)])
    x = torch.from_numpy(rows).to(dev)
    ctx = torch.zeros((B, n // 8), device=dev)
    out = {}
    for algo in ("direct", "winograd", "winograd"):
        eng.set_option("enc0", algo)
        g = eng.debug_frontend(x, sr, ctx).cpu().numpy()      # [B][T][512]
        out.setdefault(algo, []).append(g)
    d, w1, w2 = out["direct"][0], out["winograd"][0], out["winograd"][1]
    print(sr, "deterministic:", np.array_equal(w1, w2), "max|w-d|", np.abs(w1 - d).max(), "scale", np.abs(d).max())
    err = np.abs(w1 - d)
    print("  err by gate block (mean):", err.reshape(B, T, 32, 16).mean((0, 1, 3)).round(3))
    print("  err by t (max):", err.max((0, 2)).round(2))
    print("  err by stream%16 (max):", err.reshape(B // 16, 16, T, 512).max((0, 2, 3)).round(2))
    print("  err by stream tile (max):", err.reshape(B // 16, 16, T, 512).max((1, 2, 3)).round(2))
