import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from silero_vad_amd import load_silero_vad
from oracle import Oracle
from conftest import state_err
from test_gpu_parity import _adversarial, rolled_rows, run_engine
m = load_silero_vad(device=0); o = Oracle()
for tag, sr, n in (("16k", 16000, 512), ("8k", 8000, 256)):
    wav = np.load(f"tests/golden/audio_{tag}.npz")["pcm"].astype(np.float32) / 32768.0
    names, adv = _adversarial(sr, 50)
    rows = np.concatenate([rolled_rows(wav, 40, 50 * n, 7919), adv])
    want, _, wst = o.forward_audio(rows, sr)
    for algo in ("winograd", "direct"):
        m.engine.set_option("enc0", algo)
        p, _, st = run_engine(m, rows, sr)
        e = np.abs(p - want).max(1)
        se = (np.abs(st - wst) / np.maximum(1, np.abs(wst))).max((0, 2))
        print(tag, algo, "probs: speech %.2e adv %.2e | state: speech %.2e adv %.2e" % (e[:40].max(), e[40:].max(), se[:40].max(), se[40:].max()),
              "worst adv:", names[int(np.argmax(se[40:]))])
