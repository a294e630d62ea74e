#!/usr/bin/env python3
"""Which term puts the engine's carried cell state further from float64 than the oracle's?  B streams of speech,
T steps from a random carried state: (1) gate pre-activations gx = W_ih enc(stft(x)) + b of the engine / of the oracle's fp32 encoder
output pushed through W_ih in fp32 / of the test build's other encoder-0 forms, each against float64; (2) the recurrence alone: the
float64 LSTM cell over the ENGINE's own gx against the engine's final state (what the recurrence kernel adds), and over the float64 gx
(what the frontend's gx error costs in the state)."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    import test_gpu_parity as T
    from oracle import Oracle
    from oracle.weights import read_container
    from silero_vad_amd import Engine, HipSileroVAD, _lib, load_silero_vad
    out = {}
    for tag, sr in (("16k", 16000), ("8k", 8000)):
        n = 512 if sr == 16000 else 256
        C = n // 8
        wav = np.load(ROOT / "tests" / "golden" / f"audio_{tag}.npz")["pcm"].astype(np.float32) / 32768.0
        B, steps = 1023, 4
        rows = T.rolled_rows(wav, B, steps * n, 4001)
        rng = np.random.default_rng(91)
        st0 = (0.3 * rng.standard_normal((2, B, 128))).astype(np.float32)
        model = load_silero_vad(device=0)
        dev = model.device
        f64 = T._F64Net(sr, dev)
        x = torch.from_numpy(rows).to(dev)
        # float64 gx, all steps
        xx = torch.cat([torch.zeros((B, C), dtype=torch.float64, device=dev), x.double()], 1)
        x1 = xx.unfold(1, n + C, n).reshape(B * steps, n + C)
        gx64 = f64.features(x1).reshape(B, steps, 512).cpu().numpy()
        forms = {"engine (F(4,3), product)": model}
        ab = HipSileroVAD(engine=Engine(0, library=_lib.lib_ab()))
        res = {}
        for name, m, opt in (("engine F(4,3) (product)", model, None), ("test build: F(2,3)", ab, "winograd2"), ("test build: direct", ab, "direct")):
            if opt:
                m.engine.set_option("enc0", opt)
            gx = m.engine.debug_frontend(x, sr, torch.zeros((B, C), device=dev)).cpu().numpy()
            if opt:
                m.engine.set_option("enc0", "winograd")
            e = np.abs(gx - gx64)
            res[name] = {"gx_abs_err_max": float(e.max()), "gx_abs_err_rms": float(np.sqrt((e ** 2).mean())), "gx_abs_max": float(np.abs(gx64).max())}
            if name.startswith("engine"):
                gx_eng = gx
        # the oracle's fp32 encoder output through W_ih in fp32 (numpy BLAS order)
        orc = Oracle()
        w = read_container(_lib.WEIGHTS_PATH.read_bytes())
        pre = "_model" if sr == 16000 else "_model_8k"
        w_ih = w[pre + ".decoder.rnn.weight_ih"]
        bias = (w[pre + ".decoder.rnn.bias_ih"] + w[pre + ".decoder.rnn.bias_hh"])
        gxo = np.zeros((B, steps, 512), np.float32)
        for t in range(steps):
            prev = rows[:, t * n - C: t * n] if t else np.zeros((B, C), np.float32)
            _, _, st = orc.step(np.concatenate([prev, rows[:, t * n:(t + 1) * n]], 1), np.zeros((2, B, 128), np.float32), sr, stages=True)
            gxo[:, t] = st["enc3"][:, :, 0] @ w_ih.T + bias
        e = np.abs(gxo - gx64)
        res["oracle encoder (fp32) -> W_ih in fp32 (BLAS)"] = {"gx_abs_err_max": float(e.max()), "gx_abs_err_rms": float(np.sqrt((e ** 2).mean()))}
        # the recurrence alone
        W_hh = w[pre + ".decoder.rnn.weight_hh"].astype(np.float64)

        def cell64(gx):
            h, c = st0[0].astype(np.float64), st0[1].astype(np.float64)
            for t in range(steps):
                g = gx[:, t].astype(np.float64) + h @ W_hh.T
                sg = lambda v: 1.0 / (1.0 + np.exp(-v))
                i, f, gg, o = sg(g[:, :128]), sg(g[:, 128:256]), np.tanh(g[:, 256:384]), sg(g[:, 384:])
                c = f * c + i * gg
                h = o * np.tanh(c)
            return np.stack([h, c])
        s_true = cell64(gx64)
        s_from_engine_gx = cell64(gx_eng)
        s_from_oracle_gx = cell64(gxo)
        ctx = torch.zeros((B, C), device=dev)
        st = torch.from_numpy(st0).to(dev)
        model.engine.forward_audio(x, sr, ctx, st)
        s_eng = st.cpu().numpy()
        _, _, s_orc = orc.forward_audio(rows, sr, state=st0)
        rel = lambda a, b: float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())
        res["state"] = {"engine vs float64": rel(s_eng, s_true), "oracle vs float64": rel(s_orc, s_true),
                        "float64 cell over the ENGINE's gx vs float64 (= cost of the frontend's gx error)": rel(s_from_engine_gx, s_true),
                        "float64 cell over the oracle-path gx vs float64": rel(s_from_oracle_gx, s_true),
                        "engine vs float64 cell over the engine's own gx (= what the recurrence kernel adds)": rel(s_eng, s_from_engine_gx)}
        out[tag] = res
        print(tag, json.dumps(res, indent=1))
    Path("gpurun_out").mkdir(exist_ok=True)
    json.dump(out, open("gpurun_out/state_term_diag.json", "w"), indent=1)


if __name__ == "__main__":
    with torch.no_grad():
        main()
