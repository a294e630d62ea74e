#!/usr/bin/env python3
"""Bring-up diagnostics on a real MI355X: stage-by-stage errors of the HIP path against the CPU
oracle, written to gpurun_out/diag.json (richer than a pass/fail test when something is off)."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    import __graft_entry__ as ge
    ge.build()
    from oracle import Oracle
    from oracle.weights import read_container
    from silero_vad_amd import _lib, load_silero_vad

    out = {"device": torch.cuda.get_device_name(0)}
    model = load_silero_vad(0)
    eng, dev = model.engine, model.device
    o = Oracle()
    w = read_container(_lib.WEIGHTS_PATH.read_bytes())
    for tag, sr in (("16k", 16000), ("8k", 8000)):
        n = 512 if sr == 16000 else 256
        wav = np.load(ROOT / "tests/golden" / f"audio_{tag}.npz")["pcm"].astype(np.float32) / 32768.0
        B, T = 19, 6
        rows = np.stack([np.roll(wav, -b * 5003)[40 * n: 40 * n + T * n] for b in range(B)])
        want, wctx, wst = o.forward_audio(rows, sr)
        res = {}
        for impl in ("reference", "mfma"):
            eng.set_option("impl", impl)
            x = torch.from_numpy(rows).to(dev)
            ctx = torch.zeros((B, n // 8), device=dev)
            st = torch.zeros((2, B, 128), device=dev)
            t0 = time.time()
            p = eng.forward_audio(x, sr, ctx, st)
            torch.cuda.synchronize()
            res[impl] = {"prob_err": float(np.abs(p.cpu().numpy() - want).max()),
                         "state_err": float(np.abs(st.cpu().numpy() - wst).max()),
                         "ctx_equal": bool(np.array_equal(ctx.cpu().numpy(), wctx)),
                         "nan": bool(torch.isnan(p).any().item()), "ms": (time.time() - t0) * 1e3,
                         "first_probs": p[0, :4].cpu().tolist(), "want_first": want[0, :4].tolist()}
        eng.set_option("impl", "mfma")
        # frontend in isolation
        x = torch.from_numpy(rows).to(dev)
        gx = eng.debug_frontend(x, sr, torch.zeros((B, n // 8), device=dev)).cpu().numpy()
        pre = "_model" if sr == 16000 else "_model_8k"
        w_ih = w[pre + ".decoder.rnn.weight_ih"].astype(np.float64)
        bias = (w[pre + ".decoder.rnn.bias_ih"] + w[pre + ".decoder.rnn.bias_hh"]).astype(np.float64)
        C = n // 8
        errs = []
        for t in range(T):
            prev = rows[:, t * n - C: t * n] if t else np.zeros((B, C), np.float32)
            x1 = np.concatenate([prev, rows[:, t * n:(t + 1) * n]], 1)
            _, _, stg = o.step(x1, np.zeros((2, B, 128), np.float32), sr, stages=True)
            ref = stg["enc3"][:, :, 0].astype(np.float64) @ w_ih.T + bias
            errs.append([float(np.abs(gx[:, t] - ref).max()), float(np.abs(ref).max())])
        res["gx_err_scale_per_t"] = errs
        res["gx_err_by_gate"] = [float(np.abs(gx[:, 1, 128 * q:128 * q + 128] - ref_q).max())
                                 for q, ref_q in enumerate(np.split(ref, 4, axis=1))] if T > 1 else None
        out[tag] = res
        print(tag, json.dumps(res)[:1500], flush=True)
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "diag.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
