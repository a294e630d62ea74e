#!/usr/bin/env python3
"""Carried (h, c) across an abrupt drop to EXACT zeros, against float64 (profiles/r06_state_rows.md).

Rows: 1 025 streams of the reference's speech fixture; from a per-stream sample offset inside chunk 2 the signal is exactly zero for
`run` chunks (a muted microphone, a DTX gap, the zero padding behind a recording's end), then speech resumes.  For each encoder-0 form
of the test build (winograd = the product's F(4,3), winograd2, direct = tap by tap) and for the CPU oracle: the state after EVERY chunk
against a float64 evaluation of the network, worst entry over streams and units (relative to max(1, |x|)).

    gpurun -- 'python tools/zero_run_study.py > gpurun_out/zero_run_study.json'
"""
import json
import os
import sys
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
    from silero_vad_amd import Engine, HipSileroVAD, _lib
    from test_gpu_parity import _F64Net, rolled_rows
    model = HipSileroVAD(engine=Engine(0, library=_lib.lib_ab()))
    eng, dev = model.engine, model.device
    orc = Oracle()
    out = {}
    for tag, sr in (("16k", 16000), ("8k", 8000)):
        n = 512 if sr == 16000 else 256
        wav = np.load(ROOT / "tests" / "golden" / f"audio_{tag}.npz")["pcm"].astype(np.float32) / 32768.0
        B, T = 1025, 9
        only = os.environ.get("ZERO_RUN_ONLY")                 # "dither": just the near-silence rows
        cases = [(0.5, None, 0), (2.5, None, 0), (2.0, 1, 0), (2.0, 7, 0), (2.0, n // 8, 0), (2.0, n // 2, 0), (2.0, n - 1, 0),
                 (2.0, 1, 1), (2.0, 7, 1), (2.0, n - 1, 1), (2.0, 1, 8)]
        for run, where, lsb in cases:
            if only == "dither" and not lsb:
                continue
            rows = rolled_rows(wav, B, T * n, 4001)
            # first zero sample: anywhere inside chunk 2 (stream 0: its first sample), or the SAME offset `where` in every stream (the
            # shape of a batch of recordings that end one / seven samples into their last chunk)
            z0 = 2 * n + ((np.arange(B) * 37) % n if where is None else np.full(B, where))
            z1 = z0 + int(run * n)
            rng = np.random.default_rng(3)
            for b in range(B):
                # exact zeros, or NEAR-silence: uniform noise of +-lsb int16 steps (dither, comfort noise) -- not an exact zero, so these
                # rows stay on the fp32 chains
                rows[b, z0[b]:z1[b]] = 0.0 if not lsb else rng.integers(-lsb, lsb + 1, size=z1[b] - z0[b]).astype(np.float32) / 32768.0
            x = torch.from_numpy(rows).to(dev)
            f64 = _F64Net(sr, dev)
            res = {}
            for t in range(1, T + 1):
                xt = x[:, :t * n].contiguous()
                _, s64 = f64.audio_forward(xt)
                s64 = s64.cpu().numpy()
                den = np.maximum(1.0, np.abs(s64))
                _, _, so = orc.forward_audio(rows[:, :t * n].copy(), sr)
                res.setdefault("oracle", []).append(float((np.abs(so - s64) / den).max()))
                for algo in ("winograd", "winograd2", "direct", "winograd+exact", "winograd+exact+silent"):
                    eng.set_option("enc0", algo.split("+")[0])
                    eng.set_option("exact_transitions", {1: "0", 2: "edges", 3: "1"}[len(algo.split("+"))])
                    try:
                        ctx = torch.zeros((B, n // 8), device=dev)
                        st = torch.zeros((2, B, 128), device=dev)
                        eng.forward_audio(xt, sr, ctx, st)
                        torch.cuda.synchronize()
                    finally:
                        eng.set_option("enc0", "winograd")
                        eng.set_option("exact_transitions", "1")
                    res.setdefault(algo, []).append(float((np.abs(st.cpu().numpy() - s64) / den).max()))
            out[f"{tag} {'zero' if not lsb else f'+-{lsb} LSB noise'} run of {run} chunks from {'anywhere' if where is None else 'sample ' + str(where)} inside chunk 2"] = {k: [float(f"{v:.3e}") for v in vs] for k, vs in res.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
