#!/usr/bin/env python3
"""Bring-up: where do the split frontend's bad chunks sit, and which gx rows are wrong?"""
import json, sys, os
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from silero_vad_amd import Engine
dev = torch.device("cuda", 0)
eng = Engine(0)
wav = torch.from_numpy(np.load(ROOT / "tests/golden/audio_16k.npz")["pcm"].astype(np.float32) / 32768.0).to(dev)
sr, n = 16000, 512
B, T = 4096, 24
idx = (torch.arange(B, device=dev)[:, None] * 7919 + torch.arange(T * n, device=dev)[None]) % len(wav)
x = wav[idx].contiguous()
ctx = torch.zeros((B, 64), device=dev)
def gx(fp):
    eng.set_option("precision_front", fp)
    g = eng.debug_frontend(x, sr, ctx)
    torch.cuda.synchronize()
    return g
ref = gx("fp32").clone()
nwg = B // 16 * T // 4
trace = torch.zeros((nwg * 4, 4), dtype=torch.int64, device=dev)
eng.set_option("trace_ptr", hex(trace.data_ptr()))
scale = float(ref.abs().max())
for rep in range(4):
    g = gx("f16x3")
    d = (g - ref).abs()
    badmask = d > 2e-4 * scale
    chunks = badmask.any(dim=2).nonzero()
    print(json.dumps({"rep": rep, "n_bad_chunks": int(len(chunks)), "max": float(d.max())}))
    tiles = {}
    for b, t in chunks.tolist()[:4000]:
        st, j = b // 16, b % 16
        wt = st * T + t
        tiles.setdefault((wt // 4, wt % 4), []).append(j)
    for (wg, wave), js in list(tiles.items())[:12]:
        b0 = (wg * 4 + wave) // T * 16
        t = (wg * 4 + wave) % T
        rows = badmask[b0 + js[0], t].nonzero().flatten().tolist()
        dd = d[b0 + js[0], t]
        print(json.dumps({"wg": wg, "wave": wave, "cols": sorted(js), "t": t, "n_rows": len(rows),
                          "rows_head": rows[:24], "gates": sorted({r // 128 for r in rows}),
                          "g_of_rows": sorted({(r % 16) // 4 for r in rows}),
                          "blocks": sorted({(r % 128) // 16 for r in rows})[:8], "maxd": float(dd.max())}))
    print(json.dumps({"n_bad_tiles": len(tiles), "waves_hist": np.bincount([w for (_, w) in tiles], minlength=4).tolist()}))
    tr = trace.cpu().numpy()
    bad = np.zeros(nwg * 4, bool)
    for (wg, wave) in tiles:
        bad[wg * 4 + wave] = True
    import collections
    for col, name in ((0, "LDS_ALLOC"), (3, "GPR_ALLOC")):
        c_all = collections.Counter((tr[:, col] & 0xffffffff).tolist())
        c_bad = collections.Counter((tr[bad, col] & 0xffffffff).tolist())
        print(name, {hex(k): (c_bad.get(k, 0), v) for k, v in sorted(c_all.items())})
    hw = tr[:, 1] & 0xffffffff
    for nm, sh, msk in (("wave_id", 0, 0xf), ("simd", 4, 0x3), ("cu", 8, 0xf), ("sh", 12, 1), ("se", 13, 0x7)):
        f = (hw >> sh) & msk
        print(nm, {int(k): (int(bad[f == k].sum()), int((f == k).sum())) for k in np.unique(f)})
