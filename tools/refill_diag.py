#!/usr/bin/env python3
"""Bring-up diagnostic (GPU box): where a refill slab's time goes, phase by phase, with synchronisation between phases.
    python tools/refill_diag.py [recordings]"""
import sys, time, ctypes
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import load_silero_vad
from silero_vad_amd.streams import RefillPlan
from silero_vad_amd._lib import lib

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sr, n, esz = 16000, 512, 2
model = load_silero_vad(device=0)
eng, dev = model.engine, model.device
rng = np.random.default_rng(101)
base_len = 8 << 20
base_i = torch.from_numpy((rng.standard_normal(base_len) * 1000).astype(np.int16))
lens = rng.integers(20 * sr, 40 * sr, size=R)
offs = rng.integers(0, base_len - 40 * sr, size=R)
audios = [base_i[o:o + m] for o, m in zip(offs, lens)]
plan = RefillPlan([int(m) for m in lens], max(64, R // 2), 64, n)
B, S, width = plan.slots, plan.slab_chunks, plan.slab_chunks * n
ptr0 = np.array([a.data_ptr() for a in audios], dtype=np.uint64)
host = torch.empty((B, width), dtype=torch.int16, pin_memory=True)
d = torch.empty((B, width), dtype=torch.int16, device=dev)
ctx = torch.zeros((B, n // 8), device=dev)
state = torch.zeros((2, B, 128), device=dev)
out_flat = torch.zeros(int(sum(plan.n_chunks(i) for i in range(R))) + 1, device=dev)
tm = {k: 0.0 for k in ("copy", "h2d", "fill", "forward", "put")}
def sync():
    torch.cuda.synchronize()
    return time.perf_counter()
for k in range(min(12, len(plan.slab_arrays))):
    e = plan.slab_arrays[k]
    sl, rec, at, take = e[:, 0], e[:, 1], e[:, 2], e[:, 3]
    rows = np.zeros(B, dtype=np.uint64); ln = np.zeros(B, dtype=np.int64)
    rows[sl] = ptr0[rec] + (at * esz).astype(np.uint64); ln[sl] = take
    rs = torch.from_numpy(np.ascontiguousarray(sl[e[:, 4] != 0])).to(dev)
    idx = torch.randint(0, out_flat.numel(), (B * S,), device=dev)
    t0 = sync()
    lib().vad_stage_rows(rows.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p)), ln.ctypes.data_as(ctypes.POINTER(ctypes.c_long)), B, width, esz, host.data_ptr(), 0)
    t1 = sync(); d.copy_(host, non_blocking=True)
    t2 = sync()
    if rs.numel():
        ctx.index_fill_(0, rs, 0.0); state.index_fill_(1, rs, 0.0)
    t3 = sync(); probs = eng.forward_audio(d, sr, ctx, state)
    t4 = sync(); out_flat.index_put_((idx,), probs.reshape(-1))
    t5 = sync()
    if k >= 2:
        for name, a, b in (("copy", t0, t1), ("h2d", t1, t2), ("fill", t2, t3), ("forward", t3, t4), ("put", t4, t5)):
            tm[name] += b - a
cnt = min(12, len(plan.slab_arrays)) - 2
print(f"R={R} slots={B} slab={S}: per slab ms:", {k: round(v / cnt * 1e3, 3) for k, v in tm.items()})
