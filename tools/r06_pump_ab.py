"""Same-lease A/B of the pump's device batch buffers (SILERO_VAD_AMD_PUMP_BUFFERS = 2 | 3) x ticks in flight (2 | 3) x tick form
(lock step / masked full rows / compact), alternating, `reps` passes of `ticks` ticks each: ticks per second, median and spread.
    python tools/r06_pump_ab.py [ticks=2000] [reps=5]     (GPU box; profiles/r06_pump_three_buffers.md)"""
import json
import os
import statistics
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from silero_vad_amd import Engine, StreamPump, _lib  # noqa: E402

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
_lib.lib().vad_bind_host_to_device(0)
eng = Engine(device=0)
out = {}
for sr in (16000, 8000):
    n = 512 if sr == 16000 else 256
    cap = 8192
    rows = np.ascontiguousarray(bench.fixture_rows_i16(sr, cap, 32 * n))
    pat = bench.gap_flags(256, cap, 41, 0.10)
    forms = {"lockstep": (None, False), "masked": (pat, False), "compact": (pat, True)}
    pumps = {}
    for nb in (2, 3):
        os.environ["SILERO_VAD_AMD_PUMP_BUFFERS"] = str(nb)
        pumps[nb] = StreamPump(eng, sr, streams=cap, parts=1, ring_slots=4)
    t0 = {2: 0, 3: 0}
    res = {}
    for nb in (2, 3):
        pumps[nb].play(rows, 600, first_tick=0, depth=2)
        t0[nb] = 600
    for rep in range(reps):
        for form, (p, c) in forms.items():
            for depth in (2, 3):
                for nb in (2, 3):
                    _, st = pumps[nb].play(rows, ticks, first_tick=t0[nb], depth=depth, pattern=p, compact=c)
                    t0[nb] += ticks
                    res.setdefault((form, depth, nb), []).append(st["chunks"] / st["wall_ms"] * 1e3 / 1e6)
    for k, v in sorted(res.items()):
        out[f"{sr // 1000}k {k[0]} depth{k[1]} nb{k[2]}"] = {"median_Mchunks_s": round(statistics.median(v), 2), "min": round(min(v), 2), "max": round(max(v), 2)}
    for nb in (2, 3):
        pumps[nb].close()
for k, v in out.items():
    print(k, v)
print(json.dumps(out))
