"""Soak of the native pump's loop (GPU box): the same long run -- `streams` live streams missing ~10 % of their ticks -- through full-row
masked ticks and through compact ticks, at 1, 2 and 3 ticks in flight with 1 ... 8 source threads: every run must give the SAME events
and the same final (h, c, context) of every stream, bit for bit (a race between sources, server loop and the copy / compute streams --
a slot rewritten too early, a buffer overwritten under a kernel -- would show as a difference).
    python tools/pump_soak.py [ticks=20000] [streams=1024]"""
import hashlib
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from silero_vad_amd import Engine, StreamPump  # noqa: E402

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sr, n = 16000, 512
rows = np.ascontiguousarray(bench.fixture_rows_i16(sr, cap, 64 * n))
pat = bench.gap_flags(509, cap, 7, 0.10)
eng = Engine(device=0)
ref = None
for compact in (False, True):
    for depth, threads in ((1, 1), (2, 3), (3, 8), (3, 2)):
        pump = StreamPump(eng, sr, streams=cap, parts=1, ring_slots=4)
        ev, st = pump.play(rows, ticks, depth=depth, fill_threads=threads, max_events=4_000_000, pattern=pat, compact=compact)
        state = np.concatenate([np.concatenate(pump.state(b)) for b in range(0, cap, max(1, cap // 64))])
        pump.close()
        key = (len(ev), hashlib.sha256(repr(sorted((s, k, v) for s, e in ev for k, v in e.items())).encode()).hexdigest()[:16],
               hashlib.sha256(state.tobytes()).hexdigest()[:16], st["chunks"])
        print(f"compact={compact} depth={depth} threads={threads}: events {key[0]} {key[1]} state {key[2]} chunks {key[3]} "
              f"wall {st['wall_ms'] / 1e3:.2f} s p95 {st['tick_ms_p95']:.3f} max {st['tick_ms_max']:.3f} ms", flush=True)
        if ref is None:
            ref = key
        if key != ref:
            raise SystemExit("MISMATCH against the first run")
print(f"soak OK: {ticks} ticks x {cap} streams, 8 runs identical")
