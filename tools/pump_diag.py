#!/usr/bin/env python3
"""Where a tick of the native pump spends its time (bench leg stream_host): vad_pump_play over a grid of (parts, depth, source threads;
silent sources = device side only), optionally after binding the host to the GPU's NUMA node.  `--trace` plays one short run (for
rocprofv3 --kernel-trace --memory-copy-trace), `--analyze DIR` prints the copy / kernel timeline of a few steady-state ticks from the
CSV files rocprofv3 left in DIR."""
import argparse
import csv
import glob
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def analyze(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40], r.get("Queue_Id", "")))
    for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "")[:24], r.get("Bytes", r.get("Size", ""))))
    rows.sort()
    if not rows:
        print("no trace rows under", d)
        return
    mid = len(rows) // 2
    t0 = rows[mid][0]
    print("steady-state window (us relative to the first row shown): start  end  dur  what")
    for s, e, what, extra in rows[mid: mid + 40]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {what}  {extra}")
    copies = [(s, e) for s, e, w, _ in rows if w.startswith("C ") and e - s > 5000]
    if len(copies) > 20:
        c = copies[len(copies) // 4: 3 * len(copies) // 4]
        busy = sum(e - s for s, e in c)
        span = c[-1][1] - c[0][0]
        print(f"copy engines busy {busy / span:.3f} of the window; mean copy {busy / len(c) / 1e3:.1f} us; mean gap {(span - busy) / len(c) / 1e3:.1f} us")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sr", type=int, default=16000)
    ap.add_argument("--streams", type=int, default=8192)
    ap.add_argument("--ticks", type=int, default=1500)
    ap.add_argument("--bind", action="store_true")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--analyze")
    ap.add_argument("--parts", type=int, nargs="*", default=[1, 2])
    ap.add_argument("--depths", type=int, nargs="*", default=[1, 2, 3])
    ap.add_argument("--fills", type=int, nargs="*", default=[-1, 4, 8])
    a = ap.parse_args()
    if a.analyze:
        return analyze(a.analyze)
    import numpy as np
    import torch
    import bench
    from silero_vad_amd import Engine, StreamPump, _lib
    node = _lib.lib().vad_bind_host_to_device(0) if a.bind else None
    n = 512 if a.sr == 16000 else 256
    rows = np.ascontiguousarray(bench.fixture_rows_i16(a.sr, a.streams, 32 * n))
    eng = Engine(device=0)
    dev = torch.device("cuda", 0)
    link = bench.h2d_rate_GBps(dev)
    ceil = link * 1e9 / (n * 2)
    print(json.dumps({"sr": a.sr, "streams": a.streams, "numa_node_bound": node, "h2d_GBps": round(link, 2), "int16_ceiling": round(ceil)}))
    if a.trace:
        pump = StreamPump(eng, a.sr, streams=a.streams, parts=a.parts[0], ring_slots=4)
        pump.play(rows, 300, depth=a.depths[0], fill_threads=a.fills[0])
        _, st = pump.play(rows, 400, first_tick=300, depth=a.depths[0], fill_threads=a.fills[0])
        print(json.dumps(st))
        pump.close()
        return
    for parts in a.parts:
        pump = StreamPump(eng, a.sr, streams=a.streams, parts=parts, ring_slots=4)
        t0 = 0
        pump.play(rows, 600, depth=2, fill_threads=-1)
        t0 += 600
        for fill in a.fills:
            for depth in a.depths:
                _, st = pump.play(rows, a.ticks, first_tick=t0, depth=depth, fill_threads=fill)
                t0 += a.ticks
                rate = a.streams * a.ticks / (st["wall_ms"] / 1e3)
                print(f"parts {parts} fill {fill:3d} depth {depth}: {st['wall_ms'] / a.ticks * 1e3:7.1f} us/tick  {rate / 1e6:6.1f} M chunks/s  of link {rate / ceil:.3f}  "
                      f"tick p50 {st['tick_ms_p50'] * 1e3:6.1f} p95 {st['tick_ms_p95'] * 1e3:6.1f} max {st['tick_ms_max'] * 1e3:7.1f} us   host/tick: fill {st['fill_ms_mean'] * 1e3:6.1f} submit "
                      f"{st['submit_ms_mean'] * 1e3:5.1f} blocked {st['wait_ms_mean'] * 1e3:6.1f}", flush=True)
        pump.close()


if __name__ == "__main__":
    main()
