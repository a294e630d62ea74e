#!/usr/bin/env python3
"""Copy timeline of the refill scheduler's window feed from a rocprofv3 --memory-copy-trace run (tools/r06_refill_window_diag.sh): the LAST
run of long H2D copies is the window feed's -- durations, the idle gaps between consecutive copies, the link's busy fraction."""
import csv
import glob
import os
import sys

d = sys.argv[1]
long_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1e6
copies = []
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]))
copies.sort()
w = [c for c in copies if c[1] - c[0] >= long_ns and "HOST_TO_DEVICE" in c[2]]
print("long H2D copies", len(w), "of", len(copies))
runs, cur = [], [w[0]]
for a, b in zip(w[:-1], w[1:]):
    if b[0] - a[1] > 100e6:                       # a pause of more than 100 ms separates two legs
        runs.append(cur)
        cur = []
    cur.append(b)
runs.append(cur)
for i, run in enumerate(runs):
    t0, t1 = run[0][0], run[-1][1]
    busy = sum(e - s for s, e, _ in run)
    gaps = sorted((run[k + 1][0] - run[k][1]) / 1e3 for k in range(len(run) - 1)) or [0]
    dur = sorted((e - s) / 1e6 for s, e, _ in run)
    print(f"run {i}: {len(run)} copies over {(t1 - t0) / 1e6:.1f} ms, link busy {busy / (t1 - t0):.3f}; copy ms p10 {dur[len(dur) // 10]:.2f} p50 {dur[len(dur) // 2]:.2f} "
          f"p90 {dur[9 * len(dur) // 10]:.2f}; gap us p50 {gaps[len(gaps) // 2]:.0f} p90 {gaps[9 * len(gaps) // 10]:.0f} max {gaps[-1]:.0f}; gaps > 1 ms: "
          f"{sum(1 for g in gaps if g > 1000)} totalling {sum(g for g in gaps if g > 1000) / 1e3:.1f} ms")
run = runs[-1]
t0 = run[0][0]
print("-- the last run's copies (ms from its start): start, duration ms, gap before us")
prev = None
for s, e, _ in run[:int(os.environ.get("ROWS", 60))]:
    print(f"{(s - t0) / 1e6:9.2f} {(e - s) / 1e6:7.2f} {0 if prev is None else (s - prev) / 1e3:9.0f}")
    prev = e
