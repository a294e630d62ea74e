#!/usr/bin/env python3
"""Timeline of the refill scheduler from a rocprofv3 --kernel-trace run (tools/r06_refill_diag.sh): per kernel family the busy time,
for the gather kernel its durations and the idle gaps between consecutive launches."""
import csv
import glob
import os
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]))
rows.sort()
if not rows:
    sys.exit("no rows")
# the refill leg is the LAST long run of gather kernels: take the last 60 % of the gather launches
g = [(s, e) for s, e, n in rows if n.startswith("vad::(anonymous namespace)::gather_rows_kernel") or "gather_rows" in n]
print("gather launches", len(g))
tail = g[len(g) // 2:]
t0, t1 = tail[0][0], tail[-1][1]
dur = sorted((e - s) / 1e3 for s, e in tail)
gaps = sorted((tail[i + 1][0] - tail[i][1]) / 1e3 for i in range(len(tail) - 1))
print(f"window {(t1 - t0) / 1e6:.1f} ms, {len(tail)} gathers: duration us p10 {dur[len(dur) // 10]:.0f} p50 {dur[len(dur) // 2]:.0f} p90 {dur[9 * len(dur) // 10]:.0f}; "
      f"gap us p10 {gaps[len(gaps) // 10]:.0f} p50 {gaps[len(gaps) // 2]:.0f} p90 {gaps[9 * len(gaps) // 10]:.0f}; busy {sum(dur) * 1e3 / (t1 - t0):.3f}")
fam = {}
for s, e, n in rows:
    if s >= t0 and e <= t1:
        k = n.split("<")[0].split("(")[0][-40:]
        fam[k] = fam.get(k, 0) + (e - s)
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {k:42s} {v / 1e6:8.1f} ms  {v / (t1 - t0):.3f}")
mid = len(tail) // 2
w0 = tail[mid][0]
print("-- three slabs in the middle (us from the first gather's start): start end dur kernel")
for s, e, n in rows:
    if w0 <= s <= tail[mid + 3][0]:
        print(f"{(s - w0) / 1e3:9.1f} {(e - w0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}")
