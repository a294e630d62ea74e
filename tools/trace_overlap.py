#!/usr/bin/env python3
"""Overlap of kernel / copy classes in a rocprofv3 --kernel-trace --memory-copy-trace CSV directory: how long each class is active,
pairwise overlap, and each class's mean duration -- to see whether an upload route runs BESIDE the compute or alternates with it."""
import csv
import glob
import os
import sys


def klass(name):
    for key, lab in (("gather", "gather"), ("scatter", "cut"), ("front_f43", "front"), ("front_lat", "front_lat"), ("rec_small", "rec"), ("rec_kernel", "rec"),
                     ("scan", "scan"), ("copyBuffer", "blit"), ("fillBuffer", "fill")):
        if key in name:
            return lab
    return "other"


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def inter(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main(d):
    ev = {}
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.setdefault(klass(r["Kernel_Name"]), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.setdefault("copy_" + r.get("Direction", "")[12:].lower(), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if not ev:
        print("no rows")
        return
    lo = min(s for v in ev.values() for s, _ in v)
    hi = max(e for v in ev.values() for _, e in v)
    # one report per burst of upload kernels (bursts are separated by > 50 ms without one): the legs of a corpus run
    ups = sorted(ev.get("gather", []) or ev.get("cut", []))
    bursts = []
    for s, e in ups:
        if bursts and s - bursts[-1][1] < 50e6:
            bursts[-1][1] = max(bursts[-1][1], e)
        else:
            bursts.append([s, e])
    if not bursts:
        bursts = [[lo + (hi - lo) * 0.3, lo + (hi - lo) * 0.9]]
    for a, b in bursts:
        report(ev, a, b, lo, hi)


def report(ev, a, b, lo, hi):
    un = {}
    print(f"== window {(a - lo) / 1e6:.1f} .. {(b - lo) / 1e6:.1f} ms ({(b - a) / 1e6:.1f} ms) of {(hi - lo) / 1e6:.1f} ms")
    for k, v in sorted(ev.items()):
        w = [(max(s, a), min(e, b)) for s, e in v if e > a and s < b]
        un[k] = union(w)
        n = len(w)
        if n:
            print(f"{k:14s} n {n:6d}  active {length(un[k]) / (b - a):6.3f} of the window   mean {sum(e - s for s, e in w) / n / 1e3:9.1f} us")
    ks = [k for k in un if un[k]]
    for i, x in enumerate(ks):
        for y in ks[i + 1:]:
            o = inter(un[x], un[y])
            if o:
                print(f"  {x} & {y}: both active {o / (b - a):.3f} of the window ({o / max(1, min(length(un[x]), length(un[y]))):.2f} of the shorter)")


if __name__ == "__main__":
    main(sys.argv[1])
