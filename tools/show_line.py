#!/usr/bin/env python3
"""Readable digest of a bench line (tools/r04_round.sh)."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])


def show(k, v, ind=0, depth=4):
    if isinstance(v, dict) and depth > 0:
        print(" " * ind + str(k) + ":")
        for a, b in v.items():
            show(a, b, ind + 2, depth - 1)
    else:
        print(" " * ind + f"{k}: {str(v)[:170]}")


for k in ("value", "ms_per_step", "kernel_ms", "parity"):
    show(k, d.get(k))
if "roofline" in d:
    show("roofline.frac", d["roofline"]["frac"])
    show("roofline.traffic", d["roofline"].get("traffic"))
    show("rec", d["roofline"].get("rec_kernel"))
show("other_arithmetic", d.get("other_arithmetic"), depth=3)
oc = d.get("other_configs") or {}
for name, leg in oc.items():
    if isinstance(leg, dict) and "legs" in leg:
        leg = dict(leg)
        leg["legs"] = {k: {a: v.get(a) for a in ("value", "wall_s", "fraction_of_pcie_ceiling", "host_upload_call_ms")} for k, v in leg["legs"].items()}
    show(name, leg, depth=3)
cb = d.get("cpu_baseline") or {}
show("cpu_baseline.value", cb.get("value"))
show("cpu_baseline.runs", {k: (v.get("chunks_per_s"), v.get("wall_ms")) for k, v in (cb.get("runs") or {}).items()})
