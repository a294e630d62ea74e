#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/prof/<pass>/...) into small tracked files
under profiles/: per-kernel duration stats and per-kernel PMC means.

    python tools/summarize_prof.py gpurun_out/prof profiles/r01 [sr streams chunks]

The optional workload triple is recorded in the JSON so that bench.py can match a PMC summary to
the workload it is running (roofline.traffic).
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path


def short(name):
    for key in ("front_f43_kernel", "front_lat_kernel", "front_b9_kernel", "rec_b9_kernel", "front_wino_kernel", "front_kernel", "rec_kernel", "rec_skew_kernel", "ref_forward_kernel",
                "gather_rows_kernel", "scan_kernel", "unpack_gx", "exact_fix_kernel", "carry_absent_kernel"):
        if key in name:
            return name[name.index(key):].split("(")[0]
    return None


def main(src, dst_prefix, workload=(16000, 4096, 256)):
    src = Path(src)
    out = {"workload": {"sr": int(workload[0]), "streams": int(workload[1]), "chunks": int(workload[2])},
           "kernel_trace": {}, "pmc": {}}
    lines = ["# rocprofv3 summary (" + dst_prefix + ")", ""]
    for f in glob.glob(str(src / "trace" / "**" / "*_kernel_stats.csv"), recursive=True):
        lines += ["## --kernel-trace --stats (engine kernels only)", "",
                  "| kernel | calls | avg ms | min ms | max ms | % of GPU time |", "|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if not k:
                continue
            out["kernel_trace"][k] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                      "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6,
                                      "pct": float(r["Percentage"])}
            t = out["kernel_trace"][k]
            lines.append(f"| {k} | {t['calls']} | {t['avg_ms']:.4f} | {t['min_ms']:.4f} | {t['max_ms']:.4f} | {t['pct']:.2f} |")
        lines.append("")
    # per-dispatch durations: statistics over the LAST `tail` dispatches of each kernel (the bench's timed steps,
    # after its clock ramp) -- what has to agree with the hipEvent time bench.py prints
    tail = 20
    for f in glob.glob(str(src / "trace" / "**" / "*_kernel_trace.csv"), recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                per[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        lines += [f"## kernel trace, last {tail} dispatches of each kernel (the timed steps after the ramp)", "",
                  "| kernel | dispatches | median ms | min ms | mean ms | max ms |", "|---|---|---|---|---|---|"]
        for k, v in per.items():
            v.sort()
            d = sorted((e - s0) / 1e6 for s0, e in v[-tail:])
            st = {"dispatches": len(d), "median_ms": d[len(d) // 2], "min_ms": d[0], "mean_ms": sum(d) / len(d),
                  "max_ms": d[-1]}
            out.setdefault("kernel_trace_tail", {})[k] = st
            lines.append(f"| {k} | {st['dispatches']} | {st['median_ms']:.4f} | {st['min_ms']:.4f} | {st['mean_ms']:.4f} | {st['max_ms']:.4f} |")
        lines.append("")
    for f in sorted(glob.glob(str(src / "*" / "**" / "*_counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[k] = {x: r[x] for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                         "Scratch_Size", "Grid_Size", "Workgroup_Size")}
        for (k, c), v in sorted(agg.items()):
            out["pmc"].setdefault(k, {})[c] = sum(v) / len(v)
        for k, m in meta.items():
            out.setdefault("dispatch", {})[k] = m
    if out["pmc"]:
        lines += ["## --pmc passes (mean per dispatch)", "", "| kernel | counter | mean |", "|---|---|---|"]
        for k, cs in out["pmc"].items():
            for c, v in cs.items():
                lines.append(f"| {k} | {c} | {v:.6g} |")
        lines.append("")
    for log in sorted(src.glob("*.log")):
        for line in open(log, errors="ignore"):
            if line.startswith('{"metric"'):
                d = json.loads(line)
                out.setdefault("bench_lines", {})[log.stem] = {k: d[k] for k in ("value", "ms_per_step", "kernel_ms", "dtype") if k in d}
    if "bench_lines" in out:
        lines += ["## bench.py line printed inside each profiled run (hipEvent timing, same process)", ""]
        for k, v in out["bench_lines"].items():
            lines.append(f"- {k}: {json.dumps(v)}")
        lines.append("")
    Path(dst_prefix + "_summary.md").write_text("\n".join(lines))
    Path(dst_prefix + "_summary.json").write_text(json.dumps(out, indent=1))
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:6] if len(sys.argv) >= 6 else (16000, 4096, 256))
