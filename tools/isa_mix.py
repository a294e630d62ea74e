#!/usr/bin/env python3
"""Static instruction mix of the device code of a HIP translation unit (the frontend kernels are straight-line
code, so the static count of a kernel is its per-wave dynamic count):

    python tools/isa_mix.py silero_vad_amd/csrc/kernel_front.hip [extra hipcc flags...]
"""
import collections
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def mix(src, flags=()):
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", str(ROOT / "include"),
                          *flags, "--cuda-device-only", "-S", str(src), "-o", "-"],
                         check=True, capture_output=True, text=True).stdout
    out = {}
    cur = None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = collections.Counter()
            out[m.group(1)] = cur
            continue
        if cur is None:
            continue
        l = line.strip()
        if l.startswith(".Lfunc_end"):
            cur = None
            continue
        if not l or l[0] in ".;/" or l.endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"):
            cur["mfma"] += 1
        elif op.startswith("v_"):
            cur["valu"] += 1
            cur["v:" + op] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
            cur["d:" + op] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
            if op in ("s_waitcnt", "s_barrier", "s_nop"):
                cur["s:" + op] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
            cur["m:" + op] += 1
    return out


if __name__ == "__main__":
    for name, c in mix(sys.argv[1], sys.argv[2:]).items():
        print(name)
        print("   ", {k: v for k, v in c.items() if ":" not in k})
        for pre in ("v:", "d:", "m:", "s:"):
            top = sorted(((v, k[2:]) for k, v in c.items() if k.startswith(pre)), reverse=True)[:18]
            print("   ", pre, " ".join(f"{k}={v}" for v, k in top))
