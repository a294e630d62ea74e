#!/usr/bin/env python3
"""Where does a page-locked arena land, and what do the two ways of reading it over PCIe make of it?  Allocates pinned arenas one after
the other (earlier ones stay alive), prints the NUMA node(s) of each (/proc/self/numa_maps) and the rate of (a) one H2D DMA of the arena
and (b) the gather kernel reading it (vad_upload_rows how = 1: 1 MiB rows)."""
import ctypes
import re
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from silero_vad_amd import Engine, _lib


def nodes_of(ptr, nbytes):
    out = {}
    for line in open("/proc/self/numa_maps"):
        f = line.split()
        a = int(f[0], 16)
        if a <= ptr < a + (1 << 40):
            cand = (a, line)
            if a <= ptr:
                best = cand if "best" not in locals() or a > best[0] else best
    if "best" not in locals():
        return "?"
    return " ".join(x for x in best[1].split() if re.match(r"N\d+=|kernelpagesize|bind|prefer|default|interleave", x))


def main():
    bind = "--nobind" not in sys.argv
    node = _lib.lib().vad_bind_host_to_device(0) if bind else None
    print("bound to node", node, flush=True)
    eng = Engine(0)
    dev = torch.device("cuda", 0)
    nbytes = 2 << 30
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    keep = []
    for trial in range(6):
        t = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        t.zero_()
        keep.append(t)
        torch.cuda.synchronize()
        best = 0
        for _ in range(3):
            t0 = time.perf_counter(); dst.copy_(t, non_blocking=True); torch.cuda.synchronize(); best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
        n, width = 2048, 1 << 19                               # 2048 rows of 1 MiB (int16 elements)
        rows = (ctypes.c_void_p * n)(*[t.data_ptr() + i * width * 2 for i in range(n)])
        lens = (ctypes.c_long * n)(*([width] * n))
        d2 = dst[: n * width * 2].view(torch.int16).view(n, width)
        g = 0
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.upload_rows(rows, lens, n, width, 2, d2, 1)
            torch.cuda.synchronize(); g = max(g, n * width * 2 / (time.perf_counter() - t0) / 1e9)
        print(f"arena {trial}: DMA {best:5.1f} GB/s   gather kernel {g:5.1f} GB/s   {nodes_of(t.data_ptr(), nbytes)}", flush=True)


if __name__ == "__main__":
    main()
