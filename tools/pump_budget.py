#!/usr/bin/env python3
"""The native pump with the CPU budget of ONE of eight ranks: run under `taskset -c <2 CPUs>` with
LOCAL_WORLD_SIZE=8 (so that vad_host_threads() sees a rank's share), one source thread; prints link fraction, tick p50 / p95 / max,
fill time per tick for both rates, lock-step and with gaps.  tools/r06_pump_budget.sh drives it beside the whole-box run."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    import numpy as np
    import torch
    import bench
    from silero_vad_amd import Engine, StreamPump, _lib
    L = _lib.lib()
    fills = int(os.environ.get("PUMP_FILL_THREADS", "1"))
    out = {"cpus_allowed": len(os.sched_getaffinity(0)), "vad_host_threads": L.vad_host_threads(), "LOCAL_WORLD_SIZE": os.environ.get("LOCAL_WORLD_SIZE"),
           "fill_threads": fills, "numa_node_bound": L.vad_bind_host_to_device(0)}
    dev = torch.device("cuda", 0)
    eng = Engine(device=0)
    link = bench.h2d_rate_GBps(dev)
    out["h2d_GBps"] = round(link, 2)
    for sr in (16000, 8000):
        n = 512 if sr == 16000 else 256
        cap = 8192
        rows = np.ascontiguousarray(bench.fixture_rows_i16(sr, cap, 32 * n))
        ceil = link * 1e9 / (n * 2)
        for gaps in (0.0, 0.10):
            pat = bench.gap_flags(256, cap, 41, gaps) if gaps else None
            pump = StreamPump(eng, sr, streams=cap, parts=1, ring_slots=4)
            t0 = 0
            pump.play(rows, 600, first_tick=t0, depth=2, fill_threads=fills, pattern=pat); t0 += 600
            res = {}
            for depth in (1, 2, 3):
                _, st = pump.play(rows, 3000, first_tick=t0, depth=depth, fill_threads=fills, pattern=pat); t0 += 3000
                res[f"depth{depth}"] = {"ticks_per_s": round(3000 / st["wall_ms"] * 1e3, 1), "of_link": round(cap * 3000 / (st["wall_ms"] / 1e3) / ceil, 3),
                                        "chunks_per_s": round(st["chunks"] / st["wall_ms"] * 1e3, 1),
                                        "tick_ms": {k: round(st[f"tick_ms_{k}"], 4) for k in ("p50", "p95", "max")},
                                        "fill_ms_per_tick": round(st["fill_ms_mean"], 4), "submit_ms": round(st["submit_ms_mean"], 4),
                                        "blocked_in_poll_ms": round(st["wait_ms_mean"], 4)}
            pump.close()
            out[f"{sr // 1000}k{'_gaps' if gaps else ''}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
