#!/usr/bin/env python3
"""Does a leg's figure depend on which legs ran before it in the same process?  python tools/leg_order_diag.py legA legB ... : runs the
bench legs in that order in ONE process and prints the figures that moved in the default line (stream_host tick latency / link fraction)."""
import json
import sys
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    import torch
    import torch.distributed as dist
    args = types.SimpleNamespace(live=8192, streams=4096, chunks=256, warmup=3, no_parity="--parity" not in sys.argv, recordings=4096, corpus_main_only=False,
                                 gpus=1, steps=None)
    torch.cuda.set_device(0)
    for name in [a for a in sys.argv[1:] if not a.startswith("--")]:
        if name in ("stream_host", "stream_host_8k"):
            d = bench.run_stream_host(args, 0, 1, 0, dist, 1000, 16000 if name == "stream_host" else 8000)
            print(name, json.dumps({"of_link": d["pcie"]["fraction_of_pcie_ceiling"], "lat": d["tick_latency_ms"]["median"], "p95": d["tick_latency_ms"]["p95"],
                                    "sustained": d["sustained"]["untimed_depth_trials"]}), flush=True)
        elif name in ("stream", "stream_8k"):
            d = bench.run_stream(args, 0, 1, 0, dist, 1000, 16000 if name == "stream" else 8000)
            print(name, d["value"], flush=True)
        elif name.startswith("corpus"):
            d = bench.run_corpus(args, 0, 1, 0, dist, int(name[6:] or 3))
            print(name, {k: v["fraction_of_pcie_ceiling"] for k, v in d["legs"].items()}, flush=True)
        elif name in ("c2", "8k"):
            d = bench.run_batch(args, 16000 if name == "c2" else 8000, 0, 1, 0, dist, 50)
            print(name, d["value"], flush=True)
        elif name.startswith("sleep"):
            import time
            time.sleep(float(name[5:]))
            print(name, flush=True)
        elif name == "gc":
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            print("gc + empty_cache", flush=True)
        elif name.startswith("plumbing"):
            bench.run_plumbing(args, 0, 8000 if name.endswith("8k") else 16000)
            print(name, "done", flush=True)


if __name__ == "__main__":
    main()
