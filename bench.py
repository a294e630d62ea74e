#!/usr/bin/env python3
"""Headline benchmark: speech-probability throughput of the Silero-VAD hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): synthetic 16 kHz PCM, 512-sample
chunks, 4096 independent streams per GPU x 256 chunks per stream, fp32, already resident in HBM.
One "step" = one pass of the hot path over that batch = ONE vad_forward_audio call through the C
ABI (zeroed context/state, like the reference's audio_forward).  Streams are sharded across ranks
with no data-path collective (weak scaling: every rank owns 4096 streams); the only communication
is the barrier + MAX-reduce of the elapsed time that the measurement contract asks for.

Prints ONE JSON line on rank 0.  `value` = chunks/s over the whole job.  Extra objects:
  roofline      dominant kernel (frontend: STFT + encoder + W_ih GEMM) against the fp32 MFMA peak,
                from hipEvents recorded by the engine around that kernel during the timed steps
  cpu_baseline  the CPU oracle (a port of the reference's arithmetic) on this box's host cores,
                on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SR = 16000
CHUNK = 512
STREAMS = 4096           # per GPU
CHUNKS_PER_STREAM = 256  # per step
# Algorithmic work per chunk, dense, exactly as the reference computes it:
FLOP_PER_CHUNK = 1_359_104            # whole path
FLOP_PER_CHUNK_FRONT = 2 * (264_192 + 198_144 + 49_152 + 12_288 + 24_576 + 65_536)   # STFT+enc+W_ih
BYTES_PER_CHUNK = 2_052               # fp32 PCM in + fp32 prob out
PEAK_F32_TFLOPS = 157.3               # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_HBM_GBPS = 8000.0


def cpu_baseline(seconds_target=12.0):
    """Time the oracle (kind 'port': plain-C restatement of the reference's dense arithmetic, OpenMP
    over streams) on this host.  Bounded sample of the same workload: S streams x 32 chunks."""
    import numpy as np
    from oracle import Oracle
    o = Oracle()
    cores = len(os.sched_getaffinity(0))
    rng = np.random.default_rng(17 + SR)
    T = 32

    def run(S):
        pcm = (rng.standard_normal((S, T * CHUNK)) * 0.03).astype(np.float32)
        t0 = time.perf_counter()
        o.forward_audio(pcm, SR)
        return S * T / (time.perf_counter() - t0)

    rate = run(cores * 2)                          # warm-up + calibration
    S = max(cores, int(rate * seconds_target / T) // cores * cores)
    rate = run(S)
    return {"value": round(rate, 1), "unit": "chunks/s", "cores": cores, "kind": "port",
            "sample": f"{S} streams x {T} chunks of the same 16 kHz synthetic workload, oracle/vad_oracle.c "
                      f"(gcc -O3 -mavx2 -mfma, OpenMP over streams, {cores} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=STREAMS, help=argparse.SUPPRESS)
    ap.add_argument("--chunks", type=int, default=CHUNKS_PER_STREAM, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from silero_vad_amd import Engine

    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    eng = Engine(device=local)
    B, T = args.streams, args.chunks
    L = T * CHUNK
    gen = torch.Generator(device=dev).manual_seed(17 + SR + rank)
    pcm = torch.empty((B, L), dtype=torch.float32, device=dev)
    # 0.03 * N(0,1) as in examples/onnx_sequence/run.py:159-162, plus a per-stream tone so that the
    # operands are not sign-symmetric noise only (throughput is data independent; DVFS is not)
    pcm.normal_(0.0, 0.03, generator=gen)
    tt = torch.arange(L, device=dev, dtype=torch.float32) / SR
    f0 = 90.0 + 3.0 * torch.arange(B, device=dev, dtype=torch.float32)[:, None]
    pcm += 0.1 * torch.sin(2 * torch.pi * f0 * tt[None, :])
    del tt, f0
    ctx = torch.zeros((B, CHUNK // 8), device=dev)
    state = torch.zeros((2, B, 128), device=dev)
    probs = torch.empty((B, T), device=dev)
    eng.reserve(SR, B, T)

    def step():
        ctx.zero_()
        state.zero_()
        eng.forward_audio(pcm, SR, ctx, state, probs)

    for _ in range(args.warmup):
        step()
    eng.set_option("profile", "1")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    front_ms, rec_ms, calls = eng.kernel_times()
    eng.set_option("profile", "0")
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ok = bool(torch.isfinite(probs).all().item())

    if rank == 0:
        chunks_per_step = B * T * world
        value = chunks_per_step * args.steps / elapsed
        front_avg_s = front_ms / 1e3 / max(calls, 1)
        achieved = B * T * FLOP_PER_CHUNK_FRONT / front_avg_s / 1e12
        out = {
            "metric": "audio-chunks/sec (32 ms @ 16 kHz)",
            "value": round(value, 1),
            "unit": "chunks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: synthetic 16 kHz PCM resident in HBM, {CHUNK}-sample chunks, "
                                   f"{B} streams/GPU x {T} chunks/stream per step, zero initial state",
                       "streams_per_gpu": B, "chunks_per_stream": T, "sample_rate": SR,
                       "sharding": f"streams x{world}, no collectives"},
            "realtime_factor": round(value * 0.032, 1),
            "outputs_finite": ok,
            "path_fraction": {"fp32_peak": round(value / world * FLOP_PER_CHUNK / (PEAK_F32_TFLOPS * 1e12), 4),
                              "hbm_peak": round(value / world * BYTES_PER_CHUNK / (PEAK_HBM_GBPS * 1e9), 6),
                              "flop_per_chunk": FLOP_PER_CHUNK, "bytes_per_chunk": BYTES_PER_CHUNK},
            "kernel_ms": {"front": round(front_ms / max(calls, 1), 4), "rec": round(rec_ms / max(calls, 1), 4)},
            "roofline": {"bound": "mfma", "kernel": "front_kernel<32,float>",
                         "achieved": round(achieved, 3), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_TFLOPS, 4),
                         "flop_per_launch": B * T * FLOP_PER_CHUNK_FRONT,
                         "avg_launch_ms": round(front_avg_s * 1e3, 4),
                         "traffic": None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
