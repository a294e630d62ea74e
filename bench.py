#!/usr/bin/env python3
"""Headline benchmark: speech-probability throughput of the Silero-VAD hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|8k|stream|corpus]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload `c2` (BASELINE.json configs[1], SURVEY.md section 8d "C2"): synthetic 16 kHz
PCM, 512-sample chunks, 4096 independent streams per GPU x 256 chunks per stream, fp32, already
resident in HBM.  One "step" = one pass of the hot path over that batch = ONE vad_forward_audio call
through the C ABI (zeroed context/state, like the reference's audio_forward).  Streams are sharded
across ranks with no data-path collective (weak scaling: every rank owns 4096 streams); the only
communication is the barrier + MAX-reduce of the elapsed time that the measurement contract asks
for.  The other configs are extra evidence lines, not the headline:
  8k      configs[2]: 8 kHz, 256-sample chunks, 4096 streams x 256 chunks (the 8 kHz net)
  stream  configs[4]: 8192 live streams per GPU (65 536 per 8-GPU node), persistent state in HBM,
          one hipGraph-captured vad_step per 32 ms tick; a step = one tick; also reports tick latency
  corpus  configs[3] (bounded sample): ragged int16 recordings in host memory -> pinned staging ->
          H2D overlapped with compute -> probs -> native batch segmenter.  PCIe-inclusive.

Prints ONE JSON line on rank 0.  `value` = chunks/s over the whole job.  Extra objects:
  roofline      dominant kernel (frontend: STFT + encoder + W_ih GEMM) against the fp32 MFMA peak,
                from hipEvents recorded by the engine around that kernel during the timed steps
  cpu_baseline  the CPU oracle (a port of the reference's arithmetic) on this box's host cores,
                on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import glob
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

STREAMS = 4096           # per GPU (c2 / 8k)
CHUNKS_PER_STREAM = 256  # per step
LIVE_STREAMS = 8192      # per GPU (stream): 65 536 per 8-GPU node
CLOCK_RAMP_STEPS = 40    # untimed steps (~130 ms) before the warm-up: DVFS ramp, see run_batch
PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak (= fp32 vector peak)
PEAK_F16_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0

# Algorithmic work per chunk.  "dense" = exactly as the reference
# computes it (DFT-basis conv, every tap); "mfma" = what our kernels execute on the matrix pipe
# (rFFT frontend on the VALU instead of the basis conv, zero-padding taps skipped).
WORK = {
    16000: {"chunk": 512, "flop": 1_359_104, "bytes": 2_052,
            "front_dense": 2 * (264_192 + 198_144 + 49_152 + 12_288 + 24_576 + 65_536),
            "front_mfma": 2 * (10 * 128 * 132 + 5 * 64 * 128 + 2 * 64 * 64 + 128 * 64 + 512 * 128),
            "front_split_mfma": 2 * 3 * (10 * 128 * 128 + 5 * 64 * 128 + 2 * 64 * 64 + 128 * 64 + 512 * 128),
            "rec_mfma": 2 * 512 * 128, "front_kernel": "front_kernel<32,float>",
            "front_split_kernel": "front_split_kernel<32,float>"},
    8000: {"chunk": 256, "flop": 767_232, "bytes": 1_028,
           "front_dense": 2 * (66_560 + 99_840 + 49_152 + 12_288 + 24_576 + 65_536),
           "front_mfma": 2 * (10 * 128 * 68 + 5 * 64 * 128 + 2 * 64 * 64 + 128 * 64 + 512 * 128),
           "front_split_mfma": 2 * 3 * (10 * 128 * 64 + 5 * 64 * 128 + 2 * 64 * 64 + 128 * 64 + 512 * 128),
           "rec_mfma": 2 * 512 * 128, "front_kernel": "front_kernel<16,float>",
           "front_split_kernel": "front_split_kernel<16,float>"},
}


def cpu_baseline(sr, seconds_target=12.0):
    """Time the oracle (kind 'port': plain-C restatement of the reference's dense arithmetic, OpenMP
    over streams) on this host.  Bounded sample of the same workload: S streams x 32 chunks."""
    import numpy as np
    from oracle import Oracle
    o = Oracle()
    cores = len(os.sched_getaffinity(0))
    rng = np.random.default_rng(17 + sr)
    T, n = 32, WORK[sr]["chunk"]

    def run(S):
        pcm = (rng.standard_normal((S, T * n)) * 0.03).astype(np.float32)
        t0 = time.perf_counter()
        o.forward_audio(pcm, sr)
        return S * T / (time.perf_counter() - t0)

    rate = run(cores * 2)                          # warm-up + calibration
    S = max(cores, int(rate * seconds_target / T) // cores * cores)
    rate = run(S)
    return {"value": round(rate, 1), "unit": "chunks/s", "cores": cores, "kind": "port",
            "sample": f"{S} streams x {T} chunks of the same {sr // 1000} kHz synthetic workload, "
                      f"oracle/vad_oracle.c (gcc -O3 -mavx2 -mfma, OpenMP over streams, {cores} threads)"}


def pmc_traffic(kernel_key, sr, B, T):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 --pmc summary
    (profiles/*_summary.json, written by tools/summarize_prof.py from a run of this same command):
    FETCH_SIZE (KiB; doubled -- gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md "HBM")
    + WRITE_SIZE (KiB).  None if no summary matches this workload."""
    best = None
    for f in sorted(glob.glob(str(ROOT / "profiles" / "*_summary.json"))):
        try:
            d = json.loads(Path(f).read_text())
        except Exception:
            continue
        wl = d.get("workload", {"sr": 16000, "streams": 4096, "chunks": 256})
        if (wl.get("sr"), wl.get("streams"), wl.get("chunks")) != (sr, B, T):
            continue
        for k, c in d.get("pmc", {}).items():
            if k.replace(" ", "").startswith(kernel_key.replace(" ", "")) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                best = {"bytes": int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
                        "fetch_bytes": int(2.0 * c["FETCH_SIZE"] * 1024), "write_bytes": int(c["WRITE_SIZE"] * 1024),
                        "source": os.path.relpath(f, ROOT)}
    return best


def setup_dist(args):
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
    return rank, world, local, dist


def timed(world, dist, dev, steps, fn):
    """barrier + synchronize on both sides of exactly `steps` calls of fn; MAX over ranks."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed


def synth_pcm(B, L, sr, dev, seed):
    """0.03 * N(0,1) as in examples/onnx_sequence/run.py:159-162, plus a per-stream tone so that the
    operands are not sign-symmetric noise only (throughput is data independent; DVFS is not)."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    pcm = torch.empty((B, L), dtype=torch.float32, device=dev)
    pcm.normal_(0.0, 0.03, generator=gen)
    tt = torch.arange(L, device=dev, dtype=torch.float32) / sr
    f0 = 90.0 + 3.0 * (torch.arange(B, device=dev, dtype=torch.float32) % 1024)[:, None]
    pcm += 0.1 * torch.sin(2 * torch.pi * f0 * tt[None, :])
    return pcm


def roofline(sr, chunks_per_launch, front_ms_avg, B, T, precision="fp32"):
    """Dominant kernel = the frontend (STFT + encoder + W_ih).  `achieved` = the reference's DENSE flop
    count for that part of the path per second, against the fp32 MFMA peak -- the precision
    the result is delivered in.  `mfma_executed` is what the matrix pipe really runs (fp32 MFMA, or
    3 f16 MFMA flops per algorithmic flop for precision=f16x3, against the f16 peak); `hbm` is the
    same launch against the HBM roofline (PCM in + gx out, 2 x 2048 B per 16 kHz chunk)."""
    w = WORK[sr]
    split = precision == "f16x3"
    s = front_ms_avg / 1e3
    dense = chunks_per_launch * w["front_dense"] / s / 1e12
    ex_flop = w["front_split_mfma"] if split else w["front_mfma"]
    ex_peak = PEAK_F16_TFLOPS if split else PEAK_F32_TFLOPS
    execd = chunks_per_launch * ex_flop / s / 1e12
    kname = w["front_split_kernel"] if split else w["front_kernel"]
    tr = pmc_traffic(kname.split(",")[0], sr, B, T)
    io_bytes = chunks_per_launch * (w["chunk"] * 4 + 2048)
    hbm = io_bytes / s / 1e9
    return {"bound": "mfma", "kernel": kname,
            "achieved": round(dense, 3), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
            "frac": round(dense / PEAK_F32_TFLOPS, 4),
            "note": "achieved/frac use the reference's DENSE flop count (SURVEY 8d) against the fp32 peak; the "
                    "kernel replaces the DFT-basis conv by an rFFT, skips zero-pad taps and (f16x3) runs "
                    "each product as 3 f16 MFMA products, so frac can exceed 1 -- mfma_executed is the "
                    "matrix-pipe utilisation, hbm the same launch against the HBM roofline",
            "flop_per_launch": chunks_per_launch * w["front_dense"],
            "mfma_executed": {"flop_per_launch": chunks_per_launch * ex_flop, "dtype": "f16" if split else "f32",
                              "achieved": round(execd, 3), "peak": ex_peak, "frac": round(execd / ex_peak, 4)},
            "hbm": {"bytes_per_launch": io_bytes, "achieved": round(hbm, 1), "peak": PEAK_HBM_GBPS,
                    "unit": "GB/s", "frac": round(hbm / PEAK_HBM_GBPS, 4)},
            "avg_launch_ms": round(front_ms_avg, 4),
            "traffic": tr["bytes"] if tr else None, "traffic_detail": tr}


def base_line(args, world, metric_sr, value, elapsed, steps):
    return {"metric": f"audio-chunks/sec (32 ms @ {metric_sr // 1000} kHz)", "value": round(value, 1),
            "unit": "chunks/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f32 sums of f16x3 split products",
            "precision": args.precision, "data": "synthetic"}


# ---- c2 / 8k: HBM-resident batch ------------------------------------------------------------------------
def run_batch(args, sr, rank, world, local, dist):
    from silero_vad_amd import Engine
    dev = torch.device("cuda", local)
    eng = Engine(device=local)
    eng.set_precision(args.precision)
    n = WORK[sr]["chunk"]
    B, T = args.streams, args.chunks
    pcm = synth_pcm(B, T * n, sr, dev, 17 + sr + rank)
    ctx = torch.zeros((B, n // 8), device=dev)
    state = torch.zeros((2, B, 128), device=dev)
    probs = torch.empty((B, T), device=dev)
    eng.reserve(sr, B, T)

    def step():
        ctx.zero_()
        state.zero_()
        eng.forward_audio(pcm, sr, ctx, state, probs)

    # the GPU takes some tens of milliseconds of load to reach its sustained clocks (measured: +4 % between the
    # 4th and the 40th step): a fixed untimed ramp precedes the W warm-up steps so that K steps time steady state
    for _ in range(max(0, CLOCK_RAMP_STEPS - args.warmup)):
        step()
    for _ in range(args.warmup):
        step()
    eng.set_option("profile", "1")
    elapsed = timed(world, dist, dev, args.steps, step)
    front_ms, rec_ms, calls = eng.kernel_times()
    eng.set_option("profile", "0")
    ok = bool(torch.isfinite(probs).all().item())
    other = None
    if world == 1:                                     # the other arithmetic, same workload, for the record
        alt = "fp32" if args.precision == "f16x3" else "f16x3"
        p_main = probs.clone()
        eng.set_precision(alt)
        for _ in range(2):
            step()
        eng.set_option("profile", "1")
        e2 = timed(world, dist, dev, max(3, args.steps // 2), step)
        f2, r2, c2 = eng.kernel_times()
        eng.set_option("profile", "0")
        other = {"precision": alt, "value": round(B * T * max(3, args.steps // 2) / e2, 1), "unit": "chunks/s",
                 "kernel_ms": {"front": round(f2 / max(c2, 1), 4), "rec": round(r2 / max(c2, 1), 4)},
                 "max_abs_prob_diff_vs_main": float((probs - p_main).abs().max().item())}
        eng.set_precision(args.precision)
    if rank != 0:
        return None
    w = WORK[sr]
    value = B * T * world * args.steps / elapsed
    out = base_line(args, world, sr, value, elapsed, args.steps)
    cfg = "configs[1]" if sr == 16000 else "configs[2]"
    out["config"] = {"workload": f"{cfg}: synthetic {sr // 1000} kHz PCM resident in HBM, {n}-sample chunks, "
                                 f"{B} streams/GPU x {T} chunks/stream per step, zero initial state",
                     "streams_per_gpu": B, "chunks_per_stream": T, "sample_rate": sr,
                     "sharding": f"streams x{world}, no collectives"}
    out["realtime_factor"] = round(value * 0.032, 1)
    out["outputs_finite"] = ok
    out["clock_ramp_steps"] = max(0, CLOCK_RAMP_STEPS - args.warmup)
    out["path_fraction"] = {"fp32_peak": round(value / world * w["flop"] / (PEAK_F32_TFLOPS * 1e12), 4),
                            "hbm_peak": round(value / world * w["bytes"] / (PEAK_HBM_GBPS * 1e9), 6),
                            "flop_per_chunk": w["flop"], "bytes_per_chunk": w["bytes"]}
    c = max(calls, 1)
    out["kernel_ms"] = {"front": round(front_ms / c, 4), "rec": round(rec_ms / c, 4)}
    out["roofline"] = roofline(sr, B * T, front_ms / c, B, T, args.precision)
    rs = rec_ms / c / 1e3
    out["rec_kernel"] = {"gx_read_GBps": round(B * T * 2048 / rs / 1e9, 1),
                         "hbm_frac": round(B * T * 2048 / rs / 1e9 / PEAK_HBM_GBPS, 4)}
    if other:
        out["other_precision"] = other
    return out


# ---- stream: live streams, hipGraph step -------------------------------------------------------------------
def run_stream(args, rank, world, local, dist):
    from silero_vad_amd import Engine, StreamPool
    sr = 16000
    dev = torch.device("cuda", local)
    eng = Engine(device=local)
    eng.set_precision(args.precision)
    n = WORK[sr]["chunk"]
    cap = args.live
    pool = StreamPool(eng, sr, capacity=cap, graph=True)
    for _ in range(cap):
        pool.open()
    ring = 8                                                   # device-side audio source, 8 ticks long
    src = synth_pcm(cap, ring * n, sr, dev, 23 + rank).view(cap, ring, n).transpose(0, 1).contiguous()
    k = [0]

    def tick():
        pool.pcm.copy_(src[k[0] % ring])
        pool.tick_staged()
        k[0] += 1

    for _ in range(max(args.warmup, 3) + 1500):                # + DVFS ramp (~130 ms of ticks), see run_batch
        tick()
    steps = args.steps
    elapsed = timed(world, dist, dev, steps, tick)
    # latency of one tick, host-visible: input staged -> probabilities readable
    lat = []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tick()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    eng.set_option("profile", "1")
    for _ in range(20):
        pool._launch()                                          # eager launches carry the hipEvents
    front_ms, rec_ms, calls = eng.kernel_times()
    eng.set_option("profile", "0")
    ok = bool(torch.isfinite(pool.prob).all().item())
    if rank != 0:
        return None
    value = cap * world * steps / elapsed
    out = base_line(args, world, sr, value, elapsed, steps)
    out["config"] = {"workload": f"configs[4]: {cap} live 16 kHz streams/GPU ({cap * 8} per 8-GPU node), one "
                                 f"hipGraph-captured vad_step per 32 ms tick, (h,c)+context persistent in HBM",
                     "streams_per_gpu": cap, "sample_rate": sr, "step": "one tick (one chunk per stream)",
                     "sharding": f"streams x{world}, no collectives"}
    out["realtime_factor"] = round(value * 0.032, 1)
    out["outputs_finite"] = ok
    out["tick_latency_ms"] = {"median": round(lat[len(lat) // 2], 4), "p95": round(lat[int(len(lat) * 0.95)], 4),
                              "budget_ms": 32.0}
    c = max(calls, 1)
    out["kernel_ms"] = {"front": round(front_ms / c, 4), "rec": round(rec_ms / c, 4)}
    out["roofline"] = roofline(sr, cap, front_ms / c, cap, 1, args.precision)
    return out


# ---- corpus: ragged recordings from host memory -------------------------------------------------------------
def run_corpus(args, rank, world, local, dist):
    import numpy as np
    from silero_vad_amd import load_silero_vad, ragged_speech_segments
    sr = 16000
    dev = torch.device("cuda", local)
    model = load_silero_vad(device=local, precision=args.precision)
    n = WORK[sr]["chunk"]
    rng = np.random.default_rng(101 + rank)
    base_len = 8 << 20
    tt = np.arange(base_len, dtype=np.float32) / sr
    base = (0.03 * rng.standard_normal(base_len).astype(np.float32)
            + 0.2 * np.sin(2 * np.pi * 170.0 * tt) * (np.sin(2 * np.pi * 0.7 * tt) > 0))
    base = torch.from_numpy((base * 32767.0).clip(-32768, 32767).astype(np.int16))
    R = args.recordings
    lens = rng.integers(20 * sr, 40 * sr, size=R)              # 20-40 s recordings, ragged
    offs = rng.integers(0, base_len - 40 * sr, size=R)
    audios = [base[o:o + m] for o, m in zip(offs, lens)]       # views: the "files" already decoded in RAM
    chunks = int(sum((m + n - 1) // n for m in lens))
    nseg = [0]

    def step():
        segs = ragged_speech_segments(audios, model, sr, max_waste=0.1, max_bytes=256 << 20)
        nseg[0] = sum(len(s) for s in segs)

    for _ in range(max(1, min(args.warmup, 1))):
        step()
    steps = max(1, min(args.steps, 3))
    elapsed = timed(world, dist, dev, steps, step)
    if rank != 0:
        return None
    value = chunks * world * steps / elapsed
    hours = float(lens.sum()) / sr / 3600.0
    out = base_line(args, world, sr, value, elapsed, steps)
    out["config"] = {"workload": f"configs[3] bounded sample: {R} ragged int16 recordings/GPU (20-40 s, {hours:.2f} h) "
                                 "in host RAM -> pinned staging -> H2D overlapped with compute -> probs -> "
                                 "native batch segmenter; PCIe- and host-inclusive",
                     "recordings_per_gpu": R, "audio_hours_per_gpu_per_step": round(hours, 3), "sample_rate": sr,
                     "sharding": f"recordings x{world}, no collectives"}
    out["realtime_factor"] = round(value * 0.032, 1)
    out["segments_found_rank0"] = nseg[0]
    out["projected_10k_hours_s"] = round(10_000.0 / (hours * world * steps / elapsed), 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["c2", "8k", "stream", "corpus"], default="c2")
    ap.add_argument("--precision", choices=["f16x3", "fp32"], default="f16x3",
                    help="f16x3 (default): fp16x3 split products on the f16 matrix cores, fp32 sums; "
                         "fp32: exact v_mfma_f32_16x16x4_f32 chain")
    ap.add_argument("--streams", type=int, default=STREAMS, help=argparse.SUPPRESS)
    ap.add_argument("--chunks", type=int, default=CHUNKS_PER_STREAM, help=argparse.SUPPRESS)
    ap.add_argument("--live", type=int, default=LIVE_STREAMS, help=argparse.SUPPRESS)
    ap.add_argument("--recordings", type=int, default=1024, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"c2": 20, "8k": 20, "stream": 200, "corpus": 2}[args.config]

    rank, world, local, dist = setup_dist(args)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    torch.cuda.set_device(torch.device("cuda", local))

    if args.config in ("c2", "8k"):
        sr = 16000 if args.config == "c2" else 8000
        out = run_batch(args, sr, rank, world, local, dist)
    elif args.config == "stream":
        sr, out = 16000, run_stream(args, rank, world, local, dist)
    else:
        sr, out = 16000, run_corpus(args, rank, world, local, dist)

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sr)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
