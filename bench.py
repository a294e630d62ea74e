#!/usr/bin/env python3
"""Headline benchmark: speech-probability throughput of the Silero-VAD hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|8k|stream|corpus] [--precision fp32|f16x3]

`--gpus N` (N > 1) works both ways the driver may start it: under `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` (RANK/LOCAL_RANK/WORLD_SIZE in the environment), and as a plain
`python bench.py --gpus N`, which re-executes itself under torch.distributed.run on 127.0.0.1.

Default workload `c2` (BASELINE.json configs[1], SURVEY.md section 8d "C2"): synthetic 16 kHz PCM, 512-sample
chunks, 4096 independent streams per GPU x 256 chunks per stream, fp32, already resident in HBM, exact fp32
arithmetic (`dtype: "f32"`, the reference's).  One "step" = one pass of the hot path over that batch = ONE
vad_forward_audio call through the C ABI (zeroed context/state, like the reference's audio_forward).  Streams
are sharded across ranks with no data-path collective (weak scaling: every rank owns 4096 streams); the only
communication is the barrier + MAX-reduce of the elapsed time that the measurement contract asks for.

At N = 1 the same JSON line also carries, for the record (none of them is the headline `value`):
  other_precision  the opt-in f16x3 arithmetic on the same data, same ramp/steps protocol
  other_configs    8k     configs[2]: 8 kHz, 256-sample chunks, 4096 streams x 256 chunks (the 8 kHz net)
                   stream configs[4]: 8192 live streams per GPU (65 536 per 8-GPU node), persistent state in
                          HBM, one hipGraph-captured vad_step per 32 ms tick; reports tick latency
                   corpus configs[3] (bounded sample): ragged int16 recordings in host memory -> pinned staging
                          -> H2D overlapped with compute -> probs -> native segmenter.  PCIe + host inclusive.
  roofline         dominant kernel (frontend: STFT + encoder + W_ih GEMM): EXECUTED fp32 MFMA flops per launch /
                   average launch duration (hipEvents recorded by the engine around that kernel on the launch
                   stream during the timed steps) against the dense fp32 MFMA peak; always <= 1
  cpu_baseline     the reference's own ATen CPU operators (oracle/aten_port.py, kind "aten-port") timed on this
                   box's host cores under BASELINE.md section 3 protocols R1-R4 (rank 0, N = 1 only)
`--config 8k|stream|corpus` runs one of the other configs as the main leg instead.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

STREAMS = 4096           # per GPU (c2 / 8k)
CHUNKS_PER_STREAM = 256  # per step
LIVE_STREAMS = 8192      # per GPU (stream): 65 536 per 8-GPU node
CLOCK_RAMP_STEPS = 40    # untimed steps before the warm-up: DVFS ramp, see run_batch
PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak (= fp32 vector peak)
PEAK_F16_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0

# Work per chunk.  "flop"/"bytes": the reference's DENSE arithmetic and its I/O --
# "useful reference work".  "front_mfma" / "rec_mfma": matrix flops our kernels EXECUTE (rFFT frontend on the VALU
# instead of the DFT-basis conv, encoder 0 as Winograd F(2,3) over frame pairs (4 instead of 5 GEMMs per pair), zero-padding
# taps skipped, the Nyquist bin applied on the VALU) = MFMA instructions per
# 16-chunk tile (tools/isa_mix.py; SQ_INSTS_MFMA / tiles in profiles/) x 2048 flop / 16 -- the numerator of
# roofline.frac.  f16x3: three f16 products per product.
WORK = {
    16000: {"chunk": 512, "flop": 1_359_104, "bytes": 2_052,
            "front_dense": 2 * (264_192 + 198_144 + 49_152 + 12_288 + 24_576 + 65_536),
            "front_mfma": 54 * 64 * 2048 // 16,       # 3 456 MFMAs / tile: 54 weight units x 64 (enc0 as one Winograd F(4,3) tile)
            "front_mfma_direct": 2 * (10 * 128 * 128 + 5 * 64 * 128 + 2 * 64 * 64 + 128 * 64 + 512 * 128),   # 4 480 (enc0 tap by tap)
            "front_split_mfma": 2 * 3 * (10 * 128 * 128 + 5 * 64 * 128 + 2 * 64 * 64 + 128 * 64 + 512 * 128),
            # VALU issue cycles per tile beside the MFMAs (same pipe): 2 612 packed x 5.4 + 4 266 other x 2.6
            # (SQ_INSTS_VALU - SQ_INSTS_MFMA per tile, profiles/r02k_fp32_summary.md; rates profiles/r02d_issue_pipes.md)
            "front_valu_cycles": 25_200,
            "rec_mfma": 2 * 512 * 128, "front_kernel": "front_f43_kernel<32, float>",
            "front_split_kernel": "front_split_kernel<32, float>"},
    8000: {"chunk": 256, "flop": 767_232, "bytes": 1_028,
           "front_dense": 2 * (66_560 + 99_840 + 49_152 + 12_288 + 24_576 + 65_536),
           "front_mfma": 42 * 64 * 2048 // 16,        # 2 688 MFMAs / tile (Winograd F(4,3))
           "front_mfma_direct": 2 * (10 * 128 * 64 + 5 * 64 * 128 + 2 * 64 * 64 + 128 * 64 + 512 * 128),    # 3 200
           "front_split_mfma": 2 * 3 * (10 * 128 * 64 + 5 * 64 * 128 + 2 * 64 * 64 + 128 * 64 + 512 * 128),
           "front_valu_cycles": 12_500,                # 1 280 packed x 5.4 + 2 144 other x 2.6 (profiles/r02k_8k_summary.md)
           "rec_mfma": 2 * 512 * 128, "front_kernel": "front_f43_kernel<16, float>",
           "front_split_kernel": "front_split_kernel<16, float>"},
}
GX_BYTES = 2048          # engine-internal: fp32 LSTM input-gate pre-activations per chunk, written and read once


# ---- launching --------------------------------------------------------------------------------------------
def relaunch_distributed(args):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N`."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def setup_dist(args):
    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    args.gpus = world
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):   # single node: rendezvous over loopback
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if args.dry:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return rank, world, local, dist


def timed(world, dist, dev, steps, fn, sync):
    """barrier + synchronize on both sides of exactly `steps` calls of fn; MAX over ranks."""
    import torch
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed


def gpu_sync():
    import torch
    torch.cuda.synchronize()


# ---- evidence helpers -------------------------------------------------------------------------------------
def cpu_baseline(sr, budget_s=24.0):
    """The reference's CPU path on this box: oracle/aten_port.py issues the ATen operators the TorchScript model
    dispatches to (bit-identical to it on the goldens, tests/test_oracle.py) under the reference's threading and
    timing rules.  Runs in its own CPU-only process (it forks one worker per core for protocol R4)."""
    try:        # every protocol runs in its own process with its own time limit (oracle/aten_port.py baseline())
        r = subprocess.run([sys.executable, "-m", "oracle.aten_port", "--sr", str(sr), "--budget-s", str(budget_s)],
                           cwd=str(ROOT), capture_output=True, text=True, timeout=8 * budget_s + 240,
                           env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
        err = (r.stderr or r.stdout)[-400:]
    except subprocess.TimeoutExpired as e:
        line, err = None, f"timed out: {e}"
    if line is None:
        return {"value": None, "unit": "chunks/s", "kind": "aten-port", "error": err}
    d = json.loads(line)
    return {"value": d["value"], "unit": "chunks/s", "cores": d["nproc"], "affinity_cpus": d["affinity_cpus"],
            "kind": "aten-port", "best_protocol": d["best"], "cpu_model": d["cpu_model"], "torch": d["torch"],
            "runs": d["runs"],
            "sample": f"same {sr // 1000} kHz synthetic workload (0.03 N(0,1)), audio_forward over B streams x T chunks "
                      f"per run as listed under runs; warm-up {d['warmup']}, median of {d['trials']}; R1 = 1 thread B=1 "
                      "(the reference's shipped default), R2 = 1 thread B=4096, R3 = nproc threads B=4096, "
                      "R4 = nproc processes x 1 thread sharing the 4096 streams; value = the best of the four"}


def pmc_traffic(kernel_key, sr, B, T):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 --pmc summary (profiles/*_summary.json,
    written by tools/summarize_prof.py from a run of this same command): FETCH_SIZE (KiB; doubled -- gfx950 tallies
    128-B requests at 64 B, MI355X_MICROARCH.md "HBM") + WRITE_SIZE (KiB).  None if no summary matches."""
    best = None
    key = kernel_key.replace(" ", "")
    for f in sorted(glob.glob(str(ROOT / "profiles" / "*_summary.json"))):
        try:
            d = json.loads(Path(f).read_text())
        except Exception:
            continue
        wl = d.get("workload", {"sr": 16000, "streams": 4096, "chunks": 256})
        if (wl.get("sr"), wl.get("streams"), wl.get("chunks")) != (sr, B, T):
            continue
        for k, c in d.get("pmc", {}).items():
            if k.replace(" ", "").startswith(key) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                best = {"bytes": int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
                        "fetch_bytes": int(2.0 * c["FETCH_SIZE"] * 1024), "write_bytes": int(c["WRITE_SIZE"] * 1024),
                        "source": os.path.relpath(f, ROOT)}
    return best


def synth_pcm(B, L, sr, dev, seed):
    """0.03 * N(0,1) as in examples/onnx_sequence/run.py:159-162, plus a per-stream tone so that the
    operands are not sign-symmetric noise only (throughput is data independent; DVFS is not)."""
    import torch
    gen = torch.Generator(device=dev).manual_seed(seed)
    pcm = torch.empty((B, L), dtype=torch.float32, device=dev)
    pcm.normal_(0.0, 0.03, generator=gen)
    tt = torch.arange(L, device=dev, dtype=torch.float32) / sr
    f0 = 90.0 + 3.0 * (torch.arange(B, device=dev, dtype=torch.float32) % 1024)[:, None]
    pcm += 0.1 * torch.sin(2 * torch.pi * f0 * tt[None, :])
    return pcm


def issue_pipe(w, chunks_per_launch, front_ms_avg):
    """Second reading of the same launch time.  On gfx950 fp32 MFMA and VALU instructions of all waves of a SIMD serialise
    on one issue pipe (profiles/r02d_issue_pipes.md), so the kernel's floor is its MFMA cycles PLUS its VALU cycles; the
    MFMA-only `frac` above cannot exceed mfma / (mfma + valu) = 0.81 for this instruction mix however well it is scheduled."""
    mfma_cyc = w["front_mfma"] * 16 // 2048 * 32            # MFMAs per 16-chunk tile x 32 cycles
    cyc = mfma_cyc + w["front_valu_cycles"]
    tiles_per_simd = chunks_per_launch / 16 / (256 * 4)
    floor_ms = cyc * tiles_per_simd / 2.4e9 * 1e3
    return {"mfma_cycles_per_tile": mfma_cyc, "valu_cycles_per_tile": w["front_valu_cycles"],
            "floor_ms_at_2.4GHz": round(floor_ms, 3), "frac": round(floor_ms / front_ms_avg, 4),
            "mfma_share_of_floor": round(mfma_cyc / cyc, 4),
            "definition": "(MFMA + VALU issue cycles per tile) x tiles per SIMD / 2.4 GHz, over the measured launch time"}


def roofline(sr, chunks_per_launch, front_ms_avg, rec_ms_avg, B, T, precision):
    """Dominant kernel = the frontend (STFT + encoder + W_ih).  achieved = matrix flops the kernel EXECUTES per
    launch / its average launch duration, peak = the dense peak of the pipe it runs on (fp32 MFMA 157.3 TF, or
    the f16 peak for the opt-in f16x3 kernels): frac <= 1.  The reference's dense flop count for the same part of
    the path ("useful work") is reported separately and is never divided into `frac`.  traffic = HBM bytes per
    launch of this kernel from the committed PMC pass; `path` relates the whole path (both kernels) to the
    algorithmic bytes of SURVEY 8(d)."""
    w = WORK[sr]
    split = precision == "f16x3"
    s = front_ms_avg / 1e3
    ex_flop = w["front_split_mfma"] if split else w["front_mfma"]
    ex_peak = PEAK_F16_TFLOPS if split else PEAK_F32_TFLOPS
    execd = chunks_per_launch * ex_flop / s / 1e12
    kname = w["front_split_kernel"] if split else w["front_kernel"]
    rname = "rec_split_kernel" if split else "rec_kernel"
    tr_f = pmc_traffic(kname.split(",")[0], sr, B, T)
    tr_r = pmc_traffic(rname, sr, B, T)
    alg = chunks_per_launch * w["bytes"]
    kio = chunks_per_launch * (w["chunk"] * 4 + GX_BYTES)
    path_traffic = (tr_f["bytes"] + tr_r["bytes"]) if (tr_f and tr_r) else None
    out = {"bound": "mfma", "kernel": kname, "dtype": "f16" if split else "f32",
           "achieved": round(execd, 3), "peak": ex_peak, "unit": "TFLOP/s", "frac": round(execd / ex_peak, 4),
           "flop_per_launch": chunks_per_launch * ex_flop, "avg_launch_ms": round(front_ms_avg, 4),
           "definition": "executed MFMA flops of the dominant kernel per launch / its hipEvent launch duration, "
                         "against the dense peak of the matrix pipe it runs on",
           "traffic": tr_f["bytes"] if tr_f else None, "traffic_detail": tr_f,
           "kernel_io_bytes": kio,
           "useful_dense": {"flop_per_launch": chunks_per_launch * w["front_dense"],
                            "tflops": round(chunks_per_launch * w["front_dense"] / s / 1e12, 3),
                            "note": "the reference's dense flop count for this part of the path (DFT-basis conv, every "
                                    "tap); the kernel executes fewer -- not a utilisation figure"},
           "issue_pipe": None if split else issue_pipe(w, chunks_per_launch, front_ms_avg),
           "hbm": {"achieved": round(kio / s / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                   "frac": round(kio / s / 1e9 / PEAK_HBM_GBPS, 4),
                   "note": "this kernel's own I/O (PCM in + gx out) against the HBM roofline: far from binding"},
           "path": {"algorithmic_bytes": alg, "algorithmic_bytes_per_chunk": w["bytes"],
                    "traffic": path_traffic,
                    "traffic_over_algorithmic": round(path_traffic / alg, 3) if path_traffic else None,
                    "traffic_detail": {"front": tr_f, "rec": tr_r},
                    "mfma_flop_per_chunk": ex_flop + w["rec_mfma"] * (3 if split else 1)}}
    if rec_ms_avg:
        rs = rec_ms_avg / 1e3
        rflop = chunks_per_launch * w["rec_mfma"] * (3 if split else 1)
        out["rec_kernel"] = {"kernel": rname, "avg_launch_ms": round(rec_ms_avg, 4),
                             "mfma_frac": round(rflop / rs / 1e12 / ex_peak, 4),
                             "gx_read_GBps": round(chunks_per_launch * GX_BYTES / rs / 1e9, 1),
                             "hbm_frac": round(chunks_per_launch * GX_BYTES / rs / 1e9 / PEAK_HBM_GBPS, 4)}
    return out


def base_line(args, world, metric_sr, value, elapsed, steps, precision):
    return {"metric": f"audio-chunks/sec (32 ms @ {metric_sr // 1000} kHz)", "value": round(value, 1),
            "unit": "chunks/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if precision == "fp32" else "f16x3 (NOT fp32: 3-term fp16 split products, fp32 sums)",
            "precision": precision, "data": "synthetic"}


# ---- c2 / 8k: HBM-resident batch ------------------------------------------------------------------------
def time_batch(eng, precision, step, probs, world, dist, dev, steps, warmup):
    """One arithmetic on one workload: clock ramp, warm-up, `steps` timed calls with the engine's hipEvents on."""
    import torch
    eng.set_precision(precision)
    if os.environ.get("VAD_BENCH_ENC0"):           # tools/slow_box_hunt.sh: time an A/B form of the fp32 frontend (the
        eng.set_option("enc0", os.environ["VAD_BENCH_ENC0"])   # roofline block of the line then does not apply)
    # the GPU takes some tens of milliseconds of load to reach its sustained clocks (measured: +4 % between the
    # 4th and the 40th step): a fixed untimed ramp precedes the W warm-up steps so that K steps time steady state
    for _ in range(max(0, CLOCK_RAMP_STEPS - warmup)):
        step()
    for _ in range(warmup):
        step()
    eng.set_option("profile", "1")
    elapsed = timed(world, dist, dev, steps, step, gpu_sync)
    front_ms, rec_ms, calls = eng.kernel_times()
    eng.set_option("profile", "0")
    ok = bool(torch.isfinite(probs).all().item())
    c = max(calls, 1)
    return elapsed, front_ms / c, rec_ms / c, ok


def run_batch(args, sr, rank, world, local, dist, steps, with_other):
    import torch
    from silero_vad_amd import Engine
    dev = torch.device("cuda", local)
    eng = Engine(device=local)
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

    elapsed, front_ms, rec_ms, ok = time_batch(eng, args.precision, step, probs, world, dist, dev, steps, args.warmup)
    other = None
    if with_other and world == 1:                      # the other arithmetic, same data, same protocol, for the record
        alt = "fp32" if args.precision == "f16x3" else "f16x3"
        p_main = probs.clone()
        e2, f2, r2, ok2 = time_batch(eng, alt, step, probs, world, dist, dev, steps, args.warmup)
        other = {"precision": alt, "dtype": base_line(args, 1, sr, 0, 1, 1, alt)["dtype"],
                 "value": round(B * T * steps / e2, 1), "unit": "chunks/s", "steps": steps,
                 "ms_per_step": round(e2 / steps * 1e3, 4), "kernel_ms": {"front": round(f2, 4), "rec": round(r2, 4)},
                 "outputs_finite": ok2,
                 "max_abs_prob_diff_vs_main": float((probs - p_main).abs().max().item()),
                 "roofline": roofline(sr, B * T, f2, r2, B, T, alt)}
        eng.set_precision(args.precision)
    if rank != 0:
        return None
    w = WORK[sr]
    value = B * T * world * steps / elapsed
    out = base_line(args, world, sr, value, elapsed, steps, args.precision)
    cfg = "configs[1]" if sr == 16000 else "configs[2]"
    out["config"] = {"workload": f"{cfg}: synthetic {sr // 1000} kHz PCM resident in HBM, {n}-sample chunks, "
                                 f"{B} streams/GPU x {T} chunks/stream per step, zero initial state",
                     "streams_per_gpu": B, "chunks_per_stream": T, "sample_rate": sr,
                     "sharding": f"streams x{world}, no collectives"}
    out["realtime_factor"] = round(value * 0.032, 1)
    out["outputs_finite"] = ok
    out["clock_ramp_steps"] = max(0, CLOCK_RAMP_STEPS - args.warmup)
    out["timed_region_s"] = round(elapsed, 4)
    out["path_fraction"] = {"dense_flop_vs_fp32_peak": round(value / world * w["flop"] / (PEAK_F32_TFLOPS * 1e12), 4),
                            "algorithmic_bytes_vs_hbm_peak": round(value / world * w["bytes"] / (PEAK_HBM_GBPS * 1e9), 6),
                            "flop_per_chunk": w["flop"], "bytes_per_chunk": w["bytes"],
                            "note": "useful reference work (dense flops, SURVEY 8d) per second per GPU; informational"}
    out["kernel_ms"] = {"front": round(front_ms, 4), "rec": round(rec_ms, 4)}
    out["roofline"] = roofline(sr, B * T, front_ms, rec_ms, B, T, args.precision)
    if other:
        out["other_precision"] = other
    return out


# ---- stream: live streams, hipGraph step -------------------------------------------------------------------
def run_stream(args, rank, world, local, dist, steps):
    import torch
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

    for _ in range(max(args.warmup, 3) + 1500):                # + DVFS ramp (~130 ms of ticks), see time_batch
        tick()
    elapsed = timed(world, dist, dev, steps, tick, gpu_sync)
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
    out = base_line(args, world, sr, value, elapsed, steps, args.precision)
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
    out["roofline"] = roofline(sr, cap, front_ms / c, rec_ms / c, cap, 1, args.precision)
    return out


# ---- corpus: ragged recordings from host memory -------------------------------------------------------------
def run_corpus(args, rank, world, local, dist, steps):
    import numpy as np
    import torch
    from silero_vad_amd import load_silero_vad, ragged_speech_segments, refill_speech_segments
    from silero_vad_amd import streams as S
    sr = 16000
    dev = torch.device("cuda", local)
    model = load_silero_vad(device=local, precision=args.precision)
    n = WORK[sr]["chunk"]
    rng = np.random.default_rng(101 + rank)
    base_len = 8 << 20
    tt = np.arange(base_len, dtype=np.float32) / sr
    base = (0.03 * rng.standard_normal(base_len).astype(np.float32)
            + 0.2 * np.sin(2 * np.pi * 170.0 * tt) * (np.sin(2 * np.pi * 0.7 * tt) > 0))
    base_f = torch.from_numpy(base.astype(np.float32))
    base_i = torch.from_numpy((base * 32767.0).clip(-32768, 32767).astype(np.int16))
    R = args.recordings
    lens = rng.integers(20 * sr, 40 * sr, size=R)              # 20-40 s recordings, ragged
    offs = rng.integers(0, base_len - 40 * sr, size=R)
    chunks = int(sum((m + n - 1) // n for m in lens))
    hours = float(lens.sum()) / sr / 3600.0
    legs = {}
    for name, src, sched in (("int16", base_i, "buckets"), ("fp32", base_f, "buckets"), ("int16_refill", base_i, "refill")):
        audios = [src[o:o + m] for o, m in zip(offs, lens)]    # views: the "files" already decoded in RAM
        nseg = [0]
        S.STATS.clear()

        def step():
            if sched == "buckets":      # length-sorted buckets, one lock-step call each, device scan per bucket
                segs = ragged_speech_segments(audios, model, sr, max_waste=0.1, max_bytes=1 << 30)
            else:                       # persistent slots refilled at slab boundaries, one device scan at the end
                segs = refill_speech_segments(audios, model, sr, slots=max(64, R // 2), slab_chunks=64)
            nseg[0] = sum(len(s) for s in segs)

        step()                                                  # warm-up: pinned buffers, scratch
        S.STATS.clear()
        elapsed = timed(world, dist, dev, steps, step, gpu_sync)
        st = dict(S.STATS)
        bytes_in = float(lens.sum()) * (4 if name == "fp32" else 2) * steps
        legs[name] = {"scheduler": sched, "value": round(chunks * world * steps / elapsed, 1), "unit": "chunks/s", "steps": steps,
                      "s_per_step": round(elapsed / steps, 4), "segments_found_rank0": nseg[0],
                      "ingest_GBps_per_gpu": round(bytes_in / elapsed / 1e9, 2),
                      "h2d_GBps_while_copying": (round(st["h2d_bytes"] / st["h2d_s"] / 1e9, 2) if st.get("h2d_s") else None),
                      "host_stage_ms_per_step": round(st.get("stage_s", 0) / steps * 1e3, 2),
                      "host_segmenter_ms_per_step": round(st.get("scan_s", 0) / steps * 1e3, 2),
                      "d2h_MB_per_step": round(st.get("d2h_bytes", 0) / steps / 1e6, 3),
                      "buckets_per_step": int(st.get("buckets", 0) / steps),
                      "padded_over_real_samples": round(st.get("padded", 0) / max(st.get("real", 1), 1), 4),
                      "projected_10k_hours_s": round(10_000.0 / (hours * world * steps / elapsed), 1)}
    # what the link allows: the H2D rate measured while copying / bytes per chunk of the leg's sample format -- the leg's
    # value can approach it (fully overlapped pipeline), never exceed it
    link = next((l["h2d_GBps_while_copying"] for l in legs.values() if l["h2d_GBps_while_copying"]), None)
    for name, l in legs.items():
        gbps = l["h2d_GBps_while_copying"] or link
        l["pcie_ceiling_chunks_per_s"] = round(gbps * 1e9 / (n * (4 if name == "fp32" else 2)) * world, 1) if gbps else None
        l["fraction_of_pcie_ceiling"] = round(l["value"] / l["pcie_ceiling_chunks_per_s"], 3) if gbps else None
    if rank != 0:
        return None
    main = legs["int16"]
    out = base_line(args, world, sr, main["value"], main["s_per_step"] * steps, steps, args.precision)
    out["config"] = {"workload": f"configs[3] bounded sample: {R} ragged recordings/GPU (20-40 s, {hours:.2f} h) "
                                 "in host RAM -> pinned staging -> H2D overlapped with compute -> probs -> "
                                 "segmenter on the GPU -> segment lists to the host; PCIe- and host-inclusive; main leg int16 "
                                 "PCM, bucket scheduler",
                     "recordings_per_gpu": R, "audio_hours_per_gpu_per_step": round(hours, 3), "sample_rate": sr,
                     "sharding": f"recordings x{world}, no collectives"}
    out["realtime_factor"] = round(main["value"] * 0.032, 1)
    out["legs"] = legs
    out["projected_10k_hours_s"] = main["projected_10k_hours_s"]
    return out


# ---- dry: the launch / reduce / print plumbing without a GPU (tests/test_sharding.py) -------------------------
def run_dry(args, rank, world, dist):
    import torch
    acc = [0.0]

    def step():
        acc[0] += float(torch.ones(1000).sum())

    elapsed = timed(world, dist, torch.device("cpu"), args.steps, step, lambda: None)
    if rank != 0:
        return None
    out = base_line(args, world, 16000, 0.0, elapsed, args.steps, args.precision)
    out.update({"dry": True, "metric": "dry run (no measurement)", "data": "none (dry run of the launch/reduce plumbing; no GPU work, not a measurement)",
                "config": {"workload": "dry", "sharding": f"streams x{world}, no collectives"}})
    return out


def small(d):
    """The part of a config's line that is kept when it is nested under other_configs."""
    if d is None:
        return None
    keep = ("value", "unit", "steps", "ms_per_step", "dtype", "kernel_ms", "tick_latency_ms", "legs",
            "projected_10k_hours_s", "outputs_finite", "realtime_factor", "timed_region_s")
    out = {k: d[k] for k in keep if k in d}
    out["workload"] = d["config"]["workload"]
    if "roofline" in d:
        out["roofline"] = {k: d["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["c2", "8k", "stream", "corpus"], default="c2")
    ap.add_argument("--precision", choices=["fp32", "f16x3"], default="fp32",
                    help="fp32 (default): exact v_mfma_f32_16x16x4_f32 chain, the reference's arithmetic; "
                         "f16x3: opt-in fp16x3 split products on the f16 matrix cores (narrower than fp32)")
    ap.add_argument("--streams", type=int, default=STREAMS, help=argparse.SUPPRESS)
    ap.add_argument("--chunks", type=int, default=CHUNKS_PER_STREAM, help=argparse.SUPPRESS)
    ap.add_argument("--live", type=int, default=LIVE_STREAMS, help=argparse.SUPPRESS)
    ap.add_argument("--recordings", type=int, default=4096, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true", help="N=1: skip other_precision / other_configs")
    ap.add_argument("--dry", action="store_true", help="no GPU: exercise launch/barrier/reduce/print only (gloo)")
    args = ap.parse_args()
    default_steps = {"c2": 200, "8k": 200, "stream": 2000, "corpus": 2}
    if args.steps is None:
        args.steps = default_steps[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_distributed(args)                      # does not return

    # the CPU baseline runs first, in its own CPU-only process, before this process touches the GPU
    want_cpu = int(os.environ.get("WORLD_SIZE", 1)) == 1 and not args.no_cpu_baseline and not args.dry
    cpu = cpu_baseline(16000 if args.config != "8k" else 8000) if want_cpu else None

    import torch
    rank, world, local, dist = setup_dist(args)
    if args.dry:
        out = run_dry(args, rank, world, dist)
    else:
        import __graft_entry__ as ge
        if rank == 0:
            ge.build()
        if world > 1:
            dist.barrier()
        torch.cuda.set_device(torch.device("cuda", local))
        extras = world == 1 and not args.no_extras
        if args.config in ("c2", "8k"):
            sr = 16000 if args.config == "c2" else 8000
            out = run_batch(args, sr, rank, world, local, dist, args.steps, with_other=extras)
        elif args.config == "stream":
            out = run_stream(args, rank, world, local, dist, args.steps)
        else:
            out = run_corpus(args, rank, world, local, dist, max(1, min(args.steps, 3)))
        if extras and args.config == "c2":              # the other BASELINE configs, short, for the record
            oc = {}
            for name, fn in (("8k", lambda: run_batch(args, 8000, rank, world, local, dist, 100, with_other=False)),
                             ("stream", lambda: run_stream(args, rank, world, local, dist, 1000)),
                             ("corpus", lambda: run_corpus(args, rank, world, local, dist, 1))):
                try:
                    oc[name] = small(fn())
                except Exception as e:                  # an extra must never cost the headline line
                    oc[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            out["other_configs"] = oc

    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
