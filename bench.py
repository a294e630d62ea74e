#!/usr/bin/env python3
"""Headline benchmark: speech-probability throughput of the Silero-VAD hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|8k|stream|stream_host|stream_8k|stream_host_8k|corpus|plumbing|plumbing_8k]

`--gpus N` (N > 1) works both ways the driver may start it: under `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` (RANK/LOCAL_RANK/WORLD_SIZE in the environment), and as a plain
`python bench.py --gpus N`, which re-executes itself under torch.distributed.run on 127.0.0.1.

Default workload `c2` (BASELINE.json configs[1], SURVEY.md section 8d "C2"): synthetic 16 kHz PCM, 512-sample
chunks, 4096 independent streams per GPU x 256 chunks per stream, fp32, already resident in HBM, fp32
arithmetic (`dtype: "f32"`, the reference's).  One "step" = one pass of the hot path over that batch = ONE
vad_forward_audio call through the C ABI (zeroed context/state, like the reference's audio_forward).  Streams
are sharded across ranks with no data-path collective (weak scaling: every rank owns 4096 streams); the only
communication is the barrier + MAX-reduce of the elapsed time that the measurement contract asks for.

Every leg CERTIFIES ITSELF: after its timed region the CPU oracle recomputes streams of the leg's own PCM (c2 / 8k: streams 0..15
and the last 16, all 256 chunks, probabilities and final (h, c); stream legs: 24 ticks of streams 0..15 through the leg's own graph;
corpus: every 10 000th recording + its segments) and the leg fails above 1e-4 (`parity`, `parity_max_abs_dp`).

The same JSON line also carries, for the record (none of them is the headline `value`):
  other_configs    8k           configs[2]: 8 kHz, 256-sample chunks, 4096 streams x 256 chunks (the 8 kHz net)            [N = 1]
                   stream       configs[4], KERNEL ONLY: 8192 live streams per GPU (65 536 per 8-GPU node), persistent state in
                                HBM, one hipGraph-captured vad_step per 32 ms tick from a device-side audio ring           [any N]
                   stream_host  configs[4] END TO END through the native pump (vad_pump_*, csrc/pump.hip; no Python on the tick path):
                                a source thread writes every tick's int16 chunks into a page-locked ring slot -> H2D copies and
                                fused step kernels on two streams ordered by events -> VADIterator logic of every stream -> events:
                                tick latency (slot written -> events) and sustained chunks/s against the int16 PCIe ceiling
                                (three ticks in flight unless two are clearly faster; median of three timed passes)            [any N]
                   corpus       configs[3]: every rank runs a FULL per-GPU shard of the 10 000 h corpus (1 250 h =
                                151 552 ragged recordings, 37 passes of 4096 over fresh offsets) from pinned host
                                memory -> device batch (no host copy) -> probs -> segmenter on the GPU -> segment
                                lists, sharded by duration and gathered to rank 0; wall time, PCIe- and host-inclusive,
                                with a 1-in-10^4 recording parity sample checked after the timed region                    [any N]
                   stream_8k, stream_host_8k, plumbing_8k: the 8 kHz net through the same legs                            [N = 1]
                   plumbing     configs[0]: the reference's default usage -- B = 1 `model(chunk, sr).item()` per-call
                                latency (eager and hipGraph) and get_speech_timestamps on the 60 s fixture                 [N = 1]
  other_arithmetic rec_bf16x9 / all_bf16x9: the same C2 workload with the opt-in exact bf16 x 9 products (N = 1)
  roofline         dominant kernel (frontend: STFT + encoder + W_ih GEMM): EXECUTED fp32 MFMA flops per launch /
                   average launch duration (hipEvents recorded by the engine around that kernel on the launch
                   stream during the timed steps) against the dense fp32 MFMA peak; always <= 1
  legs             LAST key: every leg's value, its fraction of the bound that applies, its own parity figure and max probability
  cpu_baseline     the reference's own ATen CPU operators (oracle/aten_port.py; kind "port", port "aten-operators") timed on this
                   box's host cores under BASELINE.md section 3 protocols R1-R5 (rank 0, before the process group forms; R5 =
                   get_speech_timestamps on the fixture through the per-chunk protocol, the CPU figure beside `plumbing`)
What rank 0 prints on stdout is ONE line of at most 8 192 bytes (`compact_line`: the contract's keys, `parity`, `roofline`, `cpu_baseline`,
`legs`); the full record with every leg's prose and detail goes to gpurun_out/bench_detail.json (`emit`; stderr only names the file).
`--config <name>` runs one of the other configs as the main leg instead.
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
# c10d warns once per rank that the container's hostname does not resolve (the rendezvous is 127.0.0.1); the driver's record is a tail
# of stdout + stderr, so stderr is kept quiet.  Read when torch's C++ side initialises: set before the first `import torch`.
os.environ.setdefault("TORCH_CPP_LOG_LEVEL", "ERROR")
sys.path.insert(0, str(ROOT))

STREAMS = 4096           # per GPU (c2 / 8k)
CHUNKS_PER_STREAM = 256  # per step
LIVE_STREAMS = 8192      # per GPU (stream): 65 536 per 8-GPU node
CLOCK_RAMP_STEPS = 40    # untimed steps before the warm-up: DVFS ramp, see run_batch
PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak (= fp32 vector peak)
PEAK_HBM_GBPS = 8000.0

# Work per chunk.  "flop"/"bytes": the reference's DENSE arithmetic and its I/O --
# "useful reference work".  "front_mfma" / "rec_mfma": matrix flops our kernels EXECUTE (rFFT frontend on the VALU
# instead of the DFT-basis conv, encoder 0 as ONE Winograd F(4,3) tile over the 4 frames (6 instead of 10 GEMMs), zero-padding
# taps skipped, the Nyquist bin applied on the VALU) = MFMA instructions per 16-chunk tile (SQ_INSTS_MFMA / tiles in
# profiles/) x 2048 flop / 16 -- the numerator of roofline.frac.
WORK = {
    16000: {"chunk": 512, "flop": 1_359_104, "bytes": 2_052,
            "front_dense": 2 * (264_192 + 198_144 + 49_152 + 12_288 + 24_576 + 65_536),
            "front_mfma": 54 * 64 * 2048 // 16,       # 3 456 MFMAs / tile: 54 weight units x 64
            # VALU issue cycles per tile beside the MFMAs (same pipe): 2 612 packed x 5.4 + 4 266 other x 2.6
            # (SQ_INSTS_VALU - SQ_INSTS_MFMA per tile, profiles/r02k_fp32_summary.md; rates profiles/r02d_issue_pipes.md,
            #  confirmed by profiles/r03a_issue_pipes2.md: an fp32 MFMA owns the SIMD's vector issue, the times add)
            "front_valu_cycles": 25_200,
            "rec_mfma": 2 * 512 * 128, "front_kernel": "front_f43_kernel<32, float>"},
    8000: {"chunk": 256, "flop": 767_232, "bytes": 1_028,
           "front_dense": 2 * (66_560 + 99_840 + 49_152 + 12_288 + 24_576 + 65_536),
           "front_mfma": 42 * 64 * 2048 // 16,        # 2 688 MFMAs / tile
           "front_valu_cycles": 12_500,                # 1 280 packed x 5.4 + 2 144 other x 2.6 (profiles/r02k_8k_summary.md)
           "rec_mfma": 2 * 512 * 128, "front_kernel": "front_f43_kernel<16, float>"},
}
NUMA_NODE = {}           # run_corpus / run_stream_host: the NUMA node this rank bound its host side to (per_rank record)
NATIVE_PINNED = {}       # bytes of page-locked memory held by native objects of this rank (the pump's ring)
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
    if world > 1 or os.environ.get("VAD_BENCH_FORCE_RCCL"):          # (forced at world 1: the RCCL rehearsal a one-GPU box can run)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):   # single node: rendezvous over loopback
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if args.dry:
            dist.init_process_group("gloo")
        elif os.environ.get("VAD_BENCH_SHARE_GPU"):
            # FUNCTIONAL test of the N > 1 legs on a box with ONE GPU (tests/test_gpu_parity.py): every rank drives device 0, the
            # barrier / MAX-reduce / gather go through gloo (RCCL refuses two ranks on one device).  Not a measurement.
            local = 0
            torch.cuda.set_device(0)
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
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
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
        return {"value": None, "unit": "chunks/s", "kind": "port", "port": "aten-operators", "error": err}
    d = json.loads(line)
    return {"value": d["value"], "unit": "chunks/s", "cores": d["nproc"], "affinity_cpus": d["affinity_cpus"],
            "kind": "port", "port": "aten-operators", "best_protocol": d["best"], "cpu_model": d["cpu_model"], "torch": d["torch"],
            "runs": d["runs"],
            "sample": f"same {sr // 1000} kHz synthetic workload (0.03 N(0,1)), audio_forward over B streams x T chunks "
                      f"per run as listed under runs; warm-up {d['warmup']}, median of {d['trials']}; R1 = 1 thread B=1 "
                      "(the reference's shipped default), R2 = 1 thread B=4096, R3 = nproc threads B=4096, "
                      "R4 = nproc processes x 1 thread sharing the 4096 streams; value = the best of the four; R5 (another workload, never "
                      "the value) = get_speech_timestamps on the reference's fixture through the per-chunk protocol, one thread"}


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


def live_pmc_traffic(config, kernels):
    """HBM bytes per launch of `kernels` measured NOW: two child runs of this same command under `rocprofv3 --pmc` -- FETCH_SIZE
    and WRITE_SIZE need separate passes (TCC counter budget), counters only, no trace domain -- each a few timed steps of the same
    workload; per-dispatch means of the last dispatches, corrected as MI355X_MICROARCH.md "HBM" prescribes (FETCH_SIZE counts KiB
    and, on gfx950, tallies the 128-byte requests of wide coalesced reads at 64 bytes: doubled; WRITE_SIZE in KiB as reported).
    Returns {kernel: {"bytes", "fetch_bytes", "write_bytes"}} or (None, reason)."""
    import csv
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="vad_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", VAD_BENCH_PMC_CHILD="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", out, "-o", "pmc", "--output-format", "csv", "--", sys.executable,
                   str(ROOT / "bench.py"), "--config", config, "--no-cpu-baseline", "--no-extras", "--no-parity", "--steps", "6", "--warmup", "1"]
            r = subprocess.run(cmd, env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=120)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            per = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != counter:
                        continue
                    for k in kernels:
                        if k in row["Kernel_Name"]:
                            per.setdefault(k, []).append(float(row["Counter_Value"]))
            for k, v in per.items():
                tail = v[-6:]                                  # the timed dispatches (the clock ramp's come first)
                vals.setdefault(k, {})[counter] = sum(tail) / len(tail)
    except Exception as e:  # noqa: BLE001 -- a profiler hiccup must not cost the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for k, c in vals.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            res[k] = {"bytes": int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), "fetch_bytes": int(2.0 * c["FETCH_SIZE"] * 1024),
                      "write_bytes": int(c["WRITE_SIZE"] * 1024)}
    return (res, None) if res else (None, "no dispatch of the kernels in the counter files")


PARITY_TOL = 1e-4        # BASELINE.json north_star; examples/openvino/verify.py:167


def certify(pcm_rows, got_probs, got_state, sr, what, init_state=None, init_ctx=None):
    """CHECKER, outside every timed region (protocol: examples/openvino/verify.py:157-181): the CPU oracle recomputes the given
    streams of THE BENCH'S OWN PCM over all their chunks; the leg's probabilities (and final (h, c) when given) must agree to the
    contract.  pcm_rows [S, L] float32 numpy, got_probs [S, T], got_state [2, S, 128] or None.  (oracle/ is test infrastructure:
    used here as the checker only, like `smoke()` does.)"""
    import numpy as np
    from oracle import Oracle
    want, _, wst = Oracle().forward_audio(np.ascontiguousarray(pcm_rows, dtype=np.float32), sr, ctx=init_ctx, state=init_state)
    got_probs = np.asarray(got_probs, dtype=np.float32)
    dp = float(np.abs(got_probs - want).max())
    out = {"checker": "oracle/vad_oracle.c on this leg's own PCM, after the timed region", "streams": what,
           "streams_checked": int(want.shape[0]), "chunks_checked": int(want.size), "parity_max_abs_dp": dp, "tolerance": PARITY_TOL,
           "max_prob": round(float(want.max()), 4)}
    ok = dp <= PARITY_TOL
    if got_state is not None:
        a, b = np.asarray(got_state, dtype=np.float64), np.asarray(wst, dtype=np.float64)
        out["final_state_max_rel_err"] = float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())
        ok = ok and out["final_state_max_rel_err"] <= PARITY_TOL
    out["ok"] = bool(ok)
    return out


def require_parity(par, leg):
    if not par["ok"]:
        raise RuntimeError(f"{leg}: parity check failed: {json.dumps(par)}")


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


def speech_rows(sr, streams, L, dev):
    """Real speech for the streams a leg checks itself on (the reference's fixtures, tests/golden/audio_*.npz, as float32 / 32768;
    stream i reads circularly from offset 40 chunks + i * 7919).  Synthetic noise keeps the oracle's probability under 0.05
    ("a saturated sigmoid that hides error"); timing does not depend on sample values, so the CHECKED
    streams carry speech and their probabilities span [0, 1]."""
    import numpy as np
    import torch
    pcm = np.load(ROOT / "tests" / "golden" / f"audio_{'16k' if sr == 16000 else '8k'}.npz")["pcm"]
    n = WORK[sr]["chunk"]
    idx = (40 * n + np.arange(streams, dtype=np.int64)[:, None] * 7919 + np.arange(L, dtype=np.int64)[None, :]) % len(pcm)
    return torch.from_numpy(pcm[idx].astype(np.float32) / 32768.0).to(dev)


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


def roofline(sr, chunks_per_launch, front_ms_avg, rec_ms_avg, B, T):
    """Dominant kernel = the frontend (STFT + encoder + W_ih).  achieved = matrix flops the kernel EXECUTES per
    launch / its average launch duration, peak = the dense fp32 MFMA peak (157.3 TF): frac <= 1.  The reference's
    dense flop count for the same part of the path ("useful work") is reported separately and is never divided into
    `frac`.  `traffic` is filled in by main() from a live rocprofv3 --pmc pass of this command (live_pmc_traffic; null if rocprofv3
    is not available, at N > 1, or with --no-extras); `traffic_profiled` quotes the same figure from the newest committed passes (profiles/);
    `path` relates the whole path (both kernels) to the algorithmic bytes of SURVEY 8(d) the same way."""
    w = WORK[sr]
    s = front_ms_avg / 1e3
    ex_flop = w["front_mfma"]
    execd = chunks_per_launch * ex_flop / s / 1e12
    kname = w["front_kernel"]
    tr_f = pmc_traffic(kname.split(",")[0], sr, B, T)
    tr_r = pmc_traffic("rec_kernel", sr, B, T)
    alg = chunks_per_launch * w["bytes"]
    kio = chunks_per_launch * (w["chunk"] * 4 + GX_BYTES)
    path_traffic = (tr_f["bytes"] + tr_r["bytes"]) if (tr_f and tr_r) else None
    note = "replayed from the committed rocprofv3 --pmc passes of this command (FETCH_SIZE x 2 + WRITE_SIZE); NOT measured by this run"
    out = {"bound": "mfma", "kernel": kname, "dtype": "f32",
           "achieved": round(execd, 3), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(execd / PEAK_F32_TFLOPS, 4),
           "flop_per_launch": chunks_per_launch * ex_flop, "avg_launch_ms": round(front_ms_avg, 4),
           "definition": "executed MFMA flops of the dominant kernel per launch / its hipEvent launch duration, "
                         "against the dense fp32 MFMA peak",
           "traffic": None,
           "traffic_profiled": dict(tr_f, note=note) if tr_f else None,
           "kernel_io_bytes": kio,
           "useful_dense": {"flop_per_launch": chunks_per_launch * w["front_dense"],
                            "tflops": round(chunks_per_launch * w["front_dense"] / s / 1e12, 3),
                            "note": "the reference's dense flop count for this part of the path (DFT-basis conv, every "
                                    "tap); the kernel executes fewer -- not a utilisation figure"},
           "issue_pipe": issue_pipe(w, chunks_per_launch, front_ms_avg),
           "hbm": {"achieved": round(kio / s / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                   "frac": round(kio / s / 1e9 / PEAK_HBM_GBPS, 4),
                   "note": "this kernel's own I/O (PCM in + gx out) against the HBM roofline: far from binding"},
           "path": {"algorithmic_bytes": alg, "algorithmic_bytes_per_chunk": w["bytes"],
                    "traffic": None,
                    "traffic_profiled": path_traffic,
                    "traffic_profiled_over_algorithmic": round(path_traffic / alg, 3) if path_traffic else None,
                    "traffic_profiled_detail": {"front": tr_f, "rec": tr_r, "note": note},
                    "mfma_flop_per_chunk": ex_flop + w["rec_mfma"]}}
    if rec_ms_avg:
        rs = rec_ms_avg / 1e3
        rflop = chunks_per_launch * w["rec_mfma"]
        out["rec_kernel"] = {"kernel": "rec_kernel", "avg_launch_ms": round(rec_ms_avg, 4),
                             "mfma_frac": round(rflop / rs / 1e12 / PEAK_F32_TFLOPS, 4),
                             "single_pipe_floor_note": "per step and SIMD 256 MFMAs (8 192 cycles) + the two waves' ~1 600 VALU cycles "
                                                       "add on one issue pipe (profiles/r03a_issue_pipes2.md): mfma_frac <= 0.84",
                             "gx_read_GBps": round(chunks_per_launch * GX_BYTES / rs / 1e9, 1),
                             "hbm_frac": round(chunks_per_launch * GX_BYTES / rs / 1e9 / PEAK_HBM_GBPS, 4)}
    return out


def base_line(args, world, metric_sr, value, elapsed, steps):
    return {"metric": f"audio-chunks/sec (32 ms @ {metric_sr // 1000} kHz)", "value": round(value, 1),
            "unit": "chunks/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic"}


# ---- c2 / 8k: HBM-resident batch ------------------------------------------------------------------------
def time_batch(eng, step, probs, world, dist, dev, steps, warmup):
    """Clock ramp, warm-up, `steps` timed calls with the engine's hipEvents on."""
    import torch
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


def run_batch(args, sr, rank, world, local, dist, steps, with_other=False):
    import torch
    from silero_vad_amd import Engine
    dev = torch.device("cuda", local)
    eng = Engine(device=local)
    n = WORK[sr]["chunk"]
    B, T = args.streams, args.chunks
    pcm = synth_pcm(B, T * n, sr, dev, 17 + sr + rank)
    sel = list(range(min(16, B))) + list(range(max(16, B - 16), B))          # the streams the leg checks itself on: real speech
    if not args.no_parity:
        pcm[torch.tensor(sel, device=dev)] = speech_rows(sr, len(sel), T * n, dev)
    ctx = torch.zeros((B, n // 8), device=dev)
    state = torch.zeros((2, B, 128), device=dev)
    probs = torch.empty((B, T), device=dev)
    eng.reserve(sr, B, T)

    def step():
        ctx.zero_()
        state.zero_()
        eng.forward_audio(pcm, sr, ctx, state, probs)

    elapsed, front_ms, rec_ms, ok = time_batch(eng, step, probs, world, dist, dev, steps, args.warmup)
    parity = None
    if not args.no_parity:
        # self-certification: streams 0..15 and the last 16 (speech), all T chunks, of the PCM that was just timed
        idx = torch.tensor(sel, device=dev)
        parity = certify(pcm[idx].cpu().numpy(), probs[idx].cpu().numpy(), state[:, idx].cpu().numpy(), sr,
                         f"streams 0..15 and {B - 16}..{B - 1} of {B} (real speech from the reference's fixture; the other streams carry "
                         f"synthetic noise), all {T} chunks")
    other = None
    if with_other and world == 1:
        # for the record, never the headline: the same workload with the recurrence's W_hh * h as exact bf16 x 9 piece products on
        # the bf16 matrix pipe (option rec=bf16x9, csrc/kernel_rec_b9.hip) -- the frontend stays the fp32 MFMA chain
        p_main = probs.clone()
        eng.set_option("rec", "bf16x9")
        try:
            e2, f2, r2, ok2 = time_batch(eng, step, probs, world, dist, dev, steps, args.warmup)
        finally:
            eng.set_option("rec", "fp32")
        other = {"rec_bf16x9": {"what": "recurrence only: three bf16 pieces per operand (exact), nine exact products, fp32 accumulation; "
                                        "opt-in, a different summation than the fp32 MFMA chain -- not the headline arithmetic",
                                "value": round(B * T * steps / e2, 1), "unit": "chunks/s", "ms_per_step": round(e2 / steps * 1e3, 4),
                                "kernel_ms": {"front": round(f2, 4), "rec": round(r2, 4)}, "outputs_finite": ok2,
                                "max_abs_prob_diff_vs_main": float((probs - p_main).abs().max().item()),
                                "study": "profiles/r03g_rec_bf16x9_study.json: both recurrences against float64, 8 input sets"}}
        # ... and with the frontend's matrix products the same way too (option front_mma=bf16x9, csrc/kernel_front_b9.hip)
        eng.set_option("rec", "bf16x9")
        eng.set_option("front_mma", "bf16x9")
        try:
            e3, f3, r3, ok3 = time_batch(eng, step, probs, world, dist, dev, steps, args.warmup)
        finally:
            eng.set_option("rec", "fp32")
            eng.set_option("front_mma", "fp32")
        other["all_bf16x9"] = {"what": "frontend GEMMs AND recurrence as exact bf16 x 9 piece products (FFT, transforms, activations fp32 as "
                                       "before); opt-in, not the headline arithmetic.  DESIGN.md 4.1c says why the frontend gains so little",
                               "value": round(B * T * steps / e3, 1), "unit": "chunks/s", "ms_per_step": round(e3 / steps * 1e3, 4),
                               "kernel_ms": {"front": round(f3, 4), "rec": round(r3, 4)}, "outputs_finite": ok3,
                               "max_abs_prob_diff_vs_main": float((probs - p_main).abs().max().item())}
    if rank != 0:
        return None
    w = WORK[sr]
    value = B * T * world * steps / elapsed
    out = base_line(args, world, sr, value, elapsed, steps)
    cfg = "configs[1]" if sr == 16000 else "configs[2]"
    out["config"] = {"workload": f"{cfg}: synthetic {sr // 1000} kHz PCM resident in HBM, {n}-sample chunks, "
                                 f"{B} streams/GPU x {T} chunks/stream per step, zero initial state",
                     "streams_per_gpu": B, "chunks_per_stream": T, "sample_rate": sr,
                     "sharding": f"streams x{world}, no collectives"}
    out["realtime_factor"] = round(value * 0.032, 1)
    out["outputs_finite"] = ok
    out["parity"] = parity
    if parity:
        out["parity_max_abs_dp"] = parity["parity_max_abs_dp"]
        if world == 1:
            require_parity(parity, f"{sr // 1000} kHz batch")
    out["clock_ramp_steps"] = max(0, CLOCK_RAMP_STEPS - args.warmup)
    out["timed_region_s"] = round(elapsed, 4)
    out["path_fraction"] = {"dense_flop_vs_fp32_peak": round(value / world * w["flop"] / (PEAK_F32_TFLOPS * 1e12), 4),
                            "algorithmic_bytes_vs_hbm_peak": round(value / world * w["bytes"] / (PEAK_HBM_GBPS * 1e9), 6),
                            "flop_per_chunk": w["flop"], "bytes_per_chunk": w["bytes"],
                            "note": "useful reference work (dense flops, SURVEY 8d) per second per GPU; informational"}
    out["kernel_ms"] = {"front": round(front_ms, 4), "rec": round(rec_ms, 4)}
    out["roofline"] = roofline(sr, B * T, front_ms, rec_ms, B, T)
    if other:
        out["other_arithmetic"] = other
    return out


# ---- stream: live streams, hipGraph step -------------------------------------------------------------------
def certify_pool(pool, feed, n_ticks, sr, what):
    """Self-certification of a stream pool: every slot restarts from zero state, `n_ticks` ticks of the leg's own audio go through
    the leg's own tick function `feed(k) -> probs[capacity]` (host or device), and the oracle recomputes streams 0..15.  `feed`
    also returns the float32 audio of those 16 streams for tick k."""
    import numpy as np
    pool.open_all()
    got, rows = [], []
    for k in range(n_ticks):
        p, x16 = feed(k)
        got.append(np.asarray(p[:16].cpu() if hasattr(p, "cpu") else p[:16], dtype=np.float32).copy())
        rows.append(x16)
    state = pool.state[:, :16].cpu().numpy()
    return certify(np.concatenate(rows, axis=1), np.stack(got, axis=1), state, sr, what)


def run_stream(args, rank, world, local, dist, steps, sr=16000):
    """KERNEL ONLY: the audio source is a ring in HBM, nothing crosses PCIe, nobody reads the probabilities (the `stream_host` leg is
    the end-to-end figure)."""
    import torch
    from silero_vad_amd import Engine, StreamPool
    dev = torch.device("cuda", local)
    eng = Engine(device=local)
    n = WORK[sr]["chunk"]
    cap = args.live
    pool = StreamPool(eng, sr, capacity=cap, graph=True)
    pool.open_all()
    ring = 8                                                   # device-side audio source, 8 ticks long
    flat = synth_pcm(cap, ring * n, sr, dev, 23 + rank)
    if not args.no_parity:
        flat[:16] = speech_rows(sr, 16, ring * n, dev)          # the checked streams carry speech
    src = flat.view(cap, ring, n).transpose(0, 1).contiguous()
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
    for _ in range(10):                                         # (the first eager launches create the events: not counted)
        pool._launch()
    eng.kernel_times()
    per = []
    for _ in range(8):                                          # eager launches carry the hipEvents; the MEDIAN group of 8 x 10 counts
        for _ in range(10):
            pool._launch()
        per.append(eng.kernel_times())
    per.sort(key=lambda x: x[0] / max(x[2], 1))
    front_ms, rec_ms, calls = per[len(per) // 2]
    eng.set_option("profile", "0")
    ok = bool(torch.isfinite(pool.prob).all().item())
    parity = None
    if not args.no_parity:
        def feed(j):
            pool.pcm.copy_(src[j % ring])
            pool.tick_staged()
            return pool.prob.clone(), src[j % ring][:16].cpu().numpy()
        parity = certify_pool(pool, feed, 3 * ring, sr, f"streams 0..15 of {cap}, {3 * ring} ticks from zero state through the captured graph")
    if rank != 0:
        return None
    value = cap * world * steps / elapsed
    out = base_line(args, world, sr, value, elapsed, steps)
    out["config"] = {"workload": f"configs[4], KERNEL ONLY (audio source in HBM; see stream_host for host chunks in -> events out): {cap} live "
                                 f"{sr // 1000} kHz streams/GPU ({cap * 8} per 8-GPU node), one hipGraph-captured vad_step per 32 ms tick, "
                                 "(h,c)+context persistent in HBM",
                     "streams_per_gpu": cap, "sample_rate": sr, "step": "one tick (one chunk per stream)",
                     "sharding": f"streams x{world}, no collectives"}
    out["realtime_factor"] = round(value * 0.032, 1)
    out["outputs_finite"] = ok
    out["parity"] = parity
    if parity and world == 1:
        require_parity(parity, f"stream {sr // 1000} kHz")
    out["tick_latency_ms"] = {"median": round(lat[len(lat) // 2], 4), "p95": round(lat[int(len(lat) * 0.95)], 4),
                              "budget_ms": 32.0}
    c = max(calls, 1)
    out["kernel_ms"] = {"front": round(front_ms / c, 4), "rec": round(rec_ms / c, 4),
                        "note": "one step in ONE kernel (latency frontend + LSTM cell + head, csrc/kernel_front_lat.hip): `front` is that kernel, "
                                "`rec` only the gap between the two event records"}
    rl = roofline(sr, cap, front_ms / c, 0.0, cap, 1)
    # the fused kernel executes the frontend's AND the recurrence's matrix flops
    w = WORK[sr]
    fl = cap * (w["front_mfma"] + w["rec_mfma"])
    rl.update({"kernel": f"front_lat_kernel<{32 if sr == 16000 else 16}, float, 1, true> (frontend + LSTM cell + head)", "flop_per_launch": fl,
               "achieved": round(fl / (front_ms / c / 1e3) / 1e12, 3)})
    rl["frac"] = round(rl["achieved"] / PEAK_F32_TFLOPS, 4)
    out["roofline"] = rl
    return out


def fixture_rows_i16(sr, cap, L):
    """`cap` streams of real speech for the host-fed legs: the reference's own fixture (tests/data/test.wav / examples/c++/aepyx_8k.wav,
    committed as tests/golden/audio_*.npz), stream b reading it circularly from offset b * 7919 (SURVEY.md section 8d set iii)."""
    import numpy as np
    pcm = np.load(ROOT / "tests" / "golden" / f"audio_{'16k' if sr == 16000 else '8k'}.npz")["pcm"]
    idx = (np.arange(cap, dtype=np.int64)[:, None] * 7919 + np.arange(L, dtype=np.int64)[None, :]) % len(pcm)
    return pcm[idx]


def h2d_rate_GBps(dev, nbytes=256 << 20, reps=5):
    """What the host link gives a plain pinned -> HBM copy on this box, now (the ceiling the host-fed legs are held against)."""
    import torch
    src = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
    return best


def gap_flags(ticks, streams, seed, miss=0.10, max_burst=5):
    """[ticks, streams] uint8 presence pattern of live streams that do not arrive in lock step: every stream independently misses about
    `miss` of its ticks, in bursts of 1..max_burst ticks (a late or lost packet train)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    p_start = miss / (1.0 - miss) / ((1 + max_burst) / 2.0)
    pat = np.ones((ticks, streams), np.uint8)
    left = np.zeros(streams, np.int64)                          # ticks of the current burst still to miss
    for t in range(ticks):
        start = (left == 0) & (rng.random(streams) < p_start)
        left[start] = rng.integers(1, max_burst + 1, size=int(start.sum()))
        pat[t, left > 0] = 0
        left[left > 0] -= 1
    return pat


def run_stream_host(args, rank, world, local, dist, ticks, sr=16000, gaps=0.0):
    """configs[4] END TO END through the native pump (include/silero_vad_hip.h "live streams: the pump", csrc/pump.hip; the shape of
    the reference's native streaming loop, examples/cpp/silero-vad-onnx.cpp:335-390): NO Python and no torch on the tick path.  Per
    tick a source thread WRITES the int16 chunks of all streams into a page-locked ring slot (host memory traffic included: this is
    what an audio server's receive threads do), the server thread submits the tick -- H2D copies on the copy stream, step kernels on
    the compute stream, ordered by events, probabilities stored by the kernels straight into host memory -- and retires it with the
    VADIterator logic of every stream (events).  Two measurements, both inside vad_pump_play:
      latency    one tick at a time (depth 1): slot written -> events on the host, median / p95
      sustained  ticks in flight (depth 2 and 3, the better one is the value): chunks/s against the int16 PCIe ceiling of this box"""
    import numpy as np
    import torch
    from silero_vad_amd import Engine, StreamPump
    dev = torch.device("cuda", local)
    n = WORK[sr]["chunk"]
    cap = args.live
    R = 4
    period = 32                                                 # chunks of audio per stream the sources cycle through (268 MB at 16 kHz)
    parts = max(1, int(os.environ.get("VAD_BENCH_STREAM_PARTS", "1")))
    rows = np.ascontiguousarray(fixture_rows_i16(sr, cap, period * n))     # real speech: the iterators do produce events
    # Let the device settle behind the previous leg: what that leg freed (engine scratch of its lanes, GBs) is WIPED by the driver on
    # some boxes -- there a 4 GiB hipMalloc takes 120-360 ms instead of 0.4 (tools/r05_boxes.sh) -- and the wipe runs on the copy
    # engines this leg's H2D copies need: measured right behind a 20 GB release the pump reads 0.56 of the link and 0.40 ms p95 instead
    # of 0.87 and 0.19 (profiles/r05_ingest_routes.md section 3).  Untimed, like every leg's warm-up.
    import gc
    gc.collect()
    torch.cuda.synchronize()
    time.sleep(float(os.environ.get("VAD_BENCH_SETTLE_S", "1.0")))
    eng = Engine(device=local)
    from silero_vad_amd import _lib
    NUMA_NODE["node"] = _lib.lib().vad_bind_host_to_device(local)   # the ring and the source threads on the GPU's NUMA node
    pump = StreamPump(eng, sr, streams=cap, parts=parts, ring_slots=R)
    NATIVE_PINNED["bytes"] = max(NATIVE_PINNED.get("bytes", 0), R * cap * (n * 2 + 4))
    tick0 = [0]
    # gaps > 0: the streams do not arrive in lock step -- each misses `gaps` of its ticks in bursts of up to 5 (vad_pump_play_gaps: the
    # sources write the flags, absent streams are not stepped, a stream's audio advances only when it delivers)
    pattern = gap_flags(256, cap, 41 + rank, gaps) if gaps > 0 else None
    if pattern is None and os.environ.get("VAD_BENCH_ALL_PRESENT_FLAGS"):      # (experiment: flagged ticks in which everybody delivers)
        pattern = np.ones((4, cap), np.uint8)
    # ... and the sources write COMPACT slots (vad_pump_play_compact): only the delivering streams' rows cross the link, so a tick's link
    # cost falls with the delivery rate.  The full-row form (vad_pump_play_gaps) is timed behind it, for the record.
    compact = pattern is not None and not os.environ.get("VAD_BENCH_GAPS_FULL_ROWS")

    def play(nt, depth, packed=None):
        ev, st = pump.play(rows, nt, first_tick=tick0[0], depth=depth, pattern=pattern, compact=compact if packed is None else packed)
        tick0[0] += nt
        return st

    play(600, 2)                                                # warm-up + clock ramp
    lat = play(400, 1)
    # how many ticks to keep in flight: tried untimed, the better one is used for the timed region
    # (two passes of 500 ticks each, the faster pass of a depth counts: a single 50 ms pass picked the slower depth on one run in three)
    trial = {}
    for _ in range(2):
        for d in (2, 3):
            st = play(500, d)
            if d not in trial or st["wall_ms"] < trial[d]["wall_ms"]:
                trial[d] = st
    # three ticks in flight unless two are clearly faster: with three batch buffers depth 3 is at least as fast in a same-lease A/B
    # (tools/r06_pump_ab.py), and a 2 % difference between two 80 ms passes is noise
    depth = 2 if trial[2]["wall_ms"] < 0.95 * trial[3]["wall_ms"] else 3
    if os.environ.get("VAD_BENCH_STREAM_DEPTH"):                # (A/B knob: profiles/r06_pump_three_buffers.md)
        depth = int(os.environ["VAD_BENCH_STREAM_DEPTH"])
    runs = {f"depth{d}": {"ticks": 500, "wall_ms": round(st["wall_ms"], 2), "tick_ms_p50": round(st["tick_ms_p50"], 4),
                          "tick_ms_p95": round(st["tick_ms_p95"], 4)} for d, st in trial.items()}
    # THREE timed passes of `ticks` ticks each, the median counts (all three are in the record): a leg that lasts 0.2-0.3 s on a shared
    # host reads 10-20 % low when another tenant's burst falls into its one pass (r06K: 86.6 M where the untimed trial and four other
    # leases read 104-110 M)
    passes = []
    for _ in range(3):
        b_ = {}
        passes.append((timed(world, dist, dev, 1, lambda: b_.update(play(ticks, depth)), gpu_sync), b_))
    passes.sort(key=lambda x: x[0])
    elapsed, best = passes[1]
    timed_passes_s = [round(x[0], 4) for x in passes]
    full_rows = None
    if compact:
        fr = {}
        play(100, depth, packed=False)
        t_full = timed(world, dist, dev, 1, lambda: fr.update(play(ticks, depth, packed=False)), gpu_sync)
        full_rows = {"value": round(fr["chunks"] * world / t_full, 1), "ticks_per_s": round(ticks / t_full, 1), "tick_ms_p95": round(fr["tick_ms_p95"], 4),
                     "what": "the same ticks with every stream's row crossing the link (vad_pump_play_gaps)"}
    link = h2d_rate_GBps(dev)
    parity = None
    if not args.no_parity:
        # streams 0..15 restart from zero state, 24 ticks of the leg's own audio go through the pump; the oracle recomputes them
        nt = 24
        for b in range(16):
            pump.open_stream(b)
        got = np.zeros((16, nt), np.float32)
        pos = np.zeros(cap, np.int64)                           # chunks each stream has delivered
        t = 0
        while pos[:16].min() < nt:
            r = t % R
            fl = np.ones(cap, np.uint8) if pattern is None else pattern[t % len(pattern)].copy()
            fl[:16] &= pos[:16] < nt                            # (a checked stream stops after its nt chunks)
            on = np.flatnonzero(fl)
            pump.slot(r)[np.arange(len(on)) if compact else on] = rows[on[:, None], ((pos[on] % period) * n)[:, None] + np.arange(n)[None, :]]
            pump.submit(r, present=None if pattern is None else fl, compact=compact)
            pump.poll()
            p16 = pump.probs(r)[:16]
            d = np.flatnonzero(fl[:16])
            got[d, pos[d]] = p16[d]
            if pattern is not None and not (p16[fl[:16] == 0] == -1.0).all():
                raise RuntimeError("stream_gaps: an absent stream's probability slot does not hold VAD_PROB_ABSENT")
            pos += fl
            t += 1
        st16 = np.stack([np.stack(pump.state(b)[:2]) for b in range(16)], 1)       # [2, 16, 128]
        how = (f"{nt} ticks" if pattern is None else f"{nt} delivered chunks each over {t} ticks with ~{gaps:.0%} of them missed (present flags)")
        parity = certify(rows[:16, :nt * n].astype(np.float32) / 32768.0, got, st16, sr,
                         f"streams 0..15 of {cap}, {how} of int16 speech from zero state: ring slot -> pump (copies + kernels by events) -> host")
    pump.close()
    if rank != 0:
        return None
    delivered = best["chunks"] / (cap * ticks)                  # fraction of (stream, tick) pairs that carried a chunk
    value = best["chunks"] * world / elapsed                    # chunks that were stepped per second
    out = base_line(args, world, sr, value, elapsed, ticks)
    out["ms_per_step"] = round(elapsed / ticks * 1e3, 4)
    if pattern is not None:
        out["gaps"] = {"missed_fraction": round(1.0 - delivered, 4), "max_burst": 5, "ticks_per_s": round(ticks / elapsed, 1),
                       "compact_slots": bool(compact), "full_rows": full_rows,
                       "what": "every stream independently misses ticks; absent streams keep (h, c), context and iterator state "
                               "(vad_pump_submit_compact: only the delivering streams' rows cross the link)"}
    out["config"] = {"workload": f"configs[4] END TO END{' WITH GAPS' if pattern is not None else ''}: {cap} live {sr // 1000} kHz streams/GPU ({cap * 8} per 8-GPU node), native pump: a source "
                                 f"thread writes every tick's int16 chunks into a page-locked ring slot -> H2D ({parts} parts, copy stream) -> fused "
                                 "vad_step kernels (compute stream, ordered by events; probabilities stored straight into host memory) -> VADIterator "
                                 "logic of every stream (vad_iterator_feed) -> events; (h,c)+context persistent in HBM; real-speech fixture audio; no "
                                 "Python on the tick path",
                     "streams_per_gpu": cap, "sample_rate": sr, "step": "one tick (one chunk per stream)", "parts": parts, "ring_slots": R,
                     "sharding": f"streams x{world}, no collectives"}
    out["dtype"] = "i16"
    out["realtime_factor"] = round(value * 0.032, 1)
    out["events_emitted"] = {"latency_pass": lat["events"], "sustained_pass": best["events"]}
    out["tick_latency_ms"] = {"median": round(lat["tick_ms_p50"], 4), "p95": round(lat["tick_ms_p95"], 4), "max": round(lat["tick_ms_max"], 4),
                              "budget_ms": 32.0, "what": "one tick at a time (depth 1): ring slot completely written -> its events on the host"}
    out["sustained"] = {"depth": best["depth"], "fill_threads": best["fill_threads"], "tick_ms_p50": round(best["tick_ms_p50"], 4),
                        "tick_ms_p95": round(best["tick_ms_p95"], 4), "host_ms_per_tick": {"source_writes_slot": round(best["fill_ms_mean"], 4),
                        "submit": round(best["submit_ms_mean"], 4), "blocked_in_poll": round(best["wait_ms_mean"], 4)}, "untimed_depth_trials": runs,
                        "native_wall_s": round(best["wall_ms"] / 1e3, 4), "timed_passes_s": timed_passes_s, "value_is": "median of three timed passes"}
    ceiling = link * 1e9 / (n * 2) * world
    # (full-row ticks with gaps: the whole slot crosses the link every tick, so the link is held against the ticks; compact ticks carry
    #  the delivering streams' rows and a 5-byte-per-stream header, so it is held against the delivered chunks)
    moved = best["chunks"] / ticks if compact else cap
    out["pcie"] = {"h2d_GBps_plain_copy": round(link, 2), "int16_ceiling_chunks_per_s": round(ceiling, 1),
                   "fraction_of_pcie_ceiling": round(moved * world * ticks / elapsed / ceiling, 3),
                   "bytes_per_tick": int(moved * n * 2)}
    out["parity"] = parity
    if parity and world == 1:
        require_parity(parity, f"stream_host {sr // 1000} kHz")
    return out


# ---- corpus: a full per-GPU shard of the 10 000 h corpus, from pinned host memory -----------------------------
CORPUS_HOURS_PER_GPU = 10_000.0 / 8          # BASELINE configs[3]: 10 000 h over the 8 GPUs of a node
CORPUS_PASS = 4096                           # recordings per pass and rank (one ragged_speech_segments call)


def corpus_shard(rank, world, passes, per_pass, sr, base_len, seed=101):
    """The corpus as (length, offset) of every recording, pass by pass: each pass holds world x per_pass recordings of
    20-40 s, dealt to the ranks by total duration (sharding.shard_by_duration: deterministic, every rank computes the
    same partition without communicating).  Returns this rank's [(global ids, lengths, offsets)] per pass."""
    import numpy as np
    from silero_vad_amd import shard_by_duration
    out = []
    for p in range(passes):
        rng = np.random.default_rng(seed + p)               # the same corpus on every rank
        lens = rng.integers(20 * sr, 40 * sr, size=world * per_pass)
        offs = rng.integers(0, base_len - 40 * sr, size=world * per_pass) // 8 * 8     # 16-byte aligned int16 rows
        mine = np.asarray(shard_by_duration(lens, world, rank) if world > 1 else np.arange(per_pass), dtype=np.int64)
        out.append((mine + p * world * per_pass, lens[mine], offs[mine]))
    return out


def corpus_parity_sample(model, gids, lens, offs, base_page, counts, segs, sr, every=10_000):
    """CHECKER, outside the timed region ("parity on a 1-in-10^4 stream sample"): every 10 000th recording
    of the corpus is recomputed alone through `audio_forward` and through the CPU oracle; its probabilities must agree to
    the contract and the segments the corpus run returned for it must equal the segments scanned from the oracle's
    probabilities.  (oracle/ is test infrastructure: used here as the checker only, like `smoke()` does.)"""
    import numpy as np
    import torch
    from oracle import Oracle
    from silero_vad_amd import segment_probs
    orc = Oracle()
    first = np.concatenate([[0], np.cumsum(counts)])
    worst, checked, same = 0.0, 0, True
    k = sr // 16000 if sr > 16000 else 1                            # a raw 32 / 48 kHz corpus: the oracle gets x[::k], like the reference's front door
    sample = np.flatnonzero((gids % every == 0) & (lens > 0))
    if not len(sample):                                             # a small shard that holds no 10 000th recording: its first one
        sample = np.flatnonzero(lens > 0)[:1]
    for j in sample:
        a = base_page[int(offs[j]):int(offs[j]) + int(lens[j])]      # (the arena the corpus run read)
        x = a.to(torch.float32) / 32768.0
        got = model.audio_forward(x[None], sr)[0].numpy()
        xd = np.ascontiguousarray(x.numpy()[::k])
        want = orc.audio_forward(xd[None], sr // k)[0]
        worst = max(worst, float(np.abs(got - want).max()))
        mine = [{"start": int(p), "end": int(q)} for p, q in segs[first[j]:first[j + 1]]]
        same = same and mine == segment_probs(want, len(xd), sr // k)
        checked += 1
    return {"recordings_checked": checked, "one_in": every, "parity_sample_max_abs_dp": worst if checked else None,
            "segments_identical_to_oracle_scan": bool(same), "tolerance": 1e-4}


def run_corpus(args, rank, world, local, dist, passes, sr=16000, main_only=False):
    """sr = 48000: the same corpus as RAW 48 kHz recordings (a chunk is 1 536 of their samples: three times the bytes per chunk over
    the link): nothing decimates on the host, the frontend's loads take every third sample."""
    import numpy as np
    import torch
    from silero_vad_amd import PackedRecordings, _lib, gather_to_rank0, load_silero_vad, ragged_speech_segments
    from silero_vad_amd import streams as S
    dev = torch.device("cuda", local)
    node = _lib.lib().vad_bind_host_to_device(local)            # staging threads + pinned buffers on the GPU's NUMA node
    NUMA_NODE["node"] = node
    host_threads = _lib.lib().vad_host_threads()
    model = load_silero_vad(device=local)
    if os.environ.get("VAD_BENCH_REC_FORM"):                # A/B of the recurrence's form (results are bit-identical)
        model.engine.set_option("rec_form", os.environ["VAD_BENCH_REC_FORM"])
    n = 512 * (sr // 16000)                                  # input samples per chunk
    rng = np.random.default_rng(7)
    base_len = 8 << 20
    tt = np.arange(base_len, dtype=np.float32) / sr
    base = (0.03 * rng.standard_normal(base_len).astype(np.float32)
            + 0.2 * np.sin(2 * np.pi * 170.0 * tt) * (np.sin(2 * np.pi * 0.7 * tt) > 0))
    base_i_page = torch.from_numpy((base * 32767.0).clip(-32768, 32767).astype(np.int16))
    R = max(1, args.recordings // (sr // 16000))            # (a pass -- the pinned arena -- holds the same bytes at every rate)
    shard = corpus_shard(rank, world, passes, R, sr, base_len)
    gids = np.concatenate([g for g, _, _ in shard])
    lens = np.concatenate([l for _, l, _ in shard])
    offs_rand = np.concatenate([o for _, _, o in shard])        # views into the small pageable signal (the staged leg)
    hours = float(lens.sum()) / sr / 3600.0
    # The decoder's output arena: page-locked, refilled pass by pass (a ring the size of one pass); the recordings of a pass lie
    # in it back to back, each starting on a 16-byte boundary.
    offs = np.zeros(len(lens), dtype=np.int64)
    span = 0
    for p in range(passes):
        sl = slice(sum(len(x[1]) for x in shard[:p]), sum(len(x[1]) for x in shard[:p + 1]))
        l = lens[sl]
        o = np.concatenate([[0], np.cumsum((l + 7) // 8 * 8)[:-1]])
        offs[sl] = o
        span = max(span, int(o[-1] + l[-1]))
    arena_len = (span + base_len - 1) // base_len * base_len
    arena = torch.empty(arena_len, dtype=torch.int16, pin_memory=True)
    arena.view(-1, base_len)[:] = base_i_page
    base_i = arena

    def run_leg(src, sched, mode, nrec, keep):
        """One call over the first `nrec` recordings of this rank's shard: the pipeline (upload of bucket k+1 beside the
        kernels of bucket k, two compute lanes, scan on the device) fills once and drains once."""
        os.environ["SILERO_VAD_AMD_UPLOAD"] = mode
        res = {}
        nonlocal model
        if os.environ.get("VAD_BENCH_CORPUS_FRESH"):            # diagnostic: every leg on a fresh model (new lanes, streams, staging pool)
            model = load_silero_vad(device=local)
        if os.environ.get("VAD_BENCH_CORPUS_EMPTY_CACHE"):      # diagnostic: hand the previous leg's device blocks back to the driver
            torch.cuda.synchronize()
            torch.cuda.empty_cache()

        def one(m):
            rec = PackedRecordings(src, (offs if src is base_i else offs_rand)[:m], lens[:m])
            if sched == "buckets":      # length-sorted buckets, one lock-step call each, device scan per bucket
                return ragged_speech_segments(rec, model, sr, max_waste=0.1, max_bytes=int(os.environ.get("VAD_BENCH_BUCKET_BYTES", 1 << 30)), as_arrays=True)
            # persistent slots, refilled at slab boundaries; every recording is scanned behind the slab it retires in and its segments
            # come back while later slabs run (refill_segments_stream): when did the FIRST results reach the host?
            rs, rc_ = (int(v) for v in os.environ.get("VAD_BENCH_REFILL", "2048,128").split(","))
            counts = np.zeros(m, dtype=np.int64)
            t_in = time.perf_counter()
            res["first_result_s"] = None
            for idx, cnt, _ in S.refill_segments_stream(rec, model, sr, slots=rs, slab_chunks=rc_):
                if res["first_result_s"] is None and len(idx):
                    res["first_result_s"] = time.perf_counter() - t_in
                    res["first_result_slabs"] = int(S.STATS.get("buckets", 0))      # slabs handed to the GPU by then
                counts[idx] = cnt
            return counts, None

        one(min(nrec, 2 * R))                                   # warm-up: pinned buffers, scratch, lanes
        if sched == "buckets":
            # ... and everything the FULL plan needs, allocated before the timed region (ragged_reserve: the lanes' scratch for the
            # largest bucket of all nrec recordings, the staging slots): a scratch growth inside the run is a device synchronisation plus a
            # multi-GB hipFree / hipMalloc, 1 ms on most boxes and 120-180 ms on others (profiles/r05_ingest_routes.md)
            S.ragged_reserve(PackedRecordings(src, (offs if src is base_i else offs_rand)[:nrec], lens[:nrec]), model, sr, max_waste=0.1,
                             max_bytes=int(os.environ.get("VAD_BENCH_BUCKET_BYTES", 1 << 30)))
        else:                                                    # ... the refill route's: window buffers, slots, scratch (refill_reserve)
            rs, rc_ = (int(v) for v in os.environ.get("VAD_BENCH_REFILL", "2048,128").split(","))
            S.refill_reserve(PackedRecordings(src, (offs if src is base_i else offs_rand)[:nrec], lens[:nrec]), model, sr, slots=rs, slab_chunks=rc_)
        S.STATS.clear()
        nseg = []

        # the timed region also holds the host-side gather of the results to rank 0 (north_star: "only a host-side gather
        # of timestamps"): two compact arrays per rank, one gather_object at the end of the shard
        def whole():
            counts, segs = one(nrec)
            res["counts"], res["segs"] = counts, segs
            nseg.append(int(counts.sum()))
            if keep and world > 1:
                gathered = gather_to_rank0((gids[:nrec], counts, segs))
                if gathered is not None:
                    nseg.append(sum(int(g[1].sum()) for g in gathered))

        elapsed = timed(world, dist, dev, 1, whole, gpu_sync)
        st = dict(S.STATS)
        leg_chunks = int(((lens[:nrec] + n - 1) // n).sum())
        leg_bytes = float(lens[:nrec].sum()) * 2
        d = {"scheduler": sched, "source": "pinned" if src is base_i else "pageable", "upload": mode,
             "recordings_per_gpu": int(nrec), "audio_hours_per_gpu": round(float(lens[:nrec].sum()) / sr / 3600.0, 2),
             "value": round(leg_chunks * world / elapsed, 1), "unit": "chunks/s", "wall_s": round(elapsed, 4),
             "segments_found_rank0": nseg[0], "segments_gathered_all_ranks": nseg[1] if len(nseg) > 1 else None,
             "ingest_GBps_per_gpu": round(leg_bytes / elapsed / 1e9, 2),
             "h2d_GBps_while_copying": (round(st["h2d_bytes"] / st["h2d_s"] / 1e9, 2) if st.get("h2d_s") else None),
             # every recording's bytes cross the link at least once (the arena is a ring that is refilled: equal offsets in different
             # passes are different audio): must be >= 1
             "bytes_copied_over_live_bytes": round(st.get("h2d_bytes", 0) / max(leg_bytes, 1), 4),
             "host_stage_ms": round(st.get("stage_s", 0) * 1e3, 2),
             "host_upload_call_ms": round(st.get("upload_call_s", 0) * 1e3, 2),
             "host_segmenter_ms": round(st.get("scan_s", 0) * 1e3, 2),
             "host_ms": {k[:-2]: round(st.get(k, 0) * 1e3, 2) for k in ("setup_s", "reserve_s", "slot_wait_s", "slot_alloc_s", "result_wait_s")},
             "slot_allocs": int(st.get("slot_allocs", 0)), "stream_retries": int(st.get("stream_retries", 0)),
             "d2h_MB": round(st.get("d2h_bytes", 0) / 1e6, 3),
             "buckets": int(st.get("buckets", 0)), "window_buffers": int(st.get("refill_window_buffers", 0)),
             "padded_over_real_samples": round(st.get("padded", 0) / max(st.get("real", 1), 1), 4)}
        if res.get("first_result_s") is not None:              # the refill scheduler hands results over as recordings retire
            d["first_result_at"] = round(res["first_result_s"] / elapsed, 4)     # fraction of the leg's wall time (planning included)
            d["first_result_after_slabs"] = [res.get("first_result_slabs"), int(st.get("buckets", 0))]   # ... of the shard's slabs
        return d, res

    legs = {}
    main_mode = os.environ.get("VAD_BENCH_CORPUS_UPLOAD", "window")
    short = min(len(lens), 6 * R)       # the comparison legs: 6 passes (0.5 s each: the pipeline's fill and drain, ~20 ms, are 4 % of that)
    if os.environ.get("VAD_BENCH_CORPUS_PRELEG"):               # diagnostic: one short leg BEFORE the main one (order dependence)
        pre = os.environ["VAD_BENCH_CORPUS_PRELEG"]
        legs[f"pre_{pre}"], _ = run_leg(base_i, "buckets", pre, short, False)
    if os.environ.get("VAD_BENCH_REFILL_FIRST"):               # diagnostic: the refill leg before anything else has run (order dependence)
        legs["pre_refill"], _ = run_leg(base_i, "refill", "gather", min(len(lens), 18 * R), False)
    legs["main"], res = run_leg(base_i, "buckets", main_mode, len(lens), True)
    parity = None
    if not args.no_parity:
        parity = corpus_parity_sample(model, gids, lens, offs, base_i, res["counts"], res["segs"], sr)
    if not (args.corpus_main_only or main_only):                # the other ingest routes on 6 passes' worth, for comparison
        for other in (() if os.environ.get("VAD_BENCH_ONLY_REFILL") else ("window", "gather", "dma")):
            if other != main_mode:
                legs[f"pinned_{other}"], _ = run_leg(base_i, "buckets", other, short, False)
        legs["pageable_staged"], _ = run_leg(base_i_page, "buckets", "stage", short, False)
        # (the whole shard: recordings are admitted longest first, so the first ones retire ten slabs in -- 2-3 % of a full shard's wall time,
        #  setup included; on six passes that would be 11 %)
        if not os.environ.get("VAD_BENCH_SKIP_REFILL"):           # (diagnostic knob: the bucket routes only)
            legs["pinned_refill_gather"], _ = run_leg(base_i, "refill", "gather", len(lens), False)
        # the same scheduler fed from arena windows: recordings admitted in arena order, one DMA per 256 MB window a few slabs ahead of
        # its readers, the slabs' rows cut from the windows' device copies (streams._refill_iter)
        if not os.environ.get("VAD_BENCH_SKIP_REFILL"):
            legs["pinned_refill_window"], _ = run_leg(base_i, "refill", "window", len(lens), False)
    os.environ.pop("SILERO_VAD_AMD_UPLOAD", None)
    # what the link allows: the H2D rate measured while copying / bytes per chunk -- a leg's value can approach it (fully
    # overlapped pipeline), never exceed it
    link = max((l["h2d_GBps_while_copying"] or 0) for l in legs.values()) or None
    for l in legs.values():
        l["pcie_ceiling_chunks_per_s"] = round(link * 1e9 / (n * 2) * world, 1) if link else None
        l["fraction_of_pcie_ceiling"] = round(l["value"] / l["pcie_ceiling_chunks_per_s"], 3) if link else None
    if rank != 0:
        return None
    main = legs["main"]
    out = base_line(args, world, sr, main["value"], main["wall_s"], 1)
    out["ms_per_step"] = round(main["wall_s"] * 1e3, 3)
    full = abs(hours - CORPUS_HOURS_PER_GPU) / CORPUS_HOURS_PER_GPU < 0.05
    out["config"] = {"workload": f"configs[3]{'' if sr == 16000 else f' as RAW {sr // 1000} kHz int16 (x[::{sr // 16000}] in the loads)'}: "
                                 f"{'the full' if full else 'a PARTIAL'} per-GPU shard of the 10 000 h corpus -- "
                                 f"{passes} x {R} ragged recordings/GPU (20-40 s, {hours:.1f} h of audio per GPU) back to back in a pinned "
                                 "host arena -> one DMA per 2 GiB arena window -> padded batches cut on the GPU (no host copy) -> probs -> "
                                 "segmenter on the GPU -> segment lists to the host -> gathered to rank 0; ONE step = the whole shard; "
                                 "PCIe- and host-inclusive wall time; int16 PCM, bucket scheduler, two compute lanes",
                     "recordings_per_gpu": int(len(lens)), "audio_hours_per_gpu": round(hours, 2), "sample_rate": sr,
                     "sharding": f"recordings x{world} by duration (shard_by_duration), no collectives; results gathered to rank 0",
                     "host_threads_per_rank": host_threads, "numa_node_bound": node}
    out["realtime_factor"] = round(main["value"] * 0.032, 1)
    out["wall_s"] = main["wall_s"]
    out["audio_hours_all_gpus"] = round(hours * world, 1)
    out["ten_k_hours_at_this_rate_s"] = round(10_000.0 / (hours * world / main["wall_s"]), 1)
    out["parity_sample"] = parity
    if parity:
        out["parity_sample_max_abs_dp"] = parity["parity_sample_max_abs_dp"]
    out["legs"] = legs
    return out


# ---- plumbing: the reference's default usage, one chunk per call (configs[0]) -----------------------------
def run_plumbing(args, local, sr=16000):
    """B = 1: what an unmodified caller does -- `model(chunk, sr).item()` once per 32 ms chunk
    (src/silero_vad/utils_vad.py:324-336, :528; the reference advertises "< 1 ms per chunk", README.md:103)."""
    import numpy as np
    import torch
    from silero_vad_amd import Engine, StreamPool, get_speech_timestamps, load_silero_vad
    n = WORK[sr]["chunk"]
    tag = "16k" if sr == 16000 else "8k"
    expect = {"16k": 19, "8k": 44}[tag]                       # what the reference returns for the fixture
    wav = torch.from_numpy(np.load(ROOT / "tests" / "golden" / f"audio_{tag}.npz")["pcm"].astype(np.float32) / 32768.0)
    model = load_silero_vad(device=local)

    def med(xs):
        xs = sorted(xs)
        return round(xs[len(xs) // 2] * 1e3, 4)

    # (a) eager per-call protocol, host chunk in, float out
    model.reset_states()
    lat = []
    for i in range(60 + 400):
        c = wav[i * n:(i + 1) * n]
        t0 = time.perf_counter()
        model(c, sr).item()
        lat.append(time.perf_counter() - t0)
    eager_ms = med(lat[60:])
    # (b) the same step captured in a hipGraph (StreamPool with one slot), host chunk in, float out
    pool = StreamPool(Engine(device=local), sr, capacity=1, graph=True)
    pool.open()
    lat = []
    for i in range(60 + 400):
        c = wav[i * n:(i + 1) * n][None]
        t0 = time.perf_counter()
        pool.tick(c).item()
        lat.append(time.perf_counter() - t0)
    graph_ms = med(lat[60:])
    # (c) get_speech_timestamps on the fixture: the one-call fast path, and the unmodified per-chunk protocol
    class PerChunk:                                            # hides audio_forward_device: the caller loops over chunks
        def __init__(self, m):
            self.m = m
        def reset_states(self):
            self.m.reset_states()
        def __call__(self, x, sr):
            return self.m(x, sr)
    t0 = time.perf_counter(); ts_fast = get_speech_timestamps(wav, model, sampling_rate=sr); fast_s = time.perf_counter() - t0
    t0 = time.perf_counter(); ts_fast = get_speech_timestamps(wav, model, sampling_rate=sr); fast_s = time.perf_counter() - t0
    t0 = time.perf_counter(); ts_chunk = get_speech_timestamps(wav, PerChunk(model), sampling_rate=sr); chunk_s = time.perf_counter() - t0
    if len(ts_fast) != expect or ts_chunk != ts_fast:
        raise RuntimeError(f"plumbing: expected the reference's {expect} segments, got {len(ts_fast)} / {len(ts_chunk)}")
    chunks = (len(wav) + n - 1) // n
    secs = len(wav) / sr
    return {"workload": f"configs[0]: the reference's fixture ({secs:.0f} s, {sr // 1000} kHz) through the reference's per-chunk protocol, B = 1",
            "per_call_latency_ms": {"eager_model_call_item": eager_ms, "hipgraph_step_item": graph_ms,
                                    "note": "host chunk in -> float out, median of 400 calls.  eager: 2 KB into page-locked memory, ONE kernel (one workgroup per stream) that reads it in place and stores the probability into page-locked memory, the call returns when that slot has changed (vad_step_host_sync); hipgraph: a 1-stream StreamPool tick (H2D, fused step, D2H, stream wait)"},
            f"get_speech_timestamps_{secs:.0f}s": {"segments": len(ts_fast), "one_call_fast_path_ms": round(fast_s * 1e3, 2),
                                                  "per_chunk_protocol_ms": round(chunk_s * 1e3, 1), "chunks": chunks,
                                                  "per_chunk_protocol_ms_per_chunk": round(chunk_s * 1e3 / chunks, 4),
                                                  "identical_segments": True,
                                                  "cpu_beside_it": "cpu_baseline.runs.R5_get_speech_timestamps_fixture (16 kHz fixture: the same "
                                                                   "call over the reference's ATen operators, one thread)"},
            "reference_claim": "< 1 ms per chunk on one CPU thread (README.md:103); measured CPU R1 in cpu_baseline"}


# ---- dry: the launch / shard / gather / reduce / print plumbing without a GPU (tests/test_sharding.py) --------------------
class _DryModel:
    """No GPU, no oracle: probabilities are a cheap deterministic function of the audio, enough to drive the sharding,
    the ragged scheduler's CPU branch, the native host scanner and the gather."""
    def audio_forward_device(self, x, sr):
        import torch
        n = 512 if sr == 16000 else 256
        x = x.to(torch.float32)
        T = (x.shape[1] + n - 1) // n
        x = torch.nn.functional.pad(x, (0, T * n - x.shape[1]))
        return torch.sigmoid(40.0 * x.view(x.shape[0], T, n).abs().mean(-1) - 2.0)


def run_dry(args, rank, world, dist):
    import numpy as np
    import torch
    if args.config == "corpus":
        from silero_vad_amd import gather_to_rank0, ragged_speech_segments
        sr, R, passes, base_len = 16000, 12, 2, 1 << 20
        rng = np.random.default_rng(7)
        base = torch.from_numpy((0.2 * rng.standard_normal(base_len) * (np.sin(np.arange(base_len) / 9000.0) > 0)).astype(np.float32))
        shard = corpus_shard(rank, world, passes, R, sr, base_len)
        res = []

        def whole():
            for _, lens, offs in shard:
                res.append(ragged_speech_segments([base[o:o + m] for o, m in zip(offs.tolist(), lens.tolist())], _DryModel(), sr,
                                                  threshold=0.4))
        elapsed = timed(world, dist, torch.device("cpu"), 1, whole, lambda: None)
        mine = {int(g): r for (gids, _, _), rr in zip(shard, res) for g, r in zip(gids, rr)}
        parts = gather_to_rank0(mine)
        if rank != 0:
            return None
        merged = {}
        for part in parts:
            merged.update(part)
        out = base_line(args, world, sr, 0.0, elapsed, 1)
        out.update({"dry": True, "metric": "dry run (no measurement)", "data": "none (dry run of the corpus sharding / gather plumbing; no GPU work)",
                    "config": {"workload": "dry corpus", "sharding": f"recordings x{world} by duration, results gathered to rank 0"},
                    "recordings_total": world * R * passes, "recordings_gathered": len(merged),
                    "ids_complete": sorted(merged) == list(range(world * R * passes)),
                    "segments_total": sum(len(v) for v in merged.values())})
        return out
    acc = [0.0]

    def step():
        acc[0] += float(torch.ones(1000).sum())

    elapsed = timed(world, dist, torch.device("cpu"), args.steps, step, lambda: None)
    if rank != 0:
        return None
    out = base_line(args, world, 16000, 0.0, elapsed, args.steps)
    out.update({"dry": True, "metric": "dry run (no measurement)", "data": "none (dry run of the launch/reduce plumbing; no GPU work, not a measurement)",
                "config": {"workload": "dry", "sharding": f"streams x{world}, no collectives"}})
    return out


def small(d):
    """The part of a config's line that is kept when it is nested under other_configs."""
    if d is None:
        return None
    keep = ("value", "unit", "steps", "ms_per_step", "dtype", "kernel_ms", "tick_latency_ms", "legs", "wall_s", "n_gpus",
            "audio_hours_all_gpus", "ten_k_hours_at_this_rate_s", "parity_sample", "parity_sample_max_abs_dp",
            "outputs_finite", "realtime_factor", "timed_region_s", "parity", "parity_max_abs_dp", "pcie", "events_emitted", "sustained", "gaps")
    out = {k: d[k] for k in keep if k in d}
    out["workload"] = d["config"]["workload"]
    for k in ("sharding", "host_threads_per_rank", "numa_node_bound", "recordings_per_gpu", "audio_hours_per_gpu", "parts", "ring_slots"):
        if k in d["config"]:
            out[k] = d["config"][k]
    if "roofline" in d:
        out["roofline"] = {k: d["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms")}
    return out


def compact_legs(out):
    """Every leg of the run in one small object: value in chunks/s, its fraction of what bounds it (fp32 MFMA peak for the kernels, this
    box's int16 PCIe ceiling for the host-fed legs), the leg's own parity check (max |dp| against the oracle on the leg's own audio, and the
    largest probability among the checked chunks: a check on probabilities that never leave 0 proves little)."""
    def one(d):
        if not isinstance(d, dict):
            return None
        if "error" in d:
            return {"error": d["error"][:80]}
        e = {"value": d.get("value")}
        if isinstance(d.get("roofline"), dict) and "frac" in d["roofline"]:
            e["of_mfma_peak"] = d["roofline"]["frac"]
        if isinstance(d.get("pcie"), dict):
            e["of_link"] = d["pcie"].get("fraction_of_pcie_ceiling")
        if isinstance(d.get("tick_latency_ms"), dict):
            e["tick_ms_p95"] = d["tick_latency_ms"].get("p95")
            if "max" in d["tick_latency_ms"]:
                e["tick_ms_max"] = d["tick_latency_ms"]["max"]
        par = d.get("parity")
        if isinstance(par, dict):
            e["dp"] = float(f"{par.get('parity_max_abs_dp', 0):.2e}")
            e["max_prob"] = par.get("max_prob")
        elif d.get("parity_sample_max_abs_dp") is not None:
            e["dp"] = float(f"{d['parity_sample_max_abs_dp']:.2e}")
        if isinstance(d.get("gaps"), dict):              # live streams that miss ticks: the fraction of (stream, tick) pairs without a chunk
            e["missed"] = d["gaps"].get("missed_fraction")
            if isinstance(d["gaps"].get("full_rows"), dict):   # (the same ticks without compact slots)
                e["full_rows_value"] = d["gaps"]["full_rows"].get("value")
        if "first_result_at" in d:                       # corpus routes that hand results over while the shard runs
            e["first_result_at"] = d["first_result_at"]
        return e
    legs = {"c2": one(out)}
    for name, d in (out.get("other_configs") or {}).items():
        if name.startswith("plumbing"):
            pc = (d or {}).get("per_call_latency_ms") or {}
            gst = next((v for k, v in (d or {}).items() if k.startswith("get_speech_timestamps")), {}) or {}
            legs[name] = {"call_ms": pc.get("eager_model_call_item"), "per_chunk_protocol_ms": gst.get("per_chunk_protocol_ms"),
                          "identical_segments": gst.get("identical_segments")}
            if isinstance(d, dict) and "error" in d:
                legs[name] = {"error": d["error"][:80]}
            continue
        legs[name] = one(d)
        if name.startswith("corpus") and isinstance(d, dict) and isinstance(d.get("legs"), dict):
            legs[name]["of_link"] = d["legs"].get("main", {}).get("fraction_of_pcie_ceiling")
            routes = {k: v.get("fraction_of_pcie_ceiling") for k, v in d["legs"].items() if k != "main"}
            if routes:
                legs[name]["routes_of_link"] = routes
            fr = [d["legs"].get(k, {}).get("first_result_at") for k in ("pinned_refill_gather", "pinned_refill_window")]
            if any(f is not None for f in fr):
                legs[name]["refill_first_result_at"] = fr[0] if fr[1] is None else fr
    return legs


LINE_CAP = 8192          # bytes: what the driver keeps of stdout; the line must fit in it whole
DETAIL_FILE = "gpurun_out/bench_detail.json"


def _num(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(out):
    """THE line: everything the measurement contract names, in at most LINE_CAP bytes whatever the run held (9 legs, 8 ranks).  Per-leg
    workload prose, traffic detail, issue-pipe reading, untimed trials and per-rank records are NOT here: `emit` writes the full record to
    DETAIL_FILE (stderr only names it).  Strings stay under 150 characters (the driver's parser clips there)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out[k] for k in keep if k in out}
    cfg = dict(out.get("config") or {})
    if isinstance(cfg.get("workload"), str):
        cfg["workload"] = cfg["workload"][:148]
    line["config"] = {k: (v[:148] if isinstance(v, str) else v) for k, v in cfg.items()}
    for k in ("dry", "realtime_factor", "timed_region_s", "outputs_finite", "wall_s", "audio_hours_all_gpus", "ten_k_hours_at_this_rate_s",
              "recordings_total", "recordings_gathered", "ids_complete", "segments_total"):
        if k in out:
            line[k] = out[k]
    if isinstance(out.get("kernel_ms"), dict):
        line["kernel_ms"] = {k: v for k, v in out["kernel_ms"].items() if k != "note"}
    if isinstance(out.get("tick_latency_ms"), dict):
        line["tick_latency_ms"] = {k: v for k, v in out["tick_latency_ms"].items() if k != "what"}
    if isinstance(out.get("pcie"), dict):
        line["pcie"] = out["pcie"]
    par = out.get("parity")
    if isinstance(par, dict):
        line["parity"] = {"max_abs_dp": float(f"{par.get('parity_max_abs_dp', 0):.3e}"), "max_prob": par.get("max_prob"),
                          "final_state": (float(f"{par['final_state_max_rel_err']:.3e}") if "final_state_max_rel_err" in par else None),
                          "streams_checked": par.get("streams_checked"), "chunks_checked": par.get("chunks_checked"),
                          "tolerance": par.get("tolerance"), "checker": "oracle/vad_oracle.c on the timed PCM", "ok": par.get("ok")}
    ps = out.get("parity_sample")
    if isinstance(ps, dict):
        line["parity"] = {"max_abs_dp": float(f"{(ps.get('parity_sample_max_abs_dp') or 0):.3e}"), "recordings_checked": ps.get("recordings_checked"),
                          "one_in": ps.get("one_in"), "segments_identical": ps.get("segments_identical_to_oracle_scan"), "tolerance": ps.get("tolerance")}
    rl = out.get("roofline")
    if isinstance(rl, dict):
        r = {k: rl.get(k) for k in ("bound", "kernel", "dtype", "achieved", "peak", "unit", "frac", "avg_launch_ms", "flop_per_launch", "traffic",
                                    "kernel_io_bytes", "traffic_over_kernel_io") if k in rl}
        if rl.get("traffic") is None and isinstance(rl.get("traffic_profiled"), dict):
            r["traffic_profiled"] = rl["traffic_profiled"].get("bytes")      # the committed profile's figure, not this run's
        if isinstance(rl.get("issue_pipe"), dict):
            r["issue_pipe_frac"] = rl["issue_pipe"].get("frac")
        if isinstance(rl.get("hbm"), dict):
            r["hbm_frac_kernel_io"] = rl["hbm"].get("frac")
        p = rl.get("path")
        if isinstance(p, dict):
            r["path"] = {k: p.get(k) for k in ("algorithmic_bytes", "algorithmic_bytes_per_chunk", "traffic", "traffic_over_algorithmic",
                                               "mfma_flop_per_chunk") if k in p}
            if p.get("traffic") is None:
                r["path"]["traffic_profiled"] = p.get("traffic_profiled")
            pf = out.get("path_fraction") or {}
            r["path"]["dense_flop_vs_fp32_peak"] = pf.get("dense_flop_vs_fp32_peak")
            r["path"]["algorithmic_bytes_vs_hbm_peak"] = pf.get("algorithmic_bytes_vs_hbm_peak")
            km = out.get("kernel_ms") or {}
            if km.get("front") and "n_gpus" in out and "value" in out and out.get("value"):
                # executed matrix flops of BOTH kernels over the whole step time, against the fp32 MFMA peak
                r["path"]["mfma_frac"] = round(out["value"] / max(out["n_gpus"], 1) * p.get("mfma_flop_per_chunk", 0) / (PEAK_F32_TFLOPS * 1e12), 4)
        rk = rl.get("rec_kernel")
        if isinstance(rk, dict):
            r["rec_kernel"] = {k: rk.get(k) for k in ("avg_launch_ms", "mfma_frac", "hbm_frac")}
        line["roofline"] = r
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c = {k: cb.get(k) for k in ("value", "unit", "cores", "cpu_model", "kind", "port", "best_protocol", "torch", "cached_from") if k in cb}
        if isinstance(cb.get("runs"), dict):
            c["runs"] = {k.split("_")[0]: _num(v.get("chunks_per_s") if isinstance(v, dict) else v, 1) for k, v in cb["runs"].items()}
        if "error" in cb:
            c["error"] = str(cb["error"])[:140]
        c["sample"] = "0.03 N(0,1) PCM, audio_forward B x T per protocol R1-R4 (BASELINE.md 3), warm-up 3, median of 5; R5 = fixture per chunk"
        line["cpu_baseline"] = c
    oa = out.get("other_arithmetic")
    if isinstance(oa, dict):
        line["other_arithmetic"] = {k: {"value": v.get("value"), "max_abs_prob_diff_vs_main": _num(v.get("max_abs_prob_diff_vs_main"), 9)}
                                    for k, v in oa.items() if isinstance(v, dict)}
    if "node_totals" in out:
        line["node_totals"] = out["node_totals"]
    if "plumbing" in out and isinstance(out["plumbing"], dict):
        pl = out["plumbing"]
        line["plumbing"] = {"per_call_latency_ms": {k: v for k, v in (pl.get("per_call_latency_ms") or {}).items() if k != "note"}}
        for k, v in pl.items():
            if k.startswith("get_speech_timestamps") and isinstance(v, dict):
                line["plumbing"][k] = {kk: vv for kk, vv in v.items() if kk != "cpu_beside_it"}
    line["detail"] = DETAIL_FILE
    if "legs" in out and not out.get("dry"):
        legs = out["legs"]
        line["legs"] = legs if "c2" in legs else {k: {"value": v.get("value"), "of_link": v.get("fraction_of_pcie_ceiling"),
                                                      "segments_found_rank0": v.get("segments_found_rank0"),
                                                      "segments_gathered_all_ranks": v.get("segments_gathered_all_ranks")} for k, v in legs.items()}
    # safety valve: whatever a future leg adds, the line fits -- shed the least important parts first
    for shed in (("other_arithmetic",), ("plumbing",), ("cpu_baseline", "sample"), ("roofline", "path"), ("legs",), ("node_totals",), ("pcie",)):
        if len(json.dumps(line)) <= LINE_CAP - 64:
            break
        tgt = line
        for k in shed[:-1]:
            tgt = tgt.get(k, {})
        tgt.pop(shed[-1], None)
        line["shed"] = line.get("shed", []) + ["/".join(shed)]
    return line


def emit(out, stream=None):
    """Rank 0's output: the full record to DETAIL_FILE (next to this file; `gpurun_out/` is what comes back from a GPU box; stderr
    only names it), then ONE line of at most LINE_CAP bytes on stdout, last."""
    try:
        path = ROOT / DETAIL_FILE
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(json.dumps(out, indent=1) + "\n")
    except OSError as e:
        print(f"bench: could not write {DETAIL_FILE}: {e}", file=sys.stderr)
    # stderr stays SHORT: the driver's record is a tail of "stdout, then stderr" -- a 27 KB record dumped there would push the stdout
    # line out of that tail (what made BENCH_r05 unparseable was size, not shape).  VAD_BENCH_STDERR_DETAIL=1 brings the dump back.
    if os.environ.get("VAD_BENCH_STDERR_DETAIL"):
        print("bench detail: " + json.dumps(out), file=sys.stderr, flush=True)
    else:
        print(f"bench detail: {DETAIL_FILE}", file=sys.stderr, flush=True)
    text = json.dumps(compact_line(out))
    assert len(text) <= LINE_CAP, len(text)
    print(text, file=stream or sys.stdout, flush=True)
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["c2", "8k", "stream", "stream_host", "stream_gaps", "stream_8k", "stream_host_8k", "corpus", "plumbing", "plumbing_8k"],
                    default="c2")
    ap.add_argument("--streams", type=int, default=STREAMS, help=argparse.SUPPRESS)
    ap.add_argument("--chunks", type=int, default=CHUNKS_PER_STREAM, help=argparse.SUPPRESS)
    ap.add_argument("--live", type=int, default=LIVE_STREAMS, help=argparse.SUPPRESS)
    ap.add_argument("--recordings", type=int, default=CORPUS_PASS, help=argparse.SUPPRESS)
    ap.add_argument("--corpus-passes", type=int, default=37, help="corpus: passes of --recordings per GPU (37 x 4096 x 30 s = 1 250 h)")
    ap.add_argument("--corpus-main-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-parity", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true", help="skip other_configs")
    ap.add_argument("--dry", action="store_true", help="no GPU: exercise launch/shard/gather/barrier/reduce/print only (gloo)")
    args = ap.parse_args()
    default_steps = {"c2": 200, "8k": 200, "stream": 2000, "stream_8k": 2000, "stream_host": 2000, "stream_host_8k": 2000, "stream_gaps": 2000,
                     "corpus": 1, "plumbing": 1, "plumbing_8k": 1}   # corpus: one step = the whole shard
    if args.steps is None:
        args.steps = default_steps[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_distributed(args)                      # does not return

    # the CPU baseline runs first, in its own CPU-only process, before this process touches the GPU
    # (at N > 1 too, on rank 0: the other ranks wait for it in the rendezvous -- the multi-rank line carries its own baseline)
    want_cpu = int(os.environ.get("RANK", 0)) == 0 and not args.no_cpu_baseline and not args.dry
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
        extras = not args.no_extras
        if args.config in ("c2", "8k"):
            sr = 16000 if args.config == "c2" else 8000
            out = run_batch(args, sr, rank, world, local, dist, args.steps, with_other=extras and args.config == "c2")
        elif args.config in ("stream", "stream_8k"):
            out = run_stream(args, rank, world, local, dist, args.steps, 16000 if args.config == "stream" else 8000)
        elif args.config in ("stream_host", "stream_host_8k", "stream_gaps"):
            out = run_stream_host(args, rank, world, local, dist, args.steps, 8000 if args.config == "stream_host_8k" else 16000,
                                  gaps=0.10 if args.config == "stream_gaps" else 0.0)
        elif args.config == "corpus":
            out = run_corpus(args, rank, world, local, dist, args.corpus_passes)
        else:
            out = {"metric": "per-call latency (plumbing, no throughput claim)", "value": None, "n_gpus": 1,
                   "config": {"workload": "configs[0]"}, "plumbing": run_plumbing(args, local, 16000 if args.config == "plumbing" else 8000)}
        if extras and args.config == "c2":              # the other BASELINE configs, for the record
            oc = {}
            legs = [("stream", lambda: run_stream(args, rank, world, local, dist, 1000)),
                    ("stream_host", lambda: run_stream_host(args, rank, world, local, dist, 2000)),
                    ("stream_gaps", lambda: run_stream_host(args, rank, world, local, dist, 2000, gaps=0.10)),
                    ("corpus", lambda: run_corpus(args, rank, world, local, dist, args.corpus_passes)),
                    ("corpus_48k", lambda: run_corpus(args, rank, world, local, dist, min(6, args.corpus_passes), sr=48000, main_only=True))]
            if world == 1:
                legs = [("8k", lambda: run_batch(args, 8000, rank, world, local, dist, 100))] + legs + \
                       [("stream_8k", lambda: run_stream(args, rank, world, local, dist, 1000, 8000)),
                        ("stream_host_8k", lambda: run_stream_host(args, rank, world, local, dist, 2000, 8000)),
                        ("plumbing", lambda: {"config": {"workload": "configs[0]"}, **run_plumbing(args, local)}),
                        ("plumbing_8k", lambda: {"config": {"workload": "configs[0], 8 kHz"}, **run_plumbing(args, local, 8000)})]
            def hand_back():
                """between legs: the finished leg's device blocks (window buffers, staging slots, GiBs) go back to the driver instead of
                sitting in torch's cache beside the next leg's own -- eight ranks rehearsed on ONE GPU ran out of memory that way"""
                import gc
                gc.collect()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()

            for name, fn in legs:
                if not args.dry:
                    hand_back()
                if world > 1:                           # every rank takes part in a leg's barriers: no swallowing of errors there
                    r = fn()
                    if rank == 0:
                        oc[name] = small(r)
                    continue
                try:
                    r = fn()
                    oc[name] = r if name.startswith("plumbing") else small(r)
                except Exception as e:                  # an extra must never cost the headline line
                    oc[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            if rank == 0:
                out["other_configs"] = oc
        under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        if (extras and world == 1 and args.config in ("c2", "8k") and not os.environ.get("VAD_BENCH_PMC_CHILD")
                and not os.environ.get("VAD_BENCH_NO_PMC") and not under_profiler):   # (a profiled run does not start a profiler of its own)
            # roofline.traffic, measured by THIS run: two short child runs of this command under rocprofv3 --pmc
            fk = WORK[16000 if args.config == "c2" else 8000]["front_kernel"].split("<")[0]
            t0 = time.perf_counter()
            got, why = live_pmc_traffic(args.config, [fk, "rec_kernel"])
            rl = out["roofline"]
            if got and fk in got:
                rl["traffic"] = got[fk]["bytes"]
                rl["traffic_detail"] = dict(got[fk], unit="bytes per launch", how="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one child run of "
                                            "this command each (6 timed steps), mean of the timed dispatches; FETCH_SIZE x 2 (gfx950 tallies "
                                            "128-byte requests at 64 bytes, MI355X_MICROARCH.md HBM) + WRITE_SIZE, KiB -> bytes",
                                            seconds=round(time.perf_counter() - t0, 1))
                rl["traffic_over_kernel_io"] = round(got[fk]["bytes"] / rl["kernel_io_bytes"], 3)
                if "rec_kernel" in got:
                    pt = got[fk]["bytes"] + got["rec_kernel"]["bytes"]
                    rl["path"]["traffic"] = pt
                    rl["path"]["traffic_over_algorithmic"] = round(pt / rl["path"]["algorithmic_bytes"], 3)
                    rl["path"]["traffic_detail"] = {"front": got[fk], "rec": got["rec_kernel"]}
            else:
                rl["traffic_note"] = f"live PMC pass unavailable ({why}); traffic_profiled is the committed profile's figure"

    if world > 1 and not args.dry:
        # what every rank of the node held and used, gathered on the host (for the N = 8 rehearsal and the scaling run's record)
        import resource
        from silero_vad_amd import _lib, gather_to_rank0
        hs = torch.cuda.host_memory_stats() if hasattr(torch.cuda, "host_memory_stats") else {}
        mine = {"rank": rank, "local_rank": local, "host_threads": _lib.lib().vad_host_threads(), "numa_node_bound": NUMA_NODE.get("node"),
                "torch_pinned_peak_bytes": int(hs.get("allocated_bytes.peak", 0) or 0), "native_pinned_bytes": int(NATIVE_PINNED.get("bytes", 0)),
                "device_peak_bytes_torch": int(torch.cuda.max_memory_allocated()), "max_rss_mb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024}
        per_rank = gather_to_rank0(mine)
        if rank == 0:
            out["per_rank"] = per_rank
            out["node_totals"] = {"pinned_bytes": sum(r["torch_pinned_peak_bytes"] + r["native_pinned_bytes"] for r in per_rank),
                                  "host_threads": sum(r["host_threads"] for r in per_rank),
                                  "device_peak_bytes_torch": sum(r["device_peak_bytes_torch"] for r in per_rank)}
    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if os.environ.get("VAD_BENCH_SHARE_GPU") and world > 1:
            out["data"] = "synthetic; FUNCTIONAL run of the multi-rank legs with all ranks on ONE GPU (gloo) -- not a measurement"
        if not args.dry and args.config == "c2" and "legs" not in out:
            out["legs"] = compact_legs(out)             # LAST key of the line
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
