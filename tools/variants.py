#!/usr/bin/env python3
"""Build and time A/B variants of the kernels (compile-time knobs in csrc/*.hip; any tool or test takes a variant through
SILERO_VAD_AMD_LIB=build/variants/lib_<name>.so -- tools/b9_time.py, tools/trace_b9.py, tools/lat_time.py ...).

    python tools/variants.py build            # here (no GPU): build/variants/lib_<name>.so
    python tools/variants.py run [names...]   # on the GPU box: bench each, write gpurun_out/variants.json

A variant is a set of -D flags.  Variants whose name starts with "abl" are timing-only ablations
(wrong results by construction); every other variant is parity-checked against the golden vectors before
it is timed.  The product library is always silero_vad_amd/libsilero_vad_hip.so (default knobs).
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "silero_vad_amd" / "csrc"
OUT = ROOT / "build" / "variants"
sys.path.insert(0, str(ROOT))
from __graft_entry__ import CPP_SOURCES as CPP, HIP_SOURCES as HIP      # noqa: E402  (the product's translation units)

VARIANTS = {
    "base": [],
    "lat_wg2": ["-DVAD_LAT_WG_PER_CU=2", "-DVAD_LAT_DEPTH=8"], "lat_wg2_d4": ["-DVAD_LAT_WG_PER_CU=2", "-DVAD_LAT_DEPTH=4"],
    "lat_d8": ["-DVAD_LAT_DEPTH=8"], "lat_d24": ["-DVAD_LAT_DEPTH=24"], "lat_d32": ["-DVAD_LAT_DEPTH=32"],
    "lat_noload": ["-DVAD_LAT_ABLATE=1"], "lat_nofft": ["-DVAD_LAT_ABLATE=2"], "lat_neither": ["-DVAD_LAT_ABLATE=3"],
    "rec_skew": ["-DVAD_REC_SKEW=1"],                  # fp32 recurrence with the two waves of a SIMD half a step apart (slower: r03a)
    "rec_skew_noprio": ["-DVAD_REC_SKEW=1", "-DVAD_REC_PRIO=0"],
    "nobvbatch": ["-DVAD_BV_BATCH=0"],                 # frontend: B-operand transforms interleaved with the MFMAs (round-2 form)
    "xbasis": ["-DVAD_F43_EF=0"],                       # F(4,3) input transform from x0..x3 instead of (E, F, x1, x2) (f43 form only)
    "trace": ["-DVAD_TRACE=1"],
    "fftprio1": ["-DVAD_F43_FFT_PRIO=1"], "fftprio3": ["-DVAD_F43_FFT_PRIO=3"],
    "trace_fftprio3": ["-DVAD_TRACE=1", "-DVAD_F43_FFT_PRIO=3"],
    "gemmprio3": ["-DVAD_F43_GEMM_PRIO=3"], "trace_gemmprio3": ["-DVAD_TRACE=1", "-DVAD_F43_GEMM_PRIO=3"],
    "trace_noload": ["-DVAD_TRACE=1", "-DVAD_ABLATE=4"],
    "abl_nobar": ["-DVAD_ABLATE=1"],
    "abl_nofft": ["-DVAD_ABLATE=2"],
    "abl_noload": ["-DVAD_ABLATE=4"],
    "abl_noring": ["-DVAD_ABLATE=8"],
    "abl_mfma_only": ["-DVAD_ABLATE=15"],
    "abl_coalesced": ["-DVAD_ABLATE=16"],
    "abl_seg64": ["-DVAD_ABLATE=32"],
    # bf16 x 9 frontend (tools/b9_time.py): timing-only ablations
    "abl_b9_nosplit": ["-DVAD_ABLATE=64"], "abl_b9_nofrag": ["-DVAD_ABLATE=128"], "abl_b9_mfma3": ["-DVAD_ABLATE=256"],
    "abl_b9_nosplit_nofft": ["-DVAD_ABLATE=66"], "abl_b9_valu_none": ["-DVAD_ABLATE=70"], "abl_b9_noring_nobar": ["-DVAD_ABLATE=9"],
    "abl_b9_mfma_only": ["-DVAD_ABLATE=207"],
    "nopk_all": [],                                     # every knob unit without packed fp32 (the bf16 x 9 recurrence beside plain VALU only)
    "b9_w4": ["-DVAD_B9_WAVES=4"],                     # bf16 x 9 frontend: two 4-wave workgroups per CU (default: one 8-wave workgroup)
    "pk_b9": [],
    "noexact": ["-DVAD_NO_EXACT=1"],
    "gather128": ["-DVAD_GATHER_WAVES=128"], "gather192": ["-DVAD_GATHER_WAVES=192"], "gather64": ["-DVAD_GATHER_WAVES=64"],   # ingest kernel's footprint                   # frontends without the silent-frame test (what does it cost at C2?  profiles/r06_exact.md)
}


def build(names):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    OUT.mkdir(parents=True, exist_ok=True)
    common = ["-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-I", str(ROOT / "include")]
    shared = OUT / "obj_shared"
    shared.mkdir(exist_ok=True)
    procs = []
    # translation units without knobs are compiled once
    knob_units = {"kernel_front_f43.hip", "kernel_front_lat.hip", "kernel_front_b9.hip", "kernel_rec.hip", "kernel_rec_b9.hip"}
    if os.environ.get("VAD_VARIANT_UNITS"):      # only these translation units carry the knobs (the others come from the shared build)
        knob_units = set(os.environ["VAD_VARIANT_UNITS"].split(","))
    for src in HIP + CPP:
        if src in knob_units:
            continue
        o = shared / (src + ".o")
        cmd = ([hipcc, "--offload-arch=gfx950"] + common + ["-c", str(CSRC / src), "-o", str(o)]
               if src.endswith(".hip") else [hipcc] + common + ["-x", "c++", "-c", str(CSRC / src), "-o", str(o)])
        procs.append(subprocess.Popen(cmd))
    nopk = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    for name in names:
        d = OUT / ("obj_" + name)
        d.mkdir(exist_ok=True)
        for src in knob_units:
            extra = nopk if (name.startswith("nopk") or (src == "kernel_front_b9.hip" and not name.startswith("pk_"))) else []
            procs.append(subprocess.Popen([hipcc, "--offload-arch=gfx950"] + common + extra + VARIANTS[name]
                                          + ["-c", str(CSRC / src), "-o", str(d / (src + ".o"))]))
    for p in procs:
        if p.wait() != 0:
            sys.exit("compile failed")
    for name in names:
        objs = [str(shared / (s + ".o")) for s in HIP + CPP if s not in knob_units] + \
               [str(OUT / ("obj_" + name) / (s + ".o")) for s in knob_units]
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o",
                        str(OUT / f"lib_{name}.so")] + objs, check=True)
        print("built", name)


CHECK = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import silero_vad_amd
m = silero_vad_amd.load_silero_vad(device=0)
g = np.load(%r)
wav = np.load(%r)["pcm"].astype(np.float32) / 32768.0
B, T, L, stride = (int(v) for v in g["batch_meta"])
rows = np.stack([np.roll(wav, -b * stride)[:L] for b in range(B)])
got = m.audio_forward(torch.from_numpy(rows), 16000).numpy()
print("PARITY", float(np.abs(got - g["probs_batch"]).max()))        # golden vectors of the reference model
"""


def run(names):
    res = {}
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    for name in names:
        lib = OUT / f"lib_{name}.so"
        env = dict(os.environ, SILERO_VAD_AMD_LIB=str(lib))
        r = {}
        if not name.startswith("abl"):
            chk = subprocess.run([sys.executable, "-c", CHECK % (str(ROOT), str(ROOT / "tests/golden/golden_16k.npz"),
                                                                  str(ROOT / "tests/golden/audio_16k.npz"))],
                                 env=env, capture_output=True, text=True, timeout=600)
            r["parity"] = next((float(l.split()[1]) for l in chk.stdout.splitlines() if l.startswith("PARITY")), None)
            if r["parity"] is None:
                r["error"] = chk.stderr[-500:]
        b = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "60", "--warmup", "2"]
                           + (["--precision", os.environ["VARIANT_PRECISION"]] if os.environ.get("VARIANT_PRECISION") else []),
                           env=env, capture_output=True, text=True, timeout=600)
        line = next((l for l in b.stdout.splitlines() if l.startswith('{"metric"')), None)
        if line:
            d = json.loads(line)
            r.update(value=d["value"], ms_per_step=d["ms_per_step"], kernel_ms=d["kernel_ms"])
        else:
            r["error"] = (b.stderr or b.stdout)[-500:]
        res[name] = r
        print(name, json.dumps(r), flush=True)
        (ROOT / "gpurun_out" / "variants.json").write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "build"
    names = sys.argv[2:] or list(VARIANTS)
    (build if cmd == "build" else run)(names)
