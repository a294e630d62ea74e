#!/usr/bin/env python3
"""Export the published Silero-VAD v6 weights to the flat container the engine loads.

Run ONCE in the authoring container (needs /root/reference); the output travels with
the repo because /root/reference does not exist on the GPU box.

Source of truth: ``state_dict()`` of ``src/silero_vad/data/silero_vad.jit`` (SURVEY.md §0.3:
the safetensors file in the same directory holds a *different* weight set and must not be
used).  30 tensors, 545 282 fp32 values, two independent nets (16 kHz: ``_model.*``,
8 kHz: ``_model_8k.*``).

Container layout (little endian) -- parsed by ``silero_vad_amd/csrc/weights.cpp`` and
``oracle/weights.py``:

    char[8]   magic   = b"SVADW001"
    u32       n_tensors
    u32       reserved (0)
    char[64]  source sha256 hex of the .jit
    n_tensors x {
        char[64] name (zero padded)      e.g. "_model.encoder.0.reparam_conv.weight"
        u32      ndim
        u32      dims[4]                 (unused dims = 1)
        u64      byte offset of the fp32 payload from the start of the file (64-B aligned)
        u64      n_elements
    }
    payloads
"""
import hashlib
import struct
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
JIT = REF / "src/silero_vad/data/silero_vad.jit"
OUT = Path(__file__).resolve().parents[1] / "silero_vad_amd/data/silero_vad_v6.weights"


def main() -> None:
    import torch

    sha = hashlib.sha256(JIT.read_bytes()).hexdigest()
    model = torch.jit.load(str(JIT), map_location="cpu")
    model.eval()
    sd = model.state_dict()
    names = list(sd.keys())
    assert len(names) == 30, len(names)

    rec = struct.Struct("<64sI4IQQ")
    header_len = 8 + 4 + 4 + 64 + rec.size * len(names)
    off = (header_len + 63) // 64 * 64
    table, blobs = [], []
    for n in names:
        a = sd[n].detach().cpu().contiguous().numpy().astype("<f4")
        dims = list(a.shape) + [1] * (4 - a.ndim)
        table.append(rec.pack(n.encode(), a.ndim, *dims, off, a.size))
        blobs.append((off, a.tobytes()))
        off = (off + a.nbytes + 63) // 64 * 64

    buf = bytearray(off)
    buf[0:8] = b"SVADW001"
    struct.pack_into("<II", buf, 8, len(names), 0)
    buf[16:80] = sha.encode()
    p = 80
    for t in table:
        buf[p:p + rec.size] = t
        p += rec.size
    for o, b in blobs:
        buf[o:o + len(b)] = b
    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_bytes(bytes(buf))
    total = sum(int(np.prod(sd[n].shape)) for n in names)
    print(f"wrote {OUT} ({len(buf)} bytes, {total} params, jit sha256 {sha})")


if __name__ == "__main__":
    sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
    main()
