#!/usr/bin/env python3
"""How do torch's pool streams map onto hardware queues?  Overlap matrix (vad_streams_overlap) of the current stream and the next 10
streams torch hands out, before and after some of them have been used."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from silero_vad_amd import Engine
eng = Engine(0)
dev = torch.device("cuda", 0)
sts = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(10)]
for rnd in range(2):
    print("round", rnd)
    for i, a in enumerate(sts):
        print(i, "".join("1" if eng.streams_overlap(a, b) else ("-" if a is b else "0") for b in sts))
