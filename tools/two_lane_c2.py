#!/usr/bin/env python3
"""C2 (4096 streams x 256 chunks, fp32 PCM resident in HBM) with successive steps dealt to TWO engines (vad_clone: shared weights, own
scratch) on two streams, so that one step's recurrence can run beside the next step's frontend -- against the same steps on one
engine.  Timing experiment; the probabilities of both forms are compared bit for bit."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import Engine                                    # noqa: E402
from silero_vad_amd.streams import _distinct_queue_stream            # noqa: E402

B, T, n, sr = 4096, 256, 512, 16000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
eng = Engine(device=0)
g = torch.Generator(device=dev).manual_seed(5)
pcm = 0.03 * torch.randn((B, T * n), device=dev, generator=g)
engines = [eng, eng.clone()]
cur = torch.cuda.current_stream(dev)
streams = [cur, _distinct_queue_stream(eng, dev, [cur])]
bufs = []
for e in engines:
    e.reserve(sr, B, T)
    bufs.append((torch.zeros((B, n // 8), device=dev), torch.zeros((2, B, 128), device=dev), torch.empty((B, T), device=dev)))


def step(lane):
    ctx, state, probs = bufs[lane]
    with torch.cuda.stream(streams[lane]):
        ctx.zero_()
        state.zero_()
        engines[lane].forward_audio(pcm, sr, ctx, state, probs)


def run(lanes, k):
    for s in streams:
        s.wait_stream(cur)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        step(i % lanes)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for lanes in (1, 2, 1, 2):
    run(lanes, 40)                                                   # clock ramp
    t = run(lanes, steps)
    print(f"lanes {lanes}: {t / steps * 1e3:.3f} ms per step, {B * T * steps / t / 1e6:.1f} M chunks/s")
print("identical probabilities on both lanes:", bool(torch.equal(bufs[0][2], bufs[1][2])))
