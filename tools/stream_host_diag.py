#!/usr/bin/env python3
"""Bring-up (GPU box): how should the host-fed stream pool be shaped?  8192 int16 streams, one tick = H2D + fused step + D2H.
   variants: one pool / P sub-pools with their own graphs and streams / ONE graph with P parallel branches / eager launches."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from silero_vad_amd import Engine, StreamPool

dev = torch.device("cuda", 0)
sr, n, cap, R = 16000, 512, 8192, 4
eng = Engine(0)
KEEP = []

def pools(P, graph=True):
    cuts = [cap * i // P for i in range(P + 1)]
    out = []
    for i in range(P):
        q = StreamPool(eng.clone(), sr, capacity=cuts[i + 1] - cuts[i], graph=graph, dtype=torch.int16, host_slots=R)
        q.open_all()
        q.host_pcm.random_(-3000, 3000)
        out.append(q)
    return out

def timeit(fn, iters=600, warm=300):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

for P in (1, 2, 4, 8):
    ps = pools(P)
    k = [0]
    def serial():
        for q in ps: q.submit(k[0] % R)
        for q in ps: q.wait(k[0] % R)
        k[0] += 1
    t_serial = timeit(serial)
    def host_only():
        for q in ps: q.submit(k[0] % R)
        k[0] += 1
    # host cost of the submits alone (GPU may lag behind: bounded by the sync every 50)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200):
        host_only()
        if i % 8 == 7: torch.cuda.synchronize()
    t_host = (time.perf_counter() - t0) / 200 * 1e3
    # ONE graph, P parallel branches
    main = torch.cuda.Stream(dev)
    branches = [torch.cuda.Stream(dev) for _ in ps]
    graphs = []
    torch.cuda.synchronize()
    for r in range(R):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            for q, b in zip(ps, branches):
                b.wait_stream(main)
                with torch.cuda.stream(b):
                    q._host_tick(r)
            for b in branches:
                main.wait_stream(b)
        graphs.append(g)
    done = torch.cuda.Event()
    def one_graph():
        with torch.cuda.stream(main):
            graphs[k[0] % R].replay()
            done.record(main)
        done.synchronize()
        k[0] += 1
    t_one = timeit(one_graph)
    def one_graph_depth2():
        with torch.cuda.stream(main):
            graphs[k[0] % R].replay()
        k[0] += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(400):
        one_graph_depth2()
        if i % 2 == 1: torch.cuda.synchronize()
    t_one2 = (time.perf_counter() - t0) / 400 * 1e3
    # eager: no graphs
    pe = pools(P, graph=False)
    def eager():
        for q in pe: q.submit(k[0] % R)
        for q in pe: q.wait(k[0] % R)
        k[0] += 1
    t_eager = timeit(eager, 300, 100)
    def eager_depth2():                     # tick k + 1 submitted before tick k is awaited
        for q in pe: q.submit((k[0] + 1) % R)
        for q in pe: q.wait(k[0] % R)
        k[0] += 1
    for q in pe: q.submit(k[0] % R)
    t_eager2 = timeit(eager_depth2, 600, 100)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(64):
        for q in pe: q.submit(i % R)
    t_sub = (time.perf_counter() - t0) / 64 * 1e3
    torch.cuda.synchronize()
    print(f"P={P}: own graphs+streams {t_serial:.4f} ms/tick (submit calls alone {t_host:.4f}), ONE graph with {P} branches {t_one:.4f} "
          f"(two ticks per sync {t_one2:.4f}), eager native {t_eager:.4f}, two ticks in flight {t_eager2:.4f}, its submit calls alone {t_sub:.4f}   [H2D alone at 57 GB/s: {cap * n * 2 / 57e9 * 1e3:.4f}]", flush=True)
    KEEP.append((ps, pe, graphs, branches, main))
