"""N > 1 path on CPU: stream sharding + host-side gather with the gloo backend, world_size 2 and 8 (the node the scaling run uses).
The per-rank engine is replaced by a stand-in built on the CPU oracle (tests may use the oracle;
the product never does) -- what is under test is the partition/gather logic, which has no
data-path collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_range_partitions():
    from silero_vad_amd import shard_range
    for n in (0, 1, 7, 8, 4096, 4099):
        for w in (1, 2, 3, 8):
            got = [i for r in range(w) for i in shard_range(n, w, r)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, w, r)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_shard_by_duration_partitions_and_balances():
    from silero_vad_amd import shard_by_duration
    rng = np.random.default_rng(6)
    for n, w in ((0, 3), (1, 2), (7, 2), (100, 8), (1000, 8)):
        lens = [int(v) for v in rng.integers(1, 100000, size=n)]
        parts = [shard_by_duration(lens, w, r) for r in range(w)]
        assert sorted(i for p in parts for i in p) == list(range(n))        # a partition, every rank computes it alike
        assert all(p == sorted(p) for p in parts)
        if n >= 100:
            loads = [sum(lens[i] for i in p) for p in parts]
            assert max(loads) - min(loads) <= max(lens)                      # LPT: within one item of each other
            counts = [sum(lens[i] for i in shard_range_list(n, w, r)) for r in range(w)]
            assert max(loads) <= max(counts)                                  # never worse than dealing by count
    with pytest.raises(ValueError):
        shard_by_duration([1, 2], 2, 2)


def shard_range_list(n, w, r):
    from silero_vad_amd import shard_range
    return list(shard_range(n, w, r))


class OracleModel:
    """Stand-in engine for CPU tests: model protocol + audio_forward_device over the oracle."""

    def __init__(self):
        from oracle import Oracle
        self.o = Oracle()

    def reset_states(self):
        self.o.reset_states()

    def __call__(self, x, sr):
        return torch.from_numpy(self.o(x.numpy(), sr))

    def audio_forward_device(self, x, sr):
        return torch.from_numpy(self.o.audio_forward(x.numpy(), sr))


def _audios():
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "audio_16k.npz"))["pcm"]
    wav = gold.astype(np.float32) / 32768.0
    lens = [40000, 40000, 25000, 40000, 33333, 25000, 16000]
    return [torch.from_numpy(wav[i * 50000: i * 50000 + n].copy()) for i, n in enumerate(lens)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from silero_vad_amd import batch_speech_timestamps
    res = batch_speech_timestamps(_audios(), OracleModel(), rank=rank, world_size=world, threshold=0.4)
    res2 = batch_speech_timestamps(_audios(), OracleModel(), rank=rank, world_size=world, threshold=0.4, balance="count")
    if rank == 0:
        assert res == res2                                      # the partition does not change anybody's result
        q.put(res)
    else:
        assert res is None and res2 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_world_size_n_gloo_matches_single_process(built, world):
    """(world 8 over 7 recordings: one rank owns nothing and still takes part in the gather.)"""
    from silero_vad_amd import batch_speech_timestamps, get_speech_timestamps
    audios = _audios()
    single = batch_speech_timestamps(audios, OracleModel(), threshold=0.4)
    direct = [get_speech_timestamps(a, OracleModel(), threshold=0.4) for a in audios]
    assert single == direct
    assert sum(len(s) for s in single) > 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert got == single


@pytest.mark.parametrize("launcher,gpus", [("self", 2), ("torchrun", 2), ("torchrun", 8)])
def test_bench_multi_gpu_launch_plumbing(launcher, gpus):
    """`bench.py --gpus N` (N = 2, and 8 as the scaling run starts it) as the driver may start it -- plain (it re-executes itself under torch.distributed.run)
    and under an explicit torch.distributed.run -- with --dry: gloo, no GPU, no VAD work; checks rendezvous on
    127.0.0.1, barrier + MAX-reduce timing, and that exactly one JSON line with n_gpus = 2 comes out of rank 0."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    tail = [str(root / "bench.py"), "--gpus", str(gpus), "--steps", "4", "--warmup", "1", "--dry"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    env = {k: v for k, v in __import__("os").environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == 4 and d["dry"] is True and d["scaling"] == "weak"


def test_bench_corpus_leg_shards_and_gathers_over_ranks():
    """`bench.py --gpus 2 | 8 --config corpus --dry`: the corpus leg's multi-rank plumbing on gloo, no GPU -- every pass is dealt
    out by duration (shard_by_duration), each rank runs its share through the ragged scheduler + native scanner (with a
    stand-in model), rank 0 gathers every recording's segment list exactly once; one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = {}
    for gpus in (8, 2, 1):
        r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(gpus), "--config", "corpus", "--dry"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
        assert len(lines) == 1
        res[gpus] = json.loads(lines[0])
    d = res[2]
    assert d["n_gpus"] == 2 and d["dry"] is True and d["ids_complete"] is True
    assert d["recordings_gathered"] == d["recordings_total"] == 48 and d["segments_total"] > 0
    # the corpus of a pass depends on the world size (world x per_pass recordings), so compare per recording: the first
    # rank-0-sized half is not the same set; what must hold is completeness on both and a plausible segment count
    assert res[1]["ids_complete"] is True and res[1]["recordings_gathered"] == 24
    e = res[8]
    assert e["n_gpus"] == 8 and e["ids_complete"] is True and e["recordings_gathered"] == e["recordings_total"] == 8 * 24
