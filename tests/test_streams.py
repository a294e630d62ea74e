"""Host logic of silero_vad_amd/streams.py on CPU: the ragged-corpus plan, the batch segmenter and
the batched streaming iterator.  The GPU engine is replaced by a stand-in over the CPU oracle
(tests may use the oracle; the product never does)."""
import os

import numpy as np
import pytest
import torch

from test_sharding import OracleModel


def _wav():
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "audio_16k.npz"))["pcm"]
    return gold


def test_ragged_plan_covers_everything_once_and_bounds_waste():
    from silero_vad_amd import RaggedPlan
    rng = np.random.default_rng(3)
    lens = [int(v) for v in rng.integers(1, 200_000, size=300)] + [0, 0, 512, 513]
    plan = RaggedPlan(lens, max_waste=0.1, max_bytes=8 << 20)
    seen = sorted(i for b in plan.buckets for i in b)
    assert seen == [i for i, n in enumerate(lens) if n > 0]
    assert sorted(plan.empty) == [i for i, n in enumerate(lens) if n == 0]
    for b in plan.buckets:
        top = lens[b[0]]
        assert all(lens[i] <= top for i in b)
        if len(b) > 1:
            assert 1.0 - sum(lens[i] for i in b) / (top * len(b)) <= 0.1 + 1e-9
            assert top * len(b) * 4 <= 8 << 20
    assert plan.real_samples() == sum(lens)
    assert plan.padded_samples() <= plan.real_samples() / 0.9 + max(lens)


def test_bucket_and_window_planners_equal_the_sequential_rules_fuzz():
    """RaggedPlan / WindowedPlan plan on arrays (prefix sums, searchsorted); the rules they implement are sequential: recordings by
    descending length, a bucket closes when the next member would push its padding waste above max_waste or its padded bytes above
    max_bytes; a window closes where the arena order turns back or its span would exceed window_bytes.  Both restated here as plain
    loops and compared on random sets (empty recordings, equal lengths, a ring that wraps, permuted hand-over order)."""
    from silero_vad_amd import PackedRecordings, RaggedPlan
    from silero_vad_amd.streams import WindowedPlan
    rng = np.random.default_rng(11)

    def buckets_loop(lens, max_waste, max_bytes, itemsize):
        order = sorted((i for i, n in enumerate(lens) if n > 0), key=lambda i: -lens[i])      # stable: ties in index order
        out, cur, tot = [], [], 0
        for i in order:
            if cur:
                padded = lens[cur[0]] * (len(cur) + 1)
                if 1.0 - (tot + lens[i]) / padded > max_waste or padded * itemsize > max_bytes:
                    out.append(cur)
                    cur, tot = [], 0
            cur.append(i)
            tot += lens[i]
        return out + ([cur] if cur else [])

    for trial in range(150):
        n = int(rng.integers(0, 250))
        kind = trial % 4
        lens = (rng.integers(0, 200_000, size=n) if kind == 0 else rng.integers(500, 520, size=n) if kind == 1
                else rng.choice([0, 512, 513, 16_000, 640_000], size=n) if kind == 2 else rng.integers(1, 40_000, size=n)).astype(np.int64)
        mw, mb, isz = float(rng.choice([0.0, 0.05, 0.15, 0.5])), int(rng.choice([1 << 12, 1 << 16, 1 << 20, 1 << 30])), int(rng.choice([2, 4]))
        ll = [int(v) for v in lens]
        plan = RaggedPlan(ll, mw, mb, isz)
        assert plan.buckets == buckets_loop(ll, mw, mb, isz), (trial, kind)
        assert plan.empty == [i for i, v in enumerate(ll) if v <= 0] and plan.lengths == ll
        if n == 0:
            continue
        # the same recordings in an arena: back to back with small gaps, as a ring that wraps, or handed over in another order
        offs = np.concatenate([[0], np.cumsum(lens + rng.integers(0, 40, size=n))[:-1]]).astype(np.int64)
        if kind == 1 and n > 6:
            per = n // 3
            offs = offs % int(offs[per - 1] + lens[per - 1] + 1)
        elif kind == 2:
            perm = rng.permutation(n)
            offs, lens = offs[perm], lens[perm]
        base = torch.zeros(int((offs + lens).max()) + 8, dtype=torch.int16)
        wbytes = int(rng.choice([50_000, 200_000, 1 << 22]))
        seen = {}
        wp = WindowedPlan(PackedRecordings(base, offs, lens), mw, mb, 2, wbytes, on_windows=lambda p: seen.update(n=len(p.windows), b=len(p.buckets)))
        live = [i for i in range(n) if lens[i] > 0]
        by_off = sorted(live, key=lambda i: offs[i])
        free = all(offs[b] >= offs[a] + lens[a] for a, b in zip(by_off, by_off[1:]))
        windows, spans, cur, a0, a1 = [], [], [], 0, 0
        for i in (by_off if free else live):
            o, e = int(offs[i]), int(offs[i] + lens[i])
            if cur and (o < a1 or (e - a0) * 2 > wbytes):
                windows.append(cur)
                spans.append((a0, a1))
                cur = []
            if not cur:
                a0 = o
            cur.append(i)
            a1 = e
        if cur:
            windows.append(cur)
            spans.append((a0, a1))
        assert wp.windows == windows and wp.span == spans, (trial, kind)
        want = [([w[j] for j in b], k) for k, w in enumerate(windows) for b in buckets_loop([int(lens[i]) for i in w], mw, mb, 2)]
        assert wp.buckets == [b for b, _ in want] and wp.window_of == [k for _, k in want]
        assert seen == {"n": len(windows), "b": 0}            # the hook fires once the windows are known, before any bucket is


@pytest.mark.parametrize("as_i16", [False, True])
def test_ragged_probs_equal_single_recording_runs(built, as_i16):
    """Zero padding to the bucket length must not change a recording's own probabilities."""
    from silero_vad_amd import ragged_probs
    pcm = _wav()
    lens = [40000, 40000, 25000, 39999, 33333, 512, 100, 16000, 0, 90, 110]
    audios = []
    for k, n in enumerate(lens):
        a = pcm[k * 45000: k * 45000 + n]
        audios.append(torch.from_numpy(a.copy() if as_i16 else a.astype(np.float32) / 32768.0))
    model = OracleModel()
    if as_i16:
        inner = model.audio_forward_device
        model.audio_forward_device = lambda x, sr: inner(x.to(torch.float32) / 32768.0, sr)
    got = ragged_probs(audios, model, 16000, max_waste=0.5)
    single = OracleModel()
    for a, p in zip(audios, got):
        if len(a) == 0:
            assert p.numel() == 0
            continue
        x = a.to(torch.float32) / 32768.0 if as_i16 else a
        want = single.audio_forward_device(x[None], 16000)[0]
        assert p.shape == want.shape and torch.equal(p, want)


def test_segment_probs_batch_equals_per_stream(built):
    from silero_vad_amd import segment_probs, segment_probs_batch
    rng = np.random.default_rng(11)
    B, T = 150, 400
    walk = np.cumsum(rng.standard_normal((B, T)) * 0.6, axis=1)
    probs = torch.from_numpy((1 / (1 + np.exp(-walk))).astype(np.float32))
    nck = rng.integers(0, T + 1, size=B)
    lens = [int(max(0, n * 512 - rng.integers(0, 512))) if n else 0 for n in nck]
    for kw in ({}, {"threshold": 0.3, "min_silence_duration_ms": 300, "speech_pad_ms": 100},
               {"max_speech_duration_s": 3.0}, {"max_speech_duration_s": 2.0, "use_max_poss_sil_at_max_speech": False}):
        got = segment_probs_batch(probs, nck, lens, 16000, threads=4, **kw)
        for i in range(B):
            assert got[i] == segment_probs(probs[i, : nck[i]], lens[i], 16000, **kw), (i, kw)
    assert sum(len(g) for g in got) > 50
    with pytest.raises(ValueError):
        segment_probs_batch(probs, nck, lens, 44100)
    with pytest.raises(ValueError):
        segment_probs_batch(probs, nck[:-1], lens, 16000)


@pytest.mark.parametrize("sr", [16000, 8000])
def test_batch_vad_iterator_equals_reference_iterator_per_stream(built, sr):
    from silero_vad_amd import BatchVADIterator, VADIterator
    rng = np.random.default_rng(5)
    B, T = 6, 700
    walk = np.cumsum(rng.standard_normal((B, T)) * 0.5, axis=1)
    probs = (1 / (1 + np.exp(-walk))).astype(np.float32)
    win = 512 if sr == 16000 else 256

    class Replay:                                   # a model object that replays one row of probs
        def __init__(self, row):
            self.row, self.i = row, 0

        def reset_states(self):
            self.i = 0

        def __call__(self, x, sr):
            self.i += 1
            return torch.tensor([[self.row[self.i - 1]]])

    want = {}
    for b in range(B):
        it = VADIterator(Replay(probs[b]), sampling_rate=sr, threshold=0.55, min_silence_duration_ms=160)
        want[b] = [e for t in range(T) if (e := it(torch.zeros(win)))]
    bit = BatchVADIterator(B, threshold=0.55, sampling_rate=sr, min_silence_duration_ms=160)
    got = {b: [] for b in range(B)}
    for t in range(T):
        for slot, ev in bit.feed(probs[:, t]):
            got[slot].append(ev)
    assert got == want and sum(len(v) for v in want.values()) > 20
    # inactive slots do not advance
    bit.reset()
    bit.feed(probs[:, 0], active=[True, False] * 3)
    assert list(bit.current_sample) == [win, 0] * 3


# ---- continuous refill ---------------------------------------------------------------------------------------------
def test_refill_plan_covers_every_sample_once():
    from silero_vad_amd import RefillPlan
    rng = np.random.default_rng(4)
    lengths = [int(v) for v in rng.integers(1, 40000, size=57)] + [0, 512, 511, 513]
    for slots, slab in ((1, 3), (5, 4), (8, 32), (100, 2)):
        plan = RefillPlan(lengths, slots, slab, 512)
        width = slab * 512
        seen = {i: 0 for i, m in enumerate(lengths) if m > 0}
        busy_prev = {}
        for entries in plan.slabs:
            assert len({e[0] for e in entries}) == len(entries) <= slots           # one recording per slot and slab
            for sl, rec, at, take, reset in entries:
                assert at == seen[rec] and 0 < take <= width                          # in order, no gaps
                assert reset == (at == 0)
                if not reset:
                    assert busy_prev.get(sl) == rec and at % width == 0              # stays in its slot
                seen[rec] += take
            busy_prev = {e[0]: e[1] for e in entries}
        assert all(seen[i] == lengths[i] for i in seen)
        assert plan.empty == [i for i, m in enumerate(lengths) if m == 0]
        # a slot idles for less than one slab per recording it serves (+ the drain at the very end)
        assert plan.padded_chunks() >= plan.real_chunks()


def test_refill_plan_in_a_given_order_admits_first_come_first_served():
    """RefillPlan(order=): what the window feed plans on.  Every sample once, in order, like the default plan; recordings are ADMITTED in
    the given order (start slabs never decrease along it), empty recordings are skipped, and an order that misses or repeats a recording
    is refused."""
    from silero_vad_amd import RefillPlan
    rng = np.random.default_rng(9)
    lengths = np.concatenate([rng.integers(1, 30000, size=41), [0, 512, 0, 513]])
    order = rng.permutation(len(lengths))
    for slots, slab in ((1, 3), (4, 4), (7, 16), (64, 2)):
        plan = RefillPlan(lengths, slots, slab, 512, order=order)
        seen = np.zeros(len(lengths), dtype=np.int64)
        for k, entries in enumerate(plan.slabs):
            for sl, rec, at, take, reset in entries:
                assert at == seen[rec] and plan.first_slab[rec] <= k <= plan.last_slab[rec]
                seen[rec] += take
        assert np.array_equal(seen, lengths)
        live = order[lengths[order] > 0]
        assert np.all(np.diff(plan.first_slab[live]) >= 0)                           # first come, first served
        assert np.all(plan.first_slab[lengths == 0] == -1) and np.all(plan.last_slab[live] >= plan.first_slab[live])
        width = slab * 512
        assert np.array_equal(plan.last_slab[live] - plan.first_slab[live] + 1, (lengths[live] + width - 1) // width)
    # ramp: the slots start over `ramp` slabs, a 1 / ramp of them per slab, lowest slots first; still every sample once, in the given order
    for slots, slab, ramp in ((8, 4, 4), (16, 2, 8), (7, 4, 3), (4, 4, 8)):
        plan = RefillPlan(lengths, slots, slab, 512, order=order, ramp=ramp)
        seen = np.zeros(len(lengths), dtype=np.int64)
        for k, entries in enumerate(plan.slabs):
            for sl, rec, at, take, reset in entries:
                assert at == seen[rec]
                seen[rec] += take
        assert np.array_equal(seen, lengths)
        live = order[lengths[order] > 0]
        assert np.all(np.diff(plan.first_slab[live]) >= 0)
        g = slots // ramp
        if g >= 1:
            for k in range(ramp - 1):
                used = {e[0] for e in plan.slabs[k]}
                assert used <= set(range((k + 1) * g)) and len(used) > 0        # slab k: only the first (k + 1) g slots have started
            assert len(plan.slabs[0]) == g
        else:                                                                    # fewer slots than ramp steps: no stagger
            assert len(plan.slabs[0]) == slots
    with pytest.raises(ValueError):
        RefillPlan(lengths, 4, 4, 512, order=order[:-1] if lengths[order[-1]] > 0 else order[1:])
    with pytest.raises(ValueError):
        RefillPlan(lengths, 4, 4, 512, order=np.concatenate([order, order[:1]]))


def test_arena_windows_with_small_leading_windows():
    """_arena_windows(lead_limit=): the first windows are cut at lead_limit, 2 lead_limit, 4 lead_limit ... until the full limit -- the same
    recordings, in the same order, every one in exactly one window, no window above its limit unless it is a single recording."""
    from silero_vad_amd.streams import _arena_windows
    rng = np.random.default_rng(3)
    lens = rng.integers(1, 5000, 400)
    lens[::37] = 0
    offs = np.concatenate([[0], np.cumsum(lens + rng.integers(0, 9, 400))[:-1]])
    order0, b0, o0, e0 = _arena_windows(offs, lens, 40000)
    order1, b1, o1, e1 = _arena_windows(offs, lens, 40000, lead_limit=5000)
    assert np.array_equal(order0, order1) and len(b1) > len(b0)
    assert [a for a, _ in b1] == [0] + [b for _, b in b1[:-1]] and b1[-1][1] == len(order1)      # a partition, in order
    for i, (a, b) in enumerate(b1):
        lim = min(40000, 5000 << i)
        assert e1[b - 1] - o1[a] <= lim or b - a == 1
    assert [(a, b) for a, b in b1[5:]] != [] and all(e1[b - 1] - o1[a] <= 40000 for a, b in b1)


def test_window_buffers_are_never_shared_by_two_live_windows_fuzz():
    """_assign_window_buffers (the refill route's window feed): a buffer takes a new window only if the window it held was last read by a
    slab BEFORE the slab at which the new window's DMA is issued; windows are issued in order, each at or before its first reader; and
    the count is what a greedy interval colouring needs (the largest number of windows alive at once)."""
    from silero_vad_amd.streams import _assign_window_buffers
    rng = np.random.default_rng(21)
    for trial in range(200):
        nw = int(rng.integers(1, 60))
        first = np.sort(rng.integers(0, 80, nw))
        last = first + rng.integers(0, 1 + int(rng.integers(1, 50)), nw)
        if trial % 10 == 0:
            last[int(rng.integers(0, nw))] += 500                                   # one window pinned by a very long recording
        ahead = int(rng.integers(0, 9))
        slack = int(rng.integers(1, 4))
        buf, n_buf, issue = _assign_window_buffers(first, last, ahead, slack=slack)
        assert np.all(issue <= first) and np.all(np.diff(issue) >= 0) and np.all(issue >= 0)
        assert buf.min() == 0 and buf.max() == n_buf - 1
        for j in range(n_buf):
            ws = np.flatnonzero(buf == j)
            for a, b in zip(ws[:-1], ws[1:]):
                assert last[a] <= issue[b] - slack                                  # released `slack` slabs before the next DMA into it is issued
        alive = max(int(np.sum((issue <= t) & (last + slack - 1 >= t))) for t in range(int(last.max()) + slack + 1))
        assert n_buf == alive
    buf, n_buf, issue = _assign_window_buffers(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), 3)
    assert n_buf == 0 and len(buf) == 0


def test_refill_table_equals_its_numpy_definition():
    """vad_refill_table (native, two counting passes) against the definition it replaced: repeat every queue entry over its slabs, sort
    the rows by (slab, slot)."""
    import ctypes
    from silero_vad_amd._lib import lib
    rng = np.random.default_rng(9)
    lp = ctypes.POINTER(ctypes.c_long)
    for n, slots, width in ((0, 3, 1024), (1, 1, 512), (300, 7, 2048), (2000, 64, 4096)):
        lens = rng.integers(1, 40 * width // 3, size=n).astype(np.int64)
        rec = rng.permutation(n).astype(np.int64)
        need = np.ascontiguousarray((lens + width - 1) // width)
        start, slot = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        assert lib().vad_refill_schedule(need.ctypes.data_as(lp), n, slots, start.ctypes.data_as(lp), slot.ctypes.data_as(lp)) == 0
        n_slabs = int((start + need).max()) if n else 0
        rows = np.empty((int(need.sum()), 5), dtype=np.int64)
        cuts = np.zeros(n_slabs + 1, dtype=np.int64)
        got = lib().vad_refill_table(need.ctypes.data_as(lp), start.ctypes.data_as(lp), slot.ctypes.data_as(lp), rec.ctypes.data_as(lp),
                                     lens.ctypes.data_as(lp), n, slots, width, n_slabs, rows.ctypes.data_as(lp), cuts.ctypes.data_as(lp))
        assert got == len(rows)
        reps = np.repeat(np.arange(n), need)
        j = np.arange(len(reps)) - np.repeat(np.cumsum(need) - need, need)
        at = j * width
        want = np.stack([slot[reps], rec[reps], at, np.minimum(width, lens[reps] - at), (j == 0).astype(np.int64)], 1).reshape(-1, 5)
        slab_of = start[reps] + j
        order = np.lexsort((want[:, 0], slab_of))
        assert np.array_equal(rows, want[order])
        assert np.array_equal(cuts, np.searchsorted(slab_of[order], np.arange(n_slabs + 1)))
    assert lib().vad_refill_table(None, None, None, None, None, 1, 1, 512, 1, None, cuts.ctypes.data_as(lp)) < 0


def test_refill_plan_retires_something_early():
    """Longest-first admission alone retires nothing before the longest recordings end; a sixteenth of the slots start with the
    shortest recordings instead, without lengthening the schedule."""
    from silero_vad_amd import RefillPlan
    rng = np.random.default_rng(5)
    lengths = [int(v) for v in rng.integers(20 * 512, 900 * 512, size=4000)]
    slots, slab = 128, 16
    plan = RefillPlan(lengths, slots, slab, 512)
    width = slab * 512
    need = {i: (m + width - 1) // width for i, m in enumerate(lengths)}
    first_retired = None
    seen = {}
    for k, entries in enumerate(plan.slabs):
        for sl, rec, at, take, reset in entries:
            seen[rec] = seen.get(rec, 0) + 1
            if seen[rec] == need[rec] and first_retired is None:
                first_retired = k
    shortest = min(need.values())
    assert first_retired == shortest - 1                                   # the shortest recording runs from slab 0
    assert first_retired < max(need.values()) // 4
    # the schedule is no longer than longest-first's lower bound allows: total work / slots, rounded up, plus one recording's tail
    assert len(plan.slab_arrays) <= -(-sum(need.values()) // slots) + max(need.values())


@pytest.mark.parametrize("sr", [16000, 8000])
def test_refill_equals_one_recording_at_a_time(oracle, sr):
    """The scheduler's carried state / per-slot reset protocol, driven with the oracle behind the engine's Python
    surface (tests/replay_engine.py): every recording must get exactly the probabilities of its own audio_forward."""
    from replay_engine import ReplayEngine
    from silero_vad_amd import refill_probs, refill_speech_segments
    from silero_vad_amd.engine import HipSileroVAD
    from silero_vad_amd.timestamps import segment_probs
    n = 512 if sr == 16000 else 256
    rng = np.random.default_rng(12)
    lens = [int(v) for v in rng.integers(1, 30 * n, size=21)] + [n, 2 * n, n - 1, 0]
    t = np.arange(30 * n + 5) / sr
    base = (0.3 * np.sin(2 * np.pi * 200 * t) * (np.sin(2 * np.pi * 1.5 * t) > 0) + 0.02 * rng.standard_normal(len(t))).astype(np.float32)
    audios = [torch.from_numpy(np.roll(base, -37 * i)[:m].copy()) for i, m in enumerate(lens)]
    model = HipSileroVAD(engine=ReplayEngine(oracle))
    for slots, slab in ((4, 3), (7, 8)):
        got = refill_probs(audios, model, sr, slots=slots, slab_chunks=slab)
        for a, p in zip(audios, got):
            if len(a) == 0:
                assert p.numel() == 0
                continue
            want = oracle.audio_forward(np.pad(a.numpy(), (0, max(0, n - len(a))))[None], sr)[0]
            assert p.shape == want.shape and np.abs(p.numpy() - want).max() < 1e-6
    from silero_vad_amd import RefillPlan, refill_reserve
    from silero_vad_amd.streams import chunk_size
    plan = refill_reserve(audios, model, sr, slots=5, slab_chunks=4)          # plans (and, on a GPU, allocates); runs nothing
    assert isinstance(plan, RefillPlan) and plan.slots == 5 and plan.real_chunks() == sum((len(a) + chunk_size(sr) - 1) // chunk_size(sr) for a in audios)
    segs = refill_speech_segments(audios, model, sr, slots=5, slab_chunks=4, threshold=0.4, min_speech_duration_ms=64)
    for a, sg, p in zip(audios, segs, got):
        assert sg == (segment_probs(p, len(a), sr, threshold=0.4, min_speech_duration_ms=64) if len(a) else [])
    # int16 ingest takes the same route
    a16 = [(a * 32767).to(torch.int16) for a in audios]
    g16 = refill_probs(a16, model, sr, slots=6, slab_chunks=5)
    for a, p in zip(a16, g16):
        if len(a):
            want = oracle.audio_forward(np.pad(a.numpy().astype(np.float32) / 32768.0, (0, max(0, n - len(a))))[None], sr)[0]
            assert np.abs(p.numpy() - want).max() < 1e-6


def test_refill_and_buckets_accept_strided_int16_views(oracle):
    """Advisor finding (round 2): `batch_speech_timestamps(int16 audios, sampling_rate=32000, scheduler="refill")` hands
    `a[::2]` views to the scheduler; a non-contiguous int16 recording is the SAME dtype and must simply be made
    contiguous, not rejected as "mixed int16 / float"."""
    import warnings
    from replay_engine import ReplayEngine
    from silero_vad_amd import batch_speech_timestamps, refill_probs, ragged_probs
    from silero_vad_amd.engine import HipSileroVAD
    sr, n = 16000, 512
    rng = np.random.default_rng(4)
    t = np.arange(40 * n * 2) / (2 * sr)
    raw = ((0.3 * np.sin(2 * np.pi * 210 * t) * (np.sin(2 * np.pi * 1.2 * t) > 0) + 0.02 * rng.standard_normal(len(t))) * 32767)
    raws = [torch.from_numpy(np.roll(raw, -911 * i)[: 2 * m].astype(np.int16)) for i, m in enumerate((9 * n, 17 * n + 5, 30 * n, n))]
    views = [a[::2] for a in raws]
    assert not views[0].is_contiguous()
    model = HipSileroVAD(engine=ReplayEngine(oracle))
    want = [oracle.audio_forward(np.pad(v.numpy().astype(np.float32) / 32768.0, (0, max(0, n - len(v))))[None], sr)[0] for v in views]
    for got in (refill_probs(views, model, sr, slots=3, slab_chunks=4), ragged_probs(views, model, sr)):
        for p, w in zip(got, want):
            assert np.abs(p.numpy() - w).max() < 1e-6
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = batch_speech_timestamps(raws, model, sampling_rate=32000, scheduler="refill", threshold=0.1, min_speech_duration_ms=64)
        b = batch_speech_timestamps(raws, model, sampling_rate=32000, scheduler="buckets", threshold=0.1, min_speech_duration_ms=64)
    assert a == b
    with pytest.raises(TypeError, match="mixed int16 / float"):
        refill_probs([views[0], views[1].to(torch.float32)], model, sr)


def test_host_thread_budget_is_divided_among_local_ranks(built):
    """One process per GPU (torchrun --nproc-per-node N exports LOCAL_WORLD_SIZE): every rank may use 1/N of the node's
    CPU budget for its staging / scan workers, never all of it (8 x 16 threads under a 16-CPU quota is the
    oversubscription csrc/host_threads.hpp measured at 13x)."""
    import subprocess
    import sys
    code = "from silero_vad_amd import _lib; print(_lib.lib().vad_host_threads())"
    def run(**env):
        import os
        e = dict(os.environ, **{k: str(v) for k, v in env.items()})
        for k in ("LOCAL_WORLD_SIZE", "SILERO_VAD_AMD_HOST_THREADS"):
            if k not in env:
                e.pop(k, None)
        return int(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout.split()[-1])
    one = run()
    assert one >= 1
    assert run(LOCAL_WORLD_SIZE=8) == max(1, one // 8)
    assert run(LOCAL_WORLD_SIZE=2) == max(1, one // 2)
    assert run(LOCAL_WORLD_SIZE=8, SILERO_VAD_AMD_HOST_THREADS=5) == 5


def test_stage_rows_on_the_persistent_pool(built):
    """vad_stage_rows on many threads, repeatedly (the pool's workers are created once and reused), with a thread
    count that changes between calls."""
    import ctypes
    from silero_vad_amd import _lib
    rng = np.random.default_rng(8)
    base = rng.integers(-3000, 3000, 1 << 22).astype(np.int16)
    width, n = 300_000, 40
    for threads in (7, 3, 12, 1, 7):
        lens = rng.integers(0, width + 1, n)
        offs = rng.integers(0, len(base) - width, n)
        rows = (ctypes.c_void_p * n)(*[base.ctypes.data + 2 * int(o) for o in offs])
        clens = (ctypes.c_long * n)(*[int(v) for v in lens])
        dst = np.full((n, width), 9, np.int16)
        assert _lib.lib().vad_stage_rows(rows, clens, n, width, 2, dst.ctypes.data, threads) == 0
        for i in range(n):
            assert np.array_equal(dst[i, :lens[i]], base[offs[i]:offs[i] + lens[i]]) and not dst[i, lens[i]:].any()


def test_host_pool_survives_fork(built):
    """The process-wide helper pool after fork(): the child inherits handles of worker threads that do not exist in it.  It must (a)
    stage correctly with a pool of its own and (b) leave through exit() -- static destructors run -- without hanging on a join."""
    import ctypes
    import os
    import time
    from silero_vad_amd import _lib
    rng = np.random.default_rng(9)
    base = rng.integers(-3000, 3000, 1 << 21).astype(np.int16)
    width, n = 200_000, 24
    lens = rng.integers(0, width + 1, n)
    offs = rng.integers(0, len(base) - width, n)
    rows = (ctypes.c_void_p * n)(*[base.ctypes.data + 2 * int(o) for o in offs])
    clens = (ctypes.c_long * n)(*[int(v) for v in lens])

    def stage_ok():
        dst = np.full((n, width), 9, np.int16)
        rc = _lib.lib().vad_stage_rows(rows, clens, n, width, 2, dst.ctypes.data, 6)
        return rc == 0 and all(np.array_equal(dst[i, :lens[i]], base[offs[i]:offs[i] + lens[i]]) and not dst[i, lens[i]:].any()
                               for i in range(n))

    assert stage_ok()                                       # the parent's pool exists now (5 workers)
    pid = os.fork()
    if pid == 0:
        code = 0 if (stage_ok() and stage_ok()) else 3
        ctypes.CDLL(None).exit(code)                        # libc exit(): atexit handlers and static destructors run
    deadline = time.time() + 30
    status = None
    while time.time() < deadline:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        time.sleep(0.05)
    else:
        os.kill(pid, 9)
        os.waitpid(pid, 0)
        pytest.fail("the forked child hung (joining worker threads that do not exist in it?)")
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    assert stage_ok()                                       # and the parent's pool still works


def test_packed_recordings_windows_and_plan():
    """PackedRecordings: arena-order runs become windows of bounded span; a recording that starts before its
    predecessor ends (a ring that wrapped, overlapping views) starts a new window; WindowedPlan buckets every recording
    exactly once, inside its own window."""
    from silero_vad_amd import PackedRecordings
    from silero_vad_amd.streams import WindowedPlan
    rng = np.random.default_rng(2)
    lens = rng.integers(1000, 9000, 500).astype(np.int64)
    lens[17] = 0
    offs = np.concatenate([[0], np.cumsum((lens + 7) // 8 * 8)[:-1]])
    ring = int(offs[250])                                        # the arena is refilled from the start half way through
    offs[250:] -= ring
    base = torch.zeros(int(max(offs + lens)) + 8, dtype=torch.int16)
    rec = PackedRecordings(base, offs, lens)
    assert len(rec) == 500 and rec[3].shape[0] == lens[3] and rec[17].numel() == 0
    wins = rec.windows(max_bytes=200_000)
    assert wins[0][0] == 0 and wins[-1][1] == 500 and all(a[1] == b[0] for a, b in zip(wins, wins[1:]))
    assert any(hi == 250 for _, hi in wins)                      # the wrap is a window boundary
    for lo, hi in wins:
        assert (offs[hi - 1] + lens[hi - 1] - offs[lo]) * 2 <= 200_000 or hi - lo == 1
        assert np.all(offs[lo + 1:hi] >= offs[lo:hi - 1] + lens[lo:hi - 1])
    plan = WindowedPlan(rec, max_waste=0.2, max_bytes=60_000, itemsize=2, window_bytes=200_000)
    seen = sorted(i for b in plan.buckets for i in b)
    assert seen == [i for i in range(500) if lens[i] > 0] and plan.empty == [17]
    assert sorted(i for w in plan.windows for i in w) == seen
    for w, (a0, a1) in zip(plan.windows, plan.span):                # arena order inside a window, bounded span, everything inside it
        assert all(offs[i] <= offs[j] for i, j in zip(w, w[1:]))
        assert (a1 - a0) * 2 <= 200_000 or len(w) == 1
        assert all(a0 <= offs[i] and offs[i] + lens[i] <= a1 for i in w)
    for b, w in zip(plan.buckets, plan.window_of):
        assert set(b) <= set(plan.windows[w])
    assert 0.9 < plan.density <= 1.0                                 # a refilled ring: no byte is copied once for two recordings
    assert any(w[-1] == 249 for w in plan.windows)                  # ... so the wrap of the ring ends a window
    # an overlap-free set handed over in any order is walked in arena order: the same windows
    free = PackedRecordings(base, offs[:250], lens[:250])
    planf = WindowedPlan(free, 0.2, 60_000, 2, 200_000)
    perm = rng.permutation(250)
    plan2 = WindowedPlan(PackedRecordings(base, offs[:250][perm], lens[:250][perm]), 0.2, 60_000, 2, 200_000)
    assert sorted(tuple(sorted(perm[i] for i in w)) for w in plan2.windows) == sorted(tuple(sorted(w)) for w in planf.windows)
    # a list of views of one tensor is recognised as a packed set
    from silero_vad_amd.streams import _as_packed
    views = [base[o:o + m] for o, m in zip(offs, lens)]
    pk = _as_packed(views)
    assert pk is not None and np.array_equal(pk.offsets, offs) and np.array_equal(pk.lengths, lens) and pk.base.data_ptr() == base.data_ptr()
    assert _as_packed(views + [torch.zeros(5, dtype=torch.int16)]) is None and _as_packed([v.clone() for v in views]) is None
    with pytest.raises(ValueError):
        PackedRecordings(base, offs, lens + base.numel())
    with pytest.raises(ValueError):
        PackedRecordings(base.to(torch.float64), offs, lens)


def test_packed_recordings_through_the_cpu_stand_in(oracle):
    """The corpus functions take a PackedRecordings wherever they take a list; results as arrays or as lists of dicts."""
    from replay_engine import ReplayEngine
    from silero_vad_amd import PackedRecordings, ragged_probs, ragged_speech_segments
    from silero_vad_amd.engine import HipSileroVAD
    sr, n = 16000, 512
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "audio_16k.npz"))["pcm"]
    lens = np.array([9 * n, 17 * n + 5, 30 * n, n, 0, 12 * n + 400])
    offs = np.array([0, 20000, 90000, 300000, 0, 500000])
    base = torch.from_numpy(gold.copy())
    rec = PackedRecordings(base, offs, lens)
    lst = [base[o:o + m] for o, m in zip(offs, lens)]
    model = HipSileroVAD(engine=ReplayEngine(oracle))
    a, b = ragged_probs(rec, model, sr), ragged_probs(lst, model, sr)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    segs = ragged_speech_segments(lst, model, sr, threshold=0.4)
    counts, flat = ragged_speech_segments(rec, model, sr, threshold=0.4, as_arrays=True)
    first = np.concatenate([[0], np.cumsum(counts)])
    assert [[{"start": int(p), "end": int(q)} for p, q in flat[first[i]:first[i + 1]]] for i in range(len(lens))] == segs
    assert sum(len(s) for s in segs) > 0 and counts[4] == 0


# ---- callers' protocol details (CPU stand-in engine: the oracle behind the engine's Python surface) ----------------------------
def _stand_in(oracle):
    from replay_engine import ReplayEngine
    from silero_vad_amd.engine import HipSileroVAD
    return HipSileroVAD(engine=ReplayEngine(oracle))


def test_progress_callback_fires_slab_by_slab_with_the_reference_values(oracle):
    """src/silero_vad/utils_vad.py:330-336: one report per chunk, min(start + window, n) / n * 100.  With the one-call fast path
    the reports come slab by slab WHILE the recording is processed (audio_forward_slabs: 256 chunks per slab, carried state),
    not after the fact -- and the probabilities are those of one audio_forward call."""
    from silero_vad_amd import get_speech_timestamps
    model = _stand_in(oracle)
    wav = torch.from_numpy(_wav()[:16000 * 21 + 77])                  # 657 chunks: three slabs, the last chunk partial
    n = len(wav)
    seen, calls_at_report = [], []

    def cb(pct):
        seen.append(pct)
        calls_at_report.append(model.engine.calls["forward_audio"])

    with_cb = get_speech_timestamps(wav, model, progress_tracking_callback=cb)
    assert seen == [min(s + 512, n) / n * 100 for s in range(0, n, 512)] and seen[-1] == 100.0
    assert calls_at_report[0] == 1 and calls_at_report[255] == 1 and calls_at_report[256] == 2 and calls_at_report[-1] == 3
    model.engine.calls["forward_audio"] = 0
    assert get_speech_timestamps(wav, model) == with_cb and model.engine.calls["forward_audio"] == 1
    # 32 kHz input: slabs are cut at multiples of the RAW chunk (1024 samples)
    seen.clear()
    wav32 = torch.from_numpy(np.repeat(_wav()[:16000 * 9], 2))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = get_speech_timestamps(wav32, model, sampling_rate=32000, progress_tracking_callback=cb)
        b = get_speech_timestamps(wav32, model, sampling_rate=32000)
    assert a == b and len(seen) == (16000 * 9 + 511) // 512


def test_model_call_returns_a_tensor_where_the_input_lives(oracle):
    """src/silero_vad/utils_vad.py:91-92: the reference's model objects hand back a CPU tensor for a CPU chunk -- `.item()`
    (utils_vad.py:328, :528) and `.numpy()` callers both work."""
    model = _stand_in(oracle)
    out = model(torch.from_numpy(_wav()[:512]), 16000)
    assert out.device.type == "cpu" and out.shape == (1, 1) and out.dtype == torch.float32
    assert 0.0 <= float(out.numpy()[0, 0]) <= 1.0 and isinstance(out.item(), float)
    # rows that overlap in memory (expand: stride 0; unfold: row stride < row length) are materialised, not rejected
    row = torch.from_numpy(_wav()[:512])
    model.reset_states()
    p_expand = model(row.expand(3, -1), 16000)
    model.reset_states()
    assert torch.equal(p_expand, model(row.repeat(3, 1), 16000))
    sig = torch.from_numpy(_wav()[:512 + 2 * 256])
    x = model._to_device(sig.unfold(0, 512, 256))
    assert x.shape == (3, 512) and x.stride(0) >= 512


def test_iterator_feed_argument_errors_and_event_capacity(built):
    """vad_iterator_feed (the C entry point behind BatchVADIterator): bad arguments are refused, and a too small event buffer is
    reported by the return value (the number of events there were), never overrun."""
    import ctypes
    from silero_vad_amd import _lib
    L = _lib.lib()
    n = 6
    probs = np.full(n, 0.9, np.float32)                   # every stream starts speaking on this tick
    trig = np.zeros(n, np.uint8)
    tend = np.zeros(n, np.int64)
    cur = np.zeros(n, np.int64)
    ev = (_lib.IterEvent * 2)()                           # room for two of the six events
    ev[1].slot = -7
    args = (probs.ctypes.data, None, n, 512, 0.5, 1600.0, 480.0, trig.ctypes.data, tend.ctypes.data, cur.ctypes.data)
    m = L.vad_iterator_feed(*args, ev, 2)
    assert m == n and [ev[0].slot, ev[1].slot] == [0, 1] and ev[0].kind == 0 and ev[0].sample == 0       # max(0, 512 - 480 - 512)
    assert trig.all() and (cur == 512).all()
    assert L.vad_iterator_feed(probs.ctypes.data, None, -1, 512, 0.5, 1600.0, 480.0, trig.ctypes.data, tend.ctypes.data, cur.ctypes.data, ev, 2) == -1
    assert L.vad_iterator_feed(None, None, n, 512, 0.5, 1600.0, 480.0, trig.ctypes.data, tend.ctypes.data, cur.ctypes.data, ev, 2) == -1
    assert L.vad_iterator_feed(probs.ctypes.data, None, n, 0, 0.5, 1600.0, 480.0, trig.ctypes.data, tend.ctypes.data, cur.ctypes.data, ev, 2) == -1
    assert L.vad_iterator_feed(probs.ctypes.data, None, n, 512, 0.5, 1600.0, 480.0, trig.ctypes.data, tend.ctypes.data, cur.ctypes.data, None, 2) == -1
    # the Python wrapper checks its own arguments
    from silero_vad_amd import BatchVADIterator
    it = BatchVADIterator(4)
    with pytest.raises(ValueError, match="expected 4 probabilities"):
        it.feed(np.zeros(5, np.float32))
    with pytest.raises(ValueError, match="expected 4 active flags"):
        it.feed(np.zeros(4, np.float32), active=[True] * 3)


def test_ragged_reserve_is_harmless_without_a_gpu(built):
    """`ragged_reserve` (allocate up front what a run needs) plans and returns when the model is a CPU stand-in; the run after it is
    the run without it."""
    from silero_vad_amd import ragged_probs, ragged_reserve
    wav = _wav().astype(np.float32) / 32768.0
    audios = [torch.from_numpy(wav[i * 7000: i * 7000 + n].copy()) for i, n in enumerate([9000, 4000, 12000, 700])]
    m = OracleModel()
    assert ragged_reserve(audios, m, 16000) is None
    got = ragged_probs(audios, m, 16000)
    want = ragged_probs(audios, OracleModel(), 16000)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
