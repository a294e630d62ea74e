"""Host logic of silero_vad_amd/streams.py on CPU: the ragged-corpus plan, the batch segmenter and
the batched streaming iterator.  The GPU engine is replaced by a stand-in over the CPU oracle
(tests may use the oracle; the product never does)."""
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
