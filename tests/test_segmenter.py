"""Native segmenter (csrc/segmenter.cpp) and the Python callers against outputs of the reference's
own get_speech_timestamps / VADIterator (tests/golden/golden_segments.json), fed with the
reference's own probabilities -- "identical segments" is a graded parity criterion."""
import math
import warnings

import numpy as np
import pytest
import torch

from conftest import SRS


class ReplayModel:
    """Duck-typed model object that replays recorded probabilities (model protocol:
    src/silero_vad/utils_vad.py:57-92)."""

    def __init__(self, probs):
        self.probs = list(map(float, probs))
        self.i = 0

    def reset_states(self):
        self.i = 0

    def __call__(self, x, sr):
        p = self.probs[self.i]
        self.i += 1
        return torch.tensor([[p]], dtype=torch.float32)


def _variants(golden):
    for tag in ("16k", "8k"):
        for name in golden["segments"][tag]["timestamps"]:
            yield tag, name


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_timestamps_all_variants(built, golden, tag):
    from silero_vad_amd import get_speech_timestamps
    sr = SRS[tag]
    info = golden["segments"][tag]
    wav = torch.from_numpy(golden[tag]["wav"])
    for name, rec in info["timestamps"].items():
        kw = dict(rec["kwargs"])
        if name == "sr32000":
            continue
        kw.setdefault("sampling_rate", sr)
        got = get_speech_timestamps(wav, ReplayModel(golden[tag]["probs_wav"]), **kw)
        assert got == rec["out"], f"{tag}/{name}"
    assert len(info["timestamps"]["default"]["out"]) == {"16k": 19, "8k": 44}[tag]


def test_sr_multiple_of_16000(built, golden):
    """sampling_rate=32000: audio[::2], warning, outputs scaled by step (utils_vad.py:301-305,447-450).
    The probabilities of the decimated signal are recomputed by the oracle in the GPU test; here
    the replayed ones come from the golden json itself via a per-chunk oracle run."""
    from oracle import Oracle
    from silero_vad_amd import get_speech_timestamps
    wav = torch.from_numpy(golden["16k"]["wav"])
    rec = golden["segments"]["16k"]["timestamps"]["sr32000"]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = get_speech_timestamps(wav, _OracleModel(Oracle()), sampling_rate=32000)
    assert any("multiply of 16000" in str(x.message) for x in w)
    assert got == rec["out"]


class _OracleModel:
    """Per-chunk protocol over the CPU oracle (tests only)."""

    def __init__(self, o):
        self.o = o

    def reset_states(self):
        self.o.reset_states()

    def __call__(self, x, sr):
        return torch.from_numpy(self.o(x.numpy(), sr))


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_vad_iterator(built, golden, tag):
    from silero_vad_amd import VADIterator
    sr = SRS[tag]
    n = 512 if sr == 16000 else 256
    wav = torch.from_numpy(golden[tag]["wav"])
    for name, rec in golden["segments"][tag]["iterator"].items():
        it = VADIterator(ReplayModel(golden[tag]["probs_wav"]), sampling_rate=sr, **rec["init"])
        ev = []
        for s in range(0, (len(wav) // n) * n, n):
            e = it(wav[s:s + n], **rec["call"])
            if e:
                ev.append(e)
        assert ev == rec["events"], f"{tag}/{name}"
    assert len(golden["segments"][tag]["iterator"]["default"]["events"]) == {"16k": 39, "8k": 92}[tag]


def test_segmenter_edge_cases(built):
    from silero_vad_amd import segment_probs
    assert segment_probs([], 0) == []
    assert segment_probs([0.9], 100) == []                       # shorter than min_speech
    # all speech: one segment spanning the audio, padding clipped at both ends
    n = 100
    assert segment_probs([0.9] * n, n * 512) == [{"start": 0, "end": n * 512}]
    # all silence
    assert segment_probs([0.1] * n, n * 512) == []
    # exactly at threshold enters (>=), exactly at neg_threshold does not leave (<)
    # (0.5 and 0.25 are exact in float32, which is what a model returns)
    probs = [0.5] + [0.25] * 30
    assert segment_probs(probs, len(probs) * 512, neg_threshold=0.25) == [{"start": 0, "end": len(probs) * 512}]
    probs = [0.5] + [0.2499] * 30
    assert segment_probs(probs, len(probs) * 512, neg_threshold=0.25, min_speech_duration_ms=0) == \
        [{"start": 0, "end": 512 + 480}]
    with pytest.raises(ValueError):
        segment_probs([0.5], 512, sampling_rate=44100)


def test_segmenter_matches_python_scan_on_random_probs(built):
    """Property test against an independent, direct transcription of the documented rules
    (SURVEY.md section 8a 'Segmenter semantics') on random probability tracks, including the
    max_speech_duration_s branches that the fixtures never reach at their default of inf."""
    from silero_vad_amd import segment_probs
    rng = np.random.default_rng(3)
    for trial in range(60):
        n = int(rng.integers(1, 400))
        # piecewise-smooth random walk in [0,1] so that runs of speech/silence exist
        p = np.clip(np.cumsum(rng.normal(0, 0.15, n)) % 2.0, 0, 2)
        p = np.where(p > 1, 2 - p, p).astype(np.float32)
        sr = int(rng.choice([8000, 16000]))
        win = 512 if sr == 16000 else 256
        L = n * win - int(rng.integers(0, win))
        kw = dict(threshold=float(rng.choice([0.3, 0.5, 0.7])),
                  min_speech_duration_ms=int(rng.choice([0, 100, 250])),
                  max_speech_duration_s=float(rng.choice([0.5, 1.0, 3.0, math.inf])),
                  min_silence_duration_ms=int(rng.choice([0, 64, 100, 300])),
                  speech_pad_ms=int(rng.choice([0, 30, 100])),
                  min_silence_at_max_speech=int(rng.choice([20, 98])),
                  use_max_poss_sil_at_max_speech=bool(rng.integers(0, 2)))
        assert segment_probs(p, L, sr, **kw) == _python_scan(p.tolist(), L, sr, **kw), (trial, kw)


def _python_scan(probs, audio_len, sr, threshold=0.5, neg_threshold=None, min_speech_duration_ms=250,
                 max_speech_duration_s=math.inf, min_silence_duration_ms=100, speech_pad_ms=30,
                 min_silence_at_max_speech=98, use_max_poss_sil_at_max_speech=True):
    win = 512 if sr == 16000 else 256
    min_speech = sr * min_speech_duration_ms / 1000
    pad = sr * speech_pad_ms / 1000
    max_speech = sr * max_speech_duration_s - win - 2 * pad
    min_sil = sr * min_silence_duration_ms / 1000
    min_sil_max = sr * min_silence_at_max_speech / 1000
    neg = max(threshold - 0.15, 0.01) if neg_threshold is None else neg_threshold
    on, segs, cur, tend, prev_end, nxt, cands = False, [], None, 0, 0, 0, []
    for i, p in enumerate(probs):
        pos = win * i
        if p >= threshold and tend:
            d = pos - tend
            if d > min_sil_max:
                cands.append((tend, d))
            tend = 0
            if nxt < prev_end:
                nxt = pos
        if p >= threshold and not on:
            on, cur = True, pos
            continue
        if on and pos - cur > max_speech:
            if use_max_poss_sil_at_max_speech and cands:
                prev_end, d = max(cands, key=lambda c: c[1])
                segs.append([cur, prev_end])
                nxt = prev_end + d
                if nxt < prev_end + pos:
                    cur = nxt
                else:
                    on, cur = False, None
                prev_end = nxt = tend = 0
                cands = []
            elif prev_end:
                segs.append([cur, prev_end])
                if nxt < prev_end:
                    on, cur = False, None
                else:
                    cur = nxt
                prev_end = nxt = tend = 0
                cands = []
            else:
                segs.append([cur, pos])
                on, cur, prev_end, nxt, tend, cands = False, None, 0, 0, 0, []
                continue
        if p < neg and on:
            if not tend:
                tend = pos
            quiet = pos - tend
            if not use_max_poss_sil_at_max_speech and quiet > min_sil_max:
                prev_end = tend
            if quiet < min_sil:
                continue
            if tend - cur > min_speech:
                segs.append([cur, tend])
            on, cur, prev_end, nxt, tend, cands = False, None, 0, 0, 0, []
    if cur is not None and audio_len - cur > min_speech:
        segs.append([cur, audio_len])
    for i, s in enumerate(segs):
        if i == 0:
            s[0] = int(max(0, s[0] - pad))
        if i != len(segs) - 1:
            gap = segs[i + 1][0] - s[1]
            if gap < 2 * pad:
                s[1] += int(gap // 2)
                segs[i + 1][0] = int(max(0, segs[i + 1][0] - gap // 2))
            else:
                s[1] = int(min(audio_len, s[1] + pad))
                segs[i + 1][0] = int(max(0, segs[i + 1][0] - pad))
        else:
            s[1] = int(min(audio_len, s[1] + pad))
    return [{"start": a, "end": b} for a, b in segs]


def test_collect_drop_chunks(built):
    from silero_vad_amd import collect_chunks, drop_chunks
    wav = torch.arange(100.0)
    ts = [{"start": 10, "end": 20}, {"start": 50, "end": 55}]
    assert collect_chunks(ts, wav).tolist() == list(range(10, 20)) + list(range(50, 55))
    kept = drop_chunks(ts, wav).tolist()
    assert kept == list(range(0, 10)) + list(range(20, 50)) + list(range(55, 100))
    sec = [{"start": 0.1, "end": 0.2}]
    assert collect_chunks(sec, wav, seconds=True, sampling_rate=100).tolist() == list(range(10, 20))
    with pytest.raises(ValueError):
        collect_chunks(sec, wav, seconds=True)
