import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SRS = {"16k": 16000, "8k": 8000}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """The in-tree shared library and the oracle (built once per session if stale)."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def golden():
    out = {}
    for tag in SRS:
        d = dict(np.load(GOLD / f"golden_{tag}.npz"))
        d["pcm_i16"] = np.load(GOLD / f"audio_{tag}.npz")["pcm"]
        d["wav"] = d["pcm_i16"].astype(np.float32) / 32768.0
        d.update(np.load(GOLD / f"golden_ext_{tag}.npz"))       # round-5 protocols: nonfinite, gain (make_golden.py extended())
        out[tag] = d
    out["segments"] = json.loads((GOLD / "golden_segments.json").read_text())
    out["ext"] = json.loads((GOLD / "golden_ext.json").read_text())
    out["misc"] = dict(np.load(GOLD / "golden_ext_misc.npz"))    # decim (16 kHz fixture [::2] at 8 kHz), srswitch
    return out


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import Oracle
    return Oracle()


def synthetic_audio(sr, rng):
    """examples/openvino/verify.py:31-51 signal (same construction as tests/golden/make_golden.py)."""
    def t(sec):
        return np.arange(int(sec * sr)) / sr
    parts = [np.zeros(int(3 * sr), dtype=np.float32)]
    tt = t(4)
    parts.append((0.02 * np.sin(2 * np.pi * 60 * tt) + 0.01 * np.sin(2 * np.pi * 120 * tt)
                  + 0.005 * np.sin(2 * np.pi * 180 * tt)).astype(np.float32))
    parts.append((0.05 * rng.standard_normal(int(3 * sr))).astype(np.float32))
    tt = t(4)
    env = 0.5 * (1 + np.sign(np.sin(2 * np.pi * 4 * tt)))
    carrier = np.sin(2 * np.pi * 220 * tt) + 0.6 * np.sin(2 * np.pi * 710 * tt) \
        + 0.3 * np.sin(2 * np.pi * 2400 * tt)
    parts.append((0.15 * env * carrier + 0.02 * rng.standard_normal(len(tt))).astype(np.float32))
    tt = t(3)
    parts.append((0.1 * np.sin(2 * np.pi * (100 + 900 * tt) * tt)).astype(np.float32))
    parts.append((0.3 * rng.standard_normal(int(3 * sr))).astype(np.float32))
    parts.append(np.zeros(int(2 * sr), dtype=np.float32))
    return np.concatenate(parts)


def kat_segments(probs, thr=0.5, min_chunks=8):
    """examples/openvino/verify.py:116-127."""
    segs, start = [], None
    for i, p in enumerate(probs):
        if p >= thr and start is None:
            start = i
        elif p < thr and start is not None:
            if i - start >= min_chunks:
                segs.append((start, i))
            start = None
    if start is not None and len(probs) - start >= min_chunks:
        segs.append((start, len(probs)))
    return segs


def state_err(a, b):
    """Error of an LSTM (h, c) state relative to max(1, |ref|): the cell state c is unbounded
    (up to ~50 on the fixtures), so its tolerance is relative; h in [-1, 1] stays absolute."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()) if a.size else 0.0


def check_nonfinite(g, probs, state, probs_after, tol_prob, tol_state):
    """The reference's behaviour on NaN / Inf / overflowing samples (make_golden.py, protocol nonfinite): NaN exactly where the
    reference is NaN (sticky from the poisoned chunk on, whole (h, c) rows), everything else within tolerance."""
    want = g["nf_probs"]
    assert np.array_equal(np.isnan(probs), np.isnan(want))
    assert np.isnan(want).sum() == 9 + 7 + 8 + 6                      # streams 1-4, from chunks 3, 5, 4, 6 of 12
    ok = ~np.isnan(want)
    assert np.abs(probs[ok] - want[ok]).max() < tol_prob
    assert np.array_equal(np.isnan(state), np.isnan(g["nf_state"]))
    rows = ~np.isnan(g["nf_state"]).any(axis=(0, 2))
    assert rows.tolist() == [True, False, False, False, False, True]
    assert state_err(state[:, rows], g["nf_state"][:, rows]) < tol_state
    # after reset_states() the stream is clean again
    assert not np.isnan(probs_after).any()
    assert np.abs(probs_after - g["nf_probs_after_reset"]).max() < tol_prob
