#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the authoring container: it imports the unmodified reference package from
/root/reference/src (the absent ``torchaudio`` is stubbed -- it is used only by
read_audio/save_audio, utils_vad.py:138-191) and drives the TorchScript model
``silero_vad.jit`` on CPU.  /root/reference does not exist on the GPU box, so everything a
test needs (audio + expected numbers) is committed here.

Protocols (all trace to the reference):
  * wav      -- per-chunk ``model(chunk, sr)`` over a real-speech fixture exactly as
               get_speech_timestamps does (utils_vad.py:323-336), final LSTM state kept
  * synth    -- examples/openvino/verify.py:31-51 synthetic signal, default_rng(42), chained state
  * noise    -- examples/onnx_sequence/run.py:159-169: 0.03*N(0,1) audio (rng 17+sr),
               random initial state (rng 29+sr)*0.01, explicit-state entry of the net
  * batch    -- ``audio_forward`` on B tiled streams, ragged length (right zero pad)
  * stage    -- per-layer activations of 16 chunks (mag, enc0..3, h, c) for kernel bring-up
  * segments -- get_speech_timestamps / VADIterator outputs for several argument sets,
               and the examples/openvino/verify.py:116-127 segment-count KATs (29 / 79)
Round-5 additions (golden_ext_{16k,8k}.npz + golden_ext.json; the files above are untouched and still reproduce bit for bit):
  * nonfinite -- ``model(x[B, N], sr)`` on 6 speech streams: clean | NaN sample in chunk 3 | +Inf in chunk 5 | -Inf in the
               last (context) samples of chunk 4 | 1e20 (finite, overflows the magnitude) in chunk 6 | clean; then
               ``reset_states()`` and 4 clean chunks.  torch.relu / aten::lstm_cell propagate NaN
               (JIT!/vad/utils/model_utils.py:19-25, JIT!/torch/nn/modules/rnn.py:69): the poisoned stream reads NaN from
               that chunk until the reset.  Plus get_speech_timestamps on the fixture with one NaN sample inside speech
               and inside silence (utils_vad.py:328,352-361: a NaN probability passes neither threshold test).
  * gain      -- the wav protocol at gain 0.1 and 0.01 (quiet speech), probabilities + final state + default segments
  * decim     -- (16 kHz fixture only) test.wav[::2] through the 8 kHz net, examples/onnx_sequence/README.md:61
  * sr48000   -- get_speech_timestamps(np.repeat(wav, 3), sampling_rate=48000) (x[::3] front door, utils_vad.py:301-305)
  * srswitch  -- a 4-stream batch whose calls switch sr 16000 -> 8000 -> 16000 (auto reset,
               JIT!/vad/model/vad_annotator.py:37-57)
"""
import json
import sys
import types
import wave
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent

ta = types.ModuleType("torchaudio")
ta.__version__ = "2.8.0"
sys.modules["torchaudio"] = ta
sys.path.insert(0, str(REF / "src"))

import torch  # noqa: E402
import silero_vad  # noqa: E402
from silero_vad import VADIterator, get_speech_timestamps  # noqa: E402

torch.set_num_threads(1)

WAVS = {16000: REF / "tests/data/test.wav", 8000: REF / "examples/c++/aepyx_8k.wav"}
CHUNK = {16000: 512, 8000: 256}
CTX = {16000: 64, 8000: 32}


def load_wav_i16(path, sr):
    w = wave.open(str(path))
    assert w.getframerate() == sr and w.getnchannels() == 1 and w.getsampwidth() == 2
    return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).copy()


def synthetic_audio(sr, rng):
    # examples/openvino/verify.py:31-51 (same construction, re-expressed)
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
    # examples/openvino/verify.py:116-127 rule
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


def chained_probs(model, audio_f32, sr, pad_tail):
    n = CHUNK[sr]
    model.reset_states()
    probs = []
    L = len(audio_f32)
    end = L if pad_tail else (L // n) * n
    for s in range(0, end, n):
        c = audio_f32[s:s + n]
        if len(c) < n:
            c = np.concatenate([c, np.zeros(n - len(c), np.float32)])
        probs.append(model(torch.from_numpy(c), sr).item())
    return np.asarray(probs, np.float32), model._state.numpy().copy(), model._context.numpy().copy()


def main():
    model = silero_vad.load_silero_vad()
    seg_json = {}
    for sr in (16000, 8000):
        n, ctx = CHUNK[sr], CTX[sr]
        tag = "16k" if sr == 16000 else "8k"
        pcm = load_wav_i16(WAVS[sr], sr)
        np.savez_compressed(HERE / f"audio_{tag}.npz", pcm=pcm)
        wav = pcm.astype(np.float32) / 32768.0
        out = {}

        # --- wav protocol -------------------------------------------------------------
        p, st, cx = chained_probs(model, wav, sr, pad_tail=True)
        out["probs_wav"], out["state_wav"], out["ctx_wav"] = p, st, cx
        kat = kat_segments(p)
        print(sr, "wav chunks", len(p), "kat segments", len(kat), "mean", p.mean())

        # --- synth protocol -----------------------------------------------------------
        syn = synthetic_audio(sr, np.random.default_rng(42))
        p, st, cx = chained_probs(model, syn, sr, pad_tail=False)
        out["probs_synth"], out["state_synth"] = p, st
        out["synth_checksum"] = np.asarray([float(np.abs(syn).sum()), float(syn[sr * 8 + 17])], np.float64)
        print(sr, "synth chunks", len(p), "max", p.max())

        # --- noise protocol (explicit initial state through the inner net) -------------
        rng = np.random.default_rng(17 + sr)
        noise = (rng.standard_normal(round(8.0 * sr)) * 0.03).astype(np.float32)
        rng = np.random.default_rng(29 + sr)
        st0 = (rng.standard_normal((2, 1, 128)) * 0.01).astype(np.float32)
        net = model._model if sr == 16000 else model._model_8k
        state = torch.from_numpy(st0.copy())
        cxt = torch.zeros(1, ctx)
        probs = []
        for s in range(0, (len(noise) // n) * n, n):
            x1 = torch.cat([cxt, torch.from_numpy(noise[s:s + n])[None]], 1)
            o, state = net(x1, state)
            cxt = x1[:, -ctx:]
            probs.append(o.item())
        out["probs_noise"] = np.asarray(probs, np.float32)
        out["state_noise_init"], out["state_noise_final"] = st0, state.numpy().copy()

        # --- batch protocol (audio_forward, ragged length) ------------------------------
        B, T = 6, 96
        L = T * n - 100
        rows = np.stack([np.roll(wav, -b * 7919)[:L] for b in range(B)])
        pb = model.audio_forward(torch.from_numpy(rows), sr).numpy()
        out["probs_batch"], out["state_batch"] = pb.astype(np.float32), model._state.numpy().copy()
        out["batch_meta"] = np.asarray([B, T, L, 7919], np.int64)

        # --- stage protocol: per-layer activations of 16 consecutive chunks -------------
        off = 40 * n  # inside speech for both fixtures
        xs = np.stack([wav[off + i * n - ctx: off + (i + 1) * n] for i in range(16)])
        xt = torch.from_numpy(xs)
        mag = net.stft(xt)
        feats = [mag]
        h = mag
        for i in range(4):
            h = getattr(net.encoder, str(i))(h)
            feats.append(h)
        rng = np.random.default_rng(5 + sr)
        st_in = (rng.standard_normal((2, 16, 128)) * 0.5).astype(np.float32)
        o, st_out = net(xt, torch.from_numpy(st_in.copy()))
        out["stage_x"] = xs
        out["stage_mag"] = feats[0].numpy()
        for i in range(4):
            out[f"stage_enc{i}"] = feats[i + 1].numpy()
        out["stage_state_in"], out["stage_state_out"] = st_in, st_out.numpy()
        out["stage_prob"] = o.numpy()

        np.savez_compressed(HERE / f"golden_{tag}.npz", **out)

        # --- segment / iterator goldens ---------------------------------------------------
        wav_t = torch.from_numpy(wav)
        variants = {
            "default": {},
            "thr03": dict(threshold=0.3),
            "thr07_neg02": dict(threshold=0.7, neg_threshold=0.2),
            "sil300": dict(min_silence_duration_ms=300),
            "pad100": dict(speech_pad_ms=100),
            "pad0": dict(speech_pad_ms=0),
            "minspeech1000": dict(min_speech_duration_ms=1000),
            "max6": dict(max_speech_duration_s=6),
            "max6_legacy": dict(max_speech_duration_s=6, use_max_poss_sil_at_max_speech=False),
            "max3_sil40": dict(max_speech_duration_s=3, min_silence_at_max_speech=40),
            "seconds": dict(return_seconds=True),
            "seconds_res3": dict(return_seconds=True, time_resolution=3),
        }
        segs = {}
        for name, kw in variants.items():
            segs[name] = {"kwargs": kw,
                          "out": get_speech_timestamps(wav_t, model, sampling_rate=sr, **kw)}
        if sr == 16000:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                segs["sr32000"] = {"kwargs": dict(sampling_rate=32000),
                                   "out": get_speech_timestamps(wav_t, model, sampling_rate=32000)}
        it_out = {}
        for name, (ikw, ckw) in {"default": ({}, {}),
                                 "seconds": ({}, dict(return_seconds=True)),
                                 "thr03_sil300_pad100": (dict(threshold=0.3, min_silence_duration_ms=300,
                                                              speech_pad_ms=100), {})}.items():
            it = VADIterator(model, sampling_rate=sr, **ikw)
            ev = []
            for s in range(0, (len(wav) // n) * n, n):
                e = it(wav_t[s:s + n], **ckw)
                if e:
                    ev.append(e)
            it_out[name] = {"init": ikw, "call": ckw, "events": ev}
        seg_json[tag] = {"n_samples": int(len(wav)), "kat_segments_thr05_min8": len(kat),
                         "timestamps": segs, "iterator": it_out}
        print(sr, "default segments", len(segs["default"]["out"]), "iterator events",
              len(it_out["default"]["events"]))

    (HERE / "golden_segments.json").write_text(json.dumps(seg_json, indent=1))
    extended(model)


NONFINITE = [  # (stream, chunk, sample position as a fraction of the chunk, value)
    (1, 3, 0.20, float("nan")),
    (2, 5, 0.50, float("inf")),
    (3, 4, 0.98, float("-inf")),      # inside the last C samples: also part of chunk 5's context
    (4, 6, 0.40, 1e20),               # finite; (w * 1e20)^2 overflows fp32 in the magnitude -> Inf -> NaN in encoder 0
]


def first_gap(segs, lo, min_len):
    """start of the first silence of at least min_len samples after sample lo (from the clean default segments)."""
    for a, b in zip(segs[:-1], segs[1:]):
        if a["end"] >= lo and b["start"] - a["end"] >= min_len:
            return a["end"]
    raise RuntimeError("no gap")


def extended(model):
    """Round-5 protocols (module docstring).  Everything is derived from the two committed audio fixtures."""
    import warnings
    ext_json = {}
    wavs = {}
    for sr in (16000, 8000):
        n, ctx = CHUNK[sr], CTX[sr]
        tag = "16k" if sr == 16000 else "8k"
        wav = load_wav_i16(WAVS[sr], sr).astype(np.float32) / 32768.0
        wavs[sr] = wav
        wav_t = torch.from_numpy(wav)
        out, js = {}, {}

        # --- nonfinite --------------------------------------------------------------------------------
        B, T, TR = 6, 12, 4
        off = 40 * n
        rows = np.stack([np.roll(wav, -b * 7919)[off: off + (T + TR) * n] for b in range(B)]).copy()
        pos = []
        for b, t, frac, val in NONFINITE:
            i = t * n + int(frac * n)
            rows[b, i] = val
            pos.append([b, t, i])
        model.reset_states()
        probs = [model(torch.from_numpy(rows[:, t * n:(t + 1) * n]), sr).numpy()[:, 0] for t in range(T)]
        out["nf_rows"], out["nf_pos"] = rows, np.asarray(pos, np.int64)
        out["nf_probs"], out["nf_state"] = np.stack(probs, 1), model._state.numpy().copy()
        model.reset_states()
        probs = [model(torch.from_numpy(rows[:, t * n:(t + 1) * n]), sr).numpy()[:, 0] for t in range(T, T + TR)]
        out["nf_probs_after_reset"], out["nf_state_after_reset"] = np.stack(probs, 1), model._state.numpy().copy()
        nanmap = np.isnan(out["nf_probs"])
        print(sr, "nonfinite: first NaN chunk per stream", [int(r.argmax()) if r.any() else -1 for r in nanmap],
              "state NaN rows", np.isnan(out["nf_state"]).all(axis=(0, 2)).tolist())
        clean = get_speech_timestamps(wav_t, model, sampling_rate=sr)
        speech_at = (clean[3]["start"] + clean[3]["end"]) // 2
        silence_at = first_gap(clean, clean[3]["end"], 3 * n) + n + n // 2
        js["nonfinite_timestamps"] = {}
        for name, at in (("nan_in_speech", speech_at), ("nan_in_silence", silence_at), ("inf_in_speech", speech_at)):
            w2 = wav.copy()
            w2[at] = np.inf if name.startswith("inf") else np.nan
            js["nonfinite_timestamps"][name] = {
                "at": int(at), "value": "inf" if name.startswith("inf") else "nan",
                "out": get_speech_timestamps(torch.from_numpy(w2), model, sampling_rate=sr)}
        print(sr, "nonfinite segments", {k: len(v["out"]) for k, v in js["nonfinite_timestamps"].items()},
              "clean", len(clean))

        # --- gain ---------------------------------------------------------------------------------------
        js["gain"] = {}
        for g, gt in ((0.1, "g01"), (0.01, "g001")):
            q = (wav * np.float32(g)).astype(np.float32)
            p, st, _ = chained_probs(model, q, sr, pad_tail=True)
            out[f"probs_{gt}"], out[f"state_{gt}"] = p, st
            js["gain"][gt] = {"gain": g, "out": get_speech_timestamps(torch.from_numpy(q), model, sampling_rate=sr)}
            print(sr, "gain", g, "mean prob", float(p.mean()), "segments", len(js["gain"][gt]["out"]))

        # --- sr 48000 front door (16 kHz net only: 48000 % 16000 == 0) -----------------------------------
        if sr == 16000:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                # np.repeat(wav, 3) is a 48 kHz signal whose x[::3] is the fixture itself: timestamps come back in 48 kHz samples
                js["sr48000"] = {"out": get_speech_timestamps(torch.from_numpy(np.repeat(wav, 3)), model, sampling_rate=48000)}
            print(sr, "sr48000 segments", len(js["sr48000"]["out"]))
        np.savez_compressed(HERE / f"golden_ext_{tag}.npz", **out)
        ext_json[tag] = js

    # --- decim: the 16 kHz fixture decimated by 2 through the 8 kHz net -------------------------------------
    dec = np.ascontiguousarray(wavs[16000][::2])
    p, st, cx = chained_probs(model, dec, 8000, pad_tail=True)
    segs = get_speech_timestamps(torch.from_numpy(dec), model, sampling_rate=8000)
    print("decim chunks", len(p), "mean", float(p.mean()), "segments", len(segs), "kat", len(kat_segments(p)))
    ext_json["decim_16k_to_8k"] = {"out": segs, "kat_segments_thr05_min8": len(kat_segments(p))}

    # --- srswitch: 4 streams, calls alternate between the two nets ----------------------------------------
    plan = [16000] * 3 + [8000] * 3 + [16000] * 2 + [8000] * 1
    model.reset_states()
    sw, cur = [], {16000: 40 * 512, 8000: 40 * 256}
    for sr in plan:
        n = CHUNK[sr]
        x = np.stack([np.roll(wavs[sr], -b * 7919)[cur[sr]: cur[sr] + n] for b in range(4)])
        cur[sr] += n
        sw.append(model(torch.from_numpy(x), sr).numpy()[:, 0])
    np.savez_compressed(HERE / "golden_ext_misc.npz", probs_decim=p, state_decim=st, ctx_decim=cx,
                        srswitch_plan=np.asarray(plan, np.int64), srswitch_probs=np.stack(sw, 0),
                        srswitch_state=model._state.numpy().copy())
    (HERE / "golden_ext.json").write_text(json.dumps(ext_json, indent=1))


if __name__ == "__main__":
    with torch.no_grad():
        main()
