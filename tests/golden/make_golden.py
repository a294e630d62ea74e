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


if __name__ == "__main__":
    with torch.no_grad():
        main()
