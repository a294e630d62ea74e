"""Callers of the hot path: `get_speech_timestamps`, `VADIterator`, chunk utilities.

Same signatures, defaults, warnings and error texts as the reference
(src/silero_vad/utils_vad.py:211-455, :458-549, :552-655), so user code switches by changing the
import.  Differences in HOW, not WHAT:

  * with a `HipSileroVAD` model the whole recording goes through ONE `audio_forward` on the GPU
    (frontend parallel over time, LSTM sequential) instead of one `model(chunk).item()` round trip
    per 32 ms chunk (utils_vad.py:324-336); any other duck-typed model object still works through
    the per-chunk protocol;
  * the hysteresis scan + padding pass (utils_vad.py:338-440) runs in the native segmenter
    (csrc/segmenter.cpp via `vad_segment_probs`), not in a Python loop.
"""
import ctypes
import warnings
from typing import Callable, List

import torch

from . import _lib
from ._lib import lib


def _speech_probs(audio, model, sampling_rate, window, progress_cb, step=1):
    """Speech probability of every `window`-sample chunk of `audio` (utils_vad.py:323-336).  `audio` is at
    `sampling_rate * step` Hz when step > 1 (a multiple of 16 kHz): a model with `audio_forward_device` is then handed
    the RAW signal and rate and reads every step-th sample on the GPU (the reference's `audio[::step]`,
    utils_vad.py:301-307, without the decimated copy); any other model object gets the decimated view."""
    n = (len(audio) + step - 1) // step                  # len(audio[::step])

    def report(first, count):                            # the reference's per-chunk progress values (utils_vad.py:332-336)
        for start in range(first * window, min((first + count) * window, n), window):
            progress_cb(min(start + window, n) / n * 100)

    fast = getattr(model, "audio_forward_device", None)
    slabs = getattr(model, "audio_forward_slabs", None)
    if fast is not None and n > 0:
        x = audio.unsqueeze(0)
        if n < window:      # the reference pads every chunk to a full window (utils_vad.py:326-327), so a
            x = torch.nn.functional.pad(x, (0, window * step - len(audio)))   # recording shorter than one window is legal here
        if progress_cb and slabs is not None:
            # progress while the recording is being processed: one report per chunk, slab by slab (a slab = 256 chunks = 8 s)
            parts = []
            for first, p in slabs(x, sampling_rate * step):
                parts.append(p[0].cpu())
                report(first, parts[-1].numel())
            probs = torch.cat(parts)
        else:
            probs = fast(x, sampling_rate * step)[0].cpu()
            if progress_cb:
                report(0, probs.numel())
    else:
        if step > 1:
            audio = audio[::step]
        model.reset_states()
        vals = []
        for start in range(0, n, window):
            chunk = audio[start:start + window]
            if len(chunk) < window:
                chunk = torch.nn.functional.pad(chunk, (0, int(window - len(chunk))))
            vals.append(model(chunk, sampling_rate).item())
            if progress_cb:
                progress_cb(min(start + window, n) / n * 100)
        probs = torch.tensor(vals, dtype=torch.float32)
    return probs


def segment_probs(probs, audio_length_samples, sampling_rate=16000, threshold=0.5, neg_threshold=None,
                  min_speech_duration_ms=250, max_speech_duration_s=float("inf"),
                  min_silence_duration_ms=100, speech_pad_ms=30, min_silence_at_max_speech=98,
                  use_max_poss_sil_at_max_speech=True):
    """Speech probabilities (one per chunk) -> list of {'start','end'} in samples (native scan)."""
    probs = torch.as_tensor(probs, dtype=torch.float32).contiguous().cpu()
    p = _lib.SegmentParams()
    lib().vad_segment_params_default(ctypes.byref(p), int(sampling_rate))
    p.threshold = float(threshold)
    p.neg_threshold = -1.0 if neg_threshold is None else float(neg_threshold)
    p.min_speech_duration_ms = int(min_speech_duration_ms)
    p.max_speech_duration_s = float(max_speech_duration_s)
    p.min_silence_duration_ms = int(min_silence_duration_ms)
    p.speech_pad_ms = int(speech_pad_ms)
    p.min_silence_at_max_speech_ms = int(min_silence_at_max_speech)
    p.use_max_poss_sil_at_max_speech = 1 if use_max_poss_sil_at_max_speech else 0
    n = probs.numel()
    ptr = ctypes.cast(probs.data_ptr(), _lib.f32p) if n else None
    cap = n // 2 + 2
    while True:     # the scanner reports how many segments it found: a too small buffer is grown and the scan redone
        out = (_lib.Segment * cap)()
        m = lib().vad_segment_probs(ptr, n, int(audio_length_samples), ctypes.byref(p), out, cap)
        if m <= cap:
            break
        cap = int(m)
    _raise_scan_error(m)
    return [{"start": int(out[i].start), "end": int(out[i].end)} for i in range(m)]


def _raise_scan_error(rc):
    """Negative return codes of the native scanner (csrc/segmenter.cpp)."""
    if rc == -2:
        raise ValueError("Currently silero VAD models support 8000 and 16000 (or multiply of 16000) sample rates")
    if rc < 0:
        raise ValueError(f"native segmenter: bad arguments (code {rc})")


@torch.no_grad()
def get_speech_timestamps(audio: torch.Tensor,
                          model,
                          threshold: float = 0.5,
                          sampling_rate: int = 16000,
                          min_speech_duration_ms: int = 250,
                          max_speech_duration_s: float = float('inf'),
                          min_silence_duration_ms: int = 100,
                          speech_pad_ms: int = 30,
                          return_seconds: bool = False,
                          time_resolution: int = 1,
                          visualize_probs: bool = False,
                          progress_tracking_callback: Callable[[float], None] = None,
                          neg_threshold: float = None,
                          window_size_samples: int = 512,
                          min_silence_at_max_speech: int = 98,
                          use_max_poss_sil_at_max_speech: bool = True):
    """Split a recording into speech segments.  Arguments and return value: see the reference
    docstring (src/silero_vad/utils_vad.py:229-288); `window_size_samples` is ignored there too."""
    if not torch.is_tensor(audio):
        try:
            audio = torch.Tensor(audio)
        except Exception:
            raise TypeError("Audio cannot be casted to tensor. Cast it manually")
    if audio.dim() > 1:
        for _ in range(audio.dim()):
            audio = audio.squeeze(0)
        if audio.dim() > 1:
            raise ValueError("More than one dimension in audio. Are you trying to process audio with 2 channels?")

    step = 1
    if sampling_rate > 16000 and (sampling_rate % 16000 == 0):
        step = sampling_rate // 16000
        sampling_rate = 16000                            # (the decimation itself, audio[::step], happens in _speech_probs)
        warnings.warn('Sampling rate is a multiply of 16000, casting to 16000 manually!')
    if sampling_rate not in [8000, 16000]:
        raise ValueError("Currently silero VAD models support 8000 and 16000 (or multiply of 16000) sample rates")

    window = 512 if sampling_rate == 16000 else 256
    total = (len(audio) + step - 1) // step              # len(audio[::step]), utils_vad.py:305
    probs = _speech_probs(audio, model, sampling_rate, window, progress_tracking_callback, step)
    speeches = segment_probs(probs, total, sampling_rate, threshold, neg_threshold,
                             min_speech_duration_ms, max_speech_duration_s, min_silence_duration_ms,
                             speech_pad_ms, min_silence_at_max_speech, use_max_poss_sil_at_max_speech)

    if return_seconds:
        seconds_total = total / sampling_rate
        for seg in speeches:
            seg['start'] = max(round(seg['start'] / sampling_rate, time_resolution), 0)
            seg['end'] = min(round(seg['end'] / sampling_rate, time_resolution), seconds_total)
    elif step > 1:
        for seg in speeches:
            seg['start'] *= step
            seg['end'] *= step

    if visualize_probs:
        make_visualization(probs.tolist(), window / sampling_rate)
    return speeches


def make_visualization(probs, step):
    import pandas as pd
    pd.DataFrame({'probs': probs}, index=[i * step for i in range(len(probs))]).plot(
        figsize=(16, 8), kind='area', ylim=[0, 1.05], xlim=[0, len(probs) * step],
        xlabel='seconds', ylabel='speech probability', colormap='tab20')


class VADIterator:
    """Streaming start/end event emitter (reference: src/silero_vad/utils_vad.py:458-549).

    One chunk in, `{'start': n}` / `{'end': n}` / None out.  Note the reference's conventions, kept
    here: `current_sample` counts the END of the chunk, the exit threshold is fixed at
    `threshold - 0.15`, and reported positions are shifted back by one window."""

    def __init__(self, model, threshold: float = 0.5, sampling_rate: int = 16000,
                 min_silence_duration_ms: int = 100, speech_pad_ms: int = 30):
        if sampling_rate not in [8000, 16000]:
            raise ValueError('VADIterator does not support sampling rates other than [8000, 16000]')
        self.model = model
        self.threshold = threshold
        self.sampling_rate = sampling_rate
        self.min_silence_samples = sampling_rate * min_silence_duration_ms / 1000
        self.speech_pad_samples = sampling_rate * speech_pad_ms / 1000
        self.reset_states()

    def reset_states(self):
        self.model.reset_states()
        self.triggered = False
        self.temp_end = 0
        self.current_sample = 0

    def _fmt(self, pos, return_seconds, time_resolution):
        return round(pos / self.sampling_rate, time_resolution) if return_seconds else int(pos)

    @torch.no_grad()
    def __call__(self, x, return_seconds=False, time_resolution: int = 1):
        if not torch.is_tensor(x):
            try:
                x = torch.Tensor(x)
            except Exception:
                raise TypeError("Audio cannot be casted to tensor. Cast it manually")
        win = x.shape[-1]
        self.current_sample += win
        prob = self.model(x, self.sampling_rate).item()
        loud = prob >= self.threshold

        if loud and self.temp_end:
            self.temp_end = 0
        if loud and not self.triggered:
            self.triggered = True
            start = max(0, self.current_sample - self.speech_pad_samples - win)
            return {'start': self._fmt(start, return_seconds, time_resolution)}
        if self.triggered and prob < self.threshold - 0.15:
            if not self.temp_end:
                self.temp_end = self.current_sample
            if self.current_sample - self.temp_end >= self.min_silence_samples:
                end = self.temp_end + self.speech_pad_samples - win
                self.temp_end = 0
                self.triggered = False
                return {'end': self._fmt(end, return_seconds, time_resolution)}
        return None


def _bounds(ts, seconds, sampling_rate):
    if seconds and not sampling_rate:
        raise ValueError('sampling_rate must be provided when seconds is True')
    for seg in ts:
        if seconds:          # round(), like the reference's _seconds_to_samples_tss (utils_vad.py:648-655)
            yield round(seg['start'] * sampling_rate), round(seg['end'] * sampling_rate)
        else:
            yield seg['start'], seg['end']


def collect_chunks(tss: List[dict], wav: torch.Tensor, seconds: bool = False,
                   sampling_rate: int = None) -> torch.Tensor:
    """Concatenate the audio inside the given segments (reference: utils_vad.py:552-600)."""
    parts = [wav[a:b] for a, b in _bounds(tss, seconds, sampling_rate)]
    return torch.cat(parts)        # an empty list raises, as in the reference (utils_vad.py:594)


def drop_chunks(tss: List[dict], wav: torch.Tensor, seconds: bool = False,
                sampling_rate: int = None) -> torch.Tensor:
    """Concatenate the audio OUTSIDE the given segments (reference: utils_vad.py:603-655)."""
    parts, cur = [], 0
    for a, b in _bounds(tss, seconds, sampling_rate):
        parts.append(wav[cur:a])
        cur = b
    parts.append(wav[cur:])
    return torch.cat(parts)


def read_audio(path: str, sampling_rate: int = 16000) -> torch.Tensor:
    """PCM16 mono/stereo WAV -> float tensor in [-1, 1] (÷32768, examples/cpp/wav.h:113-118).
    The reference decodes through torchaudio/torchcodec (utils_vad.py:138-172), which this image
    does not ship; other containers and resampling are out of scope here."""
    import wave

    import numpy as np
    with wave.open(str(path)) as w:
        if w.getsampwidth() != 2:
            raise ValueError("read_audio: only 16-bit PCM WAV is supported in this build")
        if w.getframerate() != sampling_rate:
            raise ValueError(f"read_audio: file is {w.getframerate()} Hz, asked for {sampling_rate} Hz "
                             "(resampling is not available in this build)")
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).reshape(-1, w.getnchannels())
    return torch.from_numpy(pcm.astype(np.float32).mean(axis=1) / 32768.0)


def save_audio(path: str, tensor: torch.Tensor, sampling_rate: int = 16000):
    """float tensor in [-1, 1] -> PCM16 mono WAV (the reference writes through torchaudio, utils_vad.py:175-191; same file)."""
    import wave

    import numpy as np
    pcm = (tensor.detach().cpu().reshape(-1).numpy() * 32768.0).clip(-32768, 32767).astype(np.int16)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sampling_rate)
        w.writeframes(pcm.tobytes())
