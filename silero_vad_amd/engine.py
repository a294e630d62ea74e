"""Host-side mirror of the reference model object, backed by the HIP engine.

`HipSileroVAD` keeps the reference's duck-typed model protocol -- the surface
`get_speech_timestamps` / `VADIterator` / user code rely on:

    model(x, sr) -> Tensor[B, 1]       JIT!/vad/model/vad_annotator.py:14-90,
                                       src/silero_vad/utils_vad.py:57-92 (OnnxWrapper.__call__)
    model.reset_states()               vad_annotator.py:157-162, utils_vad.py:51-55
    model.audio_forward(x, sr)         vad_annotator.py:128-156, utils_vad.py:94-110
    model.sample_rates, model._state, model._context

with the same validation rules and ValueError texts (vad_annotator.py:17,91-127).  Arithmetic is
done by the gfx950 kernels behind the C ABI (include/silero_vad_hip.h); PyTorch is used only for
device memory and streams.  There is no CPU fallback: constructing the model without a usable
GPU raises.
"""
import contextlib
import ctypes
import os

import torch

from . import _lib
from ._lib import check, lib

_MSG_DIMS = "Too many dimensions for input audio chunk {}"
_MSG_RATES = "Supported sampling rates: {} (or multiply of 16000)"
_MSG_SHORT = "Input audio chunk is too short"
_MSG_SAMPLES = ("Provided number of samples is {} (Supported values: 256 for 8000 sample rate, "
                "512 for 16000)")


class Engine:
    """Thin RAII wrapper of a `vad_engine*` (one per GPU)."""

    def __init__(self, device=0, weights_path=None):
        if not torch.cuda.is_available():
            raise RuntimeError("silero_vad_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() "
                               "is False and there is no CPU fallback")
        blob = open(weights_path or _lib.WEIGHTS_PATH, "rb").read()
        self.device = int(device)
        self._h = ctypes.c_void_p()
        status = lib().vad_create(blob, len(blob), self.device, ctypes.byref(self._h))
        if status != _lib.VAD_OK:
            self._h = None
            raise _lib.VadError(status, f"vad_create(device={device})")
        impl = os.environ.get("SILERO_VAD_AMD_IMPL")
        if impl:
            self.set_option("impl", impl)
        self.precision = "fp32"                      # the engine's default (include/silero_vad_hip.h)
        prec = os.environ.get("SILERO_VAD_AMD_PRECISION")
        if prec:
            self.set_precision(prec)

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().vad_destroy(self._h)
            except Exception:        # interpreter shutdown: module globals may already be gone
                pass
            self._h = None

    __del__ = close

    def set_option(self, name, value):
        check(self._h, lib().vad_set_option(self._h, name.encode(), str(value).encode()))

    def set_precision(self, precision):
        """"fp32" (exact, the default) | "f16x3" (opt-in: fp16 x 3 split products on the f16 matrix
        cores, fp32 sums; narrower than fp32, see include/silero_vad_hip.h)."""
        self.set_option("precision", precision)
        self.precision = precision

    def reserve(self, sr, B, T):
        check(self._h, lib().vad_reserve(self._h, sr, B, T))

    def scratch_generation(self):
        """Changes whenever the engine reallocated scratch: captured hipGraphs must then be re-captured."""
        return int(lib().vad_scratch_generation(self._h))

    def kernel_times(self):
        """(front_ms, rec_ms, calls): kernel GPU time summed over the calls since the last query."""
        f, r, n = ctypes.c_float(), ctypes.c_float(), ctypes.c_long()
        check(self._h, lib().vad_kernel_times(self._h, ctypes.byref(f), ctypes.byref(r), ctypes.byref(n)))
        return f.value, r.value, n.value

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def forward_audio(self, pcm, sr, ctx, state, probs=None):
        """pcm [B, L] (float32 or int16, cuda), ctx [B, C], state [2, B, 128] updated in place.
        Returns probs [B, T] (cuda)."""
        assert pcm.is_cuda and pcm.dim() == 2 and pcm.stride(1) == 1
        B, L = pcm.shape
        n = 512 if sr == 16000 else 256
        T = (L + n - 1) // n
        if probs is None:
            probs = torch.empty((B, T), dtype=torch.float32, device=pcm.device)
        fn = lib().vad_forward_audio if pcm.dtype == torch.float32 else lib().vad_forward_audio_i16
        if pcm.dtype not in (torch.float32, torch.int16):
            raise TypeError(f"pcm dtype must be float32 or int16, got {pcm.dtype}")
        ld = pcm.stride(0) if B > 1 else L          # a size-1 dim may carry any stride (e.g. 0)
        ldp = probs.stride(0) if B > 1 else T
        check(self._h, fn(self._h, sr, B, L, pcm.data_ptr(), ld, ctx.data_ptr(),
                          state.data_ptr(), probs.data_ptr(), ldp, self._stream()))
        return probs

    def step(self, pcm, sr, ctx, state, prob):
        B = pcm.shape[0]
        check(self._h, lib().vad_step(self._h, sr, B, pcm.data_ptr(), pcm.stride(0) if B > 1 else pcm.shape[1], ctx.data_ptr(),
                                      state.data_ptr(), prob.data_ptr(), self._stream()))
        return prob

    def debug_frontend(self, pcm, sr, ctx):
        B, L = pcm.shape
        n = 512 if sr == 16000 else 256
        gx = torch.empty((B, L // n, 512), dtype=torch.float32, device=pcm.device)
        check(self._h, lib().vad_debug_frontend(self._h, sr, B, L, pcm.data_ptr(), pcm.stride(0) if B > 1 else L,
                                                ctx.data_ptr(), gx.data_ptr(), self._stream()))
        return gx


class HipSileroVAD:
    """Drop-in for the reference's model object (TorchScript `VADRNNJITMerge` / `OnnxWrapper`)."""

    def __init__(self, device=0, engine=None, precision="fp32"):
        """precision: "fp32" (default) = the reference's arithmetic on the exact fp32 matrix kernels.
        Opt-in: "f16x3" pins the fp16x3 split kernels (narrower than fp32; include/silero_vad_hip.h,
        option "precision"); "auto" runs f16x3 and transparently reruns a call in fp32 if it reports an
        out-of-fp16-range input (NaN probability)."""
        if precision not in ("auto", "f16x3", "fp32"):
            raise ValueError("precision must be auto|f16x3|fp32")
        self.engine = engine or Engine(device)
        self.precision = precision
        if engine is None and precision != "auto":
            self.engine.set_precision(precision)
        self.device = getattr(self.engine, "torch_device", None) or torch.device("cuda", self.engine.device)
        self.sample_rates = [8000, 16000]
        self.reset_states()

    def _device_ctx(self):
        # tests drive the host logic with a stand-in engine on the CPU (tests/replay_engine.py); the product
        # engine is always a GPU (Engine() raises otherwise)
        return torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()

    @contextlib.contextmanager
    def _selected(self):
        """An engine may be shared by several wrappers (and by direct users): select this wrapper's kernels
        for the duration of one call and put the engine back the way it was found."""
        want = "f16x3" if self.precision == "auto" else self.precision
        before = self.engine.precision
        if before != want:
            self.engine.set_precision(want)
        try:
            yield
        finally:
            if self.engine.precision != before:
                self.engine.set_precision(before)

    def _guarded(self, run):
        """Run `run()` (which advances self._state / self._context in place and returns probabilities);
        in "auto" mode, if the f16x3 kernels flag the input (NaN), restore the carried state and rerun
        in fp32.  The check reads the probabilities back, i.e. synchronises, as every caller of the
        reference protocol does anyway (`.item()` / `.cpu()`)."""
        if self.precision != "auto" or self.engine.precision != "f16x3":
            return run()
        st0, ctx0 = self._state.clone(), self._context.clone()
        out = run()
        if bool(torch.isnan(out).any()):
            self._state.copy_(st0)
            self._context.copy_(ctx0)
            self.engine.set_precision("fp32")
            try:
                out = run()
            finally:
                self.engine.set_precision("f16x3")
        return out

    # -- reference: vad_annotator.py:91-127 / utils_vad.py:33-49 --------------------------------------
    def _validate_input(self, x, sr: int):
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        if x.dim() == 1:
            x = x.unsqueeze(0)
        if x.dim() > 2:
            raise ValueError(_MSG_DIMS.format(x.dim()))
        if sr != 16000 and (sr % 16000 == 0):
            x = x[:, ::sr // 16000]
            sr = 16000
        if sr not in self.sample_rates:
            raise ValueError(_MSG_RATES.format(self.sample_rates))
        if sr / x.shape[1] > 31.25:
            raise ValueError(_MSG_SHORT)
        return x, sr

    def reset_states(self, batch_size=1):
        self._state = torch.zeros(0)
        self._context = torch.zeros(0)
        self._last_sr = 0
        self._last_batch_size = 0

    def _to_device(self, x):
        if x.dtype != torch.int16:
            x = x.to(torch.float32)
        return x.to(self.device, non_blocking=True).contiguous()

    def _ensure_state(self, sr, batch_size):
        if self._last_sr and self._last_sr != sr:
            self.reset_states()
        if self._last_batch_size and self._last_batch_size != batch_size:
            self.reset_states()
        if not len(self._context):
            ctx = 64 if sr == 16000 else 32
            self._context = torch.zeros((batch_size, ctx), dtype=torch.float32, device=self.device)
        if not len(self._state):
            self._state = torch.zeros((2, batch_size, 128), dtype=torch.float32, device=self.device)

    # -- reference: vad_annotator.py:14-90 ------------------------------------------------------------
    def __call__(self, x, sr: int):
        x, sr = self._validate_input(x, sr)
        num_samples = 512 if sr == 16000 else 256
        if x.shape[-1] != num_samples:
            raise ValueError(_MSG_SAMPLES.format(x.shape[-1]))
        batch_size = x.shape[0]
        self._ensure_state(sr, batch_size)
        with self._device_ctx(), self._selected():
            xd = self._to_device(x)
            if xd.dtype == torch.int16:
                xd = xd.to(torch.float32) / 32768.0
            out = torch.empty((batch_size, 1), dtype=torch.float32, device=self.device)
            out = self._guarded(lambda: self.engine.step(xd, sr, self._context, self._state, out))
        self._last_sr = sr
        self._last_batch_size = batch_size
        return out

    forward = __call__

    # -- reference: vad_annotator.py:128-156 (returns a CPU tensor, like the reference) ----------------
    def audio_forward(self, x, sr: int):
        return self.audio_forward_device(x, sr).cpu()

    def audio_forward_device(self, x, sr: int, guarded=None):
        """audio_forward that leaves the probabilities in HBM.  No host synchronisation -- except in "auto" mode,
        where the f16x3 range check has to read them back (pass guarded=False to skip the check and handle flagged
        rows yourself, as the ragged corpus path does)."""
        if guarded is None:
            guarded = self.precision == "auto"
        x, sr = self._validate_input(x, sr)
        self.reset_states()
        batch_size = x.shape[0]
        self._ensure_state(sr, batch_size)
        with self._device_ctx(), self._selected():
            xd = self._to_device(x)
            run = lambda: self.engine.forward_audio(xd, sr, self._context, self._state)
            probs = self._guarded(run) if guarded else run()
        self._last_sr = sr
        self._last_batch_size = batch_size
        return probs


def load_silero_vad(onnx=False, opset_version=16, device=0, precision="fp32"):
    """Reference signature (src/silero_vad/model.py:6) plus `device` and `precision`.
    `onnx`/`opset_version` are accepted for source compatibility and ignored: there is one backend,
    the HIP engine."""
    return HipSileroVAD(device=device, precision=precision)
