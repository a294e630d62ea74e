"""Host-side mirror of the reference model object, backed by the HIP engine.

`HipSileroVAD` keeps the reference's duck-typed model protocol -- the surface
`get_speech_timestamps` / `VADIterator` / user code rely on:

    model(x, sr) -> Tensor[B, 1]       JIT!/vad/model/vad_annotator.py:14-90,
                                       src/silero_vad/utils_vad.py:57-92 (OnnxWrapper.__call__)
    model.reset_states()               vad_annotator.py:157-162, utils_vad.py:51-55
    model.audio_forward(x, sr)         vad_annotator.py:128-156, utils_vad.py:94-110
    model.sample_rates, model._state, model._context

with the same validation rules and ValueError texts (vad_annotator.py:17,91-127).  Arithmetic is
done by the gfx950 kernels behind the C ABI (include/silero_vad_hip.h), in fp32; PyTorch is used only
for device memory and streams.  There is no CPU fallback: constructing the model without a usable
GPU raises.
"""
import contextlib
import ctypes
import os

import torch

from . import _lib
from ._lib import lib

_MSG_DIMS = "Too many dimensions for input audio chunk {}"
_MSG_RATES = "Supported sampling rates: {} (or multiply of 16000)"
_MSG_SHORT = "Input audio chunk is too short"
_MSG_SAMPLES = ("Provided number of samples is {} (Supported values: 256 for 8000 sample rate, "
                "512 for 16000)")


def _net_rate(sr: int):
    """(rate of the net that serves `sr`, decimation step): multiples of 16 kHz run on the 16 kHz net
    (vad_annotator.py:104-112)."""
    if sr > 16000 and sr % 16000 == 0:
        return 16000, sr // 16000
    return sr, 1


class Engine:
    """Thin RAII wrapper of a `vad_engine*` (one per GPU; `clone()` gives further handles that share the weights)."""

    def __init__(self, device=0, weights_path=None, library=None, _handle=None):
        if not torch.cuda.is_available():
            raise RuntimeError("silero_vad_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() "
                               "is False and there is no CPU fallback")
        self._L = library or lib()
        self.device = int(device)
        self.precision = "fp32"                      # the engine's one arithmetic (include/silero_vad_hip.h)
        self.options = {}                            # what set_option was given (clones start with the same)
        if _handle is not None:
            self._h = _handle
            return
        blob = open(weights_path or _lib.WEIGHTS_PATH, "rb").read()
        self._h = ctypes.c_void_p()
        status = self._L.vad_create(blob, len(blob), self.device, ctypes.byref(self._h))
        if status != _lib.VAD_OK:
            self._h = None
            raise _lib.VadError(status, f"vad_create(device={device})")
        impl = os.environ.get("SILERO_VAD_AMD_IMPL")
        if impl:
            self.set_option("impl", impl)

    def _check(self, status):
        _lib.check(self._h, status, self._L)

    def clone(self):
        """A second handle on the same GPU: shared weight images, the same options, its own scratch -- so that two
        calls can be in flight on two streams (vad_clone)."""
        h = ctypes.c_void_p()
        self._check(self._L.vad_clone(self._h, ctypes.byref(h)))
        e = Engine(self.device, library=self._L, _handle=h)
        e.options = dict(self.options)
        return e

    def close(self):
        if getattr(self, "_h", None):
            try:
                self._L.vad_destroy(self._h)
            except Exception:        # interpreter shutdown: module globals may already be gone
                pass
            self._h = None

    __del__ = close

    def set_option(self, name, value):
        self._check(self._L.vad_set_option(self._h, name.encode(), str(value).encode()))
        if name not in ("profile", "trace_ptr"):
            self.options[name] = str(value)

    def set_transient(self, name, value):
        """`set_option` for a setting a caller applies for the duration of one call and takes back: not recorded in
        `options` (clones made later do not inherit it, and nothing that keys on `options` sees a change)."""
        self._check(self._L.vad_set_option(self._h, name.encode(), str(value).encode()))

    def set_precision(self, precision):
        """The engine computes in fp32 only; kept so that callers may state it."""
        self.set_option("precision", precision)

    def _retry_alloc(self, call):
        """The library's scratch is allocated outside torch's caching allocator: when it does not fit (VAD_ERR_ALLOC, returned before
        anything was launched) while torch sits on freed blocks -- earlier calls' window buffers and staging slots, other ranks on the
        same device -- those go back to the driver and the call is made once more."""
        status = call()
        if status == _lib.VAD_ERR_ALLOC and torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            status = call()
        return status

    def reserve(self, sr, B, T):
        self._check(self._retry_alloc(lambda: self._L.vad_reserve(self._h, sr, B, T)))

    def scratch_generation(self):
        """Changes whenever the engine reallocated scratch: captured hipGraphs must then be re-captured."""
        return int(self._L.vad_scratch_generation(self._h))

    def kernel_times(self):
        """(front_ms, rec_ms, calls): kernel GPU time summed over the calls since the last query."""
        f, r, n = ctypes.c_float(), ctypes.c_float(), ctypes.c_long()
        self._check(self._L.vad_kernel_times(self._h, ctypes.byref(f), ctypes.byref(r), ctypes.byref(n)))
        return f.value, r.value, n.value

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def forward_audio(self, pcm, sr, ctx, state, probs=None):
        """pcm [B, L] (float32 or int16, cuda) at `sr` (8000, 16000, or a multiple of 16000: decimated on the
        device), ctx [B, C], state [2, B, 128] updated in place.  Returns probs [B, T] (cuda)."""
        assert pcm.is_cuda and pcm.dim() == 2 and pcm.stride(1) == 1
        B, L = pcm.shape
        net, step = _net_rate(sr)
        n = 512 if net == 16000 else 256
        T = ((L + step - 1) // step + n - 1) // n
        if probs is None:
            probs = torch.empty((B, T), dtype=torch.float32, device=pcm.device)
        if pcm.dtype not in (torch.float32, torch.int16):
            raise TypeError(f"pcm dtype must be float32 or int16, got {pcm.dtype}")
        fn = self._L.vad_forward_audio if pcm.dtype == torch.float32 else self._L.vad_forward_audio_i16
        ld = pcm.stride(0) if B > 1 else L          # a size-1 dim may carry any stride (e.g. 0)
        ldp = probs.stride(0) if B > 1 else T
        self._check(self._retry_alloc(lambda: fn(self._h, sr, B, L, pcm.data_ptr(), ld, ctx.data_ptr(),
                                                 state.data_ptr(), probs.data_ptr(), ldp, self._stream())))
        return probs

    def step(self, pcm, sr, ctx, state, prob):
        B = pcm.shape[0]
        self._check(self._L.vad_step(self._h, sr, B, pcm.data_ptr(), pcm.stride(0) if B > 1 else pcm.shape[1],
                                     ctx.data_ptr(), state.data_ptr(), prob.data_ptr(), self._stream()))
        return prob

    def step_present(self, pcm, sr, ctx, state, prob, present=None, ctx_out=None):
        """vad_step_present: one step of B live streams (pcm [B, N] float32 or int16, cuda) of which only the rows with
        present[b] != 0 (uint8 / bool cuda tensor [B]; None = all) have a chunk this tick.  An absent row's ctx / state are left
        exactly as they are and prob[b] = -1.0 (VAD_PROB_ABSENT).  ctx_out: a second context buffer (else in place)."""
        B = pcm.shape[0]
        if present is not None and (not present.is_cuda or present.numel() != B or present.element_size() != 1 or not present.is_contiguous()):
            raise ValueError("present must be a contiguous 1-byte cuda tensor of B flags")
        self._check(self._L.vad_step_present(self._h, sr, B, pcm.data_ptr(), pcm.element_size(), pcm.stride(0) if B > 1 else pcm.shape[1],
                                             ctx.data_ptr(), None if ctx_out is None else ctx_out.data_ptr(), state.data_ptr(), prob.data_ptr(),
                                             None if present is None else present.data_ptr(), self._stream()))
        return prob

    def step_host(self, host_pcm, dev_pcm, sr, ctx, state, dev_prob, host_prob, stream=None, host_present=None, dev_present=None):
        """One tick from page-locked host chunks to page-locked host probabilities on `stream` (a raw stream handle; default:
        torch's current stream): H2D, the step, D2H -- vad_step_host, asynchronous.  dev_prob None: the kernel writes the
        probabilities straight into host_prob.  host_present / dev_present (page-locked [B] bytes and its device staging): the
        flags of vad_step_host_present (absent rows keep their state)."""
        B = host_pcm.shape[0]
        s = self._stream() if stream is None else ctypes.c_void_p(stream)
        dp = None if dev_prob is None else dev_prob.data_ptr()
        if dev_pcm is None:         # no staging copy: the kernel reads the page-locked chunks where they lie (a handful of streams)
            self._check(self._L.vad_step_host(self._h, sr, B, host_pcm.data_ptr(), host_pcm.element_size(), None,
                                              ctx.data_ptr(), state.data_ptr(), dp, host_prob.data_ptr(), s))
        elif host_present is None:
            self._check(self._L.vad_step_host(self._h, sr, B, host_pcm.data_ptr(), host_pcm.element_size(), dev_pcm.data_ptr(),
                                              ctx.data_ptr(), state.data_ptr(), dp, host_prob.data_ptr(), s))
        else:
            self._check(self._L.vad_step_host_present(self._h, sr, B, host_pcm.data_ptr(), host_pcm.element_size(), dev_pcm.data_ptr(),
                                                      ctx.data_ptr(), state.data_ptr(), dp, host_prob.data_ptr(), host_present.data_ptr(),
                                                      dev_present.data_ptr(), s))

    def step_host_sync(self, host_pcm, sr, ctx, state, host_prob, stream=None):
        """vad_step_host_sync: the step on page-locked chunks read in place, returning when the B probabilities are in `host_prob`
        (the blocking `model(chunk, sr)` of the reference's callers; the wait watches the page-locked slots, not the stream)."""
        s = self._stream() if stream is None else ctypes.c_void_p(stream)
        self._check(self._L.vad_step_host_sync(self._h, sr, host_pcm.shape[0], host_pcm.data_ptr(), host_pcm.element_size(),
                                               ctx.data_ptr(), state.data_ptr(), host_prob.data_ptr(), s))

    def upload_rows(self, rows, lens, n, width, elem_size, dst, how=0):
        """Ragged rows in PINNED host memory -> dst[n, width] on the GPU, zero padded, on the current stream
        (vad_upload_rows; how 0: copy engines, 1: gather kernel).  rows / lens: ctypes arrays."""
        self._check(self._L.vad_upload_rows(self._h, rows, lens, n, width, elem_size, dst.data_ptr(), how, self._stream()))

    def streams_overlap(self, a, b) -> bool:
        """vad_streams_overlap: do kernels on torch streams a and b run beside each other (distinct hardware queues)?"""
        rc = self._L.vad_streams_overlap(self._h, a.cuda_stream, b.cuda_stream)
        if rc < 0:
            self._check(-rc)
        return rc == 1

    def debug_frontend(self, pcm, sr, ctx):
        B, L = pcm.shape
        n = 512 if sr == 16000 else 256
        gx = torch.empty((B, L // n, 512), dtype=torch.float32, device=pcm.device)
        self._check(self._L.vad_debug_frontend(self._h, sr, B, L, pcm.data_ptr(), pcm.stride(0) if B > 1 else L,
                                               ctx.data_ptr(), gx.data_ptr(), self._stream()))
        return gx


def _raw_current_stream(device):
    """The current stream's handle on `device` without building a torch.cuda.Stream object (2 us of a 40 us call)."""
    get = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if get is not None:
        return get(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


class HipSileroVAD:
    """Drop-in for the reference's model object (TorchScript `VADRNNJITMerge` / `OnnxWrapper`)."""

    def __init__(self, device=0, engine=None, precision="fp32"):
        """precision: "fp32" -- the reference's arithmetic, the only one the engine has (the parameter is kept for
        source compatibility with earlier rounds)."""
        if precision != "fp32":
            raise ValueError("precision must be fp32")
        self.engine = engine or Engine(device)
        self.precision = "fp32"
        self.device = getattr(self.engine, "torch_device", None) or torch.device("cuda", self.engine.device)
        self.sample_rates = [8000, 16000]
        self._small = None           # page-locked (chunk, probability) buffers of the B <= 16 call path (__call__)
        self._fast = None            # the last B = 1 call of that path, ready to be issued again (__call__)
        self.reset_states()

    def _device_ctx(self):
        # tests drive the host logic with a stand-in engine on the CPU (tests/replay_engine.py); the product
        # engine is always a GPU (Engine() raises otherwise)
        return torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()

    # -- reference: vad_annotator.py:91-127 / utils_vad.py:33-49 --------------------------------------
    def _validate_input(self, x, sr: int):
        """The reference's rules, returning what the reference returns (the decimated view for multiples of 16 kHz).
        The forward paths below apply the same rules through `_front_door`, which does NOT materialise `x[:, ::k]`:
        the kernels read every k-th sample themselves."""
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

    def _front_door(self, x, sr: int):
        """`_validate_input` with the decimation left to the device: (x2d at the RAW rate, raw rate, net rate,
        samples per row at the net's rate).  Same errors, in the same order."""
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        if x.dim() == 1:
            x = x.unsqueeze(0)
        if x.dim() > 2:
            raise ValueError(_MSG_DIMS.format(x.dim()))
        net, step = (16000, sr // 16000) if (sr != 16000 and sr % 16000 == 0) else (sr, 1)
        if net not in self.sample_rates:
            raise ValueError(_MSG_RATES.format(self.sample_rates))
        n_net = (x.shape[1] + step - 1) // step          # len(x[0, ::step])
        if n_net == 0 or net / n_net > 31.25:
            raise ValueError(_MSG_SHORT)
        if not getattr(self.engine, "fused_front_door", True) and step > 1:   # CPU stand-in engines (tests)
            return x[:, ::step], net, net, n_net
        return x, sr, net, n_net

    def reset_states(self, batch_size=1):
        self._fast = None
        self._state = torch.zeros(0)
        self._context = torch.zeros(0)
        self._last_sr = 0
        self._last_batch_size = 0

    def _to_device(self, x):
        if x.dtype != torch.int16:
            x = x.to(torch.float32)
        x = x.to(self.device, non_blocking=True)
        # rows may be strided (the engine takes a row pitch) -- but not overlapping: `x.expand(B, -1)` (stride 0) or an
        # `unfold` view (row stride < row length) is materialised, as the reference's operators would read it
        if x.stride(-1) == 1 and (x.shape[0] == 1 or x.stride(0) >= x.shape[1]):
            return x
        return x.contiguous()

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
        """-> Tensor[B, 1] on the device the INPUT lives on: a CPU chunk gives a CPU tensor, like the reference's model
        objects (src/silero_vad/utils_vad.py:91-92: `.item()` and `.numpy()` callers both work); a CUDA chunk keeps the
        result in HBM (no host synchronisation)."""
        fp = self._fast
        if (fp is not None and type(x) is torch.Tensor and sr == fp[0] and x.dim() == 1 and x.shape[0] == fp[1] and x.dtype == fp[2]
                and x.device.type == "cpu" and self._context is fp[3] and self._state is fp[4]):
            # the same call as the last one -- a 1-D CPU chunk of the same rate, length and dtype, the state it left: every check of
            # the general path below would come out as it did then, so go straight to the launch (2-3 us of a 36 us call)
            fp[5].copy_(x)
            stream = _raw_current_stream(self.device)
            if stream != fp[6]:
                fp[6] = stream
                fp[7][-1] = ctypes.c_void_p(stream)
            rc = fp[8](*fp[7])
            if rc:
                self.engine._check(rc)
            return fp[9].clone()
        x, sr_raw, sr, n_net = self._front_door(x, sr)
        home = x.device
        num_samples = 512 if sr == 16000 else 256
        if n_net != num_samples:
            raise ValueError(_MSG_SAMPLES.format(n_net))
        batch_size = x.shape[0]
        self._ensure_state(sr, batch_size)
        if home.type == "cpu" and self.device.type == "cuda" and sr_raw == sr and batch_size <= 16 and hasattr(self.engine, "_h"):
            # What every unmodified caller does -- `model(chunk, sr).item()` on a CPU chunk, once per 32 ms (src/silero_vad/utils_vad.py:
            # 324-336, :528): the chunk is copied (2 KB) into the model's own page-locked buffer, ONE launch reads it from there and writes
            # the probability into page-locked memory, one stream wait.  No H2D operation, no D2H operation, no device tensor made.
            sm = self._small
            dt = torch.int16 if x.dtype == torch.int16 else torch.float32
            if sm is None or sm[2] != (num_samples, dt, batch_size):
                # (buffers of 16 rows, views of the caller's batch size: made once per (rate, dtype, B), not per call)
                if sm is None or sm[2][:2] != (num_samples, dt):
                    full = (torch.empty((16, num_samples), dtype=dt, pin_memory=True), torch.empty((16,), dtype=torch.float32, pin_memory=True))
                else:
                    full = sm[3]
                sm = self._small = (full[0][:batch_size], full[1][:batch_size].unsqueeze(1), (num_samples, dt, batch_size), full)
            pcm, prob = sm[0], sm[1]                # prob: the [B, 1] view of the page-locked slots
            pcm.copy_(x)
            eng = self.engine
            # (vad_step_host_sync returns when the probabilities are in `prob`: it watches the page-locked slots, not the stream)
            rc = eng._L.vad_step_host_sync(eng._h, sr, batch_size, pcm.data_ptr(), pcm.element_size(), self._context.data_ptr(),
                                           self._state.data_ptr(), prob.data_ptr(), ctypes.c_void_p(_raw_current_stream(self.device)))
            if rc:
                eng._check(rc)
            self._last_sr = sr
            self._last_batch_size = batch_size
            if batch_size == 1 and x.dtype == dt:        # remember the call: [rate, samples, dtype, context, state, chunk row, stream, args, fn, result view]
                stream = _raw_current_stream(self.device)
                self._fast = [sr, num_samples, dt, self._context, self._state, pcm[0], stream,
                              [eng._h, sr, 1, pcm.data_ptr(), pcm.element_size(), self._context.data_ptr(), self._state.data_ptr(),
                               prob.data_ptr(), ctypes.c_void_p(stream)], eng._L.vad_step_host_sync, prob]
            return prob.clone()
        with self._device_ctx():
            xd = self._to_device(x)
            if xd.dtype == torch.int16:
                xd = xd.to(torch.float32) / 32768.0
            out = torch.empty((batch_size, 1), dtype=torch.float32, device=self.device)
            if sr_raw == sr:
                self.engine.step(xd, sr, self._context, self._state, out)
            else:       # 32 / 48 / ... kHz: one chunk through the front door (every k-th sample is read on the device)
                self.engine.forward_audio(xd, sr_raw, self._context, self._state, out)
        self._last_sr = sr
        self._last_batch_size = batch_size
        return out if home.type == self.device.type else out.to(home)

    forward = __call__

    # -- reference: vad_annotator.py:128-156 (returns a CPU tensor, like the reference) ----------------
    def audio_forward(self, x, sr: int):
        return self.audio_forward_device(x, sr).cpu()

    def audio_forward_device(self, x, sr: int):
        """audio_forward that leaves the probabilities in HBM: no host synchronisation."""
        x, sr_raw, sr, _ = self._front_door(x, sr)
        self.reset_states()
        batch_size = x.shape[0]
        self._ensure_state(sr, batch_size)
        with self._device_ctx():
            xd = self._to_device(x)
            probs = self.engine.forward_audio(xd, sr_raw, self._context, self._state)
        self._last_sr = sr
        self._last_batch_size = batch_size
        return probs


    def audio_forward_slabs(self, x, sr: int, slab_chunks: int = 256):
        """`audio_forward_device` in time slabs of `slab_chunks` chunks with the state carried between them: yields
        (index of the slab's first chunk, probs[B, t] in HBM) slab by slab -- the same probabilities as one call (the engine slabs
        long inputs internally the same way).  What `get_speech_timestamps` uses to fire its `progress_tracking_callback` while
        the recording is being processed, as the reference's per-chunk loop does (src/silero_vad/utils_vad.py:330-336)."""
        x, sr_raw, sr, _ = self._front_door(x, sr)
        self.reset_states()
        batch_size = x.shape[0]
        self._ensure_state(sr, batch_size)
        raw = (512 if sr == 16000 else 256) * (sr_raw // sr)       # raw samples per chunk
        with self._device_ctx():
            xd = self._to_device(x)
            for s in range(0, xd.shape[1], slab_chunks * raw):
                yield s // raw, self.engine.forward_audio(xd[:, s:s + slab_chunks * raw], sr_raw, self._context, self._state)
        self._last_sr = sr
        self._last_batch_size = batch_size


def load_silero_vad(onnx=False, opset_version=16, device=0, precision="fp32"):
    """Reference signature (src/silero_vad/model.py:6) plus `device`.
    `onnx`/`opset_version` are accepted for source compatibility and ignored: there is one backend,
    the HIP engine, and one arithmetic, fp32."""
    return HipSileroVAD(device=device, precision=precision)
