"""ctypes front end of oracle/vad_oracle.c (TEST INFRASTRUCTURE, never imported by the product).

``Oracle`` mirrors the reference model protocol (``__call__``, ``reset_states``,
``audio_forward``; JIT!/vad/model/vad_annotator.py:14-162) on numpy arrays so tests can drive
it exactly like the reference object, plus the functional ``step`` (ONNX-graph I/O,
src/silero_vad/utils_vad.py:80-82).
"""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
WEIGHTS = HERE.parent / "silero_vad_amd" / "data" / "silero_vad_v6.weights"
_LIB = None


def build_oracle(force=False):
    so, src = HERE / "libvad_oracle.so", HERE / "vad_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "-B", "libvad_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(str(build_oracle()))
        f32p = ctypes.POINTER(ctypes.c_float)
        lib.oracle_create.restype = ctypes.c_void_p
        lib.oracle_create.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        lib.oracle_destroy.argtypes = [ctypes.c_void_p]
        lib.oracle_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, f32p, f32p, f32p,
                                    f32p, ctypes.c_int]
        lib.oracle_forward_audio.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_long, f32p, ctypes.c_long, f32p, f32p, f32p]
        lib.oracle_stage_floats.argtypes = [ctypes.c_int]
        _LIB = lib
    return _LIB


def load_weights_blob(path=WEIGHTS):
    return Path(path).read_bytes()


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


class Oracle:
    sample_rates = [8000, 16000]

    def __init__(self, weights_path=WEIGHTS):
        blob = load_weights_blob(weights_path)
        self._h = _lib().oracle_create(blob, len(blob))
        if not self._h:
            raise RuntimeError("oracle: bad weight container")
        self.reset_states()

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().oracle_destroy(self._h)
            self._h = None

    # ---- functional forms -------------------------------------------------------------------
    def step(self, x1, state, sr, stages=False):
        """x1 [B, C+N] float32, state [2, B, 128] -> (prob [B], new_state[, stage dict])."""
        x1 = np.ascontiguousarray(x1, np.float32)
        st = np.array(state, np.float32, copy=True, order="C")
        B = x1.shape[0]
        prob = np.empty(B, np.float32)
        nst = _lib().oracle_stage_floats(sr) if stages else 0
        stage = np.empty((B, nst), np.float32) if stages else None
        rc = _lib().oracle_step(self._h, sr, B, _p(x1), _p(st), _p(prob),
                                _p(stage) if stages else None, nst)
        if rc:
            raise ValueError(f"oracle: unsupported sr {sr}")
        if not stages:
            return prob, st
        K = 129 if sr == 16000 else 65
        cuts = np.cumsum([0, K * 4, 128 * 4, 64 * 2, 64, 128])
        shp = [(K, 4), (128, 4), (64, 2), (64, 1), (128, 1)]
        names = ["mag", "enc0", "enc1", "enc2", "enc3"]
        d = {nm: stage[:, cuts[i]:cuts[i + 1]].reshape(B, *shp[i]) for i, nm in enumerate(names)}
        return prob, st, d

    def forward_audio(self, pcm, sr, ctx=None, state=None):
        """pcm [B, L] -> probs [B, ceil(L/N)], ctx, state (explicit carried state)."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        B, L = pcm.shape
        N = 512 if sr == 16000 else 256
        C = N // 8
        T = (L + N - 1) // N
        ctx = np.zeros((B, C), np.float32) if ctx is None else np.array(ctx, np.float32, order="C")
        state = np.zeros((2, B, 128), np.float32) if state is None else \
            np.array(state, np.float32, order="C")
        probs = np.empty((B, T), np.float32)
        rc = _lib().oracle_forward_audio(self._h, sr, B, L, _p(pcm), pcm.shape[1], _p(ctx),
                                         _p(state), _p(probs))
        if rc:
            raise ValueError(f"oracle: unsupported sr {sr}")
        return probs, ctx, state

    # ---- the reference's stateful protocol ------------------------------------------------------
    def reset_states(self, batch_size=1):
        self._state = np.zeros((2, 0, 128), np.float32)
        self._context = np.zeros((0, 0), np.float32)
        self._last_sr = 0
        self._last_batch_size = 0

    def __call__(self, x, sr):
        x = np.asarray(x, np.float32)
        if x.ndim == 1:
            x = x[None]
        N = 512 if sr == 16000 else 256
        C = N // 8
        assert x.shape[1] == N and sr in (8000, 16000)
        B = x.shape[0]
        if (self._last_sr and self._last_sr != sr) or \
                (self._last_batch_size and self._last_batch_size != B):
            self.reset_states()
        if self._context.size == 0:
            self._context = np.zeros((B, C), np.float32)
            self._state = np.zeros((2, B, 128), np.float32)
        x1 = np.concatenate([self._context, x], 1)
        prob, self._state = self.step(x1, self._state, sr)
        self._context = x1[:, -C:].copy()
        self._last_sr, self._last_batch_size = sr, B
        return prob[:, None]

    def audio_forward(self, x, sr):
        x = np.asarray(x, np.float32)
        if x.ndim == 1:
            x = x[None]
        self.reset_states()
        probs, self._context, self._state = self.forward_audio(x, sr)
        self._last_sr, self._last_batch_size = sr, x.shape[0]
        return probs
