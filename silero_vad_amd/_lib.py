"""ctypes binding of include/silero_vad_hip.h (the in-tree libsilero_vad_hip.so).

The product path has no CPU fallback: if the shared library is missing this module raises at
import of the symbols, and `Engine(...)` raises if no gfx950 device can be opened.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int16, c_int64,
                    c_long, c_size_t, c_void_p)
from pathlib import Path

PKG = Path(__file__).resolve().parent
# SILERO_VAD_AMD_LIB selects another build of the SAME library (tools/variants.py A/B kernels)
LIB_PATH = Path(os.environ.get("SILERO_VAD_AMD_LIB") or PKG / "libsilero_vad_hip.so")
# the test build: the product's sources plus the superseded A/B forms of the frontend (option enc0=direct|winograd2);
# loaded only by the parity tests that compare those forms with the product (Engine(..., library=lib_ab()))
LIB_AB_PATH = PKG / "libsilero_vad_hip_ab.so"
WEIGHTS_PATH = PKG / "data" / "silero_vad_v6.weights"

VAD_OK = 0
VAD_ERR_ALLOC = 6
STATUS_NAMES = {1: "ARG", 2: "SAMPLE_RATE", 3: "WEIGHTS", 4: "NO_DEVICE", 5: "HIP", 6: "ALLOC",
                7: "CAPTURE", 8: "OPTION"}


class SegmentParams(Structure):
    _fields_ = [("threshold", c_double), ("neg_threshold", c_double), ("sampling_rate", c_int),
                ("min_speech_duration_ms", c_int), ("max_speech_duration_s", c_double),
                ("min_silence_duration_ms", c_int), ("speech_pad_ms", c_int),
                ("min_silence_at_max_speech_ms", c_int), ("use_max_poss_sil_at_max_speech", c_int)]


class Segment(Structure):
    _fields_ = [("start", c_int64), ("end", c_int64)]


class IterEvent(Structure):
    _fields_ = [("slot", ctypes.c_int32), ("kind", ctypes.c_int32), ("sample", c_int64)]


class PumpParams(Structure):
    _fields_ = [("sampling_rate", c_int), ("streams", c_int), ("parts", c_int), ("ring_slots", c_int), ("threshold", c_double),
                ("min_silence_duration_ms", c_int), ("speech_pad_ms", c_int)]


class PumpStats(Structure):
    _fields_ = [("ticks", c_long), ("events", c_long), ("wall_ms", c_double), ("tick_ms_p50", c_double), ("tick_ms_p95", c_double),
                ("tick_ms_max", c_double), ("fill_ms_mean", c_double), ("submit_ms_mean", c_double), ("wait_ms_mean", c_double),
                ("fill_threads", c_int), ("depth", c_int), ("chunks", c_long)]


# every symbol include/silero_vad_hip.h declares: name -> (restype, argtypes)
f32p, i16p = POINTER(c_float), POINTER(c_int16)
SYMBOLS = {
    "vad_create": (c_int, [c_void_p, c_size_t, c_int, POINTER(c_void_p)]),
    "vad_destroy": (None, [c_void_p]),
    "vad_clone": (c_int, [c_void_p, POINTER(c_void_p)]),
    "vad_strerror": (c_char_p, [c_int]),
    "vad_last_error": (c_char_p, [c_void_p]),
    "vad_device": (c_int, [c_void_p]),
    "vad_geometry": (c_int, [c_int, POINTER(c_int), POINTER(c_int)]),
    "vad_set_option": (c_int, [c_void_p, c_char_p, c_char_p]),
    "vad_step": (c_int, [c_void_p, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vad_step_host_sync": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vad_step_host": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vad_step_split": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vad_step_present": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vad_step_host_present": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "vad_pump_params_default": (None, [POINTER(PumpParams), c_int, c_int]),
    "vad_pump_create": (c_int, [c_void_p, POINTER(PumpParams), POINTER(c_void_p)]),
    "vad_pump_destroy": (None, [c_void_p]),
    "vad_pump_last_error": (c_char_p, [c_void_p]),
    "vad_pump_geometry": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "vad_pump_slot": (c_void_p, [c_void_p, c_int]),
    "vad_pump_submit": (c_int, [c_void_p, c_int]),
    "vad_pump_present": (c_void_p, [c_void_p, c_int]),
    "vad_pump_submit_present": (c_int, [c_void_p, c_int, c_void_p]),
    "vad_pump_play_gaps": (c_long, [c_void_p, c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_long, c_int, c_int, c_void_p, c_long,
                                    POINTER(PumpStats)]),
    "vad_pump_submit_compact": (c_int, [c_void_p, c_int, c_void_p]),
    "vad_pump_submit_rows": (c_int, [c_void_p, c_int, c_void_p, c_long]),
    "vad_pump_play_compact": (c_long, [c_void_p, c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_long, c_int, c_int, c_void_p, c_long,
                                       POINTER(PumpStats)]),
    "vad_pump_poll": (c_long, [c_void_p, c_int, c_void_p, c_long, POINTER(c_int)]),
    "vad_pump_probs": (c_void_p, [c_void_p, c_int]),
    "vad_pump_open": (c_int, [c_void_p, c_int]),
    "vad_pump_close": (c_int, [c_void_p, c_int]),
    "vad_pump_state": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "vad_pump_play": (c_long, [c_void_p, c_void_p, c_long, c_long, c_long, c_long, c_int, c_int, c_void_p, c_long, POINTER(PumpStats)]),
    "vad_forward_audio": (c_int, [c_void_p, c_int, c_int, c_long, c_void_p, c_long, c_void_p, c_void_p,
                                  c_void_p, c_long, c_void_p]),
    "vad_forward_audio_i16": (c_int, [c_void_p, c_int, c_int, c_long, c_void_p, c_long, c_void_p,
                                      c_void_p, c_void_p, c_long, c_void_p]),
    "vad_reserve": (c_int, [c_void_p, c_int, c_int, c_long]),
    "vad_scratch_bytes": (c_size_t, [c_void_p]),
    "vad_scratch_generation": (ctypes.c_ulong, [c_void_p]),
    "vad_kernel_times": (c_int, [c_void_p, POINTER(c_float), POINTER(c_float), POINTER(c_long)]),
    "vad_segment_params_default": (None, [POINTER(SegmentParams), c_int]),
    "vad_segment_probs": (c_long, [f32p, c_long, c_long, POINTER(SegmentParams), POINTER(Segment), c_long]),
    "vad_segment_probs_batch": (c_long, [f32p, c_long, c_long, POINTER(c_long), POINTER(c_long),
                                         POINTER(SegmentParams), POINTER(Segment), c_long,
                                         POINTER(c_long), c_int]),
    "vad_segment_probs_device": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p,
                                         POINTER(SegmentParams), c_void_p, c_long, c_void_p, c_void_p]),
    "vad_iterator_feed": (c_long, [c_void_p, c_void_p, c_long, c_int, c_double, c_double, c_double, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_long]),
    "vad_stage_rows": (c_int, [POINTER(c_void_p), POINTER(c_long), c_long, c_long, c_size_t, c_void_p, c_int]),
    "vad_upload_rows": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_long), c_long, c_long, c_size_t, c_void_p, c_int,
                                c_void_p]),
    "vad_refill_schedule": (c_int, [POINTER(c_long), c_long, c_long, POINTER(c_long), POINTER(c_long)]),
    "vad_refill_table": (c_long, [POINTER(c_long)] * 5 + [c_long, c_long, c_long, c_long, POINTER(c_long), POINTER(c_long)]),
    "vad_streams_overlap": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vad_host_register": (c_int, [c_void_p, c_size_t]),
    "vad_host_unregister": (c_int, [c_void_p]),
    "vad_host_threads": (c_int, []),
    "vad_bind_host_to_device": (c_int, [c_int]),
    "vad_debug_packed_floats": (c_long, [c_void_p, c_int, c_int]),
    "vad_debug_packed_copy": (c_int, [c_void_p, c_int, c_int, f32p, c_long]),
    "vad_create_host_only": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "vad_debug_activation": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_long, c_void_p]),
    "vad_debug_foreign_load": (c_int, [c_void_p, c_int, c_int, c_long, c_void_p]),
    "vad_debug_frontend": (c_int, [c_void_p, c_int, c_int, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
}

_lib = None
_lib_ab = None


def _load(path):
    if not path.exists():
        raise OSError(f"{path} not built: run `python __graft_entry__.py` (hipcc, gfx950)")
    handle = ctypes.CDLL(str(path))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(handle, name)
        fn.restype, fn.argtypes = res, args
    return handle


def lib():
    """Load the product library once; raises OSError/AttributeError if it or a symbol is missing."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def lib_ab():
    """The test build (A/B forms of the frontend); only tests ask for it."""
    global _lib_ab
    if _lib_ab is None:
        _lib_ab = _load(LIB_AB_PATH)
    return _lib_ab


class VadError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().vad_strerror(status).decode()
        super().__init__(f"{msg}{': ' + detail if detail else ''} [VAD_ERR_{STATUS_NAMES.get(status, status)}]")


def check(handle, status, library=None):
    if status != VAD_OK:
        detail = (library or lib()).vad_last_error(handle).decode() if handle else ""
        raise VadError(status, detail)
