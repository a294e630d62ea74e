"""The C-ABI shared library on a machine WITHOUT a GPU: it loads, exports every symbol the header
declares, and its host-only entry points behave (no compute calls here)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def header_functions():
    text = (ROOT / "include" / "silero_vad_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vad_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(built):
    from silero_vad_amd import _lib
    names = header_functions()
    assert len(names) >= 18
    handle = ctypes.CDLL(str(_lib.LIB_PATH))
    for n in names:
        assert hasattr(handle, n), f"{n} declared in the header but not exported"
    assert sorted(_lib.SYMBOLS) == names, "python binding and header disagree"
    _lib.lib()


def test_strerror_and_geometry(built):
    from silero_vad_amd import _lib
    L = _lib.lib()
    assert L.vad_strerror(0) == b"ok"
    for code in range(1, 9):
        assert L.vad_strerror(code) not in (b"ok", b"unknown status")
    c, x = ctypes.c_int(), ctypes.c_int()
    assert L.vad_geometry(16000, ctypes.byref(c), ctypes.byref(x)) == 0 and (c.value, x.value) == (512, 64)
    assert L.vad_geometry(8000, ctypes.byref(c), ctypes.byref(x)) == 0 and (c.value, x.value) == (256, 32)
    assert L.vad_geometry(44100, None, None) == 2


def test_weights_container_errors(built):
    from silero_vad_amd import _lib
    L = _lib.lib()
    h = ctypes.c_void_p()
    assert L.vad_create_host_only(b"garbage" * 40, 280, ctypes.byref(h)) == 3       # VAD_ERR_WEIGHTS
    blob = bytearray(_lib.WEIGHTS_PATH.read_bytes())
    blob[100:110] = b"corrupted!"                                                    # break a tensor name
    assert L.vad_create_host_only(bytes(blob), len(blob), ctypes.byref(h)) == 3
    good = _lib.WEIGHTS_PATH.read_bytes()
    assert L.vad_create_host_only(good, len(good), ctypes.byref(h)) == 0
    # a host-only engine refuses device work loudly
    assert L.vad_reserve(h, 16000, 4, 4) == 4                                        # VAD_ERR_NO_DEVICE
    assert b"host-only" in L.vad_last_error(h)
    assert L.vad_set_option(h, b"impl", b"bogus") == 8
    assert L.vad_set_option(h, b"impl", b"reference") == 0
    # every option states its values; an unknown value or name is VAD_ERR_OPTION with a message, never a silent default
    for name, good_vals, bad in ((b"front_mma", (b"bf16x9", b"fp32"), b"bf16"), (b"rec", (b"bf16x9", b"fp32"), b"fp16"),
                                 (b"front", (b"latency", b"throughput", b"auto"), b"fast"), (b"rec_form", (b"mfma", b"auto"), b"scalar"), (b"precision", (b"fp32",), b"bf16x9"),
                                 (b"enc0", (b"winograd",), b"direct"), (b"gx_cap_mib", (b"64",), b"0")):
        for v in good_vals:
            assert L.vad_set_option(h, name, v) == 0, (name, v)
        assert L.vad_set_option(h, name, bad) == 8, (name, bad)
        assert name in L.vad_last_error(h)
    assert L.vad_set_option(h, b"no_such_option", b"1") == 8 and b"unknown option" in L.vad_last_error(h)
    L.vad_destroy(h)


def test_pump_and_split_step_refuse_without_a_device(built):
    """The round-5 entry points on a machine without a GPU: null / host-only handles are refused with a status, never a crash, and the
    pump's value-returning accessors return NULL / VAD_PUMP_ERROR."""
    from silero_vad_amd import _lib
    L = _lib.lib()
    good = _lib.WEIGHTS_PATH.read_bytes()
    h = ctypes.c_void_p()
    assert L.vad_create_host_only(good, len(good), ctypes.byref(h)) == 0
    prm = _lib.PumpParams()
    L.vad_pump_params_default(ctypes.byref(prm), 16000, 64)
    assert (prm.sampling_rate, prm.streams, prm.threshold, prm.min_silence_duration_ms, prm.speech_pad_ms) == (16000, 64, 0.5, 100, 30)
    p = ctypes.c_void_p()
    assert L.vad_pump_create(None, ctypes.byref(prm), ctypes.byref(p)) == 1 and not p.value            # VAD_ERR_ARG
    assert L.vad_pump_create(h, ctypes.byref(prm), ctypes.byref(p)) == 4 and not p.value               # VAD_ERR_NO_DEVICE (host-only engine)
    prm.sampling_rate = 44100
    assert L.vad_pump_create(h, ctypes.byref(prm), ctypes.byref(p)) == 2                               # VAD_ERR_SAMPLE_RATE
    prm.sampling_rate, prm.streams = 16000, 0
    assert L.vad_pump_create(h, ctypes.byref(prm), ctypes.byref(p)) == 1
    assert L.vad_pump_slot(None, 0) is None and L.vad_pump_probs(None, 0) is None
    assert L.vad_pump_submit(None, 0) == 1 and L.vad_pump_open(None, 0) == 1 and L.vad_pump_close(None, 0) == 1
    assert L.vad_pump_poll(None, 1, None, 0, None) == -3                                                # VAD_PUMP_ERROR
    assert L.vad_pump_play(None, None, 0, 0, 0, 0, 1, 0, None, 0, None) == -3
    assert L.vad_pump_last_error(None) == b"null pump"
    L.vad_pump_destroy(None)
    assert L.vad_step_split(h, 16000, 4, None, 2, 512, None, None, None, None, None) == 4               # host-only engine
    assert L.vad_step_split(None, 16000, 4, None, 2, 512, None, None, None, None, None) == 1
    assert L.vad_step_present(h, 16000, 4, None, 2, 512, None, None, None, None, None, None) == 4
    assert L.vad_step_present(None, 16000, 4, None, 2, 512, None, None, None, None, None, None) == 1
    assert L.vad_step_host_present(h, 16000, 4, None, 2, None, None, None, None, None, None, None, None) == 4
    assert L.vad_step_host_sync(h, 16000, 4, None, 2, None, None, None, None) == 4 and L.vad_step_host_sync(None, 16000, 4, None, 2, None, None, None, None) == 1
    assert L.vad_pump_present(None, 0) is None and L.vad_pump_submit_present(None, 0, None) == 1
    assert L.vad_pump_play_gaps(None, None, 0, 0, None, 0, 0, 0, 1, 0, None, 0, None) == -3
    assert L.vad_pump_submit_rows(None, 0, None, 0) == 1
    assert L.vad_pump_submit_compact(None, 0, None) == 1 and L.vad_pump_play_compact(None, None, 0, 0, None, 0, 0, 0, 1, 0, None, 0, None) == -3
    assert L.vad_streams_overlap(None, None, None) == -1 and L.vad_streams_overlap(h, None, None) == -4
    L.vad_destroy(h)


def test_no_gpu_means_loud_failure(built):
    """The product path has no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from silero_vad_amd import _lib, load_silero_vad
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        load_silero_vad()
    good = _lib.WEIGHTS_PATH.read_bytes()
    h = ctypes.c_void_p()
    assert _lib.lib().vad_create(good, len(good), 0, ctypes.byref(h)) == 4           # VAD_ERR_NO_DEVICE


def test_packed_image_sizes(built):
    from silero_vad_amd import _lib
    L = _lib.lib()
    good = _lib.WEIGHTS_PATH.read_bytes()
    h = ctypes.c_void_p()
    assert L.vad_create_host_only(good, len(good), ctypes.byref(h)) == 0
    # 16 kHz frontend stream: E0 3x(9 kg x 8 mb) + E1 3x(8x4) + E2 2x(4x4) + E3 (4x8) + IH 4x(8x8) KiB
    assert L.vad_debug_packed_floats(h, 16000, 0) == (3 * 72 + 3 * 32 + 2 * 16 + 32 + 4 * 64) * 256
    assert L.vad_debug_packed_floats(h, 8000, 0) == (3 * 40 + 3 * 32 + 2 * 16 + 32 + 4 * 64) * 256
    assert L.vad_debug_packed_floats(h, 16000, 5) == (4 * 4 + 26) * 16 * 256        # Winograd image: whole 16 KiB units
    assert L.vad_debug_packed_floats(h, 8000, 5) == (4 * 2 + 26) * 16 * 256
    assert L.vad_debug_packed_floats(h, 16000, 6) == 54 * 16 * 256                  # F(4,3) image = the program, 54 | 42 units
    assert L.vad_debug_packed_floats(h, 8000, 6) == 42 * 16 * 256
    assert L.vad_debug_packed_floats(h, 16000, 1) == 128 * 512
    n = L.vad_debug_packed_floats(h, 16000, 1)
    whh = np.empty(n, np.float32)
    assert L.vad_debug_packed_copy(h, 16000, 1, whh.ctypes.data_as(_lib.f32p), n) == 0
    # the packed recurrent image is a permutation of W_hh
    from oracle.weights import read_container
    w = read_container(good)["_model.decoder.rnn.weight_hh"]
    assert np.array_equal(np.sort(whh), np.sort(w.ravel()))
    L.vad_destroy(h)


def build_c_client(tmp_path):
    """tests/c_client/client.c -> an executable linked against the in-tree library: strict C99 against the header alone."""
    import subprocess
    from silero_vad_amd import _lib
    exe = tmp_path / "client"
    obj = tmp_path / "client.o"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", str(ROOT / "include"), "-c",
                    str(ROOT / "tests" / "c_client" / "client.c"), "-o", str(obj)], check=True)
    libdir = _lib.LIB_PATH.parent
    subprocess.run(["gcc", str(obj), "-o", str(exe), "-L", str(libdir), f"-l:{_lib.LIB_PATH.name}", "-L", "/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_header_is_valid_c_and_a_plain_c_client_links(built, tmp_path):
    """The boundary is a C ABI: the header must compile as strict C99 (no C++, no HIP headers) and a client that uses the streaming
    entry points (vad_create, vad_reserve, vad_host_register, vad_step_host, vad_iterator_feed, ...) must link against the library.
    Without a GPU the client fails loudly at vad_create (there is no CPU fallback behind the ABI either)."""
    import subprocess
    import torch
    from silero_vad_amd import _lib
    exe = build_c_client(tmp_path)
    assert subprocess.run([str(exe)], capture_output=True).returncode == 64          # usage
    if not torch.cuda.is_available():
        pcm = tmp_path / "pcm.raw"
        np.zeros(2048, np.int16).tofile(pcm)
        r = subprocess.run([str(exe), str(_lib.WEIGHTS_PATH), str(pcm), "16000", "2"], capture_output=True, text=True)
        assert r.returncode == 1 and "vad_create" in r.stderr, (r.returncode, r.stderr)
