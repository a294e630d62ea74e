"""Lane-accurate emulation (tests/emu_wave.py) of the MFMA wave programs, run on the packed weight
images the C++ packer really produces, against activations recorded from the reference model.
Pins -- without a GPU -- the fragment packing, the K-permutation ("chain layout") trick, the
4-lane cooperative FFT algebra and the persistent-RNN data flow."""
import ctypes

import numpy as np
import pytest

import emu_wave as E
from conftest import SRS, state_err


@pytest.fixture(scope="module")
def packed(built):
    from silero_vad_amd import _lib
    L = _lib.lib()
    blob = _lib.WEIGHTS_PATH.read_bytes()
    h = ctypes.c_void_p()
    assert L.vad_create_host_only(blob, len(blob), ctypes.byref(h)) == 0
    out = {}
    for sr in (16000, 8000):
        for which in (0, 1, 2, 5, 6):
            n = L.vad_debug_packed_floats(h, sr, which)
            a = np.empty(n, np.float32)
            assert L.vad_debug_packed_copy(h, sr, which, a.ctypes.data_as(_lib.f32p), n) == 0
            out[sr, which] = a
    L.vad_destroy(h)
    return out


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_wave_program_matches_reference_activations(packed, golden, tag):
    sr, g = SRS[tag], golden[tag]
    emu = E.FrontEmu(sr, packed[sr, 0], packed[sr, 2])
    out = emu.run(g["stage_x"])
    Q = emu.Q
    mag = np.stack([E.mag_from_layout(out["X"][v], Q) for v in range(4)], -1)       # [16][K][4]
    ref = g["stage_mag"]
    assert np.abs(mag - ref).max() < 2e-6 * max(1.0, np.abs(ref).max()) + 2e-6
    feat = E.chain_to_dense(out["feat"])
    ref = g["stage_enc3"][:, :, 0]
    assert np.abs(feat - ref).max() < 3e-5
    prob, hn, cn = E.rec_step(packed[sr, 1], packed[sr, 2], emu.tb, out["gx"],
                              g["stage_state_in"][0], g["stage_state_in"][1])
    assert np.abs(prob - g["stage_prob"][:, 0]).max() < 1e-5
    assert state_err(np.stack([hn, cn]), g["stage_state_out"]) < 2e-5


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_winograd_wave_program_matches_reference_activations(packed, golden, tag):
    """front_wino_kernel's program (encoder 0 as two Winograd F(2,3) transforms over the frame pairs, row parts, K-half
    encoder 1, whole-unit weight stream walked by make_wsched) on the Winograd image the C++ packer produces: same
    encoder output and gate pre-activations as the reference, to fp32 round-off -- and as the tap-by-tap program."""
    sr, g = SRS[tag], golden[tag]
    emu = E.FrontWinoEmu(sr, packed[sr, 5], packed[sr, 2])
    out = emu.run(g["stage_x"])
    feat = E.chain_to_dense(out["feat"])
    ref = g["stage_enc3"][:, :, 0]
    assert np.abs(feat - ref).max() < 3e-5
    direct = E.FrontEmu(sr, packed[sr, 0], packed[sr, 2]).run(g["stage_x"])
    assert np.abs(out["gx"] - direct["gx"]).max() < 1e-4 * max(1.0, np.abs(direct["gx"]).max())
    prob, hn, cn = E.rec_step(packed[sr, 1], packed[sr, 2], emu.tb, out["gx"],
                              g["stage_state_in"][0], g["stage_state_in"][1])
    assert np.abs(prob - g["stage_prob"][:, 0]).max() < 1e-5
    assert state_err(np.stack([hn, cn]), g["stage_state_out"]) < 2e-5


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_f43_wave_program_matches_reference_activations(packed, golden, tag):
    """front_f43_kernel's program (encoder 0 as ONE Winograd F(4,3) tile over the 4 frames, row parts, encoder 1 fed part
    by part -- two taps per unit at 16 kHz, the tap-2 units spanning two parts --, units consumed in image order) on the
    F(4,3) image the C++ packer produces: same encoder output and gate pre-activations as the reference, to fp32
    round-off -- and as the tap-by-tap program."""
    sr, g = SRS[tag], golden[tag]
    emu = E.FrontF43Emu(sr, packed[sr, 6], packed[sr, 2])
    out = emu.run(g["stage_x"])
    feat = E.chain_to_dense(out["feat"])
    ref = g["stage_enc3"][:, :, 0]
    assert np.abs(feat - ref).max() < 3e-5
    direct = E.FrontEmu(sr, packed[sr, 0], packed[sr, 2]).run(g["stage_x"])
    assert np.abs(out["gx"] - direct["gx"]).max() < 1e-4 * max(1.0, np.abs(direct["gx"]).max())
    prob, hn, cn = E.rec_step(packed[sr, 1], packed[sr, 2], emu.tb, out["gx"],
                              g["stage_state_in"][0], g["stage_state_in"][1])
    assert np.abs(prob - g["stage_prob"][:, 0]).max() < 1e-5
    assert state_err(np.stack([hn, cn]), g["stage_state_out"]) < 2e-5


def test_mfma_emulation_is_transpose_detecting():
    """Asymmetric operands: a swapped A/B or row/col map in the emulator would not reproduce A @ B."""
    rng = np.random.default_rng(0)
    A = rng.standard_normal((16, 4)).astype(np.float32)
    B = rng.standard_normal((4, 16)).astype(np.float32)
    a = np.zeros(64, np.float32)
    b = np.zeros(64, np.float32)
    for lane in range(64):
        a[lane] = A[lane & 15, lane >> 4]
        b[lane] = B[lane >> 4, lane & 15]
    acc = E.mfma_16x16x4(a, b, np.zeros((4, 64), np.float32))
    D = A @ B
    for lane in range(64):
        for r in range(4):
            assert abs(acc[r, lane] - D[4 * (lane >> 4) + r, lane & 15]) < 1e-5


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_bf16x9_recurrent_image_is_exact(built, tag):
    """The three-piece bf16 image of W_hh (option rec=bf16x9, csrc/kernel_rec_b9.hip): the pieces of every weight add up to
    the fp32 weight EXACTLY, each piece is a bf16 (low 16 bits of its fp32 pattern are zero), and slot (g, e) of K32 step u
    of wave w, gate q, row i holds W_hh[128 q + 16 w + i][32 u + 8 g + e]."""
    from oracle.weights import read_container
    from silero_vad_amd import _lib
    sr = SRS[tag]
    L = _lib.lib()
    blob = _lib.WEIGHTS_PATH.read_bytes()
    h = ctypes.c_void_p()
    assert L.vad_create_host_only(blob, len(blob), ctypes.byref(h)) == 0
    n = L.vad_debug_packed_floats(h, sr, 7)
    raw = np.empty(n, np.float32)
    assert L.vad_debug_packed_copy(h, sr, 7, raw.ctypes.data_as(_lib.f32p), n) == 0
    L.vad_destroy(h)
    img = raw.view(np.uint16).reshape(8, 3, 4, 4, 64, 8)                   # [wave][piece][gate][u][lane][e]
    pieces = (img.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    W = read_container(blob)[("_model" if sr == 16000 else "_model_8k") + ".decoder.rnn.weight_hh"].astype(np.float64)
    w, q, u, lane, e = np.meshgrid(np.arange(8), np.arange(4), np.arange(4), np.arange(64), np.arange(8), indexing="ij")
    want = W[128 * q + 16 * w + (lane & 15), 32 * u + 8 * (lane >> 4) + e]
    assert np.array_equal(pieces.sum(1), want)                              # exact: float64 holds the three pieces' sum
    assert np.all(np.abs(pieces[:, 1]) <= np.abs(pieces[:, 0]) * 2.0 ** -8 + 1e-45)
    assert np.all(np.abs(pieces[:, 2]) <= np.abs(pieces[:, 0]) * 2.0 ** -16 + 1e-45)


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_bf16x9_frontend_image_is_the_f43_program_exactly(packed, built, tag):
    """The three-piece bf16 image of the frontend (option front_mma=bf16x9, csrc/kernel_front_b9.hip) is the F(4,3) program
    unit for unit: the pieces of every weight add up to the fp32 weight EXACTLY, and slot (g, e) of K32 step kp of a unit
    holds what the fp32 unit holds for k-step 8 kp + e at k = g (same rows) -- so the wave program the emulator validates
    on the fp32 image (test_f43_wave_program_matches_reference_activations) is the program of the bf16 x 9 kernel too."""
    from silero_vad_amd import _lib
    sr = SRS[tag]
    Q = 32 if sr == 16000 else 16
    L = _lib.lib()
    blob = _lib.WEIGHTS_PATH.read_bytes()
    h = ctypes.c_void_p()
    assert L.vad_create_host_only(blob, len(blob), ctypes.byref(h)) == 0
    n = L.vad_debug_packed_floats(h, sr, 8)
    raw = np.empty(n, np.float32)
    assert L.vad_debug_packed_copy(h, sr, 8, raw.ctypes.data_as(_lib.f32p), n) == 0
    L.vad_destroy(h)
    f32 = packed[sr, 6].reshape(-1, 8, 2, 64, 4)                            # [unit][fp32 step][rb][lane][ks]
    NU = f32.shape[0]
    img = raw.view(np.uint16).reshape(NU, 4, 3, 2, 64, 8)                   # [unit][step][piece][rb][lane][e]
    pieces = (img.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    RB, P = 64 // Q, 8 // (64 // Q)
    per_part = [6 + ((2 + (p & 1)) if Q == 32 else 5) for p in range(P)]
    unit_m = []
    for p in range(P):
        unit_m += [RB] * 6 + [4] * (per_part[p] - 6)
    unit_m += [4, 4] + [8] * 18
    assert len(unit_m) == NU == (54 if Q == 32 else 42)
    for u in range(NU):
        H = unit_m[u] // 2
        for ib in range(4):
            kp, mh = ib // H, ib % H
            for e in range(8):
                i_f = (2 * kp + e // 4) * H + mh
                want = f32[u, i_f, :, :, e % 4].astype(np.float64)           # [rb][lane]
                got = pieces[u, ib, :, :, :, e]                             # [piece][rb][lane]
                assert np.array_equal(got.sum(0), want)
                assert np.all(np.abs(got[1]) <= np.abs(got[0]) * 2.0 ** -8 + 1e-45)
                assert np.all(np.abs(got[2]) <= np.abs(got[0]) * 2.0 ** -16 + 1e-45)
