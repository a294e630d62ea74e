"""The CPU oracle (oracle/vad_oracle.c) against vectors produced by the real reference
(tests/golden/make_golden.py) and against the reference's published known answers."""
import numpy as np
import pytest

from conftest import SRS, check_nonfinite, kat_segments, state_err, synthetic_audio

# fp32 restatement vs ATen fp32 kernels: summation order differs, nothing else
TOL_PROB = 2e-5
TOL_STATE = 1e-4


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_wav_protocol(oracle, golden, tag):
    g, sr = golden[tag], SRS[tag]
    probs, ctx, state = oracle.forward_audio(g["wav"][None], sr)
    assert probs.shape[1] == len(g["probs_wav"])
    assert np.abs(probs[0] - g["probs_wav"]).max() < TOL_PROB
    assert state_err(state, g["state_wav"]) < TOL_STATE
    assert np.array_equal(ctx, g["ctx_wav"])
    # published known-answer segment counts (examples/openvino/README.md:62)
    assert len(kat_segments(probs[0])) == {"16k": 29, "8k": 79}[tag]
    assert len(kat_segments(probs[0])) == golden["segments"][tag]["kat_segments_thr05_min8"]


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_synth_protocol_stateful_calls(oracle, golden, tag):
    g, sr = golden[tag], SRS[tag]
    syn = synthetic_audio(sr, np.random.default_rng(42))
    assert abs(float(np.abs(syn).sum()) - g["synth_checksum"][0]) < 1e-3 * g["synth_checksum"][0]
    n = 512 if sr == 16000 else 256
    oracle.reset_states()
    probs = [oracle(syn[s:s + n], sr)[0, 0] for s in range(0, (len(syn) // n) * n, n)]
    assert np.abs(np.asarray(probs) - g["probs_synth"]).max() < TOL_PROB
    assert state_err(oracle._state, g["state_synth"]) < TOL_STATE


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_noise_protocol_explicit_state(oracle, golden, tag):
    g, sr = golden[tag], SRS[tag]
    rng = np.random.default_rng(17 + sr)
    noise = (rng.standard_normal(round(8.0 * sr)) * 0.03).astype(np.float32)
    n = 512 if sr == 16000 else 256
    L = (len(noise) // n) * n
    probs, _, state = oracle.forward_audio(noise[None, :L], sr, state=g["state_noise_init"])
    assert np.abs(probs[0] - g["probs_noise"]).max() < TOL_PROB
    assert state_err(state, g["state_noise_final"]) < TOL_STATE


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_batch_ragged(oracle, golden, tag):
    g, sr = golden[tag], SRS[tag]
    B, T, L, stride = (int(v) for v in g["batch_meta"])
    rows = np.stack([np.roll(g["wav"], -b * stride)[:L] for b in range(B)])
    probs = oracle.audio_forward(rows, sr)
    assert probs.shape == (B, T)
    assert np.abs(probs - g["probs_batch"]).max() < TOL_PROB
    assert state_err(oracle._state, g["state_batch"]) < TOL_STATE


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_stages(oracle, golden, tag):
    g, sr = golden[tag], SRS[tag]
    prob, state, st = oracle.step(g["stage_x"], g["stage_state_in"], sr, stages=True)
    assert np.abs(st["mag"] - g["stage_mag"]).max() < 3e-5 * max(1.0, np.abs(g["stage_mag"]).max())
    for i in range(4):
        ref = g[f"stage_enc{i}"]
        assert np.abs(st[f"enc{i}"] - ref).max() < 3e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(prob - g["stage_prob"][:, 0]).max() < TOL_PROB
    assert state_err(state, g["stage_state_out"]) < TOL_STATE


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_nonfinite_protocol(oracle, golden, tag):
    g, sr = golden[tag], SRS[tag]
    n = 512 if sr == 16000 else 256
    rows, T = g["nf_rows"], g["nf_probs"].shape[1]
    oracle.reset_states()
    probs = np.stack([oracle(rows[:, t * n:(t + 1) * n], sr)[:, 0] for t in range(T)], 1)
    state = oracle._state.copy()
    oracle.reset_states()
    after = np.stack([oracle(rows[:, t * n:(t + 1) * n], sr)[:, 0] for t in range(T, rows.shape[1] // n)], 1)
    check_nonfinite(g, probs, state, after, TOL_PROB, TOL_STATE)
    # the time-loop entry agrees with the per-chunk one
    p2, _, s2 = oracle.forward_audio(rows[:, :T * n], sr)
    assert np.array_equal(np.isnan(p2), np.isnan(probs)) and np.array_equal(np.isnan(s2), np.isnan(state))


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_any_single_overflowing_bin_is_nan(oracle, golden, tag):
    """The HIP path's rule is 'any non-finite STFT magnitude => the chunk is NaN' (csrc/front_common.hpp poison()); in the
    reference an Inf magnitude becomes NaN through Inf - Inf in the encoder sums, which depends on the weights' signs.
    With these weights it always does: every single (bin, frame) set to +Inf alone gives a NaN probability."""
    sr = SRS[tag]
    from oracle.oracle import load_weights_blob
    from oracle.weights import read_container
    g = golden[tag]
    w = read_container(load_weights_blob())
    pre = "_model" if sr == 16000 else "_model_8k"
    w0, b0 = w[f"{pre}.encoder.0.reparam_conv.weight"], w[f"{pre}.encoder.0.reparam_conv.bias"]     # [128, K, 3]
    w1, b1 = w[f"{pre}.encoder.1.reparam_conv.weight"], w[f"{pre}.encoder.1.reparam_conv.bias"]     # [64, 128, 3]
    K = w0.shape[1]
    mag0 = g["stage_mag"][0]                                            # [K, 4], a real frame set
    relu = lambda v: np.where(v <= 0, np.float32(0), v)                 # NaN-propagating, like torch.relu
    with np.errstate(invalid="ignore"):
        for m in range(4):
            for k in range(K):
                mag = np.pad(mag0, ((0, 0), (1, 1)))
                mag[k, m + 1] = np.inf
                e0 = relu(b0[:, None] + sum(w0[:, :, t] @ mag[:, t:t + 4] for t in range(3)))           # [128, 4]
                e0 = np.pad(e0, ((0, 0), (1, 1)))
                e1 = relu(b1[:, None] + sum(w1[:, :, t] @ e0[:, t:t + 4:2] for t in range(3)))          # [64, 2]
                # encoder 2 (T 2 -> 1) sums both positions of every channel: one NaN there is NaN everywhere after it
                assert np.isnan(e1).any(), (m, k)
    # and end to end through the oracle: a sample of 1e20 anywhere in the chunk
    n = 512 if sr == 16000 else 256
    base = g["nf_rows"][0, : n].copy()
    xs = np.tile(base, (n, 1))
    xs[np.arange(n), np.arange(n)] = 1e20
    oracle.reset_states()
    assert np.isnan(oracle(xs, sr)).all()


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_gain_protocol(oracle, golden, tag):
    """Quiet speech (gain 0.1 / 0.01): small magnitudes, probabilities in the sigmoid's steep part."""
    g, sr = golden[tag], SRS[tag]
    for gt, gain in (("g01", 0.1), ("g001", 0.01)):
        q = (g["wav"] * np.float32(gain)).astype(np.float32)
        probs, _, state = oracle.forward_audio(q[None], sr)
        assert np.abs(probs[0] - g[f"probs_{gt}"]).max() < 5e-5, gt
        assert state_err(state, g[f"state_{gt}"]) < TOL_STATE, gt


def test_decim_and_srswitch(oracle, golden):
    """test.wav[::2] through the 8 kHz net (examples/onnx_sequence/README.md:61) and calls that alternate between the nets
    (auto reset, JIT!/vad/model/vad_annotator.py:37-57)."""
    mz = golden["misc"]
    dec = np.ascontiguousarray(golden["16k"]["wav"][::2])
    probs, ctx, state = oracle.forward_audio(dec[None], 8000)
    assert np.abs(probs[0] - mz["probs_decim"]).max() < TOL_PROB
    assert state_err(state, mz["state_decim"]) < TOL_STATE and np.array_equal(ctx, mz["ctx_decim"])
    assert len(kat_segments(probs[0])) == golden["ext"]["decim_16k_to_8k"]["kat_segments_thr05_min8"]
    oracle.reset_states()
    cur = {16000: 40 * 512, 8000: 40 * 256}
    for i, sr in enumerate(int(v) for v in mz["srswitch_plan"]):
        n = 512 if sr == 16000 else 256
        wav = golden["16k" if sr == 16000 else "8k"]["wav"]
        x = np.stack([np.roll(wav, -b * 7919)[cur[sr]: cur[sr] + n] for b in range(4)])
        cur[sr] += n
        assert np.abs(oracle(x, sr)[:, 0] - mz["srswitch_probs"][i]).max() < TOL_PROB, i
    assert state_err(oracle._state, mz["srswitch_state"]) < TOL_STATE


def test_edge_cases(oracle):
    # empty batch / empty audio
    p, c, s = oracle.forward_audio(np.zeros((0, 1024), np.float32), 16000)
    assert p.shape == (0, 2)
    p, c, s = oracle.forward_audio(np.zeros((2, 0), np.float32), 8000)
    assert p.shape == (2, 0) and np.all(s == 0)
    with pytest.raises(ValueError):
        oracle.forward_audio(np.zeros((1, 512), np.float32), 44100)
    # a single sample is zero-padded to one chunk
    p, _, _ = oracle.forward_audio(np.full((1, 1), 0.5, np.float32), 16000)
    assert p.shape == (1, 1) and 0 < p[0, 0] < 1


# ---- the ATen port (oracle/aten_port.py): the reference's operators issued from the weights container -----
@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_aten_port_is_the_reference(golden, tag):
    """Same ATen kernels in the same order as the TorchScript archive => the goldens recorded from the
    reference are reproduced to the last ulp-level digit (this is what bench.py times as cpu_baseline)."""
    import torch
    from oracle.aten_port import AtenVAD
    g, sr = golden[tag], SRS[tag]
    keep = torch.get_num_threads()
    torch.set_num_threads(1)                                   # the reference's setting (src/silero_vad/model.py:3)
    try:
        m = AtenVAD()
        probs = m.audio_forward(torch.from_numpy(g["wav"]), sr)[0].numpy()
        assert np.abs(probs - g["probs_wav"]).max() <= 1e-6
        assert state_err(m._state.numpy(), g["state_wav"]) <= 1e-6
        assert np.array_equal(m._context.numpy(), g["ctx_wav"])
        B, T, L, stride = (int(v) for v in g["batch_meta"])
        rows = np.stack([np.roll(g["wav"], -b * stride)[:L] for b in range(B)])
        pb = m.audio_forward(torch.from_numpy(rows), sr).numpy()
        assert pb.shape == (B, T) and np.abs(pb - g["probs_batch"]).max() <= 1e-6
        assert state_err(m._state.numpy(), g["state_batch"]) <= 1e-5
        # per-chunk stateful protocol, and agreement with the plain-C oracle's tolerance band
        n = 512 if sr == 16000 else 256
        m.reset_states()
        p = [m(torch.from_numpy(g["wav"][s:s + n]), sr).item() for s in range(0, 40 * n, n)]
        assert np.abs(np.asarray(p, np.float32) - g["probs_wav"][:40]).max() <= 1e-6
    finally:
        torch.set_num_threads(keep)
