"""Parity tests proper: the HIP path, called through the C ABI (silero_vad_amd/_lib.py), against
  (1) golden vectors recorded from the reference's TorchScript model (tests/golden/),
  (2) the CPU oracle on the same seeded inputs,
  (3) size-independent properties at BASELINE.json's full batch (4096 streams).
Tolerance: |dp| <= 1e-4 on speech probabilities (BASELINE.json north_star; the reference's own
cross-runtime check uses the same bound, examples/openvino/verify.py:167) and identical segments.
Everything here needs a real MI355X:  python -m pytest tests -m gpu
"""
import ctypes
import warnings

import numpy as np
import pytest
import torch

from conftest import SRS, kat_segments, state_err, synthetic_audio

pytestmark = pytest.mark.gpu

TOL = 1e-4          # the contract
TIGHT = 2e-5        # what fp32 in a different summation order actually delivers; regression guard


@pytest.fixture(scope="module", params=["f16x3", "fp32"])
def model(built, request):
    """Every parity test runs against both arithmetic implementations of the engine: the default
    fp16x3 split MFMA kernels and the exact-fp32 MFMA kernels (option "precision")."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback to silently pass on)")
    from silero_vad_amd import load_silero_vad
    m = load_silero_vad(device=0, precision=request.param)
    assert m.engine._h, "native engine not created"
    assert m.engine.precision == request.param
    return m


def chunk_of(sr):
    return 512 if sr == 16000 else 256


def rolled_rows(wav, B, L, stride=7919):
    return np.stack([np.roll(wav, -b * stride)[:L] for b in range(B)])


def run_engine(model, rows, sr, state=None, ctx=None):
    eng = model.engine
    dev = model.device
    x = torch.as_tensor(rows).to(dev)
    B = x.shape[0]
    n = chunk_of(sr)
    ctx_t = torch.zeros((B, n // 8), device=dev) if ctx is None else torch.as_tensor(ctx).to(dev).clone()
    st_t = torch.zeros((2, B, 128), device=dev) if state is None else torch.as_tensor(state).to(dev).clone()
    probs = eng.forward_audio(x.contiguous(), sr, ctx_t, st_t)
    torch.cuda.synchronize()
    return probs.cpu().numpy(), ctx_t.cpu().numpy(), st_t.cpu().numpy()


# ---- (0) the slow on-device reference implementation, then the MFMA frontend in isolation ---------
@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_impl_reference_matches_oracle(model, oracle, golden, tag):
    sr, g = SRS[tag], golden[tag]
    B, T, L, stride = (int(v) for v in g["batch_meta"])
    rows = rolled_rows(g["wav"], B, L, stride)
    model.engine.set_option("impl", "reference")
    try:
        probs, ctx, st = run_engine(model, rows, sr)
    finally:
        model.engine.set_option("impl", "mfma")
    want, wctx, wst = oracle.forward_audio(rows, sr)
    assert np.abs(probs - want).max() < TIGHT
    assert np.abs(probs - g["probs_batch"]).max() < TIGHT
    assert state_err(st, wst) < TOL and np.array_equal(ctx, wctx)


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_frontend_gate_preactivations(model, oracle, golden, tag):
    """STFT + encoder + W_ih GEMM (kernel_front.hip) against the oracle's encoder output pushed
    through W_ih in float64."""
    from oracle.weights import read_container
    from silero_vad_amd import _lib
    sr, g = SRS[tag], golden[tag]
    n = chunk_of(sr)
    B, T = 19, 5
    rows = rolled_rows(g["wav"], B, T * n, 5003)
    x = torch.from_numpy(rows).to(model.device)
    ctx = torch.zeros((B, n // 8), device=model.device)
    gx = model.engine.debug_frontend(x, sr, ctx).cpu().numpy()            # [B][T][512]
    w = read_container(_lib.WEIGHTS_PATH.read_bytes())
    pre = "_model" if sr == 16000 else "_model_8k"
    w_ih = w[pre + ".decoder.rnn.weight_ih"].astype(np.float64)
    bias = (w[pre + ".decoder.rnn.bias_ih"] + w[pre + ".decoder.rnn.bias_hh"]).astype(np.float64)
    C = n // 8
    for t in range(T):
        prev = rows[:, t * n - C: t * n] if t else np.zeros((B, C), np.float32)
        x1 = np.concatenate([prev, rows[:, t * n:(t + 1) * n]], 1)
        _, _, st = oracle.step(x1, np.zeros((2, B, 128), np.float32), sr, stages=True)
        want = st["enc3"][:, :, 0].astype(np.float64) @ w_ih.T + bias
        err = np.abs(gx[:, t] - want).max()
        assert err < 1e-4 * max(1.0, np.abs(want).max()), (t, err)


# ---- (1) golden vectors from the reference ----------------------------------------------------------
@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_wav_protocol_and_segments(model, golden, tag):
    from silero_vad_amd import get_speech_timestamps
    sr, g = SRS[tag], golden[tag]
    wav = torch.from_numpy(g["wav"])
    probs = model.audio_forward(wav, sr).numpy()[0]
    err = np.abs(probs - g["probs_wav"]).max()
    assert err < TOL, err
    assert err < TIGHT, f"within contract but looser than expected: {err}"
    assert state_err(model._state.cpu().numpy(), g["state_wav"]) < TOL
    assert np.array_equal(model._context.cpu().numpy(), g["ctx_wav"])
    # decision margins: how far the closest probability is from the thresholds
    margin = min(np.abs(g["probs_wav"] - 0.5).min(), np.abs(g["probs_wav"] - 0.35).min())
    assert margin > 10 * err
    assert len(kat_segments(probs)) == {"16k": 29, "8k": 79}[tag]        # published known answers
    info = golden["segments"][tag]["timestamps"]
    for name, rec in info.items():
        kw = dict(rec["kwargs"])
        kw.setdefault("sampling_rate", sr)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = get_speech_timestamps(wav, model, **kw)
        assert got == rec["out"], f"{tag}/{name}: segments differ from the reference"


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_synth_protocol_per_chunk_calls(model, golden, tag):
    """examples/openvino/verify.py protocol through the stateful model(chunk, sr) boundary."""
    sr, g = SRS[tag], golden[tag]
    n = chunk_of(sr)
    syn = torch.from_numpy(synthetic_audio(sr, np.random.default_rng(42)))
    model.reset_states()
    probs = [model(syn[s:s + n], sr).item() for s in range(0, (len(syn) // n) * n, n)]
    assert np.abs(np.asarray(probs, np.float32) - g["probs_synth"]).max() < TIGHT
    assert state_err(model._state.cpu().numpy(), g["state_synth"]) < TOL


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_noise_protocol_explicit_state(model, golden, tag):
    """examples/onnx_sequence/run.py:159-216 protocol: random initial state, probs AND final state."""
    sr, g = SRS[tag], golden[tag]
    n = chunk_of(sr)
    noise = (np.random.default_rng(17 + sr).standard_normal(round(8.0 * sr)) * 0.03).astype(np.float32)
    L = (len(noise) // n) * n
    probs, _, st = run_engine(model, noise[None, :L], sr, state=g["state_noise_init"])
    assert np.abs(probs[0] - g["probs_noise"]).max() < TIGHT
    assert state_err(st, g["state_noise_final"]) < TOL


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_batch_ragged_audio_forward(model, golden, tag):
    sr, g = SRS[tag], golden[tag]
    B, T, L, stride = (int(v) for v in g["batch_meta"])
    rows = rolled_rows(g["wav"], B, L, stride)
    probs = model.audio_forward(torch.from_numpy(rows), sr).numpy()
    assert probs.shape == (B, T)
    assert np.abs(probs - g["probs_batch"]).max() < TIGHT
    assert state_err(model._state.cpu().numpy(), g["state_batch"]) < TOL


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_vad_iterator_events(model, golden, tag):
    from silero_vad_amd import VADIterator
    sr, g = SRS[tag], golden[tag]
    n = chunk_of(sr)
    wav = torch.from_numpy(g["wav"][: 600 * n])
    rec = golden["segments"][tag]["iterator"]["default"]
    it = VADIterator(model, sampling_rate=sr)
    ev = [e for s in range(0, len(wav), n) if (e := it(wav[s:s + n]))]
    limit = 600 * n
    want = [e for e in rec["events"] if list(e.values())[0] <= limit]
    assert ev[:len(want) - 1] == want[:len(want) - 1]
    assert len(ev) >= len(want) - 1 and len(ev) > 4


# ---- (2) oracle on seeded inputs: shapes and entry points the goldens do not cover -------------------
@pytest.mark.parametrize("B,T", [(1, 1), (3, 2), (16, 1), (17, 3), (33, 9), (64, 4)])
@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_batch_shapes_vs_oracle(model, oracle, golden, tag, B, T):
    sr = SRS[tag]
    n = chunk_of(sr)
    rows = rolled_rows(golden[tag]["wav"], B, T * n, 3571)
    rng = np.random.default_rng(B * 100 + T)
    st0 = (rng.standard_normal((2, B, 128)) * 0.3).astype(np.float32)
    cx0 = (rng.standard_normal((B, n // 8)) * 0.05).astype(np.float32)
    probs, ctx, st = run_engine(model, rows, sr, state=st0, ctx=cx0)
    want, wctx, wst = oracle.forward_audio(rows, sr, ctx=cx0, state=st0)
    assert np.abs(probs - want).max() < TIGHT
    assert state_err(st, wst) < TOL
    assert np.array_equal(ctx, wctx)


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_step_chain_equals_forward_audio(model, oracle, golden, tag):
    sr = SRS[tag]
    n = chunk_of(sr)
    B, T = 5, 40
    rows = rolled_rows(golden[tag]["wav"], B, T * n, 9001)
    model.reset_states()
    per_step = np.concatenate([model(torch.from_numpy(rows[:, t * n:(t + 1) * n]), sr).cpu().numpy()
                               for t in range(T)], 1)
    whole = model.audio_forward(torch.from_numpy(rows), sr).numpy()
    want = oracle.audio_forward(rows, sr)
    assert np.abs(per_step - want).max() < TIGHT and np.abs(whole - want).max() < TIGHT
    assert np.abs(per_step - whole).max() < 1e-6        # same kernels, same order of operations


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_carried_state_across_calls(model, golden, tag):
    """ctx/state written by one call feed the next: 8 chunks == 4 + 4 (exercises ctx_out and the
    in-place state update, and the zero-padded tail chunk)."""
    sr = SRS[tag]
    n = chunk_of(sr)
    B = 7
    rows = rolled_rows(golden[tag]["wav"], B, 8 * n - 37, 6007)
    whole, ctx_w, st_w = run_engine(model, rows, sr)
    a, ctx_a, st_a = run_engine(model, rows[:, :4 * n], sr)
    b, ctx_b, st_b = run_engine(model, rows[:, 4 * n:], sr, state=st_a, ctx=ctx_a)
    assert np.array_equal(np.concatenate([a, b], 1), whole)
    assert np.array_equal(st_b, st_w) and np.array_equal(ctx_b, ctx_w)
    tail = np.concatenate([rows[:, -(n // 8 - 37):], np.zeros((B, 37), np.float32)], 1) \
        if n // 8 > 37 else None
    if tail is not None:
        assert np.array_equal(ctx_w, tail)              # context = last C samples of the PADDED chunk


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_int16_ingest(model, golden, tag):
    sr, g = SRS[tag], golden[tag]
    n = chunk_of(sr)
    B, L = 4, 50 * n + 5
    rows_i = np.stack([np.roll(g["pcm_i16"], -b * 4001)[:L] for b in range(B)])
    p_i, c_i, s_i = run_engine(model, rows_i, sr)
    p_f, c_f, s_f = run_engine(model, rows_i.astype(np.float32) / 32768.0, sr)
    assert np.array_equal(p_i, p_f) and np.array_equal(s_i, s_f) and np.array_equal(c_i, c_f)


@pytest.mark.parametrize("k", [2, 3])
def test_sample_rate_front_door(model, golden, k):
    """32 / 48 kHz input through the C ABI: decimated on the device exactly like the reference's
    x[:, ::sr // 16000] (vad_annotator.py:104-112), then the 16 kHz path -- bit-identical to handing over
    the decimated signal, for float and int16 PCM and for single steps."""
    from silero_vad_amd import _lib
    eng, dev = model.engine, model.device
    B, T = 5, 7
    L = k * (T * 512 - 100) + 1                              # ragged: the last chunk is partial after decimation
    for src, fn in ((golden["16k"]["wav"], _lib.lib().vad_forward_audio),
                    (golden["16k"]["pcm_i16"], _lib.lib().vad_forward_audio_i16)):
        x = torch.from_numpy(np.stack([np.roll(src, -b * 3001)[:L] for b in range(B)])).to(dev).contiguous()
        xd = x[:, ::k].contiguous()
        outs = []
        for inp, sr in ((x, 16000 * k), (xd, 16000)):
            ctx = torch.zeros((B, 64), device=dev)
            st = torch.zeros((2, B, 128), device=dev)
            p = torch.empty((B, T), device=dev)
            _lib.check(eng._h, fn(eng._h, sr, B, inp.shape[1], inp.data_ptr(), inp.stride(0), ctx.data_ptr(),
                                   st.data_ptr(), p.data_ptr(), T, None))
            torch.cuda.synchronize()
            outs.append((p.clone(), st.clone(), ctx.clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    # one step: a 512 k-sample chunk
    x = torch.from_numpy(np.stack([np.roll(golden["16k"]["wav"], -b * 77)[:512 * k] for b in range(B)])).to(dev)
    res = []
    for inp, sr in ((x.contiguous(), 16000 * k), (x[:, ::k].contiguous(), 16000)):
        ctx = torch.zeros((B, 64), device=dev)
        st = torch.zeros((2, B, 128), device=dev)
        p = torch.empty((B,), device=dev)
        _lib.check(eng._h, _lib.lib().vad_step(eng._h, sr, B, inp.data_ptr(), inp.stride(0), ctx.data_ptr(),
                                               st.data_ptr(), p.data_ptr(), None))
        torch.cuda.synchronize()
        res.append(p.clone())
    assert torch.equal(res[0], res[1])
    # rates that are neither supported nor a multiple of 16 kHz are still refused
    assert _lib.lib().vad_forward_audio(eng._h, 22050, B, 512, x.data_ptr(), x.stride(0), ctx.data_ptr(),
                                        st.data_ptr(), p.data_ptr(), 1, None) == 2


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_time_slabs_are_transparent(model, golden, tag):
    """An input whose gate pre-activations exceed the scratch cap is processed in time slabs (here: cap 1 MiB,
    33 streams -> 10 steps per slab, 47 chunks -> 5 slabs); results are bit-identical to the un-slabbed call."""
    sr = SRS[tag]
    n = chunk_of(sr)
    B, T = 33, 47
    rows = rolled_rows(golden[tag]["wav"], B, T * n - 5, 4567)
    want = run_engine(model, rows, sr)
    model.engine.set_option("gx_cap_mib", 1)
    try:
        got = run_engine(model, rows, sr)
    finally:
        model.engine.set_option("gx_cap_mib", 6144)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_misaligned_rows_are_handled(model, oracle, golden):
    sr, n = 16000, 512
    B, L = 3, 6 * n + 3                                   # odd row stride -> rows not 16-B aligned
    rows = rolled_rows(golden["16k"]["wav"], B, L, 1237)
    big = torch.zeros((B, L + 1), device=model.device)
    view = big[:, 1:]                                     # offset view: base pointer misaligned too
    view.copy_(torch.from_numpy(rows))
    ctx = torch.zeros((B, 64), device=model.device)
    st = torch.zeros((2, B, 128), device=model.device)
    from silero_vad_amd._lib import check, lib
    T = (L + n - 1) // n
    probs = torch.empty((B, T), device=model.device)
    check(model.engine._h, lib().vad_forward_audio(model.engine._h, sr, B, L, view.data_ptr(), view.stride(0),
                                                   ctx.data_ptr(), st.data_ptr(), probs.data_ptr(), T, None))
    torch.cuda.synchronize()
    want, _, _ = oracle.forward_audio(rows, sr)
    assert np.abs(probs.cpu().numpy() - want).max() < TIGHT


def test_c_abi_error_paths(model):
    from silero_vad_amd._lib import VadError, lib
    eng = model.engine
    x = torch.zeros((2, 512), device=model.device)
    ctx = torch.zeros((2, 64), device=model.device)
    st = torch.zeros((2, 2, 128), device=model.device)
    out = torch.zeros((2, 1), device=model.device)
    with pytest.raises(VadError, match="SAMPLE_RATE"):
        eng.forward_audio(x, 44100, ctx, st)
    with pytest.raises(VadError, match="OPTION"):
        eng.set_option("impl", "nope")
    rc = lib().vad_forward_audio(eng._h, 16000, 2, 512, x.data_ptr(), 512, ctx.data_ptr() + 4, st.data_ptr(),
                                 out.data_ptr(), 1, None)
    assert rc == 1 and b"aligned" in lib().vad_last_error(eng._h)
    # empty work is a no-op
    assert lib().vad_forward_audio(eng._h, 16000, 0, 512, None, 512, None, None, None, 1, None) == 0
    assert lib().vad_forward_audio(eng._h, 16000, 2, 0, x.data_ptr(), 0, ctx.data_ptr(), st.data_ptr(),
                                   out.data_ptr(), 0, None) == 0


# ---- (3) full-size properties (BASELINE.json config[1]/[2]: 4096 streams) -----------------------------
@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_full_batch_properties(model, oracle, golden, tag):
    sr = SRS[tag]
    n = chunk_of(sr)
    B, T = 4096, 24
    wav = torch.from_numpy(golden[tag]["wav"]).to(model.device)
    idx = (torch.arange(B, device=model.device)[:, None] * 7919 + torch.arange(T * n, device=model.device)[None]) % len(wav)
    x = wav[idx].contiguous()                             # stream b = circular read from offset b*7919

    def run(inp):
        ctx = torch.zeros((inp.shape[0], n // 8), device=model.device)
        st = torch.zeros((2, inp.shape[0], 128), device=model.device)
        p = model.engine.forward_audio(inp, sr, ctx, st)
        torch.cuda.synchronize()
        return p, st

    p1, s1 = run(x)
    p2, s2 = run(x)
    assert torch.equal(p1, p2) and torch.equal(s1, s2)                    # deterministic
    perm = torch.randperm(B, device=model.device, generator=torch.Generator(model.device).manual_seed(1))
    p3, s3 = run(x[perm].contiguous())
    assert torch.equal(p3, p1[perm]) and torch.equal(s3, s1[:, perm])     # streams are independent
    assert torch.isfinite(p1).all() and (p1 >= 0).all() and (p1 <= 1).all()
    sub = slice(0, 64)                                                    # parity subset vs the oracle
    want, _, wst = oracle.forward_audio(x[sub].cpu().numpy(), sr)
    assert np.abs(p1[sub].cpu().numpy() - want).max() < TIGHT
    assert state_err(s1[:, sub].cpu().numpy(), wst) < TOL
    far = slice(4000, 4032)
    want, _, _ = oracle.forward_audio(x[far].cpu().numpy(), sr)
    assert np.abs(p1[far].cpu().numpy() - want).max() < TIGHT
    assert float((p1 > 0.5).float().mean()) > 0.3                         # real speech: not a saturated test


def test_profile_option_reports_kernel_times(model, golden):
    eng = model.engine
    x = torch.from_numpy(rolled_rows(golden["16k"]["wav"], 64, 16 * 512)).to(model.device)
    ctx = torch.zeros((64, 64), device=model.device)
    st = torch.zeros((2, 64, 128), device=model.device)
    eng.set_option("profile", "1")
    try:
        eng.forward_audio(x, 16000, ctx, st)
        eng.forward_audio(x, 16000, ctx, st)
        f, r, calls = eng.kernel_times()
    finally:
        eng.set_option("profile", "0")
    assert f > 0 and r > 0 and calls == 2


# ---- (4) callers either side of the path: live streams (hipGraph) and ragged corpora -------------------
@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_streaming_step_in_hip_graph(model, oracle, golden, tag):
    """BASELINE configs[4]: persistent per-stream state in HBM, one hipGraph-captured vad_step per
    tick; streams join late / are reset mid-way.  Every slot must follow the oracle run of ITS OWN
    audio from its own start."""
    from silero_vad_amd import StreamPool
    sr = SRS[tag]
    n = chunk_of(sr)
    cap, T = 100, 12                                      # capacity deliberately not a multiple of 16
    rows = rolled_rows(golden[tag]["wav"], cap, T * n, 2711)
    pool = StreamPool(model.engine, sr, capacity=cap, graph=True)
    assert pool._graph is not None
    eager = StreamPool(model.engine, sr, capacity=cap, graph=False)
    join = {s: (0 if s < 66 else 4) for s in range(cap)}  # slots 66.. are admitted at tick 4
    got = np.zeros((cap, T), np.float32)
    for t in range(T):
        for s in range(cap):                              # open() hands out slots 0, 1, 2, ...
            if join[s] == t:
                assert pool.open() == s and eager.open() == s
        x = torch.from_numpy(rows[:, t * n:(t + 1) * n]).to(model.device)
        p = pool.tick(x).clone()
        q = eager.tick(x).clone()
        assert torch.equal(p, q)                          # graph replay == eager launch
        got[:, t] = p.cpu().numpy()
    torch.cuda.synchronize()
    for s in (0, 1, 17, 65, 66, 80, 99):
        t0 = join[s]
        want, _, wst = oracle.forward_audio(rows[s:s + 1, t0 * n:], sr)
        assert np.abs(got[s, t0:] - want[0]).max() < TIGHT, s
        assert state_err(pool.state[:, s:s + 1].cpu().numpy(), wst) < TOL
    pool.close(3)
    assert pool.open() == 3 and float(pool.state[:, 3].abs().max()) == 0.0


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_ragged_corpus_equals_single_recording_runs(model, golden, tag):
    """configs[3] plumbing: recordings of different lengths bucketed into lock-step batches give
    bit-identical probabilities and identical segments to one-recording-at-a-time calls."""
    from silero_vad_amd import batch_speech_timestamps, get_speech_timestamps, ragged_probs
    sr, g = SRS[tag], golden[tag]
    n = chunk_of(sr)
    rng = np.random.default_rng(9)
    lens = [int(v) for v in rng.integers(n // 2, 60 * n, size=40)] + [n, n + 1, 37 * n]
    starts = rng.integers(0, len(g["wav"]) - 60 * n, size=len(lens))
    for kind in ("f32", "i16"):
        src = g["wav"] if kind == "f32" else g["pcm_i16"]
        audios = [torch.from_numpy(src[s:s + m].copy()) for s, m in zip(starts, lens)]
        got = ragged_probs(audios, model, sr, max_waste=0.2, max_bytes=1 << 20)
        for a, p in zip(audios, got):
            if len(a) < n:                                # audio_forward itself rejects < 1 window (:124)
                a = torch.nn.functional.pad(a, (0, n - len(a)))
            want = model.audio_forward(a[None], sr)[0]
            assert torch.equal(p, want)
    audios = [torch.from_numpy(g["wav"][s:s + m].copy()) for s, m in zip(starts, lens)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        batch = batch_speech_timestamps(audios, model, sampling_rate=sr, threshold=0.45)
        single = [get_speech_timestamps(a, model, sampling_rate=sr, threshold=0.45) for a in audios]
        batch_s = batch_speech_timestamps(audios, model, sampling_rate=sr, return_seconds=True)
        single_s = [get_speech_timestamps(a, model, sampling_rate=sr, return_seconds=True) for a in audios]
    assert batch == single and batch_s == single_s
    assert sum(len(s) for s in single) > 10


def test_ragged_corpus_repairs_out_of_range_recordings(model, oracle, golden):
    """A recording far outside [-1, 1] inside a ragged corpus: the "auto" policy recomputes just that row in
    fp32, the others keep their f16x3 results."""
    from silero_vad_amd import HipSileroVAD, ragged_probs
    sr, n = 16000, 512
    wav = golden["16k"]["wav"]
    audios = [torch.from_numpy(wav[s:s + m].copy()) for s, m in ((0, 9 * n), (5000, 20 * n + 17), (90000, 14 * n))]
    audios[1] = audios[1] * 1.0e5
    auto = HipSileroVAD(engine=model.engine, precision="auto")
    got = ragged_probs(audios, auto, sr, max_waste=0.9)
    for a, p in zip(audios, got):
        want = oracle.audio_forward(a.numpy()[None], sr)[0]
        assert not torch.isnan(p).any() and np.abs(p.numpy() - want).max() < TOL


# ---- (5) the fp16x3 split arithmetic: what it relies on, and its range guard -------------------------
def _probe(model, a, b):
    from silero_vad_amd import _lib
    a = np.ascontiguousarray(a, np.float16)
    b = np.ascontiguousarray(b, np.float16)
    d = np.empty((64, 4), np.float32)
    _lib.check(model.engine._h, _lib.lib().vad_debug_mfma_f16(
        model.engine._h, a.ctypes.data, b.ctypes.data, d.ctypes.data))
    return d


def test_f16_mfma_slot_pairing_and_subnormals(model):
    """v_mfma_f32_16x16x32_f16 as the split kernels use it: slot (g, e) of A pairs with slot (g, e) of
    B (tests/emu_wave.py mfma_16x16x32_f16 is the specification), products and sums are exact in fp32,
    and fp16 subnormal inputs are NOT flushed -- the lo halves of small activations live there."""
    import emu_wave as E
    rng = np.random.default_rng(5)
    a = rng.standard_normal((64, 8)).astype(np.float16)
    b = rng.standard_normal((64, 8)).astype(np.float16)
    want = E.mfma_16x16x32_f16(a, b, np.zeros((4, 64), np.float32))          # [r][lane]
    got = _probe(model, a, b)
    assert np.abs(got.T - want).max() < 1e-5
    # subnormal halves (|x| < 2^-14) times normal halves
    a = (rng.integers(1, 1024, (64, 8)) * 2.0 ** -24).astype(np.float16)      # all subnormal
    assert np.all(np.abs(a.astype(np.float32)) < 2.0 ** -14) and np.all(a != 0)
    b = rng.integers(1, 64, (64, 8)).astype(np.float16)
    want = E.mfma_16x16x32_f16(a, b, np.zeros((4, 64), np.float32))
    got = _probe(model, a, b)
    assert np.all(want != 0)
    assert np.array_equal(got.T, want), "fp16 subnormal operands were flushed"
    got = _probe(model, b, a)                                                 # subnormal B operand
    want = E.mfma_16x16x32_f16(b, a, np.zeros((4, 64), np.float32))
    assert np.array_equal(got.T, want), "fp16 subnormal B operands were flushed"


@pytest.mark.parametrize("tag", ["16k", "8k"])
def test_split_range_guard(model, oracle, golden, tag):
    """An input far outside [-1, 1] drives activations past the fp16 range: the f16x3 kernels must
    answer NaN for that stream (and only that stream), the fp32 kernels and the "auto" wrapper must
    answer what the oracle answers."""
    from silero_vad_amd import HipSileroVAD
    sr, g = SRS[tag], golden[tag]
    n = chunk_of(sr)
    B, T = 18, 6
    rows = rolled_rows(g["wav"], B, T * n, 4001).copy()
    rows[5] *= 1.0e5
    rows[17, 2 * n:] *= 1.0e5                   # goes out of range from chunk 2 on
    rows[9] *= 20.0                             # loud but inside the fp16 range: must simply be right
    want, _, _ = oracle.forward_audio(rows, sr)
    probs, _, _ = run_engine(model, rows, sr)
    ok = [b for b in range(B) if b not in (5, 9, 17)]
    assert np.abs(probs[ok] - want[ok]).max() < TIGHT
    assert np.abs(probs[9] - want[9]).max() < TOL
    if model.engine.precision == "f16x3":
        assert np.isnan(probs[5]).all()
        # stream 17: right until the first chunk that leaves the range, NaN from there on (a quiet chunk
        # stays in range even when scaled)
        nan17 = np.isnan(probs[17])
        first = int(np.argmax(nan17))
        assert nan17.any() and first >= 2 and nan17[first:].all()
        assert np.abs(probs[17, :first] - want[17, :first]).max() < TOL
    else:
        assert np.abs(probs - want).max() < TOL
    auto = HipSileroVAD(engine=model.engine, precision="auto")
    before = model.engine.precision
    got = auto.audio_forward(torch.from_numpy(rows), sr).numpy()
    assert model.engine.precision == before
    assert np.abs(got - want).max() < TOL
    # per-chunk protocol through the guard
    auto.reset_states()
    for t in range(T):
        p = auto(torch.from_numpy(rows[:, t * n:(t + 1) * n]), sr).cpu().numpy()[:, 0]
        assert np.abs(p - want[:, t]).max() < TOL, t


def test_full_size_launches_are_bit_stable(model, golden):
    """BASELINE configs[1] size (4096 streams x 256 chunks = 65 536 tiles per launch), real speech:
    repeated launches must be bit-identical and the two arithmetic implementations must agree.
    Regression guard for the packed-fp32 / f16-MFMA interference described in
    silero_vad_amd/csrc/kernel_front_split.hip (it showed up as ~3 % of the tiles changing from run to
    run, only at this scale: two workgroups per CU in different phases)."""
    sr, n, B, T = 16000, 512, 4096, 256
    wav = torch.from_numpy(golden["16k"]["wav"]).to(model.device)
    idx = (torch.arange(B, device=model.device)[:, None] * 7919
           + torch.arange(T * n, device=model.device)[None]) % len(wav)
    x = wav[idx].contiguous()
    eng = model.engine

    def run():
        ctx = torch.zeros((B, n // 8), device=model.device)
        st = torch.zeros((2, B, 128), device=model.device)
        p = eng.forward_audio(x, sr, ctx, st)
        torch.cuda.synchronize()
        return p.clone(), st

    p0, s0 = run()
    for _ in range(4):
        p, s = run()
        assert torch.equal(p, p0) and torch.equal(s, s0)
    other = "fp32" if eng.precision == "f16x3" else "f16x3"
    eng.set_precision(other)
    try:
        q, _ = run()
    finally:
        eng.set_precision("f16x3" if other == "fp32" else "fp32")
    assert float((q - p0).abs().max()) < TIGHT
