"""Host logic of the drop-in model object that needs no GPU: validation rules and error texts of
the reference boundary (JIT!/vad/model/vad_annotator.py:17,91-127; utils_vad.py:33-49)."""
import pytest
import torch


class _NoEngine:
    device = 0


@pytest.fixture()
def model(built):
    from silero_vad_amd.engine import HipSileroVAD
    m = HipSileroVAD.__new__(HipSileroVAD)
    m.engine = _NoEngine()
    m.device = torch.device("cpu")
    m.sample_rates = [8000, 16000]
    m.reset_states()
    return m


def test_validate_input_rules(model):
    x, sr = model._validate_input(torch.zeros(512), 16000)
    assert x.shape == (1, 512) and sr == 16000
    x, sr = model._validate_input(torch.zeros(3, 1536), 48000)          # ::3 decimation
    assert x.shape == (3, 512) and sr == 16000
    x, sr = model._validate_input(torch.zeros(256), 8000)
    assert x.shape == (1, 256) and sr == 8000
    with pytest.raises(ValueError, match=r"Too many dimensions for input audio chunk 3"):
        model._validate_input(torch.zeros(1, 1, 512), 16000)
    with pytest.raises(ValueError, match=r"Supported sampling rates: \[8000, 16000\] \(or multiply of 16000\)"):
        model._validate_input(torch.zeros(512), 22050)
    with pytest.raises(ValueError, match="Input audio chunk is too short"):
        model._validate_input(torch.zeros(400), 16000)                   # 16000/400 = 40 > 31.25
    model._validate_input(torch.zeros(512), 16000)                       # 31.25 exactly is allowed


def test_call_rejects_wrong_chunk_size(model):
    with pytest.raises(ValueError, match=r"Provided number of samples is 640 \(Supported values: 256 for 8000 "
                                         r"sample rate, 512 for 16000\)"):
        model(torch.zeros(640), 16000)
    with pytest.raises(ValueError, match="Provided number of samples is 512"):
        model(torch.zeros(512), 8000)


def test_reset_states_protocol(model):
    model._last_sr, model._last_batch_size = 16000, 4
    model.reset_states()
    assert model._last_sr == 0 and model._last_batch_size == 0
    assert len(model._state) == 0 and len(model._context) == 0


def test_get_speech_timestamps_argument_errors(built):
    from silero_vad_amd import VADIterator, get_speech_timestamps

    class Dummy:
        def reset_states(self):
            pass

    with pytest.raises(ValueError, match="More than one dimension"):
        get_speech_timestamps(torch.zeros(2, 1024), Dummy())
    with pytest.raises(ValueError, match="Currently silero VAD models support 8000 and 16000"):
        get_speech_timestamps(torch.zeros(1024), Dummy(), sampling_rate=44100)
    with pytest.raises(ValueError, match="does not support sampling rates other than"):
        VADIterator(Dummy(), sampling_rate=48000)
    assert get_speech_timestamps(torch.zeros(0), Dummy()) == []


def test_hubconf_contract(built, tmp_path):
    """torch.hub entry (reference hubconf.py:26-56): `silero_vad()` -> (model, utils) with the reference's five utils in its order;
    without a GPU the model refuses loudly (no CPU fallback), the utils and the argument check work."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("hubconf", Path(__file__).resolve().parents[1] / "hubconf.py")
    hub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hub)
    assert "torch" in hub.dependencies
    with pytest.raises(Exception, match="Available ONNX opset_version"):
        hub.silero_vad(onnx=True, opset_version=14)
    if torch.cuda.is_available():
        model, utils = hub.silero_vad()
        assert model.sample_rates == [8000, 16000]
    else:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            hub.silero_vad()
        from silero_vad_amd import VADIterator, collect_chunks, get_speech_timestamps, read_audio, save_audio
        utils = (get_speech_timestamps, save_audio, read_audio, VADIterator, collect_chunks)
    assert [u.__name__ for u in utils] == ["get_speech_timestamps", "save_audio", "read_audio", "VADIterator", "collect_chunks"]
    x = 0.5 * torch.sin(torch.arange(8000) / 15.0)
    utils[1](str(tmp_path / "a.wav"), x, 8000)
    y = utils[2](str(tmp_path / "a.wav"), 8000)
    assert y.shape == x.shape and (x - y).abs().max() < 1.0 / 16384
    assert torch.equal(utils[4]([{"start": 10, "end": 20}, {"start": 100, "end": 130}], x), torch.cat([x[10:20], x[100:130]]))


def test_bench_line_digest_is_last_and_small():
    """the driver keeps the TAIL of the bench line -- its last key, `legs`, must carry every leg (value, fraction
    of its bound, its own parity figure, the largest checked probability) in at most 1.5 KB.  Checked on the committed line of the round
    and by recomputing the digest from it."""
    import json
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    line = (root / "profiles" / "r05_bench_line.json").read_text().strip()
    d = json.loads(line)
    assert list(d)[-1] == "legs" and line.rstrip().endswith("}}")
    legs = d["legs"]
    assert set(legs) == {"c2", "8k", "stream", "stream_host", "corpus", "stream_8k", "stream_host_8k", "plumbing", "plumbing_8k"}
    assert len(json.dumps(legs)) <= 1536
    for k in ("c2", "8k", "stream", "stream_8k", "stream_host", "stream_host_8k"):
        assert legs[k]["max_prob"] > 0.9 and legs[k]["dp"] < 1e-4, k          # the self-check spans the sigmoid
    assert legs["stream_host"]["of_link"] >= 0.85 and legs["stream_host_8k"]["of_link"] >= 0.85
    assert legs["stream_host"]["tick_ms_p95"] <= 0.3 and legs["stream_host_8k"]["tick_ms_p95"] <= 0.3
    import sys
    sys.path.insert(0, str(root))
    import bench
    assert bench.compact_legs({k: v for k, v in d.items() if k != "legs"}) == legs
