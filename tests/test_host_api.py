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


class _FakeEngine:
    """Stands in for the C-ABI engine: 'f16x3' answers NaN (and poisons the carried state) when the input is
    louder than 4, 'fp32' never does -- the contract of include/silero_vad_hip.h, option "precision"."""
    device = 0

    def __init__(self):
        self.precision = "f16x3"
        self.calls = []

    def set_precision(self, p):
        self.precision = p


def _guard_model(policy):
    from silero_vad_amd.engine import HipSileroVAD
    m = HipSileroVAD.__new__(HipSileroVAD)
    m.engine = _FakeEngine()
    m.precision = policy
    m.device = torch.device("cpu")
    m.sample_rates = [8000, 16000]
    m.reset_states()
    m._state = torch.zeros(2, 1, 128)
    m._context = torch.zeros(1, 64)
    return m


def _fake_step(m, x):
    def run():
        m.engine.calls.append(m.engine.precision)
        loud = bool(x.abs().max() > 4) and m.engine.precision == "f16x3"
        m._state += float("nan") if loud else 1.0            # the kernels update the state in place
        m._context += 1.0
        return torch.full((1, 1), float("nan") if loud else 0.25)
    return run


def test_auto_precision_reruns_out_of_range_calls_in_fp32():
    m = _guard_model("auto")
    out = m._guarded(_fake_step(m, torch.ones(512)))
    assert out.item() == 0.25 and m.engine.calls == ["f16x3"]
    out = m._guarded(_fake_step(m, 100 * torch.ones(512)))
    assert out.item() == 0.25                                   # answered by the fp32 rerun
    assert m.engine.calls == ["f16x3", "f16x3", "fp32"]
    assert m.engine.precision == "f16x3"                        # policy restored
    assert torch.all(m._state == 2.0) and torch.all(m._context == 2.0)   # state advanced exactly twice, no NaN left


def test_pinned_precision_is_not_second_guessed():
    m = _guard_model("f16x3")
    out = m._guarded(_fake_step(m, 100 * torch.ones(512)))
    assert torch.isnan(out).all() and m.engine.calls == ["f16x3"]        # NaN is the documented answer
