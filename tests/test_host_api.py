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


