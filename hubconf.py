"""torch.hub entry point, the reference's `hubconf.py` contract (/root/reference/hubconf.py:26-56): `silero_vad(...)` returns
`(model, utils)` with `utils = (get_speech_timestamps, save_audio, read_audio, VADIterator, collect_chunks)`, so that

    model, utils = torch.hub.load(repo_or_dir=<this repo>, model="silero_vad", source="local")
    (get_speech_timestamps, save_audio, read_audio, VADIterator, collect_chunks) = utils

keeps working for a caller of the reference.  The model is the MI355X engine (no CPU fallback: it raises without a gfx950 GPU)."""
dependencies = ["torch", "numpy"]
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from silero_vad_amd import VADIterator, collect_chunks, get_speech_timestamps, load_silero_vad, read_audio, save_audio  # noqa: E402


def silero_vad(onnx=False, force_onnx_cpu=False, opset_version=16, device=0):
    """Silero Voice Activity Detector on MI355X.  `onnx`, `force_onnx_cpu`, `opset_version` are accepted for signature compatibility:
    there is one backend here (hand-written HIP kernels behind the C ABI), and it reproduces the JIT model's numbers."""
    if onnx and opset_version not in (15, 16):
        raise Exception("Available ONNX opset_version: [15, 16]")
    model = load_silero_vad(onnx=onnx, opset_version=opset_version, device=device)
    return model, (get_speech_timestamps, save_audio, read_audio, VADIterator, collect_chunks)
