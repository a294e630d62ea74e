"""A stand-in for `silero_vad_amd.Engine` that answers from the CPU oracle -- TEST INFRASTRUCTURE.

It lets the host side of the drop-in (`HipSileroVAD`: validation, state ownership, auto-reset, the
`audio_forward` fast path of `get_speech_timestamps`) run in the authoring container, where there is no GPU,
behind the reference's own unmodified callers (tests/test_reference_callers.py).  It lives under tests/ and is
never importable from the product package."""
import numpy as np
import torch


class ReplayEngine:
    device = 0
    torch_device = torch.device("cpu")
    fused_front_door = False         # the oracle takes 8 / 16 kHz only: HipSileroVAD decimates on the host for this stand-in
    options = {}

    def __init__(self, oracle):
        self.oracle = oracle
        self.precision = "fp32"
        self.calls = {"step": 0, "forward_audio": 0}

    def set_precision(self, p):
        self.precision = p

    def step(self, pcm, sr, ctx, state, prob):
        self.calls["step"] += 1
        x1 = np.concatenate([ctx.numpy(), pcm.numpy().astype(np.float32)], axis=1)
        p, st = self.oracle.step(x1, state.numpy(), sr)
        state.copy_(torch.from_numpy(st))
        ctx.copy_(torch.from_numpy(x1[:, -ctx.shape[1]:].copy()))
        prob.copy_(torch.from_numpy(p).view_as(prob))
        return prob

    def forward_audio(self, pcm, sr, ctx, state, probs=None):
        self.calls["forward_audio"] += 1
        x = pcm.numpy().astype(np.float32)
        if pcm.dtype == torch.int16:
            x = x / 32768.0
        if sr > 16000 and sr % 16000 == 0:       # raw 32 / 48 kHz rows from the corpus schedulers: the reference's x[:, ::k] (utils_vad.py:39-42)
            x, sr = np.ascontiguousarray(x[:, ::sr // 16000]), 16000
        p, c, st = self.oracle.forward_audio(x, sr, ctx=ctx.numpy(), state=state.numpy())
        state.copy_(torch.from_numpy(st))
        ctx.copy_(torch.from_numpy(c))
        out = torch.from_numpy(p)
        if probs is not None:
            probs.copy_(out)
            return probs
        return out
