#!/usr/bin/env python3
"""CPU study (tests/emu_wave.py, the lane-accurate emulation of front_f43_kernel on the packed images): where the rounding error of
the engine's gate pre-activations comes from.  gx of 16-chunk tiles of speech against float64, (a) as the kernel computes it,
(b) with the FFT magnitudes replaced by the float64 DFT magnitudes rounded once to fp32 -- what is left is the matrix chain --,
and the oracle's order (dense fp32, 8-lane partial sums) beside them."""
import ctypes
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import emu_wave as E  # noqa: E402


def packed_images():
    from silero_vad_amd import _lib
    L = _lib.lib()
    blob = _lib.WEIGHTS_PATH.read_bytes()
    h = ctypes.c_void_p()
    assert L.vad_create_host_only(blob, len(blob), ctypes.byref(h)) == 0
    out = {}
    for sr in (16000, 8000):
        for which in (2, 6):
            n = L.vad_debug_packed_floats(h, sr, which)
            a = np.empty(n, np.float32)
            L.vad_debug_packed_copy(h, sr, which, a.ctypes.data_as(_lib.f32p), n)
            out[sr, which] = a
    L.vad_destroy(h)
    return out


def f64_net(sr):
    from oracle.weights import read_container
    from silero_vad_amd import _lib
    w = read_container(_lib.WEIGHTS_PATH.read_bytes())
    pre = "_model" if sr == 16000 else "_model_8k"
    g = lambda k: w[pre + "." + k].astype(np.float64)
    basis = g("stft.forward_basis_buffer")[:, 0]
    enc = [(g(f"encoder.{i}.reparam_conv.weight"), g(f"encoder.{i}.reparam_conv.bias"), s) for i, s in enumerate((1, 2, 2, 1))]
    w_ih = g("decoder.rnn.weight_ih")
    b = g("decoder.rnn.bias_ih") + g("decoder.rnn.bias_hh")

    def mags(x1):                          # [B, C + n] -> [B, K, 4]
        F = basis.shape[1]
        K = basis.shape[0] // 2
        x = np.concatenate([x1, x1[:, -(F // 4) - 1:-1][:, ::-1]], 1)
        fr = np.stack([x[:, m * F // 2: m * F // 2 + F] for m in range(4)], 1)
        y = fr @ basis.T
        return np.sqrt(y[..., :K] ** 2 + y[..., K:] ** 2).transpose(0, 2, 1)

    def head(a):                           # mags [B, K, 4] -> gx [B, 512]
        for wt, bs, s in enc:
            ap = np.pad(a, ((0, 0), (0, 0), (1, 1)))
            To = (a.shape[2] - 1) // s + 1
            out = np.zeros((a.shape[0], wt.shape[0], To))
            for u in range(To):
                out[:, :, u] = np.einsum("bct,oct->bo", ap[:, :, u * s: u * s + 3], wt) + bs
            a = np.maximum(out, 0)
        return a[:, :, 0] @ w_ih.T + b
    return mags, head


class ExactMagEmu(E.FrontF43Emu):
    """front_f43_kernel's matrix chain on magnitudes that carry ONE rounding (float64 DFT -> fp32)."""

    def fft_pass(self, x, V):
        Q = self.Q
        m = self.exact[:, :, V]                                   # [16, K]
        X = np.zeros((Q + 1, 64), np.float32)
        for lane in range(64):
            g, j = lane >> 4, lane & 15
            X[:Q, lane] = m[j, 4 * np.arange(Q) + E.P_RES[g]]
            if g == 0:
                X[Q, lane] = m[j, 4 * Q]
        return X


def gx_dense(out):
    """gx [32][4][64] in D-fragment order -> [16 chunks][512]"""
    g = np.zeros((16, 512))
    for lane in range(64):
        gg, j = lane >> 4, lane & 15
        for mb in range(32):
            for r in range(4):
                g[j, 16 * mb + 4 * gg + r] = out[mb, r, lane]
    return g


def main():
    from oracle import Oracle
    pk = packed_images()
    orc = Oracle()
    res = {}
    for tag, sr in (("16k", 16000), ("8k", 8000)):
        n = 512 if sr == 16000 else 256
        C = n // 8
        wav = np.load(ROOT / "tests" / "golden" / f"audio_{tag}.npz")["pcm"].astype(np.float32) / 32768.0
        mags, head = f64_net(sr)
        acc = {k: [] for k in ("engine", "engine_exact_mags", "oracle", "mag_engine", "mag_oracle")}
        for tile in range(6):
            off = (40 + 37 * tile) * n
            x1 = np.stack([wav[off + 7919 * i - C: off + 7919 * i + n] for i in range(16)])
            m64 = mags(x1.astype(np.float64))
            want = head(m64)
            emu = E.FrontF43Emu(sr, pk[sr, 6], pk[sr, 2])
            out = emu.run(x1)
            acc["engine"].append(gx_dense(out["gx"]) - want)
            mag_e = np.stack([E.mag_from_layout(out["X"][v], emu.Q) for v in range(4)], -1)
            acc["mag_engine"].append((mag_e - m64) / np.abs(m64).max())
            ex = ExactMagEmu(sr, pk[sr, 6], pk[sr, 2])
            ex.exact = m64.astype(np.float32)
            acc["engine_exact_mags"].append(gx_dense(ex.run(x1)["gx"]) - want)
            _, _, st = orc.step(x1, np.zeros((2, 16, 128), np.float32), sr, stages=True)
            from oracle.weights import read_container
            from silero_vad_amd import _lib
            w = read_container(_lib.WEIGHTS_PATH.read_bytes())
            pre = "_model" if sr == 16000 else "_model_8k"
            gxo = st["enc3"][:, :, 0] @ w[pre + ".decoder.rnn.weight_ih"].T + (w[pre + ".decoder.rnn.bias_ih"] + w[pre + ".decoder.rnn.bias_hh"])
            acc["oracle"].append(gxo - want)
            acc["mag_oracle"].append((st["mag"] - m64) / np.abs(m64).max())
        res[tag] = {k: {"rms": float(np.sqrt(np.mean(np.square(np.concatenate(v))))), "max": float(np.abs(np.concatenate(v)).max())} for k, v in acc.items()}
        print(tag, json.dumps(res[tag], indent=1))
    json.dump(res, open(ROOT / "gpurun_out" / "gx_error_study.json", "w"), indent=1)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


# ---- second part: what a split-K accumulation would buy (python tools/gx_error_study.py split) ---------------------------------------------
def mfma_chain(a, b, acc):
    """v_mfma_f32_16x16x4_f32 as an fmaf chain: one rounding per product-add, k ascending (what the header says the hardware does)."""
    A = a.reshape(4, 16).T.astype(np.float64)
    Bm = b.reshape(4, 16).astype(np.float64)
    out = acc.copy()
    for r in range(4):
        v = out[r].astype(np.float32)
        for k in range(4):
            v = (v.astype(np.float64) + A[4 * E.G + r, k] * Bm[k, E.J]).astype(np.float32)
        out[r] = v
    return out


class SplitEmu(E.FrontF43Emu):
    """front_f43_kernel with the k-groups of a GEMM segment dealt to `split` partial accumulators (zero-initialised, summed pairwise
    into the running accumulator at the end of the segment) for the segments named in `where`."""
    split, where = 1, ()
    reverse = ()            # segments whose k-groups are visited in DESCENDING order (high-frequency bins / late channels first)
    seg = 0

    def gemm_w(self, acc, bfun, M, KG):
        name = self.names[self.seg] if self.seg < len(self.names) else "?"
        self.seg += 1
        sp = self.split if (name in self.where and KG % self.split == 0) else 1
        rev = name in self.reverse
        if sp == 1 and not rev:
            return super().gemm_w(acc, bfun, M, KG)
        parts = [np.zeros_like(acc) for _ in range(sp)]
        steps = KG * (M // 2)
        order = range(steps)
        if rev:             # same blocks, k-groups from the last to the first (and the 4 k-steps inside a block too)
            order = [kg * (M // 2) + m for kg in reversed(range(KG)) for m in range(M // 2)]
        for i in order:
            unit = self.sched[self.pu + i // 8]
            base = unit * self.UNIT + (i % 8) * 2 * 256
            kg, mp = i // (M // 2), 2 * (i % (M // 2))
            tgt = parts[kg * sp // KG]
            for d in range(2):
                blk = self.image[base + d * 256: base + (d + 1) * 256].reshape(64, 4)
                for ks in (reversed(range(4)) if rev else range(4)):
                    tgt[mp + d] = E.mfma_16x16x4(blk[:, ks], bfun(kg * 4 + ks), tgt[mp + d])
        self.pu += steps // 8
        while len(parts) > 1:
            parts = [parts[i] + parts[i + 1] for i in range(0, len(parts), 2)]
        acc += parts[0]


def split_study():
    pk = packed_images()
    E.mfma_16x16x4 = mfma_chain
    res = {}
    for tag, sr in (("16k", 16000), ("8k", 8000)):
        n = 512 if sr == 16000 else 256
        C = n // 8
        wav = np.load(ROOT / "tests" / "golden" / f"audio_{tag}.npz")["pcm"].astype(np.float32) / 32768.0
        mags, head = f64_net(sr)
        Q = 32 if sr == 16000 else 16
        P = E.w_parts(Q)
        # segment names in program order (FrontF43Emu.run)
        per_part = ["e0"] * 6 + (["e1", "e1"] if Q == 32 else ["e1"] * 5)
        names = []
        for p in range(P):
            names += per_part + (["e1"] if (Q == 32 and p & 1) else [])
        names += ["e2", "e2", "e3", "ih", "ih", "ih", "ih"]
        variants = {"as is (one accumulator per output, k ascending)": (1, (), ()), "enc0 in 2": (2, ("e0",), ()), "enc1 in 2": (2, ("e1",), ()),
                    "enc0 + enc1 in 2": (2, ("e0", "e1"), ()), "every segment in 4": (4, ("e0", "e1", "e2", "e3", "ih"), ()),
                    "enc0 k DESCENDING": (1, (), ("e0",)), "enc0 + enc1 k descending": (1, (), ("e0", "e1")),
                    "all k descending": (1, (), ("e0", "e1", "e2", "e3", "ih"))}
        out = {}
        for vname, (sp, where, rev) in variants.items():
            errs = []
            for tile in range(3):
                off = (40 + 37 * tile) * n
                x1 = np.stack([wav[off + 7919 * i - C: off + 7919 * i + n] for i in range(16)])
                want = head(mags(x1.astype(np.float64)))
                emu = SplitEmu(sr, pk[sr, 6], pk[sr, 2])
                emu.split, emu.where, emu.names, emu.seg, emu.reverse = sp, where, names, 0, rev
                errs.append(gx_dense(emu.run(x1)["gx"]) - want)
            e = np.concatenate(errs)
            out[vname] = {"rms": float(np.sqrt(np.mean(e ** 2))), "max": float(np.abs(e).max())}
            print(tag, vname, out[vname], flush=True)
        res[tag] = out
    json.dump(res, open(ROOT / "gpurun_out" / "gx_split_study.json", "w"), indent=1)


def robust_study():
    """'as is' against 'all k descending' on other kinds of input: is the gain a property of the order or of three tiles of speech?"""
    pk = packed_images()
    E.mfma_16x16x4 = mfma_chain
    rng = np.random.default_rng(3)
    for tag, sr in (("16k", 16000), ("8k", 8000)):
        n = 512 if sr == 16000 else 256
        C = n // 8
        wav = np.load(ROOT / "tests" / "golden" / f"audio_{tag}.npz")["pcm"].astype(np.float32) / 32768.0
        mags, head = f64_net(sr)
        Q = 32 if sr == 16000 else 16
        P = E.w_parts(Q)
        per_part = ["e0"] * 6 + (["e1", "e1"] if Q == 32 else ["e1"] * 5)
        names = []
        for p in range(P):
            names += per_part + (["e1"] if (Q == 32 and p & 1) else [])
        names += ["e2", "e2", "e3", "ih", "ih", "ih", "ih"]
        tt = np.arange(16 * (n + C)).reshape(16, n + C) / sr
        inputs = {"speech, other offsets": np.stack([wav[(300 + 11 * i) * n - C: (300 + 11 * i) * n + n] for i in range(16)]),
                  "speech x 0.01": np.stack([wav[(90 + 7 * i) * n - C: (90 + 7 * i) * n + n] for i in range(16)]) * np.float32(0.01),
                  "white noise 0.1": (0.1 * rng.standard_normal((16, n + C))).astype(np.float32),
                  "tone 3 kHz + noise": (0.3 * np.sin(2 * np.pi * 3000 * tt) + 0.01 * rng.standard_normal((16, n + C))).astype(np.float32),
                  "silence + one click": np.where(np.arange(n + C)[None, :] == 200, 0.9, 0.0).astype(np.float32).repeat(16, 0)}
        for iname, x1 in inputs.items():
            want = head(mags(x1.astype(np.float64)))
            row = {}
            for vname, rev in (("as is", ()), ("all k descending", ("e0", "e1", "e2", "e3", "ih")), ("enc0+enc1 descending", ("e0", "e1"))):
                emu = SplitEmu(sr, pk[sr, 6], pk[sr, 2])
                emu.split, emu.where, emu.names, emu.seg, emu.reverse = 1, (), names, 0, rev
                e = gx_dense(emu.run(np.ascontiguousarray(x1))["gx"]) - want
                row[vname] = (float(np.sqrt(np.mean(e ** 2))), float(np.abs(e).max()))
            print(tag, iname, {k: (f"{a:.2e}", f"{b:.2e}") for k, (a, b) in row.items()}, flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "split":
    split_study()
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "robust":
    robust_study()
