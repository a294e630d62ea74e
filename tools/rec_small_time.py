import sys, torch
sys.path.insert(0, "/root/repo")
from silero_vad_amd import Engine
eng = Engine(0); dev = torch.device("cuda", 0)
for sr, n in ((16000, 512), (8000, 256)):
    for B in (1, 4, 16, 100, 256, 300, 512, 700, 1024, 1040):
        T = 300
        x = 0.1 * torch.randn((B, T * n), device=dev)
        out = []
        for form in ("mfma", "auto"):
            eng.set_option("rec_form", form)
            ctx = torch.zeros((B, n // 8), device=dev); st = torch.zeros((2, B, 128), device=dev)
            for _ in range(3): eng.forward_audio(x, sr, ctx, st)
            eng.set_option("profile", "1")
            for _ in range(5): eng.forward_audio(x, sr, ctx, st)
            f, r, c = eng.kernel_times(); eng.set_option("profile", "0")
            out.append(r / c / T * 1e3)
        print(sr, "B", B, "us per step: mfma %.2f  valu %.2f" % tuple(out), flush=True)
