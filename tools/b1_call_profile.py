import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from silero_vad_amd import load_silero_vad
m = load_silero_vad(device=0)
wav = torch.from_numpy(np.load("/root/repo/tests/golden/audio_16k.npz")["pcm"].astype(np.float32) / 32768.0)
n = 512
for i in range(100): m(wav[i*n:(i+1)*n], 16000).item()
def t(fn, reps=2000):
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
c = wav[:n]
print("full call+item us", t(lambda: m(c, 16000).item()))
print("front_door", t(lambda: m._front_door(c, 16000)))
x = c.unsqueeze(0)
print("ensure_state", t(lambda: m._ensure_state(16000, 1)))
pcm, prob = m._small[0][:1], m._small[1][:1, 0]
print("slices", t(lambda: (m._small[0][:1], m._small[1][:1])))
print("copy_", t(lambda: pcm.copy_(x)))
print("current_stream", t(lambda: torch.cuda.current_stream(m.device)))
st = torch.cuda.current_stream(m.device)
print("cuda_stream attr", t(lambda: st.cuda_stream))
def launch_sync():
    m.engine.step_host(pcm, None, 16000, m._context, m._state, None, prob, st.cuda_stream); st.synchronize()
print("step_host + sync", t(launch_sync))
def launch_only():
    m.engine.step_host(pcm, None, 16000, m._context, m._state, None, prob, st.cuda_stream)
t0=time.perf_counter()
for _ in range(2000): launch_only()
el=(time.perf_counter()-t0)/2000*1e6; st.synchronize()
print("step_host enqueue only us", el)
print("sync idle", t(lambda: st.synchronize()))
print("clone+unsqueeze", t(lambda: prob.clone().unsqueeze(1)))
print("item", t(lambda: prob.clone().unsqueeze(1).item()))
import ctypes
L = m.engine._L; h = m.engine._h
args = (h, 16000, 1, pcm.data_ptr(), 4, None, m._context.data_ptr(), m._state.data_ptr(), None, prob.data_ptr(), ctypes.c_void_p(st.cuda_stream))
def raw():
    L.vad_step_host(*args); st.synchronize()
print("raw ctypes + sync", t(raw))
m.engine.set_option("profile", "1")
for _ in range(200): launch_sync()
print("kernel times", m.engine.kernel_times())
# the same step with the chunk in HBM (what does reading it over PCIe cost the kernel?)
xd = c.unsqueeze(0).to(m.device); out = torch.empty((1, 1), device=m.device)
for _ in range(200):
    m.engine.step(xd, 16000, m._context, m._state, out); st.synchronize()
print("kernel times, chunk in HBM, prob to HBM", m.engine.kernel_times())
