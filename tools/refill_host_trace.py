import os, sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from silero_vad_amd import load_silero_vad, PackedRecordings
from silero_vad_amd import streams as S
sr=16000; R=4096; passes=12
model=load_silero_vad(device=0)
rng=np.random.default_rng(7)
base_len=8<<20
base=(0.03*rng.standard_normal(base_len)).astype(np.float32)
page=torch.from_numpy((base*32767).clip(-32768,32767).astype(np.int16))
lens=np.concatenate([np.random.default_rng(101+p).integers(20*sr,40*sr,size=R) for p in range(passes)])
offs=np.zeros(len(lens),dtype=np.int64); span=0
for p in range(passes):
    l=lens[p*R:(p+1)*R]; o=np.concatenate([[0],np.cumsum((l+7)//8*8)[:-1]]); offs[p*R:(p+1)*R]=o; span=max(span,int(o[-1]+l[-1]))
arena_len=(span+base_len-1)//base_len*base_len
arena=torch.empty(arena_len,dtype=torch.int16,pin_memory=True); arena.view(-1,base_len)[:]=page
os.environ["SILERO_VAD_AMD_UPLOAD"]="gather"
rec=PackedRecordings(arena,offs,lens)
def run(m):
    S.STATS.clear(); S.TRACE=[]
    torch.cuda.synchronize(); t0=time.perf_counter()
    n=0
    for idx,cnt,_ in S.refill_segments_stream(PackedRecordings(arena,offs[:m],lens[:m]),model,sr,slots=2048,slab_chunks=128): n+=len(idx)
    torch.cuda.synchronize(); el=time.perf_counter()-t0
    tr=S.TRACE
    per=np.diff([t[3] for t in tr])*1e3
    on=np.array([t[2]-t[1] for t in tr])*1e3; st=np.array([t[3]-t[2] for t in tr])*1e3
    chunks=int(((lens[:m]+511)//512).sum())
    print(json.dumps({"recs":m,"wall_s":round(el,4),"Mchunks_s":round(chunks/el/1e6,2),"slabs":len(tr),"per_slab_ms":{"p50":round(float(np.percentile(per,50)),2),"p90":round(float(np.percentile(per,90)),2),"max":round(float(per.max()),2)},
      "on_slab_ms":{"p50":round(float(np.percentile(on,50)),3),"p90":round(float(np.percentile(on,90)),3),"max":round(float(on.max()),2),"sum":round(float(on.sum()),1)},
      "stage_ms":{"p50":round(float(np.percentile(st,50)),3),"p90":round(float(np.percentile(st,90)),3),"max":round(float(st.max()),2),"sum":round(float(st.sum()),1)},
      "stats":{k:round(v,4) for k,v in S.STATS.items() if k.endswith("_s")}}))
    big=np.argsort(-per)[:8]
    print("slowest slabs:", [(int(i), round(float(per[i]),2), round(float(on[i+1]),2), round(float(st[i+1]),2)) for i in sorted(big)])
run(2*R); run(len(lens))
os.environ["SILERO_VAD_AMD_UPLOAD"]="window"
t0=time.perf_counter(); S.ragged_speech_segments(rec, model, sr, max_waste=0.1, max_bytes=1<<30, as_arrays=True); torch.cuda.synchronize(); print("bucket/window run", round(time.perf_counter()-t0,3))
os.environ["SILERO_VAD_AMD_UPLOAD"]="gather"
print("pool slots", model._stage_pool.slots)
run(len(lens)); run(len(lens))
