import csv,glob,sys
d=sys.argv[1]
f=glob.glob(d+"/*memory_copy_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if int(r["End_Timestamp"])-int(r["Start_Timestamp"])>50000 and r["Stream_Id"]!="0"]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tail=rows[-260:-20]
t0=int(tail[0]["Start_Timestamp"])
busy=0; last=0
iv=[(int(r["Start_Timestamp"])-t0, int(r["End_Timestamp"])-t0) for r in tail]
# union length
cur_s,cur_e=iv[0]
for a,b in iv[1:]:
    if a<=cur_e: cur_e=max(cur_e,b)
    else: busy+=cur_e-cur_s; cur_s,cur_e=a,b
busy+=cur_e-cur_s
span=iv[-1][1]-iv[0][0]
print("copies",len(iv),"span_us",round(span/1e3,1),"per_tick_us",round(span/1e3/len(iv),1),"link_busy_frac",round(busy/span,3),"mean_copy_us",round(sum(b-a for a,b in iv)/len(iv)/1e3,1))
for a,b in iv[:12]: print(round(a/1e3,1), round(b/1e3,1))
