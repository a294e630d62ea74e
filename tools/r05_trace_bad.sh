#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r05t; mkdir -p $out
python __graft_entry__.py > /dev/null 2>&1
python tools/leg_order_diag.py corpus37 corpus3 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | cut -c1-300
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/tools/leg_order_diag.py corpus37 corpus3 > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT; grep corpus $out/trace.log | cut -c1-300
python tools/trace_overlap.py $out/trace > $out/overlap.txt
grep "^==\|^gather\|^front \|^rec \|front & gather\|gather & rec" $out/overlap.txt
python - $out/trace <<'PY'
import csv, glob, os, sys, collections
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "gather" if "gather" in n else "front" if "front_f43" in n else "rec" if "rec_" in n else "scan" if "scan" in n else None
        if k: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id"), r.get("Stream_Id", "")))
rows.sort()
g = [r for r in rows if r[2] == "gather"]
# bursts of gather kernels
bursts, cur = [], [g[0]]
for r in g[1:]:
    if r[0] - cur[-1][1] > 50e6: bursts.append(cur); cur = []
    cur.append(r)
bursts.append(cur)
for bi, b in enumerate(bursts):
    if len(b) < 8: continue
    a, e = b[len(b)//2][0], b[len(b)//2][0] + 60e6
    print(f"-- burst {bi}: {len(b)} gather kernels; 60 ms from its middle (ms: start end dur kind queue)")
    for r in rows:
        if a <= r[0] <= e:
            print(f"   {(r[0]-a)/1e6:8.2f} {(r[1]-a)/1e6:8.2f} {(r[1]-r[0])/1e6:7.2f} {r[2]:7s} q{r[3]}")
PY
find $out -name "*.csv" -size +1M -delete
