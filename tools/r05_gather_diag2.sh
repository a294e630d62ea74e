#!/bin/bash
# order dependence of the gather route: alone / behind a window leg / in front of it / with more hardware queues; then the in-line case traced
set -u
tag=${1:-r05h}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: (v["fraction_of_pcie_ceiling"], v["wall_s"], v["host_upload_call_ms"], v["recordings_per_gpu"]) for k, v in d["legs"].items()})
PY
}
run() { name=$1; shift; env "$@" python bench.py --config corpus --corpus-passes 3 --no-cpu-baseline --no-parity > $out/$name.log 2> $out/$name.err || tail -3 $out/$name.err; echo "$name:"; show $out/$name.log; }
run A_gather_main_alone VAD_BENCH_CORPUS_UPLOAD=gather X=1
run B_default_order X=1
run C_gather_first VAD_BENCH_CORPUS_PRELEG=gather
run D_hwq8 GPU_MAX_HW_QUEUES=8
run E_hwq2 GPU_MAX_HW_QUEUES=2
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- \
  python $GRAFT_REPO_ROOT/bench.py --config corpus --corpus-passes 3 --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT; echo "traced default order:"; show $out/trace.log; python tools/trace_overlap.py $out/trace | tee $out/overlap.txt
python - $out/trace <<'PY' | tee $out/queues.txt
import csv, glob, os, sys, collections
q = collections.defaultdict(collections.Counter)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "gather" if "gather" in n else "front" if "front_f43" in n else "rec" if "rec_" in n else "cut" if "scatter" in n else None
        if k: q[k][r.get("Queue_Id")] += 1
print({k: dict(v) for k, v in q.items()})
PY
find $out -name "*.csv" -size +1M -delete
