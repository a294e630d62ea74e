#!/bin/bash
# Run first in a GPU session: is this one of the boxes on which the gather route behind a window leg reads half the link?  If so, collect:
# variants (fresh model per leg, emptied allocator cache, more hardware queues, gather leg first) and kernel/copy traces with queue ids.
set -u
tag=${1:-r05bad}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: (v["fraction_of_pcie_ceiling"], v["wall_s"], v["host_upload_call_ms"]) for k, v in d["legs"].items()})
print("   gather leg host ms:", d["legs"].get("pinned_gather", {}).get("host_ms"), "slot allocs", d["legs"].get("pinned_gather", {}).get("slot_allocs"))
PY
}
run() { name=$1; shift; env "$@" python bench.py --config corpus --corpus-passes 3 --no-cpu-baseline --no-parity > $out/$name.log 2> $out/$name.err || tail -3 $out/$name.err; echo "$name:"; show $out/$name.log; }
run default X=1
bad=$(python - $out/default.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(1 if d["legs"]["pinned_gather"]["fraction_of_pcie_ceiling"] < 0.7 else 0)
PY
)
echo "bad box: $bad  ($(rocm-smi --showuniqueid 2>/dev/null | grep -o '0x[0-9a-f]*' | head -1))" | tee $out/verdict.txt
[ "$bad" = "1" ] || [ -n "${FORCE_BAD:-}" ] || exit 0
run fresh_model VAD_BENCH_CORPUS_FRESH=1
run empty_cache VAD_BENCH_CORPUS_EMPTY_CACHE=1
run hwq8 GPU_MAX_HW_QUEUES=8
run gather_first VAD_BENCH_CORPUS_PRELEG=gather
run gather_main VAD_BENCH_CORPUS_UPLOAD=gather
run nosdma HSA_ENABLE_SDMA=0
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- \
  python $GRAFT_REPO_ROOT/bench.py --config corpus --corpus-passes 3 --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT; echo "traced default order:"; show $out/trace.log; python tools/trace_overlap.py $out/trace > $out/overlap.txt; grep -A12 "^==" $out/overlap.txt | grep "==\|gather  \|front  \|front & gather\|copy_host_to_device " 
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
