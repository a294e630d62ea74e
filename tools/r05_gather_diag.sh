#!/bin/bash
# gpurun --timeout 1200 -- 'bash tools/r05_gather_diag.sh <tag>': the scattered-pinned corpus routes on THIS box, alone in a fresh
# process each, then the gather route under rocprofv3 (kernel + copy trace) and the overlap of its upload kernel with the compute
set -u
tag=${1:-r05g}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: (v["fraction_of_pcie_ceiling"], v["wall_s"], v["host_upload_call_ms"]) for k, v in d["legs"].items()})
PY
}
for rep in 1 2; do
  python bench.py --config corpus --corpus-passes 3 --no-cpu-baseline > $out/corpus_$rep.log 2> $out/corpus_$rep.err || tail -3 $out/corpus_$rep.err
  echo "all routes, run $rep:"; show $out/corpus_$rep.log
done
for mode in gather window; do
  VAD_BENCH_CORPUS_UPLOAD=$mode python bench.py --config corpus --corpus-passes 6 --corpus-main-only --no-cpu-baseline > $out/only_$mode.log 2> $out/only_$mode.err
  echo "main = $mode alone (6 passes):"; show $out/only_$mode.log
done
${EXTRA_ENV:-} true
cd /tmp && VAD_BENCH_CORPUS_UPLOAD=gather rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace_gather -- \
  python $GRAFT_REPO_ROOT/bench.py --config corpus --corpus-passes 4 --corpus-main-only --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/$out/trace_gather.log 2>&1
cd $GRAFT_REPO_ROOT; echo "traced gather run:"; show $out/trace_gather.log; python tools/trace_overlap.py $out/trace_gather | tee $out/overlap_gather.txt
find $out -name "*.csv" -size +1M -delete
