#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r06_refill_window_diag.sh <tag>': copy trace of the corpus legs (the refill window feed's is the last run)
set -u
tag=${1:-r06rw}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp && VAD_BENCH_ONLY_REFILL=1 rocprofv3 --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --config corpus --no-cpu-baseline --no-parity --corpus-passes ${PASSES:-37} > $GRAFT_REPO_ROOT/$out/bench.log 2> $GRAFT_REPO_ROOT/$out/bench.err
cd $GRAFT_REPO_ROOT && python tools/refill_window_diag.py $out/trace | tee $out/timeline.txt
find $out/trace -name "*.csv" -size +1M -delete
