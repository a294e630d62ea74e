#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r06_refill_diag.sh <tag>': kernel trace of the refill leg
set -u
tag=${1:-r06r}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
cd /tmp && VAD_BENCH_ONLY_REFILL=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --config corpus --no-cpu-baseline --no-parity --corpus-passes ${PASSES:-6} > $GRAFT_REPO_ROOT/$out/bench.log 2> $GRAFT_REPO_ROOT/$out/bench.err
cd $GRAFT_REPO_ROOT && python tools/refill_diag.py $out/trace | tee $out/timeline.txt
find $out/trace -name "*.csv" -size +1M -delete
