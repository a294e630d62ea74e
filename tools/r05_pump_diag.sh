#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/r05_pump_diag.sh <tag>': the pump's grid at both rates, unbound / NUMA-bound, + one traced run
set -u
tag=${1:-r05p}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
for sr in 16000 8000; do
  echo "== sr $sr unbound"; python tools/pump_diag.py --sr $sr 2>&1 | tail -40
  echo "== sr $sr bound";   python tools/pump_diag.py --sr $sr --bind --parts 1 2>&1 | tail -20
done | tee $out/grid.txt
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/tools/pump_diag.py --trace --parts 1 --depths 3 --fills 8 > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT && tail -2 $out/trace.log && python tools/pump_diag.py --analyze $out/trace | tee $out/timeline.txt
find $out/trace -name "*.csv" -size +2M -delete
