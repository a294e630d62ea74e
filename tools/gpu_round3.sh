#!/bin/bash
# tests + the full default bench line (CPU baseline, other precision, other configs)
set -u
tag=${1:-x}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python __graft_entry__.py > "$out/build.log" 2>&1
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -p no:cacheprovider -x > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$?" >> "$out/pytest_gpu.log"
tail -n 30 "$out/pytest_gpu.log"
( time timeout 1200 python bench.py ) > "$out/bench_c2.log" 2>&1; tail -n 8 "$out/bench_c2.log" | cut -c1-9000
cp gpurun_out/foreign_load_*.json "$out/" 2>/dev/null
