#!/bin/bash
# GPU parity suite only:  gpurun --timeout 900 -- 'bash tools/r03_gpu_tests.sh <tag>'
set -u
tag=${1:-r03x}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
timeout 800 python -m pytest tests -m gpu -q --timeout 300 --timeout-method=thread -p no:cacheprovider ${2:-} > $out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -n 60 $out/pytest_gpu.log | cut -c1-400
