#!/bin/bash
# tests + kernel variants
set -u
tag=${1:-x}; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python __graft_entry__.py > "$out/build.log" 2>&1
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -p no:cacheprovider > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$?" >> "$out/pytest_gpu.log"
tail -n 15 "$out/pytest_gpu.log"
timeout 900 python tools/variants.py run "$@" > "$out/variants.log" 2>&1
cp gpurun_out/variants.json "$out/variants.json" 2>/dev/null
tail -n 12 "$out/variants.log" | cut -c1-300
cp gpurun_out/foreign_load_*.json "$out/" 2>/dev/null
