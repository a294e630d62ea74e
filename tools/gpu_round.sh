#!/bin/bash
# One GPU-box session: parity tests, bench lines, optional rocprof passes.  Run from the repo root:
#   gpurun --timeout 1800 -- 'bash tools/gpu_round.sh <tag> [prof]'
set -u
tag=${1:-x}; prof=${2:-}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python __graft_entry__.py > "$out/build.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 --timeout-method=thread -p no:cacheprovider > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$?" >> "$out/pytest_gpu.log"
tail -n 40 "$out/pytest_gpu.log"
timeout 900 python bench.py > "$out/bench_c2.log" 2>&1; tail -n 3 "$out/bench_c2.log" | cut -c1-6000
if [ -n "$prof" ]; then
  bash tools/profile.sh "$tag" > "$out/profile.log" 2>&1
  python tools/summarize_prof.py "gpurun_out/prof_$tag" "gpurun_out/$tag/summary" >> "$out/profile.log" 2>&1
  tail -n 30 "$out/profile.log" | cut -c1-600
fi
