#!/bin/bash
# rocprofv3 passes of the round-6 product: gpurun --timeout 1500 -- 'bash tools/r06_profile.sh r06w'
set -u
tag=${1:-r06w}
bash tools/r03_profile.sh $tag
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_${tag}_stream_host
rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$out" -o trace --output-format csv -- python bench.py --config stream_host --no-cpu-baseline --no-parity --steps 300 > gpurun_out/${tag}_stream_host.log 2>&1
for f in kernel_stats memory_copy_stats; do find "$out" -name "*${f}.csv" -exec cp {} gpurun_out/${tag}_sum/${tag}_stream_host_${f}.csv \; ; done
grep '"metric"' gpurun_out/${tag}_stream_host.log | cut -c1-600
head -6 gpurun_out/${tag}_sum/${tag}_stream_host_kernel_stats.csv | cut -c1-200; head -5 gpurun_out/${tag}_sum/${tag}_stream_host_memory_copy_stats.csv | cut -c1-200
rm -rf "$out"
