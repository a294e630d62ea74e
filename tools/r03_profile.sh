#!/bin/bash
# rocprofv3 passes of the round-3 product: gpurun --timeout 1500 -- 'bash tools/r03_profile.sh r03f'
set -u
tag=${1:-r03f}
bash tools/profile.sh ${tag}_fp32 > gpurun_out/${tag}_fp32.log 2>&1
bash tools/profile.sh ${tag}_8k --config 8k > gpurun_out/${tag}_8k.log 2>&1
bash tools/profile.sh ${tag}_stream --config stream --steps 200 > gpurun_out/${tag}_stream.log 2>&1
mkdir -p gpurun_out/${tag}_sum
python tools/summarize_prof.py gpurun_out/prof_${tag}_fp32 gpurun_out/${tag}_sum/${tag}_fp32 16000 4096 256 > /dev/null
python tools/summarize_prof.py gpurun_out/prof_${tag}_8k gpurun_out/${tag}_sum/${tag}_8k 8000 4096 256 > /dev/null
python tools/summarize_prof.py gpurun_out/prof_${tag}_stream gpurun_out/${tag}_sum/${tag}_stream 16000 8192 1 > /dev/null
cp gpurun_out/prof_${tag}_fp32/trace/*/*kernel_stats.csv gpurun_out/${tag}_sum/${tag}_fp32_kernel_stats.csv 2>/dev/null || find gpurun_out/prof_${tag}_fp32/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_sum/${tag}_fp32_kernel_stats.csv \;
head -30 gpurun_out/${tag}_sum/${tag}_fp32_summary.md; head -22 gpurun_out/${tag}_sum/${tag}_stream_summary.md
# keep the scratch small: the raw rocprof output is not merged back
rm -rf gpurun_out/prof_${tag}_fp32 gpurun_out/prof_${tag}_8k gpurun_out/prof_${tag}_stream
