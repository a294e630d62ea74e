#!/bin/bash
# evidence run (gpurun --timeout 1800 -- "bash tools/gpu_evidence.sh <tag>"): tests, the default bench line, ramped rocprofv3 trace + PMC passes (fp32 default and f16x3), summaries
set -u
tag=${1:-x}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python __graft_entry__.py > "$out/build.log" 2>&1
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -p no:cacheprovider > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -n 4 "$out/pytest_gpu.log"
( time timeout 1200 python bench.py ) > "$out/bench_c2.log" 2>&1; grep '"metric"' "$out/bench_c2.log" | cut -c1-600
bash tools/profile.sh "${tag}_fp32" > "$out/profile_fp32.log" 2>&1
python tools/summarize_prof.py "gpurun_out/prof_${tag}_fp32" "$out/fp32" >> "$out/profile_fp32.log" 2>&1
bash tools/profile.sh "${tag}_f16x3" --precision f16x3 > "$out/profile_f16x3.log" 2>&1
python tools/summarize_prof.py "gpurun_out/prof_${tag}_f16x3" "$out/f16x3" >> "$out/profile_f16x3.log" 2>&1
bash tools/profile.sh "${tag}_8k" --config 8k > "$out/profile_8k.log" 2>&1
python tools/summarize_prof.py "gpurun_out/prof_${tag}_8k" "$out/8k" 8000 4096 256 >> "$out/profile_8k.log" 2>&1
head -20 "$out/fp32_summary.md"
cp gpurun_out/foreign_load_*.json gpurun_out/activation_accuracy.json "$out/" 2>/dev/null
# keep the merged payload small: the raw traces are not needed once summarised
find gpurun_out/prof_${tag}_* -name "*kernel_trace.csv" -size +2M -delete
