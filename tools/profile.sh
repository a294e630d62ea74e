#!/bin/bash
# rocprofv3 passes over the headline bench command (run ON the GPU box, from the repo root):
#   tools/profile.sh <name> [bench args...]   ->  gpurun_out/prof_<name>/{trace,fetch,write,sq}/ + logs
# Counters are collected in their own runs, never together with kernel-trace/stats (gpurun refuses the
# combination), and FETCH_SIZE / WRITE_SIZE need separate passes (TCC counter budget).
# Afterwards:  python tools/summarize_prof.py gpurun_out/prof_<name> profiles/<name>
set -u
name=$1; shift
out=$PWD/gpurun_out/prof_$name
mkdir -p "$out"
export TMPDIR=/tmp
# the bench's own protocol: 40 untimed clock-ramp steps, 2 warm-up, then 20 timed (= profiled) steps; the summary
# reports median / min / mean over the LAST 20 dispatches of each kernel (tools/summarize_prof.py)
args="--no-cpu-baseline --no-extras --steps 20 --warmup 2 $*"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace --output-format csv -- python bench.py $args > "$out/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- python bench.py $args > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- python bench.py $args > "$out/write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$out/sq" -o sq --output-format csv -- python bench.py $args > "$out/sq.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d "$out/sq2" -o sq2 --output-format csv -- python bench.py $args > "$out/sq2.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 -d "$out/sq3" -o sq3 --output-format csv -- python bench.py $args > "$out/sq3.log" 2>&1
grep -h '"metric"' "$out"/*.log | cut -c1-400
