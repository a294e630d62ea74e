#!/bin/bash
# Sample sclk/power while a bench variant runs (GPU box).  usage: tools/clock_probe.sh <lib.so|-> [bench args]
lib=$1; shift
[ "$lib" != "-" ] && export SILERO_VAD_AMD_LIB=$lib
python bench.py --no-cpu-baseline --steps 2500 --warmup 3 "$@" > /tmp/bench_probe.json 2>/dev/null &
pid=$!
sleep 7
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | grep -oE "\([0-9]+Mhz\)|[0-9]+\.[0-9]+$" | tr '\n' ' '
  echo
  sleep 0.5
done
wait $pid
python -c "import json;d=json.load(open('/tmp/bench_probe.json'));print(d['ms_per_step'], d['kernel_ms'])"
