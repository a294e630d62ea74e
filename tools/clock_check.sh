#!/bin/bash
# run-to-run / precision-order variability of the bench main leg, with rocm-smi clock + power snapshots (GPU box)
export TMPDIR=/tmp
for run in f16x3 fp32 f16x3 f16x3 fp32; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -3 | tr '\n' ' '
  echo
  python bench.py --precision $run --no-cpu-baseline --no-extras --steps 100 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$run', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"
done
