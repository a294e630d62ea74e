#!/bin/bash
# About one MI355X box in ten runs straight-line kernels that do not fit the instruction cache (the F(2,3) frontend,
# build/variants/lib_wino2.so = option enc0=winograd2) 1.4 ms slower per launch (profiles/r02i_pmc).  Cheap on a normal box
# (two short benches); on a slow one it collects the comparison: the product's loop-structured frontend on the same box,
# instruction-fetch counters of both, clocks, the earlier ablations.
export TMPDIR=/tmp
out=gpurun_out/hunt_$(date +%s); mkdir -p $out
kms() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms']['front'])"; }
old=$(VAD_BENCH_ENC0=winograd2 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | kms)
new=$(python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | kms)
echo "front_ms straight-line(F23) $old  product(F43 loops) $new  id $(rocm-smi --showuniqueid 2>/dev/null | grep -o '0x[0-9a-f]*' | head -1)" | tee $out/summary.txt
slow=$(python -c "print(1 if float('$old') > 5.5 else 0)")
if [ "$slow" = "1" ] || [ -n "${HUNT_FORCE_PMC:-}" ]; then
  [ "$slow" = "1" ] && echo SLOW BOX | tee -a $out/summary.txt
  python bench.py --no-cpu-baseline --no-extras --steps 100 --config 8k 2>/dev/null | kms | sed 's/^/8k product front_ms /' | tee -a $out/summary.txt
  VAD_BENCH_ENC0=winograd2 python bench.py --no-cpu-baseline --no-extras --steps 100 --config 8k 2>/dev/null | kms | sed 's/^/8k straight-line front_ms /' | tee -a $out/summary.txt
  echo "== product" | tee -a $out/summary.txt
  bash tools/pmc_front.sh $out/pmc_product | tee -a $out/summary.txt
  echo "== straight-line" | tee -a $out/summary.txt
  VAD_BENCH_ENC0=winograd2 bash tools/pmc_front.sh $out/pmc_straight | tee -a $out/summary.txt
fi
