#!/bin/bash
# Cheap on a normal box (one short bench); on a box whose 16 kHz frontend is slow (> 5.5 ms) it collects what is needed to
# say WHY: issue-pipe ubench, clocks/power under load, scalar-FFT / no-FFT / 8 kHz / f16x3 variants, all in this one call.
export TMPDIR=/tmp
out=gpurun_out/hunt_$(date +%s); mkdir -p $out
line=$(python bench.py --no-cpu-baseline --no-extras --steps 150 2>/dev/null | tail -1)
front=$(echo "$line" | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms']['front'])")
./build/ubench/icache 2>/dev/null | tee -a $out/summary.txt
echo "front_ms $front id $(rocm-smi --showuniqueid 2>/dev/null | grep -o '0x[0-9a-f]*' | head -1)" | tee $out/summary.txt
slow=$(python -c "print(1 if float('$front') > 5.5 else 0)")
if [ "$slow" = "1" ] || [ -n "${HUNT_FORCE_PMC:-}" ]; then
  bash tools/pmc_front.sh $out/pmc | tee -a $out/summary.txt
fi
if [ "$slow" = "1" ]; then
  echo SLOW BOX | tee -a $out/summary.txt
  bash tools/box_check.sh > $out/box_check.log 2>&1
  for v in nopk_front abl_nofft abl_noload abl_mfma_only base; do
    [ -f build/variants/lib_$v.so ] && SILERO_VAD_AMD_LIB=build/variants/lib_$v.so python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', d['kernel_ms'])" | tee -a $out/summary.txt
  done
  python bench.py --no-cpu-baseline --no-extras --steps 100 --config 8k 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('8k', d['kernel_ms'])" | tee -a $out/summary.txt
  python bench.py --no-cpu-baseline --no-extras --steps 100 --precision f16x3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('f16x3', d['kernel_ms'])" | tee -a $out/summary.txt
  grep -E "mode|sclk|Unique|Power" $out/box_check.log | cut -c1-160 | tee -a $out/summary.txt
fi
