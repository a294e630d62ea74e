#!/bin/bash
# gpurun -- "bash tools/r06_refill_ramp_ab.sh": the refill route's window feed with / without the staggered start (RAMP) and the dependency-free window DMAs (SLACK), three interleaved repetitions
mkdir -p gpurun_out/r06S
for rep in 1 2 3; do for cfg in "1 1" "8 2" "8 1"; do set -- $cfg
  SILERO_VAD_AMD_REFILL_RAMP=$1 SILERO_VAD_AMD_REFILL_SLACK=$2 VAD_BENCH_ONLY_REFILL=1 timeout 300 python bench.py --config corpus --no-cpu-baseline --no-parity > gpurun_out/r06S/x.log 2> gpurun_out/r06S/x.err || tail -5 gpurun_out/r06S/x.err
  python - <<P
import json
d=json.load(open('gpurun_out/bench_detail.json'))
l=d['legs']['pinned_refill_window']; g=d['legs']['pinned_refill_gather']
print("rep $rep ramp $1 slack $2 window", l['wall_s'], l.get('fraction_of_pcie_ceiling'), l.get('window_buffers'), l.get('first_result_at'), "| gather", g['wall_s'], g.get('fraction_of_pcie_ceiling'), "numa", d['config'].get('numa_node_bound'))
P
done; done
hostname; cat /proc/cpuinfo | grep -m1 "model name"; nproc
