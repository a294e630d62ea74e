#!/bin/bash
# gpurun -- 'bash tools/r06_stage_sync_ab.sh [reps]': the corpus bench's bucket routes (6 passes) with the staged route's H2D copies waiting
# for their device buffer's previous reader on the device (SYNC=0, the tree before) or on the host behind the staging (SYNC=1), interleaved
mkdir -p gpurun_out/r06U
for rep in $(seq 1 ${1:-3}); do for sync in 0 1; do
  SILERO_VAD_AMD_STAGE_SYNC=$sync VAD_BENCH_SKIP_REFILL=1 timeout 300 python bench.py --config corpus --corpus-passes 6 --no-cpu-baseline --no-parity > gpurun_out/r06U/s$sync.log 2> gpurun_out/r06U/s$sync.err || tail -5 gpurun_out/r06U/s$sync.err | cut -c1-300
  python - <<P
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print("rep $rep sync $sync", " | ".join("%s %.4f s %.3f (%.1f GB/s)" % (k, l['wall_s'], l.get('fraction_of_pcie_ceiling') or 0, l.get('h2d_GBps_while_copying') or 0) for k, l in d['legs'].items()))
P
done; done
