#!/bin/bash
# gpurun -- 'bash tools/r06_refill_window_sizes.sh "<window bytes> ..." "<group> ..." "<ramp> ..."': the refill legs of the corpus bench per
# window size / copy group / start ramp
mkdir -p gpurun_out/r06S
for w in ${1:-1073741824}; do for g in ${2:-8}; do for r in ${3:-8}; do
  SILERO_VAD_AMD_REFILL_RAMP=$r SILERO_VAD_AMD_REFILL_GROUP=$g SILERO_VAD_AMD_REFILL_WINDOW=$w VAD_BENCH_ONLY_REFILL=1 timeout 300 python bench.py --config corpus --no-cpu-baseline --no-parity > gpurun_out/r06S/w$w.g$g.r$r.log 2> gpurun_out/r06S/w$w.g$g.r$r.err || tail -5 gpurun_out/r06S/w$w.g$g.r$r.err
  python - <<P
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for k in ('pinned_refill_gather','pinned_refill_window'):
    l=d['legs'][k]; print("window $w group $g ramp $r", k, l['value'], l['wall_s'], l.get('fraction_of_pcie_ceiling'), l.get('window_buffers'), l.get('first_result_at'), l.get('first_result_after_slabs'))
P
done; done; done
