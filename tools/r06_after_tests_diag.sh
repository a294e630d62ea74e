#!/bin/bash
# gpurun -- 'bash tools/r06_after_tests_diag.sh': the host-fed stream leg before and after the GPU test suite ran on the box (the round
# script's whole-bench runs behind pytest read 0.87-0.90 of the link where runs without it read 0.97-0.99)
mkdir -p gpurun_out/r06T
leg() { tag=$1
  python bench.py --config stream_host --no-cpu-baseline --no-parity > gpurun_out/r06T/$tag.log 2> gpurun_out/r06T/$tag.err || tail -3 gpurun_out/r06T/$tag.err | cut -c1-300
  python - <<P
import json
d=json.load(open('gpurun_out/bench_detail.json'))
s=d['sustained']; print("$tag", round(d['value']/1e6,1), d['pcie']['fraction_of_pcie_ceiling'], d['pcie']['h2d_GBps_plain_copy'], s['depth'], s.get('timed_passes_s'), s['host_ms_per_tick'], {k:v['wall_ms'] for k,v in s['untimed_depth_trials'].items()})
P
  grep -E "MemFree|AnonHugePages|^Cached|Unevictable|Mlocked" /proc/meminfo | tr -s ' ' | tr '\n' ';'; echo
}
leg before1; leg before2
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r06T/pytest.log 2>&1; tail -1 gpurun_out/r06T/pytest.log
leg after1; leg after2
sleep 20
leg later1
