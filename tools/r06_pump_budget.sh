#!/bin/bash
# The pump under one rank's CPU budget, beside the whole box.  gpurun --timeout 900 -- 'bash tools/r06_pump_budget.sh <tag>'
set -u
tag=${1:-r06p}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
{ echo "nproc: $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; lscpu | grep -E "Model name|Socket|NUMA|^CPU\(s\)|Thread"
  for c in /sys/class/drm/card*/device; do [ -e $c/numa_node ] && echo "$c numa_node=$(cat $c/numa_node) local_cpulist=$(cat $c/local_cpulist 2>/dev/null)"; done; } > $out/box.txt 2>&1
pick() { python - "$1" <<'PY'
import glob, os, sys
k = int(sys.argv[1])
allowed = sorted(os.sched_getaffinity(0))
local = None
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        node = int(open(d + "/numa_node").read()); lst = open(d + "/local_cpulist").read().strip()
    except (OSError, ValueError):
        continue
    if node < 0 or not lst:
        continue
    s = set()
    for part in lst.split(","):
        a, _, b = part.partition("-")
        s.update(range(int(a), int(b or a) + 1))
    local = [c for c in allowed if c in s]
    if local:
        break
pool = local or allowed
skip = 8 if len(pool) >= 8 + k else 0          # (not the first CPUs of the node: the kernel's housekeeping lands there)
print(",".join(map(str, pool[skip:skip + k])))
PY
}
echo "== whole box, 8 source threads";            PUMP_FILL_THREADS=8 python tools/pump_budget.py 2>/dev/null | tee $out/whole_box.json
echo "== 2 CPUs ($(pick 2)), LOCAL_WORLD_SIZE=8, 1 source thread"
LOCAL_WORLD_SIZE=8 PUMP_FILL_THREADS=1 taskset -c $(pick 2) python tools/pump_budget.py 2>/dev/null | tee $out/two_cpus.json
echo "== 3 CPUs ($(pick 3)), LOCAL_WORLD_SIZE=8, 1 source thread"
LOCAL_WORLD_SIZE=8 PUMP_FILL_THREADS=1 taskset -c $(pick 3) python tools/pump_budget.py 2>/dev/null | tee $out/three_cpus.json
echo "== 1 CPU ($(pick 1)), LOCAL_WORLD_SIZE=8, 1 source thread"
LOCAL_WORLD_SIZE=8 PUMP_FILL_THREADS=1 taskset -c $(pick 1) python tools/pump_budget.py 2>/dev/null | tee $out/one_cpu.json
echo "== whole box, 1 source thread"
PUMP_FILL_THREADS=1 python tools/pump_budget.py 2>/dev/null | tee $out/whole_box_one_source.json
