#!/bin/bash
# What N = 8 does to the HOST side of a rank, measured on one GPU: the corpus leg with the CPU budget one of eight
# ranks gets -- 2 CPUs of the GPU's NUMA node (taskset), 2 helper threads, LOCAL_WORLD_SIZE=8 -- beside the same leg with the whole
# box.  gpurun --timeout 900 -- 'bash tools/r04_n8_host.sh <tag>'
set -u
tag=${1:-r04a}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
{
  echo "nproc: $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
  lscpu | grep -E "Model name|Socket|NUMA|^CPU\(s\)|Thread"
  for c in /sys/class/drm/card*/device; do
    [ -e $c/numa_node ] && echo "$c numa_node=$(cat $c/numa_node) local_cpulist=$(cat $c/local_cpulist 2>/dev/null)"
  done
  echo "affinity: $(taskset -pc $$)"
} > $out/box.txt 2>&1
cat $out/box.txt
# two CPUs the process may use, on the GPU's NUMA node when that is known
cpus=$(python - <<'PY'
import glob, os
allowed = sorted(os.sched_getaffinity(0))
local = None
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        node = int(open(d + "/numa_node").read())
        lst = open(d + "/local_cpulist").read().strip()
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
pick = (local or allowed)[:2]
print(",".join(map(str, pick)))
PY
)
echo "two CPUs of one rank: $cpus" | tee -a $out/box.txt
run() {   # name, then the command prefix
  name=$1; shift
  ( time "$@" python bench.py --config corpus --no-cpu-baseline ${BENCH_ARGS:-} ) > $out/$name.log 2> $out/$name.err
  echo "$name rc=$?"
  python - $out/$name.log <<'PY'
import json, sys
ls = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not ls:
    print("no line"); sys.exit()
d = json.loads(ls[-1])
print(" host_threads_per_rank", d["config"].get("host_threads_per_rank"), "numa", d["config"].get("numa_node_bound"), "parity", d.get("parity_sample_max_abs_dp"))
for k, v in d["legs"].items():
    print(" ", k, {a: v[a] for a in ("value", "wall_s", "h2d_GBps_while_copying", "host_upload_call_ms", "host_stage_ms", "host_segmenter_ms", "fraction_of_pcie_ceiling", "buckets")})
PY
}
run corpus_2cpu taskset -c $cpus env SILERO_VAD_AMD_HOST_THREADS=2 LOCAL_WORLD_SIZE=8
run corpus_1cpu taskset -c ${cpus%%,*} env SILERO_VAD_AMD_HOST_THREADS=1 LOCAL_WORLD_SIZE=8
run corpus_allcpu env
