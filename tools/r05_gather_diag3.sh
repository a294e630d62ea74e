#!/bin/bash
set -u
tag=${1:-r05i}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: (v["fraction_of_pcie_ceiling"], v["wall_s"], v["host_upload_call_ms"]) for k, v in d["legs"].items()})
PY
}
run() { name=$1; shift; env "$@" > $out/$name.log 2> $out/$name.err || tail -3 $out/$name.err; echo "$name:"; show $out/$name.log; }
run noparity X=1 python bench.py --config corpus --corpus-passes 3 --no-cpu-baseline --no-parity
run parity X=1 python bench.py --config corpus --corpus-passes 3 --no-cpu-baseline
run parity_omp_passive OMP_WAIT_POLICY=passive python bench.py --config corpus --corpus-passes 3 --no-cpu-baseline
run parity_omp1 OMP_NUM_THREADS=1 python bench.py --config corpus --corpus-passes 3 --no-cpu-baseline
run noparity2 X=1 python bench.py --config corpus --corpus-passes 3 --no-cpu-baseline --no-parity
