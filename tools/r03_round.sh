#!/bin/bash
# One GPU session: parity suite + the default bench line.  gpurun --timeout 1500 -- 'bash tools/r03_round.sh <tag> [bench args]'
set -u
tag=${1:-r03x}; shift || true
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
timeout 800 python -m pytest tests -m gpu -q --timeout 300 --timeout-method=thread -p no:cacheprovider > $out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -n 25 $out/pytest_gpu.log | cut -c1-300
( time timeout 900 python bench.py "$@" ) > $out/bench.log 2> $out/bench.err; echo "bench rc=$?"
tail -n 5 $out/bench.err
grep '"metric"' $out/bench.log > $out/bench_line.json
python - "$out/bench_line.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
def show(k, v, ind=0):
    if isinstance(v, dict):
        print(" " * ind + str(k) + ":")
        for a, b in v.items():
            show(a, b, ind + 2)
    else:
        s = str(v)
        print(" " * ind + f"{k}: {s[:160]}")
for k in ("value", "ms_per_step", "kernel_ms"):
    show(k, d.get(k))
show("roofline.frac", d["roofline"]["frac"]); show("rec", d["roofline"].get("rec_kernel"))
show("other_configs", d.get("other_configs"))
show("cpu_baseline.value", (d.get("cpu_baseline") or {}).get("value"))
PY
