#!/bin/bash
# The N = 8 launch rehearsed on ONE GPU (gloo; every rank drives device 0), WITH the cpu_baseline rank 0 now runs at every N: one stdout
# line of at most 8 KB, the full record beside it.  FUNCTIONAL, not a measurement.  gpurun --timeout 1500 -- 'bash tools/r06_eight_ranks.sh <tag>'
set -u
tag=${1:-r06e8}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
t0=$(date +%s.%N)
VAD_BENCH_SHARE_GPU=1 timeout 1400 python bench.py --gpus 8 --steps 40 --corpus-passes 6 > $out/eight.log 2> $out/eight.err; rc=$?
t1=$(date +%s.%N)
echo "eight ranks rc=$rc wall $(python -c "print(round($t1-$t0,1))") s; stdout bytes $(wc -c < $out/eight.log)"; grep -v "^bench detail" $out/eight.err | tail -5
cp gpurun_out/bench_detail.json $out/eight_detail.json 2>/dev/null
python - $out/eight.log $out/eight_detail.json $out/eight_ranks.json $rc $(python -c "print(round($t1-$t0,1))") <<'PY'
import json, sys
raw = open(sys.argv[1]).read()
lines = [l for l in raw.splitlines() if l.startswith('{"metric"')]
d = json.loads(lines[-1]) if lines else {}
full = json.load(open(sys.argv[2])) if lines else {}
rec = {"what": "VAD_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --steps 40 --corpus-passes 6: 8 ranks (torch.distributed.run, gloo) on ONE MI355X; functional rehearsal of the N = 8 launch, NOT a measurement",
       "rc": int(sys.argv[4]), "wall_s": float(sys.argv[5]), "stdout_bytes": len(raw), "json_lines_from_rank0": len(lines), "line_bytes": len(lines[-1]) if lines else 0,
       "line_keys": list(d), "n_gpus": d.get("n_gpus"), "has_cpu_baseline": "cpu_baseline" in d, "has_roofline": "roofline" in d, "node_totals": d.get("node_totals"),
       "legs": d.get("legs"), "per_rank": full.get("per_rank")}
json.dump(rec, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in rec.items() if k != "per_rank"})[:2500])
PY
