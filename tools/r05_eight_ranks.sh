#!/bin/bash
# the N = 8 launch rehearsed on ONE GPU (gloo; every rank drives device 0): all eight ranks' pinned arenas, rings,
# host threads and streams live at once.  FUNCTIONAL, not a measurement.  -> gpurun_out/<tag>/eight_ranks.json
set -u
tag=${1:-r05e}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s.%N)
VAD_BENCH_SHARE_GPU=1 timeout 1200 python bench.py --gpus 8 --steps 40 --corpus-passes 6 --no-cpu-baseline > $out/eight.log 2> $out/eight.err; rc=$?
t1=$(date +%s.%N)
echo "eight ranks rc=$rc wall $(python -c "print(round($t1-$t0,1))") s"; tail -5 $out/eight.err
python - $out/eight.log $out/eight_ranks.json $rc $(python -c "print(round($t1-$t0,1))") <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
d = json.loads(lines[-1]) if lines else {}
oc = d.get("other_configs", {})
rec = {"what": "VAD_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --steps 40 --corpus-passes 6: 8 ranks (torch.distributed.run, gloo) on ONE MI355X; functional rehearsal of the N = 8 launch, NOT a measurement",
       "rc": int(sys.argv[3]), "wall_s": float(sys.argv[4]), "json_lines_from_rank0": len(lines), "n_gpus": d.get("n_gpus"), "data": d.get("data"),
       "per_rank": d.get("per_rank"), "node_totals": d.get("node_totals"),
       "legs": {k: {"value": v.get("value"), "sharding": v.get("sharding"), "error": v.get("error")} for k, v in oc.items()},
       "c2": {"value": d.get("value"), "parity": (d.get("parity") or {}).get("ok")},
       "corpus": {k: oc.get("corpus", {}).get(k) for k in ("recordings_per_gpu", "audio_hours_all_gpus", "wall_s", "host_threads_per_rank", "numa_node_bound", "parity_sample")},
       "corpus_gather": {k: (v.get("segments_found_rank0"), v.get("segments_gathered_all_ranks")) for k, v in (oc.get("corpus", {}).get("legs") or {}).items()}}
json.dump(rec, open(sys.argv[2], "w"), indent=1)
print(json.dumps(rec)[:1500])
PY
