for q in 4 8 16 4 8 16; do
export GPU_MAX_HW_QUEUES=$q
VAD_BENCH_CORPUS_UPLOAD=gather python bench.py --config corpus --no-cpu-baseline --corpus-main-only --no-parity --corpus-passes 8 > gpurun_out/corpus_g_$q.log 2>gpurun_out/corpus_g_$q.err || tail -5 gpurun_out/corpus_g_$q.err
python - gpurun_out/corpus_g_$q.log $q <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
v=d["legs"]["main"]; print("corpus gather, GPU_MAX_HW_QUEUES", sys.argv[2], {a:v[a] for a in ("value","wall_s","h2d_GBps_while_copying","host_upload_call_ms","buckets")}, "of 55.5M:", round(v["value"]/55.5e6,3))
PY
done
