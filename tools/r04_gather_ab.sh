for w in 96 48 32 24; do
VAD_GATHER_WAVES_RT=$w VAD_BENCH_CORPUS_UPLOAD=gather python bench.py --config corpus --no-cpu-baseline --corpus-main-only --no-parity --corpus-passes 8 > gpurun_out/corpus_g_$w.log 2>gpurun_out/corpus_g_$w.err || tail -5 gpurun_out/corpus_g_$w.err
python - gpurun_out/corpus_g_$w.log $w <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
v=d["legs"]["main"]; print("corpus gather waves", sys.argv[2], {a:v[a] for a in ("value","wall_s","h2d_GBps_while_copying","host_upload_call_ms","buckets")}, "of 55.5M:", round(v["value"]/55.5e6,3))
PY
done
