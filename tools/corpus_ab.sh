for mode in ${MODES:-window gather stage}; do
VAD_BENCH_CORPUS_UPLOAD=$mode python bench.py --config corpus --no-cpu-baseline --corpus-main-only --no-parity > gpurun_out/corpus_$mode.log 2>gpurun_out/corpus_$mode.err || tail -5 gpurun_out/corpus_$mode.err
python - gpurun_out/corpus_$mode.log <<PY
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
v=d["legs"]["main"]; print(v["upload"], v["source"], {a:v[a] for a in ("value","wall_s","h2d_GBps_while_copying","host_upload_call_ms","host_stage_ms","host_segmenter_ms","fraction_of_pcie_ceiling","buckets")})
PY
done
