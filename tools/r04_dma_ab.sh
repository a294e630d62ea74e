for k in 2 3; do
echo "== dma streams $k"
SILERO_VAD_AMD_DMA_STREAMS=$k python tools/ingest_diag.py 2>&1 | grep -E "dma per row|beside dma"
done
for k in 2 3; do
SILERO_VAD_AMD_DMA_STREAMS=$k VAD_BENCH_CORPUS_UPLOAD=dma python bench.py --config corpus --no-cpu-baseline --corpus-main-only --no-parity --corpus-passes 6 > gpurun_out/corpus_dma_$k.log 2>gpurun_out/corpus_dma_$k.err || tail -5 gpurun_out/corpus_dma_$k.err
python - gpurun_out/corpus_dma_$k.log $k <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
v=d["legs"]["main"]; print("corpus dma streams", sys.argv[2], {a:v[a] for a in ("value","wall_s","h2d_GBps_while_copying","host_upload_call_ms","fraction_of_pcie_ceiling","buckets")})
PY
done
