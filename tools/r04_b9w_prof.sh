cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-b9g_512_fft160 b9g_344_fft160}; do
rm -rf /tmp/prof_$v
SILERO_VAD_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/lib_$v.so VAD_B9_TIME_SR=16000 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/b9_time.py bf16x9_wide > /tmp/prof_$v.log 2>&1
tail -2 /tmp/prof_$v.log
f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
echo "== $v"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("fft_mags", "b9g", "rec_kernel")):
        print(r["Name"][:60], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min", float(r["MinNs"]) / 1e3, "max", float(r["MaxNs"]) / 1e3)
PY
done
