#!/bin/bash
# One GPU session: parity suite + the default bench line (+ optional extra commands).  gpurun --timeout 1500 -- 'bash tools/r06_round.sh <tag> [bench args]'
set -u
tag=${1:-r06w}; shift || true
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
if [ -z "${SKIP_TESTS:-}" ]; then
timeout ${PYTEST_TIMEOUT:-1200} python -m pytest tests -m gpu -q --timeout 300 --timeout-method=thread -p no:cacheprovider ${PYTEST_ARGS:-} ${PYTEST_K:+-k "$PYTEST_K"} > $out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -n ${PYTEST_TAIL:-30} $out/pytest_gpu.log | cut -c1-400
fi
if [ -z "${SKIP_BENCH:-}" ]; then
( time timeout 900 python bench.py "$@" ) > $out/bench.log 2> $out/bench.err; echo "bench rc=$?"
grep -v '^bench detail: ' $out/bench.err | tail -n 8
grep '"metric"' $out/bench.log > $out/bench_line.json
cp gpurun_out/bench_detail.json $out/bench_detail.json 2>/dev/null
echo "stdout bytes: $(wc -c < $out/bench.log)"; cat $out/bench_line.json
fi
if [ -n "${EXTRA:-}" ]; then
bash -c "$EXTRA" > $out/extra.log 2>&1; echo "extra rc=$?"; tail -n ${EXTRA_TAIL:-60} $out/extra.log | cut -c1-300
fi
