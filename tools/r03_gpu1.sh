#!/bin/bash
# round 3, first GPU session: parity with the skewed recurrence, the pipes2 micro-benchmark, recurrence A/B timing
set -u
out=gpurun_out/r03a; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread -p no:cacheprovider > $out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -n 8 $out/pytest_gpu.log
timeout 300 build/pipes2 > $out/pipes2.log 2>&1; echo "pipes2 rc=$?"; cat $out/pipes2.log
timeout 600 python tools/variants.py run base rec_inphase rec_noprio > $out/variants.log 2>&1; cat $out/variants.log | cut -c1-400
cp gpurun_out/variants.json $out/variants_rec.json
