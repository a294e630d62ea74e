#!/bin/bash
# the GPU parity suite on the opt-in bf16 x 9 arithmetic (same tests, same bounds).  gpurun --timeout 900 -- 'bash tools/r04_arith.sh <tag>'
tag=${1:-r04e}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py > $out/build.log 2>&1 || tail -20 $out/build.log
SILERO_VAD_AMD_TEST_ARITH=bf16x9 timeout 900 python -m pytest tests -m gpu -q --timeout 300 --timeout-method=thread -p no:cacheprovider -rf > $out/pytest_gpu_bf16x9.log 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu_bf16x9.log
grep -E "^FAILED|passed|failed|rc=" $out/pytest_gpu_bf16x9.log | cut -c1-300
