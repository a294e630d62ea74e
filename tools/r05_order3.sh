#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py > /dev/null 2>&1
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "streams_overlap or ragged or refill or corpus" -p no:cacheprovider 2>&1 | tail -3
for order in "corpus37 corpus3" "corpus37 sleep3 corpus3" "corpus37 c2 corpus3" "stream stream_host corpus37 corpus3 corpus3"; do
  echo "== $order"; python tools/leg_order_diag.py $order 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | cut -c1-300
done
