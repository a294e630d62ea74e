#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py > /dev/null 2>&1
for order in "stream_host_8k" "stream_8k stream_host_8k" "corpus3 stream_host_8k" "c2 stream_host_8k" "stream_host stream_host_8k" "--parity stream_8k stream_host_8k" "corpus37 corpus3"; do
  echo "== $order"; python tools/leg_order_diag.py $order 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | cut -c1-400
done
python tools/state_term_diag.py 2>&1 | grep -v amdgpu.ids | tail -60
