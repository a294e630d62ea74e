#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py > /dev/null 2>&1
for order in "corpus37 corpus3" "corpus37 sleep3 corpus3" "corpus37 c2 corpus3" "corpus37 gc corpus3" "corpus12 corpus3"; do
  echo "== $order"; python tools/leg_order_diag.py $order 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | cut -c1-300
done
