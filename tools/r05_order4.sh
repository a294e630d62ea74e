#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py > /dev/null 2>&1
for order in "c2 8k stream stream_host corpus37 stream_8k stream_host_8k" "c2 8k stream stream_host stream_8k stream_host_8k" "corpus37 stream_8k stream_host_8k" "corpus37 stream_host_8k" "corpus37 sleep2 stream_host_8k" "corpus37 gc stream_host_8k"; do
  echo "== $order"; python tools/leg_order_diag.py $order 2>&1 | grep "stream_host" | cut -c1-330
done
