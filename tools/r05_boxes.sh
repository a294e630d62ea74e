#!/bin/bash
# one lease: is it a 'bad box' (slow hipFree/hipMalloc of GBs)?  the corpus routes with the allocation-free run, and the alloc cost itself
export TMPDIR=/tmp
python __graft_entry__.py > /dev/null 2>&1
python - <<'PY'
import time, torch
torch.cuda.init(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); x = torch.empty(4 << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    del x; torch.cuda.empty_cache(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1)))
print("hipMalloc / hipFree of 4 GiB, ms:", ts)
PY
python bench.py --config corpus --corpus-passes 3 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print({k:(v['fraction_of_pcie_ceiling'], v['host_ms'].get('reserve')) for k,v in d['legs'].items()})"
rocm-smi --showuniqueid 2>/dev/null | grep -o '0x[0-9a-f]*' | head -1
