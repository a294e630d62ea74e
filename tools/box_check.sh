#!/bin/bash
# Which box is this, and is it one of the "slow" ones?  (GPU box; the same binary's 16 kHz frontend has been seen at 5.0 and at
# 6.4 ms on different boxes: VERDICT r1 "investigate the slow box".)  Prints identity / power / clocks, the issue-pipe
# micro-benchmark (MFMA, scalar VALU, packed VALU rates) and the bench's kernel times, then samples clocks under load.
export TMPDIR=/tmp
rocm-smi --showuniqueid --showserial --showvbios --showmaxpower --showpower --showperflevel --showtemp 2>/dev/null | grep -vE "^=|^$" | head -20
./build/ubench/pipes 2>/dev/null | grep -E "mode [0129] "
( python bench.py --no-cpu-baseline --no-extras --steps 3000 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fp32', d['value'], d['kernel_ms'], d['roofline']['frac'])" ) &
sleep 9
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Socket|junction" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; sleep 1.5; done
wait
SILERO_VAD_AMD_LIB=build/variants/lib_abl_nofft.so python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('nofft', d['kernel_ms'])"
