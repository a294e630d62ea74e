set -u
mkdir -p gpurun_out/r02h
python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -15
python bench.py --no-cpu-baseline --no-extras --steps 100 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('wino', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"
