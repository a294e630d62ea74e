#!/bin/bash
# SQ / SQC counter passes over the 16 kHz frontend kernel (run ON the GPU box): tools/pmc_front.sh <outdir>
# One rocprofv3 --pmc run per counter group (no trace domains); prints and stores the per-dispatch means of the
# frontend kernel.  Used by tools/slow_box_hunt.sh to compare a box whose FFT phases are slow with a normal one.
export TMPDIR=/tmp
out=$1; mkdir -p $out
args="--no-cpu-baseline --no-extras --steps 6 --warmup 2"
i=0
while read -r group; do
  i=$((i+1))
  rocprofv3 --pmc $group -d $out/g$i -o g$i --output-format csv -- python bench.py $args > $out/g$i.log 2>&1
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_CYCLES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL
SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL
SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU
GROUPS
python - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "front" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
json.dump(res, open(out + "/pmc_front.json", "w"), indent=1)
for k, v in res.items():
    print(f"pmc {k} {v:.4g}")
PY
