#!/bin/bash
# Broad SQ/LDS/TA counter sweep of the bench command (GPU box): gpurun_out/pmc_<name>/passN/
set -u
name=$1; shift
out=$PWD/gpurun_out/pmc_$name
mkdir -p "$out"; export TMPDIR=/tmp
args="--no-cpu-baseline --no-extras --steps 6 --warmup 1 $*"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line -d "$out/pass$i" -o p --output-format csv -- python bench.py $args > "$out/pass$i.log" 2>&1
done <<'LIST'
SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS
SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
LIST
python - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "front" if ("front_kernel" in k or "front_wino_kernel" in k or "front_f43_kernel" in k) else "rec" if "rec_kernel" in k else None
        if k: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:6s} {c:32s} {sum(v)/len(v):.6g}")
PY
