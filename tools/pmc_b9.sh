#!/bin/bash
# SQ / SQC counter passes over the fp32 and the bf16 x 9 throughput frontends at C2 (run ON the GPU box): tools/pmc_b9.sh <outdir>
# One rocprofv3 --pmc run per counter group (no trace domains); per-dispatch means of both kernels side by side.
export TMPDIR=/tmp
out=$1; mkdir -p $out
i=0
while read -r group; do
  i=$((i+1))
  VAD_B9_TIME_SR=${SR:-16000} rocprofv3 --pmc $group -d $out/g$i -o g$i --output-format csv -- python tools/b9_time.py fp32 bf16x9 > $out/g$i.log 2>&1
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH_LEVEL
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT
SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM
GROUPS
python - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for key in ("front_b9_kernel", "front_f43_kernel"):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]][key].append(float(r["Counter_Value"]))
res = {k: {kk: sum(v) / len(v) for kk, v in d.items()} for k, d in sorted(acc.items())}
json.dump(res, open(out + "/pmc_b9.json", "w"), indent=1)
for k, d in res.items():
    print(f"pmc {k:32s} f43 {d.get('front_f43_kernel', 0):.5g}  b9 {d.get('front_b9_kernel', 0):.5g}")
PY
rm -rf $out/g*/
