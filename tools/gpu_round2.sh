#!/bin/bash
# gpu_round.sh + the fp32 ablation variants (tools/variants.py run ...)
set -u
tag=${1:-x}
bash tools/gpu_round.sh "$tag" prof
timeout 900 python tools/variants.py run base abl_nobar abl_nofft abl_noload abl_noring abl_mfma_only slot8 slot16 > "gpurun_out/$tag/variants.log" 2>&1
cp gpurun_out/variants.json "gpurun_out/$tag/variants_fp32.json" 2>/dev/null
tail -n 12 "gpurun_out/$tag/variants.log" | cut -c1-300
