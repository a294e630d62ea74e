#!/bin/bash
# time the pair-form bf16 x 9 frontend of each variant library (GPU box)
for v in "$@"; do
  SILERO_VAD_AMD_LIB=build/variants/lib_$v.so VAD_B9_TIME_SR=${SR:-16000} python tools/b9_time.py bf16x9_pair 2>&1 | grep -v amdgpu.ids
done
