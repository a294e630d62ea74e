# does another kernel's VALU work run beside the engine's kernels?  (profiles/r04c section 2 also lists the retired wide kernel: commit e27b60d)
python tools/coexec_diag.py bf16x9 2>&1 | grep -v amdgpu
python tools/coexec_diag.py fp32 2>&1 | grep -v amdgpu
