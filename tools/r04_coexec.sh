python tools/coexec_diag.py bf16x9_wide 2>&1 | grep -v amdgpu
python tools/coexec_diag.py bf16x9_narrow 2>&1 | grep -v amdgpu
python tools/coexec_diag.py fp32 2>&1 | grep -v amdgpu
SILERO_VAD_AMD_LIB=build/variants/lib_abl_w_nofft_noload.so python tools/coexec_diag.py bf16x9_wide 2>&1 | grep -v amdgpu
