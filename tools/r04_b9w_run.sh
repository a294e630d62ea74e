python tools/b9w_check.py 2>&1 | grep -v amdgpu.ids
for v in b9g_344_fft128 b9g_344_fft160 b9g_384_fft128 b9g_512_fft160; do
  SILERO_VAD_AMD_LIB=build/variants/lib_$v.so VAD_B9_TIME_SR=16000 python tools/b9_time.py bf16x9_wide 2>&1 | grep -v amdgpu.ids
done
