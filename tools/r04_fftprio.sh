for v in base gemmprio3; do SILERO_VAD_AMD_LIB=build/variants/lib_$v.so python tools/b9_time.py fp32 2>&1 | grep -v amdgpu; done
SILERO_VAD_AMD_LIB=build/variants/lib_trace_gemmprio3.so python tools/trace_f43.py 16000 2>&1 | grep -E "waves traced|FFT|encoder|W_ih|sum of"
