// activations.hpp -- the LSTM cell's sigmoid / tanh as the recurrent kernels evaluate them, in ONE place so that the
// kernels and the accuracy probe (vad_debug_activation -> tests/test_gpu_parity.py::test_activation_accuracy, which
// writes the measured ulp / absolute errors to gpurun_out/ and pins bounds on them) cannot drift apart.
// (reference: aten::lstm_cell's sigmoid / tanh, JIT!/torch/nn/modules/rnn.py:69; head sigmoid,
//  JIT!/torch/nn/modules/container/___torch_mangle_7.py:10-19.)
#pragma once
#include <hip/hip_runtime.h>

namespace vad {

__device__ __forceinline__ float sigmoid_f(float x) {
    // 1 / (1 + e^-x), e^-x = 2^(-x log2 e); v_exp_f32 / v_rcp_f32 are 1-ulp ops
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_f(float x) {
    // tanh x = 2 sigmoid(2x) - 1: exact to ~1e-7 ABSOLUTE (the relative error grows towards 0, where the result
    // itself vanishes; what the cell needs is absolute accuracy: it multiplies a gate in [0, 1])
    return fmaf(2.0f, sigmoid_f(2.0f * x), -1.0f);
}

}  // namespace vad
