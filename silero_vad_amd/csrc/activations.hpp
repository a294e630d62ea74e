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

// The head's ReLU.  torch.relu is clamp_min(0) and PROPAGATES NaN (v_max_f32 / fmaxf would return the other operand): a stream
// whose carried (h, c) is NaN -- one NaN / Inf sample is enough, see fft_wave.hpp "non-finite input" -- must read NaN until it is
// reset, exactly as the reference does (JIT!/torch/nn/modules/container/___torch_mangle_7.py:10-19; goldens: tests/golden/
// make_golden.py protocol "nonfinite").
//   relu_f:  the plain form (compare + select).
//   relu2_f: 2 * relu(x) = x + |x| -- ONE v_add_f32 (the |.| is a source modifier), the price of the v_max_f32 it replaces, NaN in ->
// NaN out; x + |x| is exactly 2x for x > 0 and exactly +0 otherwise.  The recurrent kernels use it with HALVED head weights:
// fmaf(0.5 w, 2 relu(h), acc) has the same product and therefore the same bits as fmaf(w, relu(h), acc) (scaling by two is exact).
__device__ __forceinline__ float relu_f(float x) { return x <= 0.0f ? 0.0f : x; }
__device__ __forceinline__ float relu2_f(float x) { return x + __builtin_fabsf(x); }

}  // namespace vad
