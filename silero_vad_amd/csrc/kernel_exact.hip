// kernel_exact.hip -- the fix-up pass behind the throughput frontend: the chunks kernel_front_f43.hip listed as holding an exactly
// silent STFT frame beside one that is not get their gate pre-activations from the double-precision evaluation of exact_front.hpp,
// written over their columns of the gx scratch before the recurrence reads it.  (The latency frontend does the same inside its own
// kernel; exact_front.hpp says why the chunks exist and why the bits agree on both routes.)
//
// Shape: a fixed grid of 256-thread workgroups walks the list (count known only on the device: no host round trip); the last
// workgroup to finish zeroes the list's two counters, so the next launch needs no memset.  With an empty list -- every chunk of
// continuous audio -- the launch is a few microseconds of idle workgroups.
#include <hip/hip_runtime.h>

#include "exact_front.hpp"

namespace vad {
namespace {

constexpr int kFixGrid = 512;

template <int Q, typename PcmT, int DEC>
__global__ void __launch_bounds__(256) exact_fix_kernel(const FrontArgs a) {
    __shared__ ExactWs<Q> ws;
    __shared__ RefNet net;
    const int n = a.exact_list[0];                       // (stable: the frontend that filled it has finished, nobody resets it before
                                                         //  every workgroup of this launch has passed its own read)
    if ((int)blockIdx.x < n) {                           // (an empty list -- continuous audio -- costs a workgroup one load)
        if (threadIdx.x == 0) net = *a.exact_net;
        __syncthreads();
    }
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int entry = a.exact_list[2 + i], id = entry & ~kExactSilentBit;
        const long tile = id >> 4;
        const int j = id & 15;
        const long st = tile / a.nt, tl = tile % a.nt;
        const bool silent = (entry & kExactSilentBit) != 0;       // workgroup-uniform
        if (!silent) exact_gx<Q, PcmT, DEC>(a, net, st * 16 + j, a.t0 + tl, ws);
        // gx[tile][row block 32][lane 64][4]: row 16 mb + 4 g + r of chunk j sits at lane 16 g + j, element r (layout.hpp)
        float *gxt = a.gx + (size_t)tile * 32 * 256;
        for (int r = threadIdx.x; r < 512; r += 256)
            gxt[((size_t)(r >> 4) * 64 + ((r >> 2) & 3) * 16 + j) * 4 + (r & 3)] = silent ? a.gx_silent[r] : ws.gx[r];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&a.exact_list[1], 1) == (int)gridDim.x - 1) {
            a.exact_list[0] = 0;
            a.exact_list[1] = 0;
        }
    }
}

template <int Q>
__global__ void __launch_bounds__(256) exact_silent_kernel(const RefNet *net_dev, float *out) {
    __shared__ ExactWs<Q> ws;
    __shared__ RefNet net;
    if (threadIdx.x == 0) net = *net_dev;
    __syncthreads();
    FrontArgs a{};                                       // pcm == nullptr: every sample, the context included, is zero
    a.T = 2;
    exact_gx<Q, float, 1>(a, net, 0, 1, ws);
    for (int r = threadIdx.x; r < 512; r += 256) out[r] = ws.gx[r];
}

}  // namespace

hipError_t launch_exact_silent(int sr, const RefNet *net_dev, float *out, hipStream_t s) {
    if (sr == 16000) hipLaunchKernelGGL((exact_silent_kernel<32>), dim3(1), dim3(256), 0, s, net_dev, out);
    else hipLaunchKernelGGL((exact_silent_kernel<16>), dim3(1), dim3(256), 0, s, net_dev, out);
    return hipGetLastError();
}

template <typename PcmT>
hipError_t launch_exact_fix(int sr, const FrontArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0 || !a.exact_list || !a.exact_net) return hipSuccess;
    if (a.dec > 1 && (sr != 16000 || a.dec > 3)) return hipErrorInvalidValue;
    const long chunks = (long)((a.B + 15) / 16) * 16 * a.nt;
    const unsigned grid = (unsigned)(chunks < kFixGrid ? chunks : kFixGrid);
    if (a.dec == 3) hipLaunchKernelGGL((exact_fix_kernel<32, PcmT, 3>), dim3(grid), dim3(256), 0, s, a);
    else if (a.dec == 2) hipLaunchKernelGGL((exact_fix_kernel<32, PcmT, 2>), dim3(grid), dim3(256), 0, s, a);
    else if (sr == 16000) hipLaunchKernelGGL((exact_fix_kernel<32, PcmT, 1>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((exact_fix_kernel<16, PcmT, 1>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_exact_fix<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_exact_fix<int16_t>(int, const FrontArgs &, hipStream_t);

}  // namespace vad
