// weights.hpp -- host side: parse the SVADW001 container (tools/export_weights.py) and build the
// images the kernels read (layout.hpp).  Pure C++, no HIP.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "layout.hpp"

namespace vad {

// Canonical tensors of one net, pointing into the container
// (names: state_dict of silero_vad.jit).
struct NetTensors {
    const float *basis = nullptr;            // [2K][F]
    const float *ew[4] = {}, *eb[4] = {};    // [Cout][Cin][3], [Cout]
    const float *w_ih = nullptr, *w_hh = nullptr, *b_ih = nullptr, *b_hh = nullptr;
    const float *w_out = nullptr, *b_out = nullptr;
    size_t basis_n = 0, ew_n[4] = {}, eb_n[4] = {};
};

struct PackedNet {
    vadl::Geo geo{};
    std::vector<float> front;    // frontend GEMM stream (layout.hpp Seg order)
    std::vector<float> front_wino;   // frontend stream with enc0 in Winograd F(2,3) form (layout.hpp w_* units)
    std::vector<float> front_wino4;  // the same with enc0 as one F(4,3) tile, units in program order (layout.hpp w4_*)
    std::vector<float> whh;      // recurrent image
    std::vector<float> whh_lat;  // recurrent weights gate by gate in W_ih's block order [gate][kg 8][row block 8][lane][4] (kernel_front_lat.hip, fused step)
    std::vector<float> whh_rows; // recurrent weights row by row in the MFMA chain's summation order (layout.hpp "whh_rows", kernel_rec_small.hip)
    std::vector<uint16_t> whh_b9;    // recurrent image as three bf16 pieces per weight (layout.hpp "bf16 x 9 recurrent image")
    std::vector<uint16_t> front_b9;  // the F(4,3) program as three bf16 pieces per weight, 24 KiB units (layout.hpp "bf16 x 9 frontend image")
    std::vector<float> tables;   // biases, head, window, twiddles, Nyquist-bin weights
};

struct Weights {
    std::vector<uint8_t> blob;   // private copy of the container
    NetTensors net[2];           // 0: 16 kHz, 1: 8 kHz
    PackedNet packed[2];
    // returns empty string on success, else a description of what is wrong
    std::string load(const void *data, size_t nbytes);
};

}  // namespace vad
