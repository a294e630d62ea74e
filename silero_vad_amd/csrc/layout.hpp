// layout.hpp -- shared by the host packer (weights.cpp) and the device kernels.
//
// Everything the kernels read is laid out for the gfx950 f32 MFMA v_mfma_f32_16x16x4_f32:
//   lane l = (g, j): g = l >> 4 (lane group 0..3), j = l & 15
//   A operand  lane (g, i)  supplies A[row i][k = g]
//   B operand  lane (g, j)  supplies B[k = g][col j]
//   C/D        lane (g, j), register r  holds D[row 4g + r][col j]
// Columns are always independent chunks (one stream each); rows are channels.
//
// "Chain layout": an activation tensor Act[channel][col] with 16*NB channels lives in NB f32x4
// registers per lane; register (blk, r) of lane (g, j) = Act[16 blk + 4 g + r][j].  That is the
// D layout of one 16-row MFMA block, and it is ALSO a valid B operand if k-step s = 4 blk + r is
// defined to carry channel chan(s, g) = 16 (s/4) + 4 g + (s%4): the order in which a dot product
// is summed is free, so the permutation is folded into the A (weight) image on the host and a
// layer's MFMA output feeds the next layer's MFMA with no data movement at all.
//
// "Mag layout" (output of the in-wave FFT, input of encoder layer 0): k-step s < Q carries bin
// 4 s + P[g] (P = {0,2,1,3}, the residue class the 4-lane cross-lane radix-4 leaves in group g);
// k-step s == Q carries the Nyquist bin 4Q in group 0 and nothing (zero weight) elsewhere.
#pragma once

namespace vadl {

constexpr int HID = 128;

struct Geo {
    int sr, N, C, F, H, K, Q;
};
// N chunk, C context, F filter length, H hop, K bins, Q in-lane FFT length (F = 8Q, K = 4Q + 1)
constexpr Geo geo16{16000, 512, 64, 256, 128, 129, 32};
constexpr Geo geo8{8000, 256, 32, 128, 64, 65, 16};

constexpr int kResidue[4] = {0, 2, 1, 3};   // P[g]

// ---- frontend GEMM stream: 13 segments, each [kgroup][mblock][lane 64][4 k-steps] floats -------
enum Seg { E0T0, E0T1, E0T2, E1T0, E1T1, E1T2, E2T1, E2T2, E3T1, IH0, IH1, IH2, IH3, NSEG };

constexpr int seg_mblocks(int s) { return (s <= E0T2) ? 8 : (s <= E2T2) ? 4 : 8; }
constexpr int seg_ksteps(int s, int Q) {
    return (s <= E0T2) ? Q + 1 : (s <= E1T2) ? 32 : (s <= E3T1) ? 16 : 32;
}
constexpr int seg_kgroups(int s, int Q) { return (seg_ksteps(s, Q) + 3) / 4; }
constexpr long seg_floats(int s, int Q) { return (long)seg_kgroups(s, Q) * seg_mblocks(s) * 256; }
constexpr long seg_offset(int s, int Q) {
    long o = 0;
    for (int i = 0; i < s; ++i) o += seg_floats(i, Q);
    return o;
}
constexpr long front_floats(int Q) { return seg_offset(NSEG, Q); }

// ---- recurrent image: [wave 8][gate 4][kgroup 8][lane 64][4] ------------------------------------
constexpr long whh_floats() { return 8L * 4 * 8 * 256; }

// ---- small tables (floats) ----------------------------------------------------------------------
struct Tab {
    int b_e0, b_e1, b_e2, b_e3, b_g, w_out, b_out, window, tw1, tw2, total;
};
constexpr Tab make_tab(int F, int Q) {
    Tab t{};
    int o = 0;
    t.b_e0 = o; o += 128;
    t.b_e1 = o; o += 64;
    t.b_e2 = o; o += 64;
    t.b_e3 = o; o += 128;
    t.b_g = o; o += 512;          // b_ih + b_hh
    t.w_out = o; o += 128;
    t.b_out = o; o += 4;          // [0] used, padded to keep 16-B alignment
    t.window = o; o += F;         // row 0 of the reference's forward_basis_buffer (= periodic Hann)
    // twiddles are stored in the operand order of the packed complex multiply (kernel_front.hip):
    t.tw1 = o; o += 4 * Q * 4;    // [g][q]  (-s, s, c, 0),  c + i s = exp(-2 pi i P[g] q / (4Q))
    t.tw2 = o; o += 4 * Q * 4;    // [g][k'] (c, -c, s, 0),  c + i s = exp(-2 pi i (4k' + P[g]) / (8Q))
    t.total = o;
    return t;
}
constexpr Tab tab16 = make_tab(256, 32);
constexpr Tab tab8 = make_tab(128, 16);

}  // namespace vadl
