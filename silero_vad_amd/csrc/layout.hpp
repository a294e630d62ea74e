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

// ---- Winograd frontend image (kernel_front_wino.hip): whole 16-block (16 KiB) units ------------------------------
// enc0 is a k = 3, stride-1 conv over the 4 STFT frames; as two F(2,3) Winograd transforms over the frame pairs
// (0,1) and (2,3) it needs 4 instead of 5 GEMMs of [128 x 4Q] per pair:
//     d = (0, x0, x1, x2) | (x1, x2, x3, 0)      (inputs of the pair, zero = the conv's own zero padding)
//     m1 = (d0 - d2) G0,  m2 = (d1 + d2) GA,  m3 = (d2 - d1) GB,  m4 = (d1 - d3) G2
//     y_first = m1 + m2 + m3,   y_second = m2 - m3 - m4
// with G0 = g0, GA = (g0 + g1 + g2) / 2, GB = (g0 - g1 + g2) / 2, G2 = g2 (g_tau = tap tau of the conv weight; GA, GB
// are formed in double and rounded once).  Every product and sum is fp32; on the reference's fixtures the result is
// as close to the reference as the direct form (4e-7 / 1.7e-6 on the probabilities, measured with the ATen port).
// Each matrix is cut into P row parts of RB = 64 / Q row blocks, so that one (part, matrix) pair is exactly one unit
// [k-group Q/4][RB][lane 64][4]; encoder 1 is cut by K half (64 input channels = one unit per (tap, half)).
constexpr int w_rb(int Q) { return 64 / Q; }                  // row blocks per enc0 part: 2 (16 kHz) | 4 (8 kHz)
constexpr int w_parts(int Q) { return 8 / w_rb(Q); }          // 4 | 2
enum WMat { WG0 = 0, WGA, WGB, WG2 };
constexpr int w_e0(int p, int j, int Q) { return p * 4 + j; }                          // part p, matrix j
constexpr int w_e1(int h, int i, int Q) { return 4 * w_parts(Q) + 3 * h + i; }         // K half h; i: 0 tap 1, 1 tap 2, 2 tap 0
constexpr int w_e2(int i, int Q) { return 4 * w_parts(Q) + 6 + i; }                    // i: 0 tap 1, 1 tap 2
constexpr int w_e3(int u, int Q) { return 4 * w_parts(Q) + 8 + u; }                    // 2 units
constexpr int w_ih(int q, int u, int Q) { return 4 * w_parts(Q) + 10 + 4 * q + u; }    // gate q, 4 units each
constexpr int w_image_units(int Q) { return 4 * w_parts(Q) + 26; }
constexpr long kWUnitFloats = 16 * 256;
constexpr long front_wino_floats(int Q) { return (long)w_image_units(Q) * kWUnitFloats; }
// Program order of the units (what the kernel consumes, in order; enc0 units are walked twice, once per frame pair)
constexpr int w_program_units(int Q) { return 2 * 4 * w_parts(Q) + 6 + 4 + 2 + 2 + 16; }
struct WSched {
    int n;
    int unit[80];
};
constexpr WSched make_wsched(int Q) {
    WSched sc{};
    int n = 0;
    const int P = w_parts(Q), PH = P / 2;
    for (int pair = 0; pair < 2; ++pair)
        for (int h = 0; h < 2; ++h) {
            for (int pp = 0; pp < PH; ++pp)
                for (int j = 0; j < 4; ++j) sc.unit[n++] = w_e0(h * PH + pp, j, Q);
            for (int i = 0; i < (pair == 0 ? 3 : 2); ++i) sc.unit[n++] = w_e1(h, i, Q);
        }
    sc.unit[n++] = w_e2(0, Q);
    sc.unit[n++] = w_e2(1, Q);
    sc.unit[n++] = w_e3(0, Q);
    sc.unit[n++] = w_e3(1, Q);
    for (int q = 0; q < 4; ++q)
        for (int u = 0; u < 4; ++u) sc.unit[n++] = w_ih(q, u, Q);
    sc.n = n;
    return sc;
}

// ---- F(4,3) Winograd frontend image (kernel_front_wino.hip, front_wino4_kernel) -----------------------------------
// All 4 STFT frames of a chunk are one F(4,3) tile of the k = 3, stride-1 conv: d = (0, x0, x1, x2, x3, 0), 6 GEMMs of
// [128 x 4Q] instead of the 10 of the direct form (8 with two F(2,3) tiles).  With interpolation points 0, +-1, +-2, inf:
//     t1 = (x2 + x3) - 4 (x0 + x1)    U1 = -(g0 + g1 + g2) / 6           y0 = m0 + m1 + m2 +   m3 +   m4
//     t2 = (x3 - x2) + 4 (x0 - x1)    U2 = -(g0 - g1 + g2) / 6           y1 =      m1 - m2 + 2 m3 - 2 m4
//     t3 = (x3 - x1) + 2 (x2 - x0)    U3 = g0 / 24 + g1 / 12 + g2 / 6    y2 =      m1 + m2 + 4 m3 + 4 m4
//     t4 = (x3 - x1) - 2 (x2 - x0)    U4 = g0 / 24 - g1 / 12 + g2 / 6    y3 =      m1 - m2 + 8 m3 - 8 m4 + m5
//     t0 = x3 - 5 x1                  U0 = g0 / 4                        (m_i = U_i t_i; U_i formed in double, rounded once)
//     t5 = x0 - 1.25 x2               U5 = 4 g2
// The image IS the program: units in the order the kernel consumes them, each used once per tile.  Per row part p
// (16 RB rows, RB = w_rb(Q)): U1, U2, U3, U4, U0, U5, then encoder 1's share of the part's 16 RB input channels:
//   Q = 32 (8 k-steps per tap and part, two taps per unit):  [tap1 <- y0 | tap2 <- y1] -> out 0,
//        [tap0 <- y1 | tap1 <- y2] -> out 1, and after every odd part [tap2 <- y3 of part p-1 | tap2 <- y3 of part p] -> out 1
//   Q = 16 (16 k-steps per tap and part): tap1 <- y0, tap2 <- y1 (out 0); tap0 <- y1, tap1 <- y2, tap2 <- y3 (out 1)
enum W4Mat { W4U1 = 0, W4U2, W4U3, W4U4, W4U0, W4U5 };
constexpr int w4_e1_units(int p, int Q) { return Q == 32 ? 2 + (p & 1) : 5; }
constexpr int w4_part0(int p, int Q) {
    int u = 0;
    for (int i = 0; i < p; ++i) u += 6 + w4_e1_units(i, Q);
    return u;
}
constexpr int w4_e0(int p, int j, int Q) { return w4_part0(p, Q) + j; }
constexpr int w4_e1(int p, int i, int Q) { return w4_part0(p, Q) + 6 + i; }
constexpr int w4_tail0(int Q) { return w4_part0(w_parts(Q), Q); }          // E2 tap 1, tap 2, E3 (2), W_ih (16)
constexpr int w4_units(int Q) { return w4_tail0(Q) + 20; }                 // 54 | 42
constexpr long front_wino4_floats(int Q) { return (long)w4_units(Q) * kWUnitFloats; }

// ---- bf16 x 9 frontend image (kernel_front_b9.hip) ----------------------------------------------------------------
// The F(4,3) program above, unit for unit, with every fp32 weight as three exact bf16 pieces (as the recurrent image below) for
// v_mfma_f32_16x16x32_bf16.  One K32 step of that instruction carries the 8 B-operand values a lane holds for TWO fp32 k-groups:
// slot (g, e) of K32 step kp <-> fp32 k-step 8 kp + e at k = g, so that the chain / mag layouts stay what they are.  A unit holds the
// same rows and k range as its fp32 unit: [step 4][piece 3][row block 2][lane 64][8 bf16] = 24 KiB, step i of a unit of an M-row-block
// segment = (K32 step i / (M/2), row blocks 2 (i % (M/2)), +1).
constexpr int w4_unit_m(int u, int Q) {               // row blocks of the segment unit u belongs to
    const int T0 = w4_tail0(Q);
    if (u >= T0) return u < T0 + 2 ? 4 : 8;
    int p = 0;
    while (p + 1 < w_parts(Q) && w4_part0(p + 1, Q) <= u) ++p;
    return u - w4_part0(p, Q) < 6 ? w_rb(Q) : 4;
}
constexpr long kW9UnitHalfs = 4L * 3 * 2 * 64 * 8;   // 12 288 bf16 = 24 KiB
constexpr long front_b9_halfs(int Q) { return (long)w4_units(Q) * kW9UnitHalfs; }

// ---- recurrent image: [wave 8][gate 4][kgroup 8][lane 64][4] ------------------------------------
constexpr long whh_floats() { return 8L * 4 * 8 * 256; }

// ---- recurrent weights row by row (kernel_rec_small.hip): [row 512][128] floats, a row in the order in which rec_kernel's MFMA chain
// adds its products: position 16 kg + 4 r + g holds W_hh[row][16 kg + 4 g + r] (k-groups ascending; within a k-group the four MFMAs r,
// within an MFMA the four lane groups g = the instruction's k index)
constexpr long whh_rows_floats() { return 512L * 128; }

// ---- bf16 x 9 recurrent image (kernel_rec_b9.hip) -----------------------------------------------------------------
// Every fp32 weight w is stored as THREE bf16 pieces w = p0 + p1 + p2, exactly (p0 = bf16(w), p1 = bf16(w - p0),
// p2 = w - p0 - p1: bf16 carries 8 significand bits and fp32's exponent range, so three pieces hold all 24 bits and none
// underflows), for v_mfma_f32_16x16x32_bf16, whose A and B operands hold 8 bf16 per lane: lane (g, i | j) supplies k-slots
// (g, e), e < 8.  K32 step u, slot (g, e) <-> hidden unit 32 u + 8 g + e (natural order: the B operand is 8 consecutive
// h values of a stream).  Image: [wave 8][piece 3][gate 4][u 4][lane 64][8] bf16; wave w owns rows 128 q + 16 w + i.
constexpr long whh_b9_halfs() { return 8L * 3 * 4 * 4 * 64 * 8; }

// ---- small tables (floats) ----------------------------------------------------------------------
struct Tab {
    int b_e0, b_e1, b_e2, b_e3, b_g, w_out, b_out, window, tw1, tw2, w_nyq, total;
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
    t.w_nyq = o; o += 3 * 128;    // [tap][row] encoder-0 weights of the Nyquist bin (applied as an fp32 rank-1 update on the VALU)
    t.total = o;
    return t;
}
constexpr Tab tab16 = make_tab(256, 32);
constexpr Tab tab8 = make_tab(128, 16);

}  // namespace vadl
