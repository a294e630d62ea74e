// weights.cpp -- container parser + MFMA-fragment packer (host only).
#include "weights.hpp"

#include <cmath>
#include <cstring>

namespace vad {
namespace {

struct Rec {            // one table entry of the container, 100 bytes, little endian
    char name[64];
    uint32_t ndim, dims[4];
    uint64_t offset, count;
};

const float *find(const std::vector<uint8_t> &blob, const std::string &name, size_t expect,
                  std::string &err) {
    uint32_t n;
    std::memcpy(&n, blob.data() + 8, 4);
    for (uint32_t i = 0; i < n; ++i) {
        const uint8_t *p = blob.data() + 80 + (size_t)i * 100;
        if (80 + (size_t)(i + 1) * 100 > blob.size()) break;
        if (std::strncmp((const char *)p, name.c_str(), 64) != 0) continue;
        uint64_t off, cnt;
        std::memcpy(&off, p + 84, 8);
        std::memcpy(&cnt, p + 92, 8);
        if (cnt != expect || off % 4 != 0 || off + cnt * 4 > blob.size()) {
            err = "tensor " + name + " has unexpected size";
            return nullptr;
        }
        return reinterpret_cast<const float *>(blob.data() + off);
    }
    err = "tensor " + name + " missing";
    return nullptr;
}

bool bind(const std::vector<uint8_t> &blob, const std::string &prefix, const vadl::Geo &g,
          NetTensors &t, std::string &err) {
    const int cin[4] = {g.K, 128, 64, 64}, cout[4] = {128, 64, 64, 128};
    auto get = [&](const std::string &n, size_t cnt) { return find(blob, prefix + "." + n, cnt, err); };
    if (!(t.basis = get("stft.forward_basis_buffer", (size_t)2 * g.K * g.F))) return false;
    for (int l = 0; l < 4; ++l) {
        const std::string e = "encoder." + std::to_string(l) + ".reparam_conv.";
        if (!(t.ew[l] = get(e + "weight", (size_t)cout[l] * cin[l] * 3))) return false;
        if (!(t.eb[l] = get(e + "bias", (size_t)cout[l]))) return false;
    }
    if (!(t.w_ih = get("decoder.rnn.weight_ih", 4 * 128 * 128))) return false;
    if (!(t.w_hh = get("decoder.rnn.weight_hh", 4 * 128 * 128))) return false;
    if (!(t.b_ih = get("decoder.rnn.bias_ih", 4 * 128))) return false;
    if (!(t.b_hh = get("decoder.rnn.bias_hh", 4 * 128))) return false;
    if (!(t.w_out = get("decoder.decoder.2.weight", 128))) return false;
    if (!(t.b_out = get("decoder.decoder.2.bias", 1))) return false;
    return true;
}

// channel carried by k-step s in lane group g of a chain-layout activation (layout.hpp)
inline int chain_chan(int s, int g) { return 16 * (s / 4) + 4 * g + (s % 4); }

// dst = [kgroup][mblock][lane][4]; w(row, s, g) returns the weight that multiplies whatever lane
// group g supplies at k-step s, for output row `row` (or 0 if nothing is supplied).
template <class F>
void pack_segment(float *dst, int M, int KS, F w) {
    const int KG = (KS + 3) / 4;
    for (int kg = 0; kg < KG; ++kg)
        for (int m = 0; m < M; ++m)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int s = 4 * kg + r, g = lane >> 4, i = lane & 15;
                    dst[(((size_t)kg * M + m) * 64 + lane) * 4 + r] = s < KS ? w(16 * m + i, s, g) : 0.f;
                }
}

// Winograd image (layout.hpp "Winograd frontend image"): whole units, enc0 as the 4 transformed matrices
void pack_net_wino(const NetTensors &t, const vadl::Geo &g, PackedNet &p) {
    using namespace vadl;
    const int Q = g.Q, K = g.K, RB = w_rb(Q), P = w_parts(Q);
    p.front_wino.assign((size_t)front_wino_floats(Q), 0.f);
    auto unit = [&](int u) { return p.front_wino.data() + (size_t)u * kWUnitFloats; };
    auto g_tap = [&](int row, int bin, int tau) { return (double)t.ew[0][((size_t)row * K + bin) * 3 + tau]; };
    for (int part = 0; part < P; ++part)
        for (int j = 0; j < 4; ++j)
            pack_segment(unit(w_e0(part, j, Q)), RB, Q, [&](int row, int s, int gg) {
                const int r = 16 * RB * part + row, bin = 4 * s + kResidue[gg];
                const double g0 = g_tap(r, bin, 0), g1 = g_tap(r, bin, 1), g2 = g_tap(r, bin, 2);
                const double v = j == WG0 ? g0 : j == WGA ? 0.5 * (g0 + g1 + g2) : j == WGB ? 0.5 * (g0 - g1 + g2) : g2;
                return (float)v;
            });
    const int e1_tap[3] = {1, 2, 0};
    for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 3; ++i)
            pack_segment(unit(w_e1(h, i, Q)), 4, 16, [&](int row, int s, int gg) {
                return t.ew[1][((size_t)row * 128 + 64 * h + chain_chan(s, gg)) * 3 + e1_tap[i]];
            });
    for (int i = 0; i < 2; ++i)
        pack_segment(unit(w_e2(i, Q)), 4, 16, [&](int row, int s, int gg) {
            return t.ew[2][((size_t)row * 64 + chain_chan(s, gg)) * 3 + 1 + i];
        });
    pack_segment(unit(w_e3(0, Q)), 8, 16, [&](int row, int s, int gg) {       // 2 consecutive units
        return t.ew[3][((size_t)row * 64 + chain_chan(s, gg)) * 3 + 1];
    });
    for (int q = 0; q < 4; ++q)
        pack_segment(unit(w_ih(q, 0, Q)), 8, 32, [&](int row, int s, int gg) {  // 4 consecutive units
            return t.w_ih[((size_t)(128 * q + row)) * 128 + chain_chan(s, gg)];
        });
}

// F(4,3) image (layout.hpp "F(4,3) Winograd frontend image"): the units in program order
void pack_net_wino4(const NetTensors &t, const vadl::Geo &g, PackedNet &p) {
    using namespace vadl;
    const int Q = g.Q, K = g.K, RB = w_rb(Q), P = w_parts(Q);
    p.front_wino4.assign((size_t)front_wino4_floats(Q), 0.f);
    auto unit = [&](int u) { return p.front_wino4.data() + (size_t)u * kWUnitFloats; };
    auto g_tap = [&](int row, int bin, int tau) { return (double)t.ew[0][((size_t)row * K + bin) * 3 + tau]; };
    auto e1 = [&](int row, int chan, int tau) { return t.ew[1][((size_t)row * 128 + chan) * 3 + tau]; };
    for (int part = 0; part < P; ++part) {
        for (int j = 0; j < 6; ++j)
            pack_segment(unit(w4_e0(part, j, Q)), RB, Q, [&](int row, int s, int gg) {
                const int r = 16 * RB * part + row, bin = 4 * s + kResidue[gg];
                const double g0 = g_tap(r, bin, 0), g1 = g_tap(r, bin, 1), g2 = g_tap(r, bin, 2);
                double v = 0;
                switch (j) {
                    case W4U1: v = -(g0 + g1 + g2) / 6.0; break;
                    case W4U2: v = -(g0 - g1 + g2) / 6.0; break;
                    case W4U3: v = g0 / 24.0 + g1 / 12.0 + g2 / 6.0; break;
                    case W4U4: v = g0 / 24.0 - g1 / 12.0 + g2 / 6.0; break;
                    case W4U0: v = g0 / 4.0; break;
                    default: v = 4.0 * g2; break;
                }
                return (float)v;
            });
        const int c0 = 16 * RB * part;                    // first input channel of the part
        if (Q == 32) {
            // 8 k-steps per tap: k-steps 0..7 the first tap, 8..15 the second
            const int tapsA[2] = {1, 2}, tapsB[2] = {0, 1};
            pack_segment(unit(w4_e1(part, 0, Q)), 4, 16, [&](int row, int s, int gg) {
                return e1(row, c0 + chain_chan(s & 7, gg), tapsA[s >> 3]);
            });
            pack_segment(unit(w4_e1(part, 1, Q)), 4, 16, [&](int row, int s, int gg) {
                return e1(row, c0 + chain_chan(s & 7, gg), tapsB[s >> 3]);
            });
            if (part & 1)
                pack_segment(unit(w4_e1(part, 2, Q)), 4, 16, [&](int row, int s, int gg) {
                    return e1(row, c0 - 32 * (1 - (s >> 3)) + chain_chan(s & 7, gg), 2);
                });
        } else {
            const int taps[5] = {1, 2, 0, 1, 2};
            for (int i = 0; i < 5; ++i)
                pack_segment(unit(w4_e1(part, i, Q)), 4, 16, [&](int row, int s, int gg) {
                    return e1(row, c0 + chain_chan(s, gg), taps[i]);
                });
        }
    }
    const int T0 = w4_tail0(Q);
    for (int i = 0; i < 2; ++i)
        pack_segment(unit(T0 + i), 4, 16, [&](int row, int s, int gg) {
            return t.ew[2][((size_t)row * 64 + chain_chan(s, gg)) * 3 + 1 + i];
        });
    pack_segment(unit(T0 + 2), 8, 16, [&](int row, int s, int gg) {           // 2 consecutive units
        return t.ew[3][((size_t)row * 64 + chain_chan(s, gg)) * 3 + 1];
    });
    for (int q = 0; q < 4; ++q)
        pack_segment(unit(T0 + 4 + 4 * q), 8, 32, [&](int row, int s, int gg) {  // 4 consecutive units
            return t.w_ih[((size_t)(128 * q + row)) * 128 + chain_chan(s, gg)];
        });
}

void pack_net(const NetTensors &t, const vadl::Geo &g, PackedNet &p) {
    using namespace vadl;
    const int Q = g.Q, K = g.K;
    p.geo = g;
    p.front.assign((size_t)front_floats(Q), 0.f);
    for (int tau = 0; tau < 3; ++tau) {
        // encoder 0: input in mag layout
        pack_segment(p.front.data() + seg_offset(E0T0 + tau, Q), 8, Q + 1, [&](int row, int s, int gg) {
            int bin;
            if (s < Q) bin = 4 * s + kResidue[gg];
            else if (gg == 0) bin = 4 * Q;
            else return 0.f;
            return t.ew[0][((size_t)row * K + bin) * 3 + tau];
        });
        // encoder 1: 128 -> 64
        pack_segment(p.front.data() + seg_offset(E1T0 + tau, Q), 4, 32, [&](int row, int s, int gg) {
            return t.ew[1][((size_t)row * 128 + chain_chan(s, gg)) * 3 + tau];
        });
    }
    for (int tau = 1; tau < 3; ++tau)   // encoder 2: tap 0 only ever sees the left zero pad
        pack_segment(p.front.data() + seg_offset(E2T1 + tau - 1, Q), 4, 16, [&](int row, int s, int gg) {
            return t.ew[2][((size_t)row * 64 + chain_chan(s, gg)) * 3 + tau];
        });
    // encoder 3: T_in = 1, only the centre tap sees data
    pack_segment(p.front.data() + seg_offset(E3T1, Q), 8, 16, [&](int row, int s, int gg) {
        return t.ew[3][((size_t)row * 64 + chain_chan(s, gg)) * 3 + 1];
    });
    for (int q = 0; q < 4; ++q)         // W_ih, one gate (128 rows) per segment
        pack_segment(p.front.data() + seg_offset(IH0 + q, Q), 8, 32, [&](int row, int s, int gg) {
            return t.w_ih[((size_t)(128 * q + row)) * 128 + chain_chan(s, gg)];
        });

    // recurrent image [wave][gate][kgroup][lane][4]
    p.whh.assign((size_t)whh_floats(), 0.f);
    for (int w = 0; w < 8; ++w)
        for (int q = 0; q < 4; ++q)
            for (int kg = 0; kg < 8; ++kg)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int gg = lane >> 4, i = lane & 15;
                        p.whh[((((size_t)w * 4 + q) * 8 + kg) * 64 + lane) * 4 + r] =
                            t.w_hh[(size_t)(128 * q + 16 * w + i) * 128 + chain_chan(4 * kg + r, gg)];
                    }

    // recurrent weights in W_ih's order, gate by gate: [gate][kg][row block][lane][4] -- the fused single-step kernel appends them to a
    // wave's weight stream (kernel_front_lat.hip); same k order per row block as the recurrent image above, so the sums are the same
    p.whh_lat.assign((size_t)4 * 64 * 256, 0.f);
    for (int q = 0; q < 4; ++q)
        pack_segment(p.whh_lat.data() + (size_t)q * 64 * 256, 8, 32, [&](int row, int s, int gg) {
            return t.w_hh[((size_t)(128 * q + row)) * 128 + chain_chan(s, gg)];
        });

    // row image for the small-batch recurrence: a row's weights in the order rec_kernel's MFMA chain adds them (layout.hpp "whh_rows")
    p.whh_rows.assign((size_t)whh_rows_floats(), 0.f);
    for (int row = 0; row < 512; ++row)
        for (int kg = 0; kg < 8; ++kg)
            for (int r = 0; r < 4; ++r)
                for (int gg = 0; gg < 4; ++gg)
                    p.whh_rows[(size_t)row * 128 + 16 * kg + 4 * r + gg] = t.w_hh[(size_t)row * 128 + 16 * kg + 4 * gg + r];

    // recurrent image as three bf16 pieces per weight: [wave][piece][gate][u][lane][8] (layout.hpp)
    p.whh_b9.assign((size_t)whh_b9_halfs(), 0);
    auto bf16_rne = [](float x) -> uint16_t {
        uint32_t u;
        std::memcpy(&u, &x, 4);
        u += 0x7fffu + ((u >> 16) & 1u);                   // round to nearest even (no NaN / Inf among the weights)
        return (uint16_t)(u >> 16);
    };
    auto bf16_f32 = [](uint16_t h) -> float {
        const uint32_t u = (uint32_t)h << 16;
        float x;
        std::memcpy(&x, &u, 4);
        return x;
    };
    for (int w = 0; w < 8; ++w)
        for (int q = 0; q < 4; ++q)
            for (int u = 0; u < 4; ++u)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int gg = lane >> 4, i = lane & 15;
                        float r = t.w_hh[(size_t)(128 * q + 16 * w + i) * 128 + 32 * u + 8 * gg + e];
                        for (int piece = 0; piece < 3; ++piece) {
                            const uint16_t h = bf16_rne(r);
                            p.whh_b9[((((((size_t)w * 3 + piece) * 4 + q) * 4 + u) * 64 + lane) * 8) + e] = h;
                            r -= bf16_f32(h);              // exact: the remainder has fewer significant bits than r
                        }
                    }

    // tables
    pack_net_wino4(t, g, p);
    // the F(4,3) program as bf16 pieces: a re-indexing of the fp32 image (layout.hpp "bf16 x 9 frontend image")
    p.front_b9.assign((size_t)front_b9_halfs(Q), 0);
    for (int u = 0; u < w4_units(Q); ++u) {
        const int H = w4_unit_m(u, Q) / 2;
        const float *src = p.front_wino4.data() + (size_t)u * kWUnitFloats;
        uint16_t *dst = p.front_b9.data() + (size_t)u * kW9UnitHalfs;
        for (int ib = 0; ib < 4; ++ib)
            for (int rb = 0; rb < 2; ++rb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int kp = ib / H, mh = ib % H, kg = 2 * kp + e / 4, i_f = kg * H + mh;
                        float r = src[((size_t)(i_f * 2 + rb) * 64 + lane) * 4 + (e & 3)];
                        for (int piece = 0; piece < 3; ++piece) {
                            const uint16_t h = bf16_rne(r);
                            dst[((((size_t)ib * 3 + piece) * 2 + rb) * 64 + lane) * 8 + e] = h;
                            r -= bf16_f32(h);
                        }
                    }
    }
    const Tab tb = make_tab(g.F, Q);
    p.tables.assign((size_t)tb.total, 0.f);
    float *T = p.tables.data();
    std::memcpy(T + tb.b_e0, t.eb[0], 128 * 4);
    std::memcpy(T + tb.b_e1, t.eb[1], 64 * 4);
    std::memcpy(T + tb.b_e2, t.eb[2], 64 * 4);
    std::memcpy(T + tb.b_e3, t.eb[3], 128 * 4);
    for (int r = 0; r < 512; ++r) T[tb.b_g + r] = t.b_ih[r] + t.b_hh[r];
    std::memcpy(T + tb.w_out, t.w_out, 128 * 4);
    T[tb.b_out] = t.b_out[0];
    std::memcpy(T + tb.window, t.basis, (size_t)g.F * 4);   // basis row 0 = w[n] * cos(0)
    const double two_pi = 6.283185307179586476925286766559;
    for (int gg = 0; gg < 4; ++gg)
        for (int q = 0; q < Q; ++q) {
            const double a1 = two_pi * kResidue[gg] * q / (4.0 * Q);
            const float c1 = (float)std::cos(a1), s1 = (float)-std::sin(a1);
            float *t1 = T + tb.tw1 + (gg * Q + q) * 4;
            t1[0] = -s1; t1[1] = s1; t1[2] = c1; t1[3] = 0.f;
            const double a2 = two_pi * (4 * q + kResidue[gg]) / (8.0 * Q);
            const float c2 = (float)std::cos(a2), s2 = (float)-std::sin(a2);
            float *t2 = T + tb.tw2 + (gg * Q + q) * 4;
            t2[0] = c2; t2[1] = -c2; t2[2] = s2; t2[3] = 0.f;
        }
    for (int tau = 0; tau < 3; ++tau)
        for (int row = 0; row < 128; ++row)
            T[tb.w_nyq + tau * 128 + row] = t.ew[0][((size_t)row * K + 4 * Q) * 3 + tau];
    pack_net_wino(t, g, p);
}

}  // namespace

std::string Weights::load(const void *data, size_t nbytes) {
    if (!data || nbytes < 80 || std::memcmp(data, "SVADW001", 8) != 0)
        return "not an SVADW001 weight container";
    blob.assign((const uint8_t *)data, (const uint8_t *)data + nbytes);
    std::string err;
    if (!bind(blob, "_model", vadl::geo16, net[0], err)) return err;
    if (!bind(blob, "_model_8k", vadl::geo8, net[1], err)) return err;
    pack_net(net[0], vadl::geo16, packed[0]);
    pack_net(net[1], vadl::geo8, packed[1]);
    return "";
}

}  // namespace vad
