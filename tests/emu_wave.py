"""Lane-accurate numpy emulation of the gfx950 wave programs in silero_vad_amd/csrc
(kernel_front.hip, kernel_rec.hip).

Purpose: the authoring container has no GPU, and the two error-prone parts of the MFMA design --
the host-side fragment packing (weights.cpp) and the index algebra of the in-wave FFT / register
chaining -- are pure data-layout questions.  This module executes the SAME program the kernels
execute, on arrays of shape [64] (one value per lane), reading the SAME packed images through the
C ABI (vad_debug_packed_copy), so CPU tests can pin them against the oracle.

Every function mirrors a device function of the same name.
"""
import numpy as np

f32 = np.float32
LANE = np.arange(64)
G = LANE >> 4
J = LANE & 15
P_RES = np.array([0, 2, 1, 3])


def bitrev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def shfl_xor(v, m):
    return v[LANE ^ m]


def mfma_16x16x4(a, b, acc):
    """v_mfma_f32_16x16x4_f32: A[i][k] = a[16k+i], B[k][j] = b[16k+j]; D[4g+r][j] in acc[r][16g+j]."""
    A = a.reshape(4, 16).T            # [i][k]
    Bm = b.reshape(4, 16)             # [k][j]
    D = (A.astype(np.float64) @ Bm.astype(np.float64)).astype(f32)   # [16 rows][16 cols]
    out = acc.copy()
    for r in range(4):
        out[r] += D[4 * G + r, J]
    return out


class Tab:
    def __init__(self, F, Q):
        o = 0
        self.b_e0 = o; o += 128
        self.b_e1 = o; o += 64
        self.b_e2 = o; o += 64
        self.b_e3 = o; o += 128
        self.b_g = o; o += 512
        self.w_out = o; o += 128
        self.b_out = o; o += 4
        self.window = o; o += F
        self.tw1 = o; o += 4 * Q * 4
        self.tw2 = o; o += 4 * Q * 4
        self.w_nyq = o; o += 3 * 128
        self.total = o


SEG_NAMES = ["E0T0", "E0T1", "E0T2", "E1T0", "E1T1", "E1T2", "E2T1", "E2T2", "E3T1",
             "IH0", "IH1", "IH2", "IH3"]


def seg_mblocks(s):
    return 8 if s <= 2 else 4 if s <= 7 else 8


def seg_ksteps(s, Q):
    return Q + 1 if s <= 2 else 32 if s <= 5 else 16 if s <= 8 else 32


def seg_offset(s, Q):
    o = 0
    for i in range(s):
        o += ((seg_ksteps(i, Q) + 3) // 4) * seg_mblocks(i) * 256
    return o


class FrontEmu:
    def __init__(self, sr, front, tables):
        self.Q = 32 if sr == 16000 else 16
        self.front = front
        self.tab = tables
        self.tb = Tab(8 * self.Q, self.Q)
        assert len(tables) == self.tb.total
        assert len(front) == seg_offset(13, self.Q)

    # ---- load_slice<V>: x [16][18Q] (ctx | chunk), returns s [2Q][64] ---------------------------
    def load_slice(self, x, V):
        Q = self.Q
        SL = 2 * Q
        s = np.zeros((SL, 64), f32)
        sigma = 2 * V + G
        sg = np.minimum(sigma, 8) if V == 3 else sigma
        for lane in range(64):
            s[:, lane] = x[J[lane], SL * sg[lane]: SL * sg[lane] + SL]
        if V == 3:
            rev = G == 3
            extra = x[J, 16 * Q - 1]
            for i in range(SL // 2 - 1):
                k = SL - 2 - i
                lo, hi = s[i].copy(), s[k].copy()
                s[i] = np.where(rev, hi, lo)
                s[k] = np.where(rev, lo, hi)
            s[SL - 1] = np.where(rev, extra, s[SL - 1])
        return s

    def fft_inlane(self, re, im):
        Q = self.Q
        cos32 = np.cos(2 * np.pi * np.arange(16) / 32).astype(f32)
        sin32 = np.sin(2 * np.pi * np.arange(16) / 32).astype(f32)
        n = Q
        while n >= 2:
            half = n // 2
            for b0 in range(0, Q, n):
                for jx in range(half):
                    i0, i1 = b0 + jx, b0 + jx + half
                    ar, ai, br, bi = re[i0].copy(), im[i0].copy(), re[i1].copy(), im[i1].copy()
                    re[i0], im[i0] = ar + br, ai + bi
                    dr, di = ar - br, ai - bi
                    tw = jx * (32 // n)
                    if tw == 0:
                        re[i1], im[i1] = dr, di
                    elif tw == 8:
                        re[i1], im[i1] = di, -dr
                    else:
                        c, sn = cos32[tw], sin32[tw]
                        re[i1] = dr * c + di * sn
                        im[i1] = di * c - dr * sn
            n //= 2

    def fft_pass(self, x, V):
        Q, tb, T = self.Q, self.tb, self.tab
        SL = 2 * Q
        s = self.load_slice(x, V)
        re = np.zeros((Q, 64), f32)
        im = np.zeros((Q, 64), f32)
        for lane_g in range(4):
            m = G == lane_g
            w = T[tb.window + SL * lane_g: tb.window + SL * lane_g + SL]
            for k in range(SL // 4):
                re[2 * k][m] = s[4 * k][m] * w[4 * k]
                im[2 * k][m] = s[4 * k + 1][m] * w[4 * k + 1]
                re[2 * k + 1][m] = s[4 * k + 2][m] * w[4 * k + 2]
                im[2 * k + 1][m] = s[4 * k + 3][m] * w[4 * k + 3]
        sgnA = np.where(G < 2, 1, -1).astype(f32)
        sgnB = np.where(G & 1, -1, 1).astype(f32)
        tw1 = T[tb.tw1: tb.tw1 + 4 * Q * 4].reshape(4, Q, 4)      # (-s, s, c, 0)
        for q in range(Q):
            xr, xi = re[q], im[q]
            pr, pi = shfl_xor(xr, 32), shfl_xor(xi, 32)
            xr, xi = sgnA * xr + pr, sgnA * xi + pi
            g3 = G == 3
            xr, xi = np.where(g3, xi, xr), np.where(g3, -xr, xi)
            qr, qi = shfl_xor(xr, 16), shfl_xor(xi, 16)
            xr, xi = sgnB * xr + qr, sgnB * xi + qi
            c, sn = tw1[G, q, 2], tw1[G, q, 1]
            re[q] = xr * c - xi * sn
            im[q] = xr * sn + xi * c
        self.fft_inlane(re, im)
        LG = Q.bit_length() - 1
        tw2 = T[tb.tw2: tb.tw2 + 4 * Q * 4].reshape(4, Q, 4)      # (c, -c, s, 0)
        X = np.zeros((Q + 1, 64), f32)
        for k in range(Q):
            ur, ui = re[bitrev(k, LG)], im[bitrev(k, LG)]
            ks, kr = bitrev((Q - k) % Q, LG), bitrev(Q - 1 - k, LG)
            xr, xi = shfl_xor(re[kr], 16), shfl_xor(im[kr], 16)
            pr = np.where(G >= 2, xr, re[kr])
            pi = np.where(G >= 2, xi, im[kr])
            pr = np.where(G == 0, re[ks], pr)
            pi = np.where(G == 0, im[ks], pi)
            ar, ai = ur + pr, ui - pi
            dr, di = ui + pi, pr - ur
            c, sn = tw2[G, k, 0], tw2[G, k, 2]
            yr = ar + (dr * c - di * sn)
            yi = ai + (dr * sn + di * c)
            X[k] = f32(0.5) * np.sqrt(yr * yr + yi * yi)
        X[Q] = np.where(G == 0, np.abs(re[0] - im[0]), 0)
        return X

    # ---- gemm_seg: acc [M][4][64] += A_seg * B, bfun(s) -> [64] -----------------------------------
    def gemm_seg(self, acc, bfun, seg):
        Q = self.Q
        M, KS = seg_mblocks(seg), seg_ksteps(seg, Q)
        base = seg_offset(seg, Q)
        for kg in range((KS + 3) // 4):
            for m in range(M):
                blk = self.front[base + (kg * M + m) * 256: base + (kg * M + m + 1) * 256].reshape(64, 4)
                for ks in range(4):
                    if kg * 4 + ks < KS:
                        acc[m] = mfma_16x16x4(blk[:, ks], bfun(kg * 4 + ks), acc[m])

    def init_bias(self, M, off):
        acc = np.zeros((M, 4, 64), f32)
        for m in range(M):
            for r in range(4):
                acc[m, r] = self.tab[off + 16 * m + 4 * G + r]
        return acc

    def run(self, x):
        """x [16][18Q] -> dict(mag X0..X3 in mag layout, gx [32][4][64] D-fragment order)."""
        tb = self.tb
        X = [self.fft_pass(x, v) for v in range(4)]
        E0T0, E0T1, E0T2, E1T0, E1T1, E1T2, E2T1, E2T2, E3T1, IH0 = range(10)
        chain = lambda A: (lambda s: A[s >> 2, s & 3])
        relu = lambda A: np.maximum(A, 0)
        Y = self.init_bias(8, tb.b_e0)
        self.gemm_seg(Y, lambda s: X[0][s], E0T1)
        self.gemm_seg(Y, lambda s: X[1][s], E0T2)
        Y = relu(Y)
        Z0 = self.init_bias(4, tb.b_e1)
        self.gemm_seg(Z0, chain(Y), E1T1)
        Y = self.init_bias(8, tb.b_e0)
        self.gemm_seg(Y, lambda s: X[0][s], E0T0)
        self.gemm_seg(Y, lambda s: X[1][s], E0T1)
        self.gemm_seg(Y, lambda s: X[2][s], E0T2)
        Y = relu(Y)
        self.gemm_seg(Z0, chain(Y), E1T2)
        Z1 = self.init_bias(4, tb.b_e1)
        self.gemm_seg(Z1, chain(Y), E1T0)
        Y = self.init_bias(8, tb.b_e0)
        self.gemm_seg(Y, lambda s: X[1][s], E0T0)
        self.gemm_seg(Y, lambda s: X[2][s], E0T1)
        self.gemm_seg(Y, lambda s: X[3][s], E0T2)
        Y = relu(Y)
        self.gemm_seg(Z1, chain(Y), E1T1)
        Y = self.init_bias(8, tb.b_e0)
        self.gemm_seg(Y, lambda s: X[2][s], E0T0)
        self.gemm_seg(Y, lambda s: X[3][s], E0T1)
        Y = relu(Y)
        self.gemm_seg(Z1, chain(Y), E1T2)
        Z0, Z1 = relu(Z0), relu(Z1)
        V = self.init_bias(4, tb.b_e2)
        self.gemm_seg(V, chain(Z0), E2T1)
        self.gemm_seg(V, chain(Z1), E2T2)
        V = relu(V)
        Fe = self.init_bias(8, tb.b_e3)
        self.gemm_seg(Fe, chain(V), E3T1)
        Fe = relu(Fe)
        gx = np.zeros((32, 4, 64), f32)
        for q in range(4):
            Gq = self.init_bias(8, tb.b_g + 128 * q)
            self.gemm_seg(Gq, chain(Fe), IH0 + q)
            gx[8 * q: 8 * q + 8] = Gq
        return {"X": X, "feat": Fe, "gx": gx}


def mag_from_layout(X, Q):
    """mag-layout registers X[v] ([Q+1][64]) -> dense [16 chunks][4Q+1 bins]."""
    out = np.zeros((16, 4 * Q + 1), f32)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for s in range(Q):
            out[j, 4 * s + P_RES[g]] = X[s, lane]
        if g == 0:
            out[j, 4 * Q] = X[Q, lane]
    return out


def chain_to_dense(A):
    """chain-layout [NB][4][64] -> [16 cols][16 NB channels]."""
    NB = A.shape[0]
    out = np.zeros((16, 16 * NB), f32)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for blk in range(NB):
            for r in range(4):
                out[j, 16 * blk + 4 * g + r] = A[blk, r, lane]
    return out


def sigmoid_f(x):
    return (f32(1) / (f32(1) + np.exp2(f32(-1.4426950408889634) * x))).astype(f32)


def tanh_f(x):
    return (f32(2) * sigmoid_f(f32(2) * x) - f32(1)).astype(f32)


def rec_step(whh, tables, tb, gx, h, c):
    """kernel_rec.hip, one step, one 16-stream tile.
    gx [32][4][64] (D order), h/c [16 streams][128]  ->  prob [16], h', c'."""
    hbuf = np.zeros((8, 64, 4), f32)          # [wave][lane][r] = h[unit 16w + 4g + r][stream j]
    for w in range(8):
        for r in range(4):
            hbuf[w, :, r] = h[J, 16 * w + 4 * G + r]
    hn, cn = np.zeros_like(h), np.zeros_like(c)
    part_all = np.zeros((8, 16), f32)
    W = whh.reshape(8, 4, 8, 64, 4)
    for w in range(8):
        acc = [gx[8 * q + w].copy() for q in range(4)]
        for kg in range(8):
            hv = hbuf[kg]
            for r in range(4):
                for q in range(4):
                    acc[q] = mfma_16x16x4(W[w, q, kg, :, r], hv[:, r], acc[q])
        part = np.zeros(64, f32)
        for r in range(4):
            cl = c[J, 16 * w + 4 * G + r]
            ig, fg = sigmoid_f(acc[0][r]), sigmoid_f(acc[1][r])
            gg, og = tanh_f(acc[2][r]), sigmoid_f(acc[3][r])
            cnew = fg * cl + ig * gg
            hnew = og * tanh_f(cnew)
            cn[J, 16 * w + 4 * G + r] = cnew
            hn[J, 16 * w + 4 * G + r] = hnew
            part = part + tables[tb.w_out + 16 * w + 4 * G + r] * np.maximum(hnew, 0)
        part = part + shfl_xor(part, 16)
        part = part + shfl_xor(part, 32)
        part_all[w] = part[:16]
    p = tables[tb.b_out] + part_all.sum(0)
    return sigmoid_f(p.astype(f32)), hn, cn


# ---- kernel_front_wino.hip: encoder 0 as two Winograd F(2,3) transforms, weight stream in whole units ------------
def w_rb(Q):
    return 64 // Q


def w_parts(Q):
    return 8 // w_rb(Q)


def w_e0(p, j, Q):
    return p * 4 + j


def w_e1(h, i, Q):
    return 4 * w_parts(Q) + 3 * h + i


def w_e2(i, Q):
    return 4 * w_parts(Q) + 6 + i


def w_e3(u, Q):
    return 4 * w_parts(Q) + 8 + u


def w_ih(q, u, Q):
    return 4 * w_parts(Q) + 10 + 4 * q + u


def w_image_units(Q):
    return 4 * w_parts(Q) + 26


def make_wsched(Q):
    """layout.hpp make_wsched: image unit consumed by each program unit."""
    P = w_parts(Q)
    PH = P // 2
    sc = []
    for pair in range(2):
        for h in range(2):
            for pp in range(PH):
                sc += [w_e0(h * PH + pp, j, Q) for j in range(4)]
            sc += [w_e1(h, i, Q) for i in range(3 if pair == 0 else 2)]
    sc += [w_e2(0, Q), w_e2(1, Q), w_e3(0, Q), w_e3(1, Q)]
    sc += [w_ih(q, u, Q) for q in range(4) for u in range(4)]
    return sc


class FrontWinoEmu(FrontEmu):
    """The program of front_wino_kernel, executed unit by unit in PROGRAM order (every unit the kernel requests is
    taken from the schedule, so a packing / schedule / program mismatch shows up as wrong numbers)."""
    UNIT = 16 * 256

    def __init__(self, sr, image, tables):
        self.Q = 32 if sr == 16000 else 16
        self.image = image
        self.tab = tables
        self.tb = Tab(8 * self.Q, self.Q)
        assert len(tables) == self.tb.total
        assert len(image) == w_image_units(self.Q) * self.UNIT
        self.sched = make_wsched(self.Q)
        self.pu = 0                                   # program unit consumed next

    def gemm_w(self, acc, bfun, M, KG):
        """[k-group][row block] blocks in whole units, starting at the next program unit."""
        steps = KG * (M // 2)
        assert steps % 8 == 0
        for i in range(steps):
            unit = self.sched[self.pu + i // 8]
            base = unit * self.UNIT + (i % 8) * 2 * 256
            kg, mp = i // (M // 2), 2 * (i % (M // 2))
            for d in range(2):
                blk = self.image[base + d * 256: base + (d + 1) * 256].reshape(64, 4)
                for ks in range(4):
                    acc[mp + d] = mfma_16x16x4(blk[:, ks], bfun(kg * 4 + ks), acc[mp + d])
        self.pu += steps // 8

    def nyq(self, acc, xn, tau, row0):
        tb = self.tb
        for m in range(acc.shape[0]):
            for r in range(4):
                w = self.tab[tb.w_nyq + tau * 128 + row0 + 16 * m + 4 * G + r]
                acc[m, r] = acc[m, r] + w * xn

    def run(self, x):
        tb, Q = self.tb, self.Q
        RB, P = w_rb(Q), w_parts(Q)
        PH, KG0 = P // 2, Q // 4
        X = [self.fft_pass(x, v) for v in range(4)]
        xn = [X[v][Q][J] for v in range(4)]           # __shfl(X[Q], lane & 15): Nyquist magnitude of chunk j
        chain = lambda A: (lambda s: A[s >> 2, s & 3])
        relu = lambda A: np.maximum(A, 0)
        self.pu = 0
        Z0 = self.init_bias(4, tb.b_e1)
        Z1 = self.init_bias(4, tb.b_e1)
        for pair in range(2):
            if pair == 0:      # d = (0, x0, x1, x2)
                b = [lambda s: -X[1][s], lambda s: X[0][s] + X[1][s], lambda s: X[1][s] - X[0][s], lambda s: X[2][s] - X[0][s]]
                nyq_first, nyq_second = [(0, 1), (1, 2)], [(0, 0), (1, 1), (2, 2)]          # (frame, tap)
            else:              # d = (x1, x2, x3, 0)
                b = [lambda s: X[1][s] - X[3][s], lambda s: X[2][s] + X[3][s], lambda s: X[3][s] - X[2][s], lambda s: -X[2][s]]
                nyq_first, nyq_second = [(1, 0), (2, 1), (3, 2)], [(2, 0), (3, 1)]
            for h in range(2):
                Ya = np.zeros((4, 4, 64), f32)
                Yb = np.zeros((4, 4, 64), f32)
                for pp in range(PH):
                    row0 = 16 * RB * (h * PH + pp)
                    a0 = self.init_bias_at(RB, tb.b_e0 + row0)
                    a1 = self.init_bias_at(RB, tb.b_e0 + row0)
                    Pm = np.zeros((RB, 4, 64), f32)
                    Qm = np.zeros((RB, 4, 64), f32)
                    self.gemm_w(a0, b[0], RB, KG0)
                    self.gemm_w(Pm, b[1], RB, KG0)
                    self.gemm_w(Qm, b[2], RB, KG0)
                    self.gemm_w(a1, b[3], RB, KG0)
                    for fr, tau in nyq_first:
                        self.nyq(a0, xn[fr], tau, row0)
                    for fr, tau in nyq_second:
                        self.nyq(a1, xn[fr], tau, row0)
                    Ya[pp * RB:(pp + 1) * RB] = relu(a0 + (Pm + Qm))
                    Yb[pp * RB:(pp + 1) * RB] = relu(a1 + (Pm - Qm))
                if pair == 0:
                    self.gemm_w(Z0, chain(Ya), 4, 4)          # out 0, tap 1 <- y0
                    self.gemm_w(Z0, chain(Yb), 4, 4)          # out 0, tap 2 <- y1
                    self.gemm_w(Z1, chain(Yb), 4, 4)          # out 1, tap 0 <- y1
                else:
                    self.gemm_w(Z1, chain(Ya), 4, 4)          # out 1, tap 1 <- y2
                    self.gemm_w(Z1, chain(Yb), 4, 4)          # out 1, tap 2 <- y3
        Z0, Z1 = relu(Z0), relu(Z1)
        V = self.init_bias(4, tb.b_e2)
        self.gemm_w(V, chain(Z0), 4, 4)
        self.gemm_w(V, chain(Z1), 4, 4)
        V = relu(V)
        Fe = self.init_bias(8, tb.b_e3)
        self.gemm_w(Fe, chain(V), 8, 4)
        Fe = relu(Fe)
        gx = np.zeros((32, 4, 64), f32)
        for q in range(4):
            Gq = self.init_bias(8, tb.b_g + 128 * q)
            self.gemm_w(Gq, chain(Fe), 8, 8)
            gx[8 * q: 8 * q + 8] = Gq
        assert self.pu == len(self.sched)
        return {"X": X, "feat": Fe, "gx": gx}

    def init_bias_at(self, M, off):
        return self.init_bias(M, off)


# ---- kernel_front_f43.hip: encoder 0 as one Winograd F(4,3) tile, the image is the program --------------------------
def w4_e1_units(p, Q):
    return 2 + (p & 1) if Q == 32 else 5


def w4_units(Q):
    return sum(6 + w4_e1_units(p, Q) for p in range(w_parts(Q))) + 20


class FrontF43Emu(FrontWinoEmu):
    """The program of front_f43_kernel: units consumed strictly in image order (layout.hpp w4_*)."""

    def __init__(self, sr, image, tables):
        self.Q = 32 if sr == 16000 else 16
        self.image = image
        self.tab = tables
        self.tb = Tab(8 * self.Q, self.Q)
        assert len(tables) == self.tb.total
        assert len(image) == w4_units(self.Q) * self.UNIT
        self.sched = list(range(w4_units(self.Q)))
        self.pu = 0

    def run(self, x):
        tb, Q = self.tb, self.Q
        RB, P, KG0 = w_rb(Q), w_parts(Q), Q // 4
        X = [self.fft_pass(x, v) for v in range(4)]
        xn = [X[v][Q][J] for v in range(4)]
        chain = lambda A: (lambda s: A[s >> 2, s & 3])
        two = lambda A, B: (lambda s: A[s >> 2, s & 3] if s < 8 else B[(s - 8) >> 2, s & 3])
        relu = lambda A: np.maximum(A, 0)
        f = f32
        self.pu = 0
        # encoders 0 and 1 accumulate from zero and take the bias behind the last product (front_common.hpp add_bias)
        Z0 = np.zeros((4, 4, 64), f32)
        Z1 = np.zeros((4, 4, 64), f32)
        keep = None
        for p in range(P):
            row0 = 16 * RB * p
            m1, m2, m3, m4 = (np.zeros((RB, 4, 64), f32) for _ in range(4))
            self.gemm_w(m1, lambda s: (X[2][s] + X[3][s]) - f(4) * (X[0][s] + X[1][s]), RB, KG0)
            self.gemm_w(m2, lambda s: (X[3][s] - X[2][s]) + f(4) * (X[0][s] - X[1][s]), RB, KG0)
            self.gemm_w(m3, lambda s: (X[3][s] - X[1][s]) + f(2) * (X[2][s] - X[0][s]), RB, KG0)
            self.gemm_w(m4, lambda s: (X[3][s] - X[1][s]) - f(2) * (X[2][s] - X[0][s]), RB, KG0)
            sm, df, s2, d2 = m1 + m2, m1 - m2, m3 + m4, m3 - m4
            Y = [sm + s2, df + f(2) * d2, sm + f(4) * s2, df + f(8) * d2]
            self.gemm_w(Y[0], lambda s: X[3][s] - f(5) * X[1][s], RB, KG0)
            self.gemm_w(Y[3], lambda s: X[0][s] - f(1.25) * X[2][s], RB, KG0)
            for fr in range(4):
                for tau in range(3):
                    src = fr + tau - 1
                    if 0 <= src < 4:
                        self.nyq(Y[fr], xn[src], tau, row0)
            Y = [relu(y + self.init_bias(RB, tb.b_e0 + row0)) for y in Y]
            if Q == 32:
                self.gemm_w(Z0, two(Y[0], Y[1]), 4, 4)
                self.gemm_w(Z1, two(Y[1], Y[2]), 4, 4)
                if p & 1:
                    self.gemm_w(Z1, two(keep, Y[3]), 4, 4)
                else:
                    keep = Y[3]
            else:
                self.gemm_w(Z0, chain(Y[0]), 4, 4)
                self.gemm_w(Z0, chain(Y[1]), 4, 4)
                self.gemm_w(Z1, chain(Y[1]), 4, 4)
                self.gemm_w(Z1, chain(Y[2]), 4, 4)
                self.gemm_w(Z1, chain(Y[3]), 4, 4)
        Z0, Z1 = relu(Z0 + self.init_bias(4, tb.b_e1)), relu(Z1 + self.init_bias(4, tb.b_e1))
        V = self.init_bias(4, tb.b_e2)
        self.gemm_w(V, chain(Z0), 4, 4)
        self.gemm_w(V, chain(Z1), 4, 4)
        V = relu(V)
        Fe = self.init_bias(8, tb.b_e3)
        self.gemm_w(Fe, chain(V), 8, 4)
        Fe = relu(Fe)
        gx = np.zeros((32, 4, 64), f32)
        for q in range(4):
            Gq = self.init_bias(8, tb.b_g + 128 * q)
            self.gemm_w(Gq, chain(Fe), 8, 8)
            gx[8 * q: 8 * q + 8] = Gq
        assert self.pu == len(self.sched)
        return {"X": X, "feat": Fe, "gx": gx}
