/*
 * oracle/vad_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * A plain-C, fp32, CPU restatement of the Silero-VAD v6 hot path exactly as the reference
 * computes it (dense DFT-basis conv for the STFT, no FFT, no skipped taps).  It exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have a checker that runs on
 * the GPU box, where /root/reference does not exist.  Nothing under silero_vad_amd/ may call it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against golden vectors that
 * tests/golden/make_golden.py produced by running the reference's own TorchScript model
 * (silero_vad.jit) in the authoring container, and against the reference's published
 * known-answer segment counts (29 @16 kHz / 79 @8 kHz, examples/openvino/README.md:62).
 *
 * Reference lines followed ("JIT!/" = TorchScript source zipped inside
 * src/silero_vad/data/silero_vad.jit, archive dir VADr_v6_10_25_noths_re/code/__torch__/):
 *   framing + context       JIT!/vad/model/vad_annotator.py:58-67,86-87
 *   reflect pad (right F/4) JIT!/torch/nn/modules/padding/___torch_mangle_8.py:6,10
 *   STFT conv + magnitude   JIT!/vad/utils/pytorch_stft.py:17-34
 *   4 x ReLU(Conv1d k3 p1)  JIT!/vad/utils/model_utils.py:19-25
 *   LSTMCell, gates i,f,g,o JIT!/torch/nn/modules/rnn.py:13-76 (aten::lstm_cell :69)
 *   head ReLU-Conv1x1-Sigm  JIT!/torch/nn/modules/container/___torch_mangle_7.py:10-19
 *   mean over T=1           JIT!/vad/model/vad_annotator.py:206-207
 *   audio_forward           JIT!/vad/model/vad_annotator.py:128-156
 *   plain-torch twin        examples/onnx_sequence/export.py:22-79
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define HID 128

typedef struct {
    int sr, N, C, F, H, K;            /* chunk, context, filter length, hop, bins */
    const float *basis;               /* [2K][F]      rows [0,K) = w*cos, [K,2K) = -w*sin */
    const float *ew[4], *eb[4];       /* enc l: [Cout][Cin][3], [Cout] */
    int cin[4], cout[4], stride[4];
    const float *w_ih, *w_hh, *b_ih, *b_hh; /* [512][128] x2, [512] x2 */
    const float *w_out, *b_out;       /* [128], [1] */
} oracle_net;

typedef struct {
    uint8_t *blob;
    size_t nbytes;
    oracle_net net[2];                /* 0: 16 kHz, 1: 8 kHz */
} oracle_t;

/* ---- weight container (tools/export_weights.py) ------------------------------------------ */
static const float *find_tensor(const oracle_t *o, const char *name, uint64_t expect) {
    uint32_t n;
    memcpy(&n, o->blob + 8, 4);
    const uint8_t *rec = o->blob + 80;
    for (uint32_t i = 0; i < n; ++i, rec += 64 + 4 + 16 + 8 + 8) {
        if (strncmp((const char *)rec, name, 64) == 0) {
            uint64_t off, cnt;
            memcpy(&off, rec + 84, 8);
            memcpy(&cnt, rec + 92, 8);
            if (cnt != expect || off + cnt * 4 > o->nbytes) return NULL;
            return (const float *)(o->blob + off);
        }
    }
    return NULL;
}

static int bind_net(oracle_t *o, oracle_net *n, const char *prefix, int sr) {
    char nm[96];
    n->sr = sr;
    n->N = sr == 16000 ? 512 : 256;
    n->C = n->N / 8;
    n->F = n->N / 2;
    n->H = n->F / 2;
    n->K = n->F / 2 + 1;
    const int cin[4] = {n->K, 128, 64, 64}, cout[4] = {128, 64, 64, 128}, st[4] = {1, 2, 2, 1};
#define GET(dst, fmt_name, count)                                  \
    do {                                                           \
        snprintf(nm, sizeof nm, "%s.%s", prefix, fmt_name);        \
        dst = find_tensor(o, nm, (uint64_t)(count));               \
        if (!dst) { fprintf(stderr, "oracle: missing %s\n", nm); return -1; } \
    } while (0)
    GET(n->basis, "stft.forward_basis_buffer", 2 * n->K * n->F);
    for (int l = 0; l < 4; ++l) {
        char t[64];
        n->cin[l] = cin[l]; n->cout[l] = cout[l]; n->stride[l] = st[l];
        snprintf(t, sizeof t, "encoder.%d.reparam_conv.weight", l);
        GET(n->ew[l], t, cout[l] * cin[l] * 3);
        snprintf(t, sizeof t, "encoder.%d.reparam_conv.bias", l);
        GET(n->eb[l], t, cout[l]);
    }
    GET(n->w_ih, "decoder.rnn.weight_ih", 4 * HID * HID);
    GET(n->w_hh, "decoder.rnn.weight_hh", 4 * HID * HID);
    GET(n->b_ih, "decoder.rnn.bias_ih", 4 * HID);
    GET(n->b_hh, "decoder.rnn.bias_hh", 4 * HID);
    GET(n->w_out, "decoder.decoder.2.weight", HID);
    GET(n->b_out, "decoder.decoder.2.bias", 1);
#undef GET
    return 0;
}

oracle_t *oracle_create(const void *weights, size_t nbytes) {
    if (nbytes < 80 || memcmp(weights, "SVADW001", 8) != 0) return NULL;
    oracle_t *o = (oracle_t *)calloc(1, sizeof *o);
    o->blob = (uint8_t *)malloc(nbytes);
    memcpy(o->blob, weights, nbytes);
    o->nbytes = nbytes;
    if (bind_net(o, &o->net[0], "_model", 16000) || bind_net(o, &o->net[1], "_model_8k", 8000)) {
        free(o->blob); free(o);
        return NULL;
    }
    return o;
}

void oracle_destroy(oracle_t *o) {
    if (o) { free(o->blob); free(o); }
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
/* torch.relu (JIT!/vad/utils/model_utils.py:19-25, head JIT!/torch/nn/modules/container/___torch_mangle_7.py:10-19) is
 * clamp_min(0), which PROPAGATES NaN: one NaN / Inf sample makes the chunk's probability and the carried (h, c) NaN, and
 * every later chunk of the stream stays NaN until reset_states() (tests/golden/make_golden.py, protocol "nonfinite").
 * `x > 0 ? x : 0` would map NaN to 0 and hide that; this form keeps it. */
static inline float relu_(float x) { return x <= 0.0f ? 0.0f : x; }

/* ---- one stream, one step -------------------------------------------------------------------
 * x1    [C+N]   context followed by the chunk
 * h, c  [128]   in/out
 * stage optional dump: mag[K*4], e0[128*4], e1[64*2], e2[64], e3[128]  (channel-major like torch)
 */
static float step_one(const oracle_net *n, const float *x1, float *h, float *c, float *stage) {
    const int F = n->F, H = n->H, K = n->K, L = n->C + n->N, P = F / 4;
    float xp[640];
    memcpy(xp, x1, sizeof(float) * L);
    for (int j = 0; j < P; ++j) xp[L + j] = x1[L - 2 - j];        /* right reflect, edge not repeated */

    float bufA[129 * 4], bufB[128 * 4];
    /* STFT: conv1d(basis, stride = hop) then magnitude */
    for (int k = 0; k < K; ++k) {
        const float *br = n->basis + (size_t)k * F, *bi = n->basis + (size_t)(K + k) * F;
        for (int m = 0; m < 4; ++m) {
            const float *f = xp + m * H;
            float re = 0.f, im = 0.f;
#pragma omp simd reduction(+ : re, im)
            for (int t = 0; t < F; ++t) { re += br[t] * f[t]; im += bi[t] * f[t]; }
            bufA[k * 4 + m] = sqrtf(re * re + im * im);
        }
    }
    if (stage) memcpy(stage, bufA, sizeof(float) * K * 4);

    /* encoder: ReLU(Conv1d(k=3, pad=1, stride s)) x4 */
    float *in = bufA, *out = bufB;
    int T = 4, so = K * 4;
    for (int l = 0; l < 4; ++l) {
        const int ci = n->cin[l], co = n->cout[l], s = n->stride[l];
        const int To = (T + 2 - 3) / s + 1;
        for (int o = 0; o < co; ++o) {
            const float *w = n->ew[l] + (size_t)o * ci * 3;
            for (int u = 0; u < To; ++u) {
                float acc = n->eb[l][o];
                for (int tau = 0; tau < 3; ++tau) {
                    const int v = u * s + tau - 1;
                    if (v < 0 || v >= T) continue;
                    float a = 0.f;
#pragma omp simd reduction(+ : a)
                    for (int i = 0; i < ci; ++i) a += w[i * 3 + tau] * in[i * T + v];
                    acc += a;
                }
                out[o * To + u] = relu_(acc);
            }
        }
        if (stage) memcpy(stage + so, out, sizeof(float) * co * To);
        so += co * To;
        float *t = in; in = out; out = t;
        T = To;
    }
    const float *z = in;                                            /* [128] (T == 1) */

    /* LSTM cell, gate order i, f, g, o */
    float g[4 * HID];
    for (int r = 0; r < 4 * HID; ++r) {
        const float *wi = n->w_ih + (size_t)r * HID, *wh = n->w_hh + (size_t)r * HID;
        float a = 0.f, b = 0.f;
#pragma omp simd reduction(+ : a, b)
        for (int j = 0; j < HID; ++j) { a += wi[j] * z[j]; b += wh[j] * h[j]; }
        g[r] = (a + n->b_ih[r]) + (b + n->b_hh[r]);
    }
    float p = n->b_out[0];
    for (int j = 0; j < HID; ++j) {
        const float ig = sigmoidf_(g[j]), fg = sigmoidf_(g[HID + j]);
        const float gg = tanhf(g[2 * HID + j]), og = sigmoidf_(g[3 * HID + j]);
        const float cn = fg * c[j] + ig * gg;
        const float hn = og * tanhf(cn);
        c[j] = cn; h[j] = hn;
        p += n->w_out[j] * relu_(hn);
    }
    return sigmoidf_(p);
}

static const oracle_net *pick(const oracle_t *o, int sr) {
    return sr == 16000 ? &o->net[0] : sr == 8000 ? &o->net[1] : NULL;
}

/* Functional step (the ONNX-graph I/O shape, utils_vad.py:80-82):
 * x1[B][C+N], state[2][B][128] in/out, prob[B].  stage may be NULL, else [B][stage_floats]. */
int oracle_step(const oracle_t *o, int sr, int B, const float *x1, float *state, float *prob,
                float *stage, int stage_floats) {
    const oracle_net *n = pick(o, sr);
    if (!n) return -1;
    const int L = n->C + n->N;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b)
        prob[b] = step_one(n, x1 + (size_t)b * L, state + (size_t)b * HID,
                           state + (size_t)B * HID + (size_t)b * HID,
                           stage ? stage + (size_t)b * stage_floats : NULL);
    return 0;
}

/* audio_forward twin with explicit carried state (vad_annotator.py:128-156 does reset + loop):
 * pcm[B][L] (row stride ld), ctx[B][C] in/out, state[2][B][128] in/out, probs[B][T], T = ceil(L/N);
 * the tail of the last chunk is zero padded. */
int oracle_forward_audio(const oracle_t *o, int sr, int B, long L, const float *pcm, long ld,
                         float *ctx, float *state, float *probs) {
    const oracle_net *n = pick(o, sr);
    if (!n) return -1;
    const int N = n->N, C = n->C;
    const long T = (L + N - 1) / N;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        float x1[576];
        float *h = state + (size_t)b * HID, *c = state + (size_t)B * HID + (size_t)b * HID;
        memcpy(x1, ctx + (size_t)b * C, sizeof(float) * C);
        for (long t = 0; t < T; ++t) {
            const long s = t * N, avail = L - s < N ? L - s : N;
            memcpy(x1 + C, pcm + (size_t)b * ld + s, sizeof(float) * avail);
            if (avail < N) memset(x1 + C + avail, 0, sizeof(float) * (N - avail));
            probs[(size_t)b * T + t] = step_one(n, x1, h, c, NULL);
            memmove(x1, x1 + N, sizeof(float) * C);                 /* ctx = last C samples of x1 */
        }
        memcpy(ctx + (size_t)b * C, x1, sizeof(float) * C);
    }
    return 0;
}

int oracle_stage_floats(int sr) {
    const int K = sr == 16000 ? 129 : 65;
    return K * 4 + 128 * 4 + 64 * 2 + 64 + 128;
}
