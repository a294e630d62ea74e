/* A non-Python client of the C ABI, as the reference's C++ / Rust / Go / Java examples are clients of ONNX Runtime
 * (examples/cpp/silero-vad-onnx.cpp:103-142: one session.run per 32 ms chunk with explicit state) -- TEST INFRASTRUCTURE.
 * Plain C99 against include/silero_vad_hip.h only: no HIP headers, no C++.  tests/test_abi.py compiles and links it everywhere (the
 * header must be valid C and every entry point it uses must resolve); on a GPU box tests/test_gpu_parity.py also runs it: it streams
 * int16 chunks of `pcm_file` through vad_step_host (page-locked buffers from vad_host_register) and vad_iterator_feed and prints the
 * probabilities and events, which the test compares with the Python path's.
 *     client <weights> <pcm_int16_file> <sr> <streams> [sync | pump | gaps | compact]
 * With `sync` every tick is ONE blocking vad_step_host_sync (chunks read in place, probabilities stored into the page-locked buffer).
 * With `pump` the same loop runs on the native pump (vad_pump_create / slot / submit / poll / probs): the client writes the chunks into
 * the pump's page-locked ring, keeps two ticks in flight and prints each tick's probabilities and events as it is retired -- no HIP
 * call of its own at all.  With `gaps` the streams do not arrive in lock step: stream b has no chunk at tick t when
 * (7 t + 13 b) % 10 == 0, the client sets its flag to 0 (vad_pump_present / vad_pump_submit_present) and the stream's audio waits for
 * its next tick -- what a caller of the reference does by not calling its model (src/silero_vad/utils_vad.py:507-549).  With `compact`
 * the same streams write only the delivered chunks, back to back, and submit with vad_pump_submit_compact.                          */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "silero_vad_hip.h"

/* the two HIP runtime calls a C client needs for device buffers; declared here so that no HIP header is required */
extern int hipMalloc(void **ptr, size_t size);
extern int hipMemset(void *dst, int value, size_t size);
extern int hipFree(void *ptr);
extern int hipStreamSynchronize(void *stream);

static void *dev_zeros(size_t bytes) {
    void *p = NULL;
    if (hipMalloc(&p, bytes) != 0 || hipMemset(p, 0, bytes) != 0) {
        fprintf(stderr, "device allocation failed\n");
        exit(2);
    }
    return p;
}

/* the pump route: stream b plays the recording from offset b * 7919 (circular), two ticks in flight */
static int run_pump(vad_engine *e, const int16_t *pcm, long samples, int sr, int B, int N, long T, int gaps) {
    vad_pump_params prm;
    vad_pump *p = NULL;
    vad_iter_event *ev = (vad_iter_event *)calloc((size_t)B, sizeof(vad_iter_event));
    long *delivered = (long *)calloc((size_t)B, sizeof(long));      /* chunks stream b has handed over so far */
    long t, done = 0;
    int rc, R = 0;
    vad_pump_params_default(&prm, sr, B);
    prm.parts = 2;
    prm.ring_slots = 3;
    if ((rc = vad_pump_create(e, &prm, &p)) != VAD_OK) {
        fprintf(stderr, "vad_pump_create: %s\n", vad_strerror(rc));
        return 1;
    }
    vad_pump_geometry(p, NULL, NULL, &R, NULL);
    for (t = 0; t <= T; ++t) {
        if (t < T) {
            int16_t *slot = vad_pump_slot(p, (int)(t % R));
            uint8_t *flags = vad_pump_present(p, (int)(t % R));
            int b, i, row = 0;
            for (b = 0; b < B; ++b) {
                flags[b] = (uint8_t)!(gaps && (7 * t + 13 * b) % 10 == 0);
                if (!flags[b]) continue;                     /* no chunk this tick: the slot keeps whatever it holds */
                /* gaps == 2: a compact slot -- the delivered chunks back to back, the i-th delivering stream's in row i */
                for (i = 0; i < N; ++i) slot[(size_t)(gaps == 2 ? row : b) * N + i] = pcm[((long)b * 7919 + delivered[b] * N + i) % samples];
                ++delivered[b];
                ++row;
            }
            rc = gaps == 2 ? vad_pump_submit_compact(p, (int)(t % R), flags)
                 : gaps    ? vad_pump_submit_present(p, (int)(t % R), flags)
                           : vad_pump_submit(p, (int)(t % R));
            if (rc != VAD_OK) {
                fprintf(stderr, "vad_pump_submit: %s\n", vad_pump_last_error(p));
                return 1;
            }
        }
        if (t > 0) {                                  /* retire tick t - 1 while tick t runs */
            int r = -1, b;
            const long m = vad_pump_poll(p, 1, ev, B, &r);
            const float *prob;
            long k;
            if (m < 0 || r != (int)((t - 1) % R)) {
                fprintf(stderr, "vad_pump_poll: %ld (%s)\n", m, vad_pump_last_error(p));
                return 1;
            }
            prob = vad_pump_probs(p, r);
            printf("P %ld", done);
            for (b = 0; b < B; ++b) printf(" %.9g", prob[b]);
            printf("\n");
            for (k = 0; k < m; ++k) printf("E %ld %d %s %lld\n", done, (int)ev[k].slot, ev[k].kind ? "end" : "start", (long long)ev[k].sample);
            ++done;
        }
    }
    if (vad_pump_poll(p, 1, ev, B, NULL) != VAD_PUMP_IDLE) return 1;
    vad_pump_destroy(p);
    free(ev);
    free(delivered);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <weights> <pcm_int16_file> <sr> <streams>\n", argv[0]);
        return 64;
    }
    const int sr = atoi(argv[3]), B = atoi(argv[4]);
    int N = 0, C = 0;
    if (vad_geometry(sr, &N, &C) != VAD_OK || B < 1) return 64;

    FILE *f = fopen(argv[1], "rb");
    if (!f) return 66;
    fseek(f, 0, SEEK_END);
    const long wbytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *blob = malloc((size_t)wbytes);
    if (fread(blob, 1, (size_t)wbytes, f) != (size_t)wbytes) return 66;
    fclose(f);
    vad_engine *e = NULL;
    int rc = vad_create(blob, (size_t)wbytes, 0, &e);
    if (rc != VAD_OK) {
        fprintf(stderr, "vad_create: %s\n", vad_strerror(rc));
        return 1;
    }

    f = fopen(argv[2], "rb");
    if (!f) return 66;
    fseek(f, 0, SEEK_END);
    const long samples = ftell(f) / 2;
    fseek(f, 0, SEEK_SET);
    int16_t *pcm = (int16_t *)malloc((size_t)samples * 2);
    if (fread(pcm, 2, (size_t)samples, f) != (size_t)samples) return 66;
    fclose(f);
    const long T = samples / N;
    if (argc > 5 && (strcmp(argv[5], "pump") == 0 || strcmp(argv[5], "gaps") == 0 || strcmp(argv[5], "compact") == 0)) {
        rc = run_pump(e, pcm, samples, sr, B, N, T, strcmp(argv[5], "gaps") == 0 ? 1 : strcmp(argv[5], "compact") == 0 ? 2 : 0);
        vad_destroy(e);
        return rc;
    }

    const int blocking = argc > 5 && strcmp(argv[5], "sync") == 0;
    /* page-locked ingest buffer and probability buffer (what an audio server's network threads would write / read) */
    int16_t *host_pcm = (int16_t *)calloc((size_t)B * N, 2);
    float *host_prob = (float *)calloc((size_t)B, sizeof(float));
    if (vad_host_register(host_pcm, (size_t)B * N * 2) != VAD_OK || vad_host_register(host_prob, (size_t)B * sizeof(float)) != VAD_OK) {
        fprintf(stderr, "vad_host_register failed\n");
        return 1;
    }
    void *dev_pcm = dev_zeros((size_t)B * N * 2);
    float *ctx = (float *)dev_zeros((size_t)B * C * sizeof(float));
    float *state = (float *)dev_zeros((size_t)2 * B * 128 * sizeof(float));
    if ((rc = vad_reserve(e, sr, B, 1)) != VAD_OK) {
        fprintf(stderr, "vad_reserve: %s\n", vad_last_error(e));
        return 1;
    }

    /* VADIterator state of every stream (src/silero_vad/utils_vad.py:500-503) and its defaults (:477-498) */
    uint8_t *triggered = (uint8_t *)calloc((size_t)B, 1);
    int64_t *temp_end = (int64_t *)calloc((size_t)B, sizeof(int64_t));
    int64_t *current = (int64_t *)calloc((size_t)B, sizeof(int64_t));
    vad_iter_event *ev = (vad_iter_event *)calloc((size_t)B, sizeof(vad_iter_event));
    const double threshold = 0.5, min_silence = sr * 100 / 1000.0, pad = sr * 30 / 1000.0;

    for (long t = 0; t < T; ++t) {
        /* stream b plays the recording from offset b * 7919 (circular), chunk by chunk */
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < N; ++i) host_pcm[(size_t)b * N + i] = pcm[((long)b * 7919 + t * N + i) % samples];
        if (blocking) {
            /* the blocking call (what session.Run is to the reference's client): the kernel reads host_pcm in place and stores the
             * probabilities into host_prob; the call returns when they are there -- no staging buffer, no HIP call in the client */
            rc = vad_step_host_sync(e, sr, B, host_pcm, 2, ctx, state, host_prob, NULL);
        } else {
            rc = vad_step_host(e, sr, B, host_pcm, 2, dev_pcm, ctx, state, NULL, host_prob, NULL);
            if (rc == VAD_OK && hipStreamSynchronize(NULL) != 0) return 1;
        }
        if (rc != VAD_OK) {
            fprintf(stderr, "vad_step_host: %s\n", vad_last_error(e));
            return 1;
        }
        printf("P %ld", t);
        for (int b = 0; b < B; ++b) printf(" %.9g", host_prob[b]);
        printf("\n");
        const long m = vad_iterator_feed(host_prob, NULL, B, N, threshold, min_silence, pad, triggered, temp_end, current, ev, B);
        for (long k = 0; k < m; ++k) printf("E %ld %d %s %lld\n", t, (int)ev[k].slot, ev[k].kind ? "end" : "start", (long long)ev[k].sample);
    }
    vad_host_unregister(host_pcm);
    vad_host_unregister(host_prob);
    hipFree(dev_pcm);
    hipFree(ctx);
    hipFree(state);
    vad_destroy(e);
    return 0;
}
