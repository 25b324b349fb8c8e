/* owwhip_demo.c -- libowwhip.so driven from plain C (no Python, no torch): the calls a non-Python host makes in place of
 * openwakeword.Model(...).predict(frame) (model.py:232-386), for S concurrent streams.
 *
 *   gcc -std=c99 -O2 -Wall -I include examples/c/owwhip_demo.c -L openwakeword_amd -lowwhip -Wl,-rpath,$PWD/openwakeword_amd -o owwhip_demo
 *   ./owwhip_demo DIR S T [use_mfma]
 *
 * DIR holds the weight blobs in the layouts of include/owwhip.h (mel.bin, embedding.bin, head_0.bin, head_1.bin, ...; written by
 * openwakeword_amd.engine.pack_*_blob(...).tofile) and pcm.bin = int16 [T][S][1280]; the program writes scores.bin =
 * float [T][S][n_labels] and prints one line per step. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "owwhip.h"

static void* slurp(const char* dir, const char* name, size_t* nbytes) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc(n > 0 ? (size_t)n : 1);
    if (!p || fread(p, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(p); return NULL; }
    fclose(f);
    *nbytes = (size_t)n;
    return p;
}

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc__ = (call);                                                                       \
        if (rc__ < 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc__, oww_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s DIR n_streams n_steps [use_mfma]\n", argv[0]); return 2; }
    const char* dir = argv[1];
    const int S = atoi(argv[2]), T = atoi(argv[3]);
    if (oww_abi_version() != OWW_ABI_VERSION) { fprintf(stderr, "library ABI %d, header ABI %d\n", oww_abi_version(), OWW_ABI_VERSION); return 1; }

    oww_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0;
    cfg.n_streams = S;
    cfg.max_chunks = 1;
    cfg.use_mfma = argc > 4 ? atoi(argv[4]) : 3;
    oww_ctx* h = NULL;
    CHECK(oww_create(&cfg, &h));

    size_t nb = 0;
    void* blob = slurp(dir, "mel.bin", &nb);
    if (!blob) { fprintf(stderr, "cannot read %s/mel.bin\n", dir); return 1; }
    CHECK(oww_load_mel(h, blob, nb));
    free(blob);
    blob = slurp(dir, "embedding.bin", &nb);
    if (!blob) { fprintf(stderr, "cannot read %s/embedding.bin\n", dir); return 1; }
    CHECK(oww_load_embedding(h, blob, nb));
    free(blob);
    int n_heads = 0;
    for (;; ++n_heads) {
        char name[64];
        snprintf(name, sizeof name, "head_%d.bin", n_heads);
        blob = slurp(dir, name, &nb);
        if (!blob) break;
        CHECK(oww_add_head(h, blob, nb));
        free(blob);
    }
    if (!n_heads) { fprintf(stderr, "no head_0.bin in %s\n", dir); return 1; }
    CHECK(oww_commit(h));
    const int NL = oww_n_labels(h);

    size_t pcm_bytes = 0;
    int16_t* pcm = (int16_t*)slurp(dir, "pcm.bin", &pcm_bytes);
    if (!pcm || pcm_bytes != (size_t)T * S * OWW_CHUNK * sizeof(int16_t)) { fprintf(stderr, "pcm.bin: expected %d x %d x 1280 int16\n", T, S); return 1; }
    float* scores = (float*)malloc((size_t)T * S * NL * sizeof(float));
    if (!scores) return 1;
    for (int t = 0; t < T; ++t) {
        /* host PCM in, host scores out, blocking: one Model.predict() for each of the S streams */
        CHECK(oww_step(h, pcm + (size_t)t * S * OWW_CHUNK, 0, 1, scores + (size_t)t * S * NL, 0));
        float top = 0.f;
        for (int i = 0; i < S * NL; ++i) if (scores[(size_t)t * S * NL + i] > top) top = scores[(size_t)t * S * NL + i];
        printf("step %d: %d streams x %d labels, highest score %.6f\n", t, S, NL, top);
    }
    char out[4096];
    snprintf(out, sizeof out, "%s/scores.bin", dir);
    FILE* f = fopen(out, "wb");
    if (!f || fwrite(scores, sizeof(float), (size_t)T * S * NL, f) != (size_t)T * S * NL) { fprintf(stderr, "cannot write %s\n", out); return 1; }
    fclose(f);
    CHECK(oww_range_status(h, 0));
    CHECK(oww_destroy(h));
    free(pcm);
    free(scores);
    return 0;
}
