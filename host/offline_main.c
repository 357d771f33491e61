/*
 * host/offline_main.c — plain-C offline separation harness written ONLY against the reference's own C API
 * (spleeter.h + stftFix.h).  It reproduces the call order and the residual arithmetic of the reference CLI
 * (/root/reference/Executable/main.c:759-798 two-stem, :845-928 three-stem, processMT single-thread :444-541)
 * without its file decoders / resampler (out of scope, SURVEY §2.1 #9,#11,#12).
 *
 * The same source links against either
 *     libspleeterrt_amd.so          (this repo: HIP kernels behind the same symbols), or
 *     oracle/_ref/libspleeter_ref.so (the real reference, test infrastructure)
 * which is how tests/test_dropin_host.py shows that the library is a drop-in for this path.
 *
 * usage: offline_main T F stems(2|3) weights.f16 in.f32 out_prefix
 *   weights.f16 : spleeterQuantized (2 sub-nets of IEEE half bit patterns, Executable/spleeter.h:59-62)
 *   in.f32      : interleaved stereo float32 PCM @ 44.1 kHz
 *   writes <out_prefix>_Vocal.f32 / _Accompaniment.f32 (/ _Drum.f32), interleaved stereo float32
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "spleeter.h"
#include "stftFix.h"

/* fp16 container -> fp32: exponent re-bias, half-denormals flushed, no Inf/NaN case (main.c:423-443) */
static float *load_coefficients(const char *path)
{
    const size_t n = sizeof(spleeterQuantized) / sizeof(uint16_t);
    uint16_t *h = (uint16_t *)malloc(n * sizeof(uint16_t));
    FILE *f = fopen(path, "rb");
    if (!f || fread(h, sizeof(uint16_t), n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    float *out = (float *)malloc(2 * getCoeffSize());
    for (size_t i = 0; i < n; ++i) {
        uint32_t v = h[i], mag = ((v & 0x7fffu) << 13) + 0x38000000u;
        if ((v & 0x7c00u) == 0) mag = 0;
        mag |= (v & 0x8000u) << 16;
        memcpy(&out[i], &mag, 4);
    }
    free(h);
    return out;
}

/* processMT, single-thread branch (main.c:444-541): |.| -> network -> mask, tile by tile, in place */
static void process_tiles(size_t F, size_t T, size_t frames, void *coeff, float unaffected, float *reL, float *imL, float *reR, float *imR, int mode)
{
    spleeter nn = (spleeter)allocateSpleeterStr();
    initSpleeter(nn, F, T, mode, coeff);
    float *mask = 0;
    getMaskPtr(nn, &mask);
    float *mag = (float *)malloc(2 * T * F * sizeof(float));
    const size_t ntiles = (frames + T - 1) / T;
    for (size_t j = 0; j < ntiles; ++j) {
        for (size_t t = 0; t < T; ++t) {
            const size_t row = j * T + t;
            for (size_t i = 0; i < F; ++i) {
                float l = 0.0f, r = 0.0f;
                if (row < frames) {
                    const size_t idx = row * FFTSIZE + i;
                    l = hypotf(reL[idx], imL[idx]) * (float)FFTSIZE;
                    r = hypotf(reR[idx], imR[idx]) * (float)FFTSIZE;
                }
                mag[t * F + i] = l; mag[T * F + t * F + i] = r;
            }
        }
        processSpleeter(nn, mag, mask);
        for (size_t t = 0; t < T && j * T + t < frames; ++t) {
            const size_t off = (j * T + t) * FFTSIZE;
            size_t i = 0;
            for (; i < F; ++i) {
                const float mL = mask[t * F + i], mR = mask[T * F + t * F + i];
                reL[off + i] *= mL; imL[off + i] *= mL; reR[off + i] *= mR; imR[off + i] *= mR;
            }
            for (; i < HALFWNDLEN; ++i) {
                reL[off + i] *= unaffected; imL[off + i] *= unaffected; reR[off + i] *= unaffected; imR[off + i] *= unaffected;
            }
        }
    }
    freeSpleeter(nn);
    free(nn);
    free(mag);
}

static void write_stereo(const char *prefix, const char *name, const float *L, const float *R, size_t n)
{
    char path[4096];
    snprintf(path, sizeof path, "%s_%s.f32", prefix, name);
    FILE *f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    for (size_t i = 0; i < n; ++i) { float v[2] = { L[i + FFTSIZE], R[i + FFTSIZE] }; fwrite(v, 4, 2, f); }   /* undo the 4096-sample pre-shift (main.c:767,806) */
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s T F stems weights.f16 in.f32 out_prefix\n", argv[0]); return 1; }
    const size_t T = (size_t)atoi(argv[1]), F = (size_t)atoi(argv[2]);
    const int stems = atoi(argv[3]);
    float *coeff2 = load_coefficients(argv[4]);                                /* net[0] = drum (ELU), net[1] = vocal (main.c:759-760) */
    void *coeffDrum = coeff2, *coeffVocal = (char *)coeff2 + getCoeffSize();
    FILE *f = fopen(argv[5], "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", argv[5]); return 2; }
    fseek(f, 0, SEEK_END); const size_t nframes = (size_t)ftell(f) / 8; fseek(f, 0, SEEK_SET);
    float *pcm = (float *)malloc(nframes * 8);
    if (fread(pcm, 8, nframes, f) != nframes) return 2;
    fclose(f);
    const size_t readcount = (nframes + FFTSIZE - 1) / FFTSIZE, finalSize = FFTSIZE * readcount + (FFTSIZE << 1);   /* main.c:762-763 */
    float *inL = (float *)calloc(finalSize, sizeof(float)), *inR = (float *)calloc(finalSize, sizeof(float));
    for (size_t i = 0; i < nframes; ++i) { inL[i + FFTSIZE] = pcm[2 * i]; inR[i + FFTSIZE] = pcm[2 * i + 1]; }    /* 4096-sample pre-shift */
    free(pcm);
    const float unaffected = 0.1f;                                                                                  /* main.c:773 */
    OfflineSTFT *st = (OfflineSTFT *)malloc(sizeof(OfflineSTFT));
    InitSTFT(st, 1);
    float *reL = 0, *imL = 0, *reR = 0, *imR = 0;
    const size_t frames = stft(st, inL, inR, finalSize, &reL, &imL, &reR, &imR);
    if (stems == 2) {
        process_tiles(F, T, frames, coeffVocal, unaffected, reL, imL, reR, imR, 0);
        float *vL = 0, *vR = 0;
        istft(st, reL, imL, reR, imR, frames, &vL, &vR);
        float *aL = (float *)malloc(finalSize * sizeof(float)), *aR = (float *)malloc(finalSize * sizeof(float));
        for (size_t i = 0; i < finalSize; ++i) { aL[i] = inL[i] - vL[i]; aR[i] = inR[i] - vR[i]; }                  /* main.c:794-798 */
        write_stereo(argv[6], "Vocal", vL, vR, nframes);
        write_stereo(argv[6], "Accompaniment", aL, aR, nframes);
        free(vL); free(vR); free(aL); free(aR);
    } else {
        const size_t ne = frames * FFTSIZE;
        float *o[4] = { (float *)malloc(ne * 4), (float *)malloc(ne * 4), (float *)malloc(ne * 4), (float *)malloc(ne * 4) };
        float *cur[4] = { reL, imL, reR, imR };
        for (int k = 0; k < 4; ++k) memcpy(o[k], cur[k], ne * 4);                                                  /* main.c:849-856 */
        process_tiles(F, T, frames, coeffDrum, unaffected, reL, imL, reR, imR, 1);                                  /* drum on the mixture */
        for (int k = 0; k < 4; ++k) for (size_t i = 0; i < ne; ++i) o[k][i] -= cur[k][i];                           /* rest = mix - drum, complex domain (:860-866) */
        float *dL = 0, *dR = 0, *rL = 0, *rR = 0, *vL = 0, *vR = 0;
        istft(st, reL, imL, reR, imR, frames, &dL, &dR);
        float *rest[4];
        for (int k = 0; k < 4; ++k) { rest[k] = (float *)malloc(ne * 4); memcpy(rest[k], o[k], ne * 4); }
        istft(st, rest[0], rest[1], rest[2], rest[3], frames, &rL, &rR);
        process_tiles(F, T, frames, coeffVocal, unaffected, o[0], o[1], o[2], o[3], 0);                             /* vocal on the rest (:911) */
        istft(st, o[0], o[1], o[2], o[3], frames, &vL, &vR);
        for (size_t i = 0; i < finalSize; ++i) { rL[i] -= vL[i]; rR[i] -= vR[i]; }                                   /* accompaniment (:924-928) */
        write_stereo(argv[6], "Drum", dL, dR, nframes);
        write_stereo(argv[6], "Vocal", vL, vR, nframes);
        write_stereo(argv[6], "Accompaniment", rL, rR, nframes);
        for (int k = 0; k < 4; ++k) { free(o[k]); free(rest[k]); }
        free(dL); free(dR); free(rL); free(rR); free(vL); free(vR);
    }
    free(reL); free(imL); free(reR); free(imR);
    FreeSTFT(st); free(st);
    free(inL); free(inR); free(coeff2);
    return 0;
}
