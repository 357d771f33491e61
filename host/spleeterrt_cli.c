/*
 * host/spleeterrt_cli.c — offline separation CLI with the reference program's command line and output files,
 * running the whole flow device resident through srtSeparateCliHost (include/spleeterrt_amd.h).
 *
 *     spleeterrt_cli spawnNthreads timeStep analyseBinLimit stems audioFile.wav [weights.f16]
 *
 * Mirrors Executable/main.c:
 *   argument order, clamps and messages' meaning        main.c:704-748
 *   spawnNthreads = how many workers share the file's tiles (main.c:544-673: one pthread, one network instance and one contiguous
 *   tile range each).  Here a worker is a GPU: min(spawnNthreads, devices of the node) engines, one host thread each, weights uploaded
 *   once and broadcast over RCCL, ranges joined at their 3072-sample seams (srtMulti*, include/spleeterrt_amd.h).  One device: the
 *   single-engine path.  $SPLEETERRT_DEVICES="0,1,1" names the engines' devices explicitly (an index may repeat).
 *   any file length: the reference walks the tiles one at a time (main.c:455-495); here the engine holds at most
 *   $SPLEETERRT_MAX_TILES (default 64) tiles and srtSeparateCliHost walks longer files chunk by chunk
 *   4096-sample pre-shift, 4096*ceil(n/4096)+8192 pad   main.c:762-767
 *   mono input duplicated to both channels              main.c:768-769
 *   stems <= 2 -> <name>_Vocal.wav, <name>_Accompaniment.wav; else + <name>_Drum.wav      main.c:812-843, 894-965
 *   outputs are IEEE-float32 stereo 44.1 kHz RIFF/WAVE, <name> = basename of the input INCLUDING its extension
 *
 * Out of scope here (SURVEY §2.1 #9,#11,#12): FLAC/MP3 decoding and the sample-rate converter — the input must be a
 * 44.1 kHz RIFF/WAVE file (PCM 8/16/24/32-bit or float32, 1 or 2 channels).  The reference embeds its weights
 * (model.c, absent from the public tree); here the fp16 container `spleeterQuantized` (spleeter.h:59-62) is read from
 * the 6th argument or $SPLEETERRT_WEIGHTS.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "spleeterrt_amd.h"

#define FFT 4096

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | p[1] << 8); }

/* RIFF/WAVE -> interleaved float32.  Returns frames, 0 on error (message printed). */
static size_t read_wav(const char *path, float **pcm, unsigned *channels, unsigned *rate)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return 0; }
    unsigned char h[12] = { 0 };
    if (fread(h, 1, 12, f) != 12) { fprintf(stderr, "%s: not a RIFF/WAVE file (shorter than its header)\n", path); fclose(f); return 0; }
    if (!memcmp(h, "RF64", 4) || !memcmp(h, "BW64", 4)) {
        fprintf(stderr, "%s: RF64/BW64 (a WAVE file beyond 4 GiB) is not supported: the reference's float32 RIFF outputs could not hold the result; split the input\n", path); fclose(f); return 0;
    }
    if (memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) {
        fprintf(stderr, "%s: not a RIFF/WAVE file (FLAC/MP3 decoding is outside this harness)\n", path); fclose(f); return 0;
    }
    unsigned fmt = 0, ch = 0, bits = 0, sr = 0;
    for (;;) {
        unsigned char ck[8];
        if (fread(ck, 1, 8, f) != 8) { fprintf(stderr, "%s: no data chunk\n", path); fclose(f); return 0; }
        const uint32_t sz = rd32(ck + 4);
        if (!memcmp(ck, "fmt ", 4)) {
            unsigned char b[40];
            const uint32_t take = sz < sizeof b ? sz : (uint32_t)sizeof b;
            if (sz < 16 || fread(b, 1, take, f) != take) { fprintf(stderr, "%s: bad fmt chunk\n", path); fclose(f); return 0; }
            fmt = rd16(b); ch = rd16(b + 2); sr = rd32(b + 4); bits = rd16(b + 14);
            if (fmt == 0xFFFE && sz >= 26) fmt = rd16(b + 24);                         /* WAVE_FORMAT_EXTENSIBLE: sub-format */
            fseek(f, (long)(sz - take) + (long)(sz & 1), SEEK_CUR);
        } else if (!memcmp(ck, "data", 4)) {
            if (!ch || !bits) { fprintf(stderr, "%s: data before fmt\n", path); fclose(f); return 0; }
            const unsigned bps = bits / 8;
            const int ok = (fmt == 1 && (bps >= 1 && bps <= 4)) || (fmt == 3 && bps == 4);
            if (!ok || ch < 1 || ch > 2) { fprintf(stderr, "%s: unsupported WAVE encoding (format %u, %u bits, %u channels)\n", path, fmt, bits, ch); fclose(f); return 0; }
            if (sz == 0xFFFFFFFFu) { fprintf(stderr, "%s: data chunk of unknown length (streamed / > 4 GiB WAVE) is not supported; rewrite the file with a sized data chunk\n", path); fclose(f); return 0; }
            const size_t frames = (size_t)sz / ((size_t)bps * ch), ns = frames * ch;
            unsigned char *raw = (unsigned char *)malloc((size_t)sz ? (size_t)sz : 1);
            float *out = (float *)malloc((ns ? ns : 1) * sizeof(float));
            if (!raw || !out || fread(raw, 1, ns * bps, f) != ns * bps) { fprintf(stderr, "%s: truncated data chunk\n", path); free(raw); free(out); fclose(f); return 0; }
            for (size_t i = 0; i < ns; ++i) {
                const unsigned char *p = raw + i * bps;
                if (fmt == 3) memcpy(&out[i], p, 4);
                else if (bps == 1) out[i] = ((float)p[0] - 128.0f) / 128.0f;
                else if (bps == 2) out[i] = (float)(int16_t)rd16(p) / 32768.0f;
                else if (bps == 3) out[i] = (float)((int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8) / 8388608.0f;
                else out[i] = (float)((double)(int32_t)rd32(p) / 2147483648.0);
            }
            free(raw); fclose(f);
            *pcm = out; *channels = ch; *rate = sr;
            return frames;
        } else {
            fseek(f, (long)sz + (long)(sz & 1), SEEK_CUR);
        }
    }
}

/* planar (with the 4096-sample pre-shift undone) -> float32 stereo RIFF/WAVE */
static int write_wav(const char *path, const float *L, const float *R, size_t frames)
{
    FILE *f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path); return -1; }
    const uint32_t data = (uint32_t)(frames * 8), rate = 44100;
    unsigned char h[58];
    memcpy(h, "RIFF", 4); memcpy(h + 8, "WAVEfmt ", 8); memcpy(h + 38, "fact", 4); memcpy(h + 50, "data", 4);
#define PUT32(o, v) do { uint32_t v_ = (v); h[o] = v_ & 255; h[o + 1] = (v_ >> 8) & 255; h[o + 2] = (v_ >> 16) & 255; h[o + 3] = (v_ >> 24) & 255; } while (0)
#define PUT16(o, v) do { uint32_t v_ = (v); h[o] = v_ & 255; h[o + 1] = (v_ >> 8) & 255; } while (0)
    PUT32(4, 50 + data); PUT32(16, 18); PUT16(20, 3); PUT16(22, 2); PUT32(24, rate); PUT32(28, rate * 8); PUT16(32, 8); PUT16(34, 32); PUT16(36, 0);
    PUT32(42, 4); PUT32(46, (uint32_t)frames); PUT32(54, data);
    fwrite(h, 1, sizeof h, f);
    float *il = (float *)malloc((frames ? frames : 1) * 8);
    if (!il) { fprintf(stderr, "out of host memory writing %s\n", path); fclose(f); return -1; }
    for (size_t i = 0; i < frames; ++i) { il[2 * i] = L[i + FFT]; il[2 * i + 1] = R[i + FFT]; }       /* channel_joinFloat(..., preshift 4096), main.c:806 */
    const size_t w = fwrite(il, 8, frames, f);
    free(il);
    fclose(f);
    return w == frames ? 0 : -1;
}

static const char *base_name(const char *p) { const char *s = strrchr(p, '/'); return s ? s + 1 : p; }   /* main.c:114-125 */
static int is_pow2(size_t x) { return x && !(x & (x - 1)); }

int main(int argc, char **argv)
{
    if (argc < 6) {
        printf("Invalid program arguments.\nExample:\n%s spawnNthreads timeStep analyseBinLimit stems audioFile.wav [weights.f16]\n%s 3 512 1024 3 musicFile.wav\n", argv[0], argv[0]);
        return -2;
    }
    double t0 = now();
    int workers = atoi(argv[1]);
    if (workers < 1) { workers = 1; printf("spawnNthreads clamp to 1\n"); }
    size_t T = 512, F = 1024;
    const int v1 = atoi(argv[2]), v2 = atoi(argv[3]), stems = atoi(argv[4]) <= 2 ? 2 : 3;
    if (v1 < 64) { T = 64; printf("timeStep clamp to 64\n"); } else T = (size_t)v1;
    if (v2 < 512) { F = 512; printf("analyseBinLimit clamp to 512\n"); } else F = (size_t)v2;
    if (!is_pow2(T) || !is_pow2(F)) printf("Value should be power of 2 or you know what you are typing\nAccepting value and continue\n");
    if (F > 2048) { F = 2048; printf("Analysis bin limit reached, clamp value to 2048\n"); }
    const char *wpath = argc > 6 ? argv[6] : getenv("SPLEETERRT_WEIGHTS");
    if (!wpath) { fprintf(stderr, "no weights: pass the fp16 container (spleeterQuantized) as the 6th argument or in $SPLEETERRT_WEIGHTS\n"); return -1; }

    float *pcm = 0; unsigned channels = 0, rate = 0;
    const size_t nframes = read_wav(argv[5], &pcm, &channels, &rate);
    if (!nframes) return -1;
    if (rate != 44100) { fprintf(stderr, "%s: %u Hz — only 44.1 kHz input is accepted (the resampler is outside this harness)\n", argv[5], rate); return -1; }
    /* the outputs are float32 stereo RIFF files: 8 bytes per frame under a 32-bit chunk size (main.c writes the same container) */
    if ((uint64_t)nframes * 8u + 58u > 0xFFFFFFFFull) { fprintf(stderr, "%s: %zu frames do not fit a float32 stereo RIFF/WAVE output (4 GiB limit); split the input\n", argv[5], nframes); return -1; }
    const size_t readcount = (nframes + FFT - 1) / FFT, finalSize = FFT * readcount + 2 * FFT;
    float *inL = (float *)calloc(finalSize, sizeof(float)), *inR = (float *)calloc(finalSize, sizeof(float));
    if (!inL || !inR) { fprintf(stderr, "out of host memory (%zu samples per channel)\n", finalSize); return -1; }
    for (size_t i = 0; i < nframes; ++i) { inL[FFT + i] = pcm[i * channels]; inR[FFT + i] = pcm[i * channels + (channels - 1)]; }
    free(pcm);

    const size_t nhalf = srtCoeffBytes() / 4;
    uint16_t *halfs = (uint16_t *)malloc(2 * nhalf * sizeof(uint16_t));
    FILE *wf = fopen(wpath, "rb");
    if (!halfs) { fprintf(stderr, "out of host memory\n"); return -1; }
    if (!wf || fread(halfs, sizeof(uint16_t), 2 * nhalf, wf) != 2 * nhalf) { fprintf(stderr, "cannot read %zu halves from %s\n", 2 * nhalf, wpath); return -1; }
    fclose(wf);
    printf("Audio & model file loading takes: %1.14lf sec\n", now() - t0);

    const size_t rows = srtStftRows(finalSize), len = srtIstftLength(rows);
    srt_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.F = (int)F; cfg.T = (int)T; cfg.n_stems = 2;
    cfg.stem_mode[0] = 1; cfg.stem_mode[1] = 0;                        /* net[0] drum: ELU; net[1] vocal: LeakyReLU/ReLU  (main.c:782,858,911) */
    cfg.oob_weight[0] = cfg.oob_weight[1] = 0.1f;                      /* unaffectedWeight, main.c:773 */
    /* Engine capacity: the whole file when it is short, otherwise a fixed number of tiles (activations of every tile of a chunk live in
     * HBM at once: ~0.1 GB per tile at 512 x 1024); srtSeparateCliHost then walks the file chunk by chunk, so any length works. */
    const char *mt = getenv("SPLEETERRT_MAX_TILES");
    size_t cap = mt && atoi(mt) > 0 ? (size_t)atoi(mt) : 64, ntiles = (rows + T - 1) / T;
    /* workers -> devices (main.c:544-575: spawnNthreads tile-range workers) */
    int devs[64], ndev = 0;
    const char *dl = getenv("SPLEETERRT_DEVICES");
    if (dl && *dl) {
        for (const char *q = dl; *q && ndev < 64; ) { devs[ndev++] = atoi(q); q = strchr(q, ','); if (!q) break; ++q; }
    } else {
        const int have = srtDeviceCount();
        ndev = workers < have ? workers : have;
        if (ndev < 1) ndev = 1;
        if (ndev > 64) ndev = 64;
        if ((size_t)ndev > ntiles) ndev = (int)ntiles;                  /* a worker without a tile would idle */
        for (int g = 0; g < ndev; ++g) devs[g] = g;
        if (workers > ndev) printf("spawnNthreads %d: %d device worker(s) used (%d device(s), %zu tile(s))\n", workers, ndev, have, ntiles);
    }
    const size_t per = (ntiles + (size_t)ndev - 1) / (size_t)ndev;       /* tiles of the largest range */
    cfg.variant = SRT_VARIANT_EXE; cfg.max_tiles = (int)(per < cap ? per : cap); cfg.impl = SRT_IMPL_MFMA; cfg.precision = SRT_PREC_F32;
    float *out = (float *)malloc((size_t)stems * 2 * len * sizeof(float));
    if (!out) { fprintf(stderr, "out of host memory (%zu output samples)\n", (size_t)stems * 2 * len); return -1; }
    if (ndev > 1 || (dl && *dl)) {
        srt_multi *m = 0;
        char info[256];
        if (srtMultiCreate(&cfg, devs, ndev, &m)) { fprintf(stderr, "%s\n", srtLastError()); return -1; }
        if (srtMultiSetCoeffFp16Host(m, 0, halfs) || srtMultiSetCoeffFp16Host(m, 1, halfs + nhalf)) { fprintf(stderr, "%s\n", srtLastError()); return -1; }
        free(halfs);
        t0 = now();
        if (srtMultiSeparateCliHost(m, inL, inR, finalSize, stems, out)) { fprintf(stderr, "%s\n", srtLastError()); return -1; }
        srtMultiInfo(m, info, sizeof info);
        printf("Inference neural networks on %d GPU worker(s) takes %1.14lf sec (%zu tiles of %zu x %zu, at most %zu per worker in chunks of %d, %d outputs; %s)\n",
               ndev, now() - t0, ntiles, T, F, per, cfg.max_tiles, stems, info);
        srtMultiDestroy(m);
    } else {
        srt_engine *e = 0;
        if (srtCreate(&cfg, 0, &e)) { fprintf(stderr, "%s\n", srtLastError()); return -1; }
        if (srtSetCoeffFp16Host(e, 0, halfs) || srtSetCoeffFp16Host(e, 1, halfs + nhalf)) { fprintf(stderr, "%s\n", srtLastError()); return -1; }
        free(halfs);
        t0 = now();
        if (srtSeparateCliHost(e, inL, inR, finalSize, stems, out)) { fprintf(stderr, "%s\n", srtLastError()); return -1; }
        printf("Inference neural networks on the GPU takes %1.14lf sec (%zu tiles of %zu x %zu in chunks of %d, %d outputs)\n", now() - t0, ntiles, T, F, cfg.max_tiles, stems);
        srtReleaseStaging(e);                                          /* whole-file device copies of a one-shot program */
        srtDestroy(e);
    }

    static const char *names2[] = { "Vocal", "Accompaniment" }, *names3[] = { "Drum", "Vocal", "Accompaniment" };
    const char **names = stems == 2 ? names2 : names3;
    for (int k = 0; k < stems; ++k) {
        char path[4096];
        t0 = now();
        snprintf(path, sizeof path, "%s_%s.wav", base_name(argv[5]), names[k]);
        if (write_wav(path, out + (size_t)(2 * k) * len, out + (size_t)(2 * k + 1) * len, nframes)) return -1;
        printf("Saving file -> %s takes %1.14lf sec\n", path, now() - t0);
    }
    free(out); free(inL); free(inR);
    return 0;
}
