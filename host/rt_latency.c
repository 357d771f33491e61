/*
 * host/rt_latency.c — wall time of every Spleeter4StemsProcessSamples call, as the plugin's audio callback sees it.
 *
 *     rt_latency F T hops weights.f32 pace_us out.json [instances]
 *
 * The reference's real-time contract (VST/Source/Spleeter4Stems.c:351-371, called from PluginProcessor.cpp:173-181): a call
 * with one hop of audio (1024 samples) never blocks longer than that hop's own transform work, except at the hop that completes a
 * batch of T hops, where the networks started one batch earlier are joined.  This program drives `instances` independent
 * Spleeter4Stems objects from as many host threads (two plugin instances in one DAW), one hop per call, `hops` calls each, and
 * records clock_gettime around every call.  pace_us = 0: calls back to back (the GPU never idles: worst case for contention between
 * the instances' hop streams and network streams); pace_us = 23220: one call per real-time hop period (the GPU idles between calls).
 * weights.f32 holds 4 spleeterCoeff blobs (drum, bass, accompaniment, vocal) of 39 290 900 bytes each.
 * Output: one JSON object with, per instance, Init time and p50 / p99 / max of ordinary hops and of the T-hop join hops.
 * Plain C against include/Spleeter4Stems.h only; used by tests/test_latency.py.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "Spleeter4Stems.h"
#include "spleeterrt_amd.h"      /* srtLastError(): why an instance came up muted, if it did */

#define COEFF_BYTES 39290900u

static double now_us(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec;
}

typedef struct {
    int id, F, T, hops, pace_us;
    void *coeff[4];
    double init_ms, *us;        /* per-call wall time */
    float peak;
    char init_error[256];       /* srtLastError() of this thread right after Init ("" = the instance is live) */
    pthread_barrier_t *start;
} Job;

static void *run(void *arg)
{
    Job *j = (Job *)arg;
    Spleeter4Stems *msr = (Spleeter4Stems *)malloc(sizeof(Spleeter4Stems));       /* PluginProcessor.cpp:123 */
    float *inL = (float *)malloc(1024 * sizeof(float)), *inR = (float *)malloc(1024 * sizeof(float));
    float *out = (float *)calloc(8 * 1024, sizeof(float)), *ptr[8];
    unsigned lcg = 12345u + 977u * (unsigned)j->id;
    double t0 = now_us();
    Spleeter4StemsInit(msr, j->F, j->T, j->coeff);
    j->init_ms = (now_us() - t0) * 1e-3;
    snprintf(j->init_error, sizeof j->init_error, "%s", srtLastError());
    for (char *c = j->init_error; *c; ++c) if (*c == '"' || *c == '\\' || *c < 32) *c = ' ';
    pthread_barrier_wait(j->start);
    double next = now_us();
    for (int h = 0; h < j->hops; ++h) {
        for (int i = 0; i < 1024; ++i) {
            lcg = lcg * 1664525u + 1013904223u; inL[i] = ((float)(lcg >> 8) / 16777216.0f - 0.5f) * 0.2f;
            lcg = lcg * 1664525u + 1013904223u; inR[i] = ((float)(lcg >> 8) / 16777216.0f - 0.5f) * 0.2f;
        }
        for (int k = 0; k < 8; ++k) ptr[k] = out + 1024 * k;
        if (j->pace_us) {                                   /* the host calls once per hop period */
            next += j->pace_us;
            double w = next - now_us();
            if (w > 0) { struct timespec ts = { (time_t)(w / 1e6), (long)((w - 1e6 * (long)(w / 1e6)) * 1e3) }; nanosleep(&ts, 0); }
        }
        t0 = now_us();
        Spleeter4StemsProcessSamples(msr, inL, inR, 1024, ptr);
        j->us[h] = now_us() - t0;
        for (int i = 0; i < 8 * 1024; ++i) { float a = out[i] < 0 ? -out[i] : out[i]; if (a > j->peak) j->peak = a; }
    }
    Spleeter4StemsFree(msr);
    free(msr); free(inL); free(inR); free(out);
    return 0;
}

static int cmp(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }
static void stats(FILE *f, const char *name, double *v, int n)
{
    if (!n) { fprintf(f, "\"%s\": null", name); return; }
    qsort(v, n, sizeof(double), cmp);
    int i99 = (int)(0.99 * (n - 1) + 0.5);
    fprintf(f, "\"%s\": {\"n\": %d, \"p50_us\": %.1f, \"p99_us\": %.1f, \"max_us\": %.1f}", name, n, v[n / 2], v[i99], v[n - 1]);
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s F T hops weights.f32 pace_us out.json [instances]\n", argv[0]); return 2; }
    const int F = atoi(argv[1]), T = atoi(argv[2]), hops = atoi(argv[3]), pace = atoi(argv[5]), ni = argc > 7 ? atoi(argv[7]) : 2;
    if (F < 64 || T < 64 || hops < 1 || ni < 1 || ni > 16) { fprintf(stderr, "bad arguments\n"); return 2; }
    FILE *wf = fopen(argv[4], "rb");
    char *blob = (char *)malloc((size_t)4 * COEFF_BYTES);
    if (!wf || !blob || fread(blob, COEFF_BYTES, 4, wf) != 4) { fprintf(stderr, "cannot read 4 coefficient blobs from %s\n", argv[4]); return 1; }
    fclose(wf);
    pthread_barrier_t start;
    pthread_barrier_init(&start, 0, (unsigned)ni);
    Job *jobs = (Job *)calloc((size_t)ni, sizeof(Job));
    pthread_t *th = (pthread_t *)calloc((size_t)ni, sizeof(pthread_t));
    for (int i = 0; i < ni; ++i) {
        Job *j = &jobs[i];
        j->id = i; j->F = F; j->T = T; j->hops = hops; j->pace_us = pace; j->start = &start;
        for (int k = 0; k < 4; ++k) j->coeff[k] = blob + (size_t)k * COEFF_BYTES;
        j->us = (double *)calloc((size_t)hops, sizeof(double));
        pthread_create(&th[i], 0, run, j);
    }
    for (int i = 0; i < ni; ++i) pthread_join(th[i], 0);
    FILE *f = fopen(argv[6], "w");
    if (!f) { fprintf(stderr, "cannot write %s\n", argv[6]); return 1; }
    fprintf(f, "{\"F\": %d, \"T\": %d, \"hops\": %d, \"pace_us\": %d, \"instances\": [", F, T, hops, pace);
    for (int i = 0; i < ni; ++i) {
        Job *j = &jobs[i];
        double *ord = (double *)malloc(sizeof(double) * (size_t)hops), *join = (double *)malloc(sizeof(double) * (size_t)hops);
        int no = 0, nj = 0;
        for (int h = 0; h < hops; ++h) { if ((h + 1) % T == 0) join[nj++] = j->us[h]; else ord[no++] = j->us[h]; }    /* hop h+1 completes a batch */
        int worst = 0;
        for (int h = 1; h < hops; ++h) if (j->us[h] > j->us[worst]) worst = h;                                        /* which call was the slowest: the first ones (warm-up) or one next to a batch boundary */
        fprintf(f, "%s{\"init_ms\": %.1f, \"init_error\": \"%s\", \"output_peak\": %.6g, \"worst_hop\": %d, ", i ? ", " : "", j->init_ms, j->init_error, j->peak, worst);
        stats(f, "ordinary_hops", ord, no); fprintf(f, ", "); stats(f, "join_hops", join, nj);
        fprintf(f, "}");
        free(ord); free(join);
    }
    fprintf(f, "]}\n");
    fclose(f);
    return 0;
}
