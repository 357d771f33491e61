// srt_compat.hip — the reference's own C entry points (include/spleeter.h, include/stftFix.h) on top of the engine.
// Host pointers in, host pointers out; one engine and one HIP stream per instance, so distinct instances can be driven
// from distinct host threads at the same time, as the reference's callers do (Executable/main.c:296-330 tile threads,
// VST/Source/Spleeter4Stems.c:135 stem threads).
//
// Failure policy.  These functions return void / a size (the reference "never fails").  Nothing here ever falls back to a
// CPU path: on a failure (no GPU, out of memory, a HIP error) the reason goes to stderr and to srtLastError(), the instance
// is marked failed, and every later call on it produces zeros / empty results instead of touching the device again.
// SPLEETERRT_ABORT_ON_ERROR=1 turns the first failure into abort() for batch jobs that prefer to die.
#include "srt_internal.h"
#include "../../include/spleeterrt_amd.h"
#include "../../include/spleeter.h"
#include "../../include/stftFix.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static bool compat_fail(const char* where, const char* why)
{
    char buf[400];
    snprintf(buf, sizeof buf, "%s: %s", where, why ? why : srtLastError());
    srt_set_error(-2, "%s", buf);
    fprintf(stderr, "libspleeterrt_amd: %s (no CPU fallback exists; output is zero)\n", buf);
    const char* a = getenv("SPLEETERRT_ABORT_ON_ERROR");
    if (a && a[0] == '1') abort();
    return false;
}
static bool hip_ok(hipError_t e, const char* where) { return e == hipSuccess ? true : compat_fail(where, hipGetErrorString(e)); }

static int env_precision()       // SPLEETERRT_PRECISION = f32 (default) | f16 | f16x2, see SRT_PREC_* in spleeterrt_amd.h
{
    const char* v = getenv("SPLEETERRT_PRECISION");
    if (v && !strcmp(v, "f16")) return SRT_PREC_F16;
    if (v && !strcmp(v, "f16x2")) return SRT_PREC_F16X2;
    return SRT_PREC_F32;
}
static int env_variant()
{
    const char* v = getenv("SPLEETERRT_VARIANT");
    return (v && (!strcmp(v, "vst") || !strcmp(v, "VST"))) ? SRT_VARIANT_VST : SRT_VARIANT_EXE;
}

// ---------------------------------------------------------------------------------------------- spleeter.h
struct _spleeter {
    srt_engine* eng;
    hipStream_t stream;         // this instance's own stream: two instances never serialise on the null stream
    float *d_x, *d_y, *h_mask;
    size_t hw2;                 // 2 * height * width
    int failed;
};

size_t getCoeffSize(void) { return sizeof(spleeterCoeff); }
void* allocateSpleeterStr(void) { return calloc(1, sizeof(struct _spleeter)); }

void initSpleeter(struct _spleeter* nn, size_t width, size_t height, int stemMode, void* coeff)
{
    if (!nn) return;
    SrtSetupLock setup;                                       // (srt_internal.h: set-up paths are serialised process-wide)
    memset(nn, 0, sizeof *nn);
    // VST callers pass int arguments (VST/Source/spleeter.h:4): only the low 32 bits are defined for them
    const int F = (int)(width & 0xffffffffu), T = (int)(height & 0xffffffffu);
    nn->hw2 = 2 * (size_t)F * T;
    nn->h_mask = (float*)calloc(nn->hw2 ? nn->hw2 : 1, sizeof(float));           // getMaskPtr stays valid even on a failed instance
    nn->failed = 1;
    if (!nn->h_mask) { compat_fail("initSpleeter", "out of host memory"); return; }
    if (!coeff) { compat_fail("initSpleeter", "null coefficient pointer"); return; }
    srt_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.F = F; cfg.T = T; cfg.n_stems = 1; cfg.stem_mode[0] = stemMode; cfg.oob_weight[0] = 1.0f;
    cfg.variant = env_variant(); cfg.max_tiles = 1; cfg.impl = SRT_IMPL_MFMA; cfg.precision = env_precision();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { compat_fail("initSpleeter", "no HIP device (this library has no CPU path)"); return; }
    // lowest stream priority: a tile-API instance is a throughput caller; the real-time surface's per-hop stream (csrc/srt_stream.hip) runs at the
    // highest, so a plugin instance sharing the GPU with tile-API workers is dispatched ahead of their queued network kernels
    int prLeast = 0, prGreatest = 0;
    if (!hip_ok(hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest), "initSpleeter")) return;
    if (!hip_ok(hipStreamCreateWithPriority(&nn->stream, hipStreamNonBlocking, prLeast), "initSpleeter")) { nn->stream = nullptr; return; }
    if (srtCreate(&cfg, nn->stream, &nn->eng)) { nn->eng = nullptr; compat_fail("initSpleeter", nullptr); return; }
    if (srtSetCoeffHost(nn->eng, 0, coeff)) { compat_fail("initSpleeter(weights)", nullptr); return; }
    srtSetGraphMode(nn->eng, 1);                              // every processSpleeter call repeats the same launch sequence on d_x / d_y
    if (!hip_ok(hipMalloc((void**)&nn->d_x, nn->hw2 * sizeof(float)), "initSpleeter")) { nn->d_x = nullptr; return; }
    if (!hip_ok(hipMalloc((void**)&nn->d_y, nn->hw2 * sizeof(float)), "initSpleeter")) { nn->d_y = nullptr; return; }
    // pre-warm here, not in the first processSpleeter (which may be a real-time thread): workspace allocation + graph capture / instantiate
    if (!hip_ok(hipMemsetAsync(nn->d_x, 0, nn->hw2 * sizeof(float), nn->stream), "initSpleeter")) return;
    if (srtPrepareForward(nn->eng, nn->d_x, 1, nn->d_y)) { compat_fail("initSpleeter(prepare)", nullptr); return; }
    nn->failed = 0;
}

void getMaskPtr(struct _spleeter* nn, float** mask) { if (nn && mask) *mask = nn->h_mask; }

void processSpleeter(struct _spleeter* nn, float* x, float* y)
{
    if (!nn || !y) return;
    if (nn->failed || !x) {
        if (!nn->failed) compat_fail("processSpleeter", "null input");
        memset(y, 0, nn->hw2 * sizeof(float));
        return;
    }
    bool ok = hip_ok(hipMemcpyAsync(nn->d_x, x, nn->hw2 * sizeof(float), hipMemcpyHostToDevice, nn->stream), "processSpleeter");
    if (ok && srtForward(nn->eng, nn->d_x, 1, nn->d_y)) ok = compat_fail("processSpleeter", nullptr);
    ok = ok && hip_ok(hipMemcpyAsync(y, nn->d_y, nn->hw2 * sizeof(float), hipMemcpyDeviceToHost, nn->stream), "processSpleeter");
    ok = ok && hip_ok(hipStreamSynchronize(nn->stream), "processSpleeter");
    if (!ok) { nn->failed = 1; memset(y, 0, nn->hw2 * sizeof(float)); }
}

void freeSpleeter(struct _spleeter* nn)
{
    if (!nn) return;
    if (nn->eng) srtDestroy(nn->eng);
    if (nn->d_x) hipFree(nn->d_x);
    if (nn->d_y) hipFree(nn->d_y);
    if (nn->stream) hipStreamDestroy(nn->stream);
    free(nn->h_mask);
    memset(nn, 0, sizeof *nn);
    nn->failed = 1;
}

// ---------------------------------------------------------------------------------------------- stftFix.h
// OfflineSTFT is a public struct (Executable/stftFix.h:19-31): its tables are filled exactly as the reference does
// (stftFix.c:302-313) for callers that read them; the engine behind stft()/istft() hangs off the opaque `threads` pointer.
struct StftCtx { srt_engine* eng; hipStream_t stream; };

void InitSTFT(OfflineSTFT* st, size_t targetCore)
{
    if (!st) return;
    SrtSetupLock setup;
    const double w0 = 6.283185307179586476925286766559 / FFTSIZE;
    const float postScale = (float)FFTSIZE * ((1.0f / 2.0f) / (3.0f / 8.0f));
    for (unsigned i = 0; i < FFTSIZE; ++i) {
        unsigned r = 0, v = i;
        for (int b = 0; b < 12; ++b) { r = (r << 1) | (v & 1); v >>= 1; }
        st->mBitRev[i] = r;
        const float hs = (float)((1.0 / FFTSIZE) * (0.5 * (1.0 - cos(w0 * (i + 0.5)))));
        st->mPreWindow[i] = hs * (2.0f / (float)LAP);
        st->mPostWindow[i] = hs * postScale * 0.5f;
        st->mSineTab[i] = (float)sin(w0 * i);
    }
    st->targetCore = targetCore;
    st->stftThreadData = st->istftThreadData = st->shared_info = nullptr;
    st->_data[0] = st->_data[1] = nullptr;
    st->threads = nullptr;
    StftCtx* c = (StftCtx*)calloc(1, sizeof(StftCtx));
    if (!c) { compat_fail("InitSTFT", "out of host memory"); return; }
    srt_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.F = 64; cfg.T = 64; cfg.n_stems = 1; cfg.stem_mode[0] = 1; cfg.oob_weight[0] = 1.0f; cfg.max_tiles = 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { free(c); compat_fail("InitSTFT", "no HIP device (this library has no CPU path)"); return; }
    if (!hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "InitSTFT")) { free(c); return; }
    if (srtCreate(&cfg, c->stream, &c->eng)) { hipStreamDestroy(c->stream); free(c); compat_fail("InitSTFT", nullptr); return; }
    st->threads = c;
}

void FreeSTFT(OfflineSTFT* st)
{
    if (!st || !st->threads) return;
    StftCtx* c = (StftCtx*)st->threads;
    srtDestroy(c->eng);
    hipStreamDestroy(c->stream);
    free(c);
    st->threads = nullptr;
}

// Both transforms hand back calloc'ed planes the caller frees (main.c:786-789), also on failure (then all zero).
size_t stft(OfflineSTFT* st, const float* dataL, const float* dataR, size_t n, float** reL, float** imL, float** reR, float** imR)
{
    const size_t rows = srtStftRows(n);
    float** out[4] = { reL, imL, reR, imR };
    bool ok = true;
    for (int k = 0; k < 4; ++k) {
        *out[k] = (float*)calloc(rows ? rows * FFTSIZE : 1, sizeof(float));       // row stride 4096, bins 2049.. stay zero (stftFix.c:368-371)
        if (!*out[k]) ok = false;
    }
    if (!ok) { compat_fail("stft", "out of host memory"); return rows; }
    StftCtx* c = st ? (StftCtx*)st->threads : nullptr;
    if (!c) { compat_fail("stft", "InitSTFT failed or was not called"); return rows; }
    if (n < FFTSIZE) { compat_fail("stft", "need at least 4096 samples (the reference underflows here, stftFix.c:378)"); return rows; }
    const size_t specFloats = 2 * rows * SRT_SPEC_LD * 2;
    float *dL = nullptr, *dS = nullptr;
    float* hs = (float*)malloc(specFloats * sizeof(float));
    ok = hs != nullptr || compat_fail("stft", "out of host memory");
    ok = ok && hip_ok(hipMalloc((void**)&dL, 2 * n * sizeof(float)), "stft");
    ok = ok && hip_ok(hipMalloc((void**)&dS, specFloats * sizeof(float)), "stft");
    ok = ok && hip_ok(hipMemcpyAsync(dL, dataL, n * sizeof(float), hipMemcpyHostToDevice, c->stream), "stft");
    ok = ok && hip_ok(hipMemcpyAsync(dL + n, dataR, n * sizeof(float), hipMemcpyHostToDevice, c->stream), "stft");
    if (ok && srtStft(c->eng, dL, dL + n, n, dS, nullptr)) ok = compat_fail("stft", nullptr);
    ok = ok && hip_ok(hipMemcpyAsync(hs, dS, specFloats * sizeof(float), hipMemcpyDeviceToHost, c->stream), "stft");
    ok = ok && hip_ok(hipStreamSynchronize(c->stream), "stft");
    if (ok) {
        for (int ch = 0; ch < 2; ++ch)
            for (size_t r = 0; r < rows; ++r) {
                const float* src = hs + ((size_t)ch * rows + r) * SRT_SPEC_LD * 2;
                float* re = *out[2 * ch] + r * FFTSIZE; float* im = *out[2 * ch + 1] + r * FFTSIZE;
                for (int k = 0; k < HALFWNDLEN; ++k) { re[k] = src[2 * k]; im[k] = src[2 * k + 1]; }
            }
    }
    if (dL) hipFree(dL);
    if (dS) hipFree(dS);
    free(hs);
    return rows;
}

size_t istft(OfflineSTFT* st, float* reL, float* imL, float* reR, float* imR, size_t rows, float** outL, float** outR)
{
    const size_t specFloats = 2 * rows * SRT_SPEC_LD * 2, n = srtIstftLength(rows);
    *outL = (float*)calloc(n, sizeof(float)); *outR = (float*)calloc(n, sizeof(float));
    if (!*outL || !*outR) { compat_fail("istft", "out of host memory"); return n; }
    StftCtx* c = st ? (StftCtx*)st->threads : nullptr;
    if (!c) { compat_fail("istft", "InitSTFT failed or was not called"); return n; }
    if (rows < 1) return n;
    float* hs = (float*)calloc(specFloats, sizeof(float));
    float *dS = nullptr, *dO = nullptr;
    bool ok = hs != nullptr || compat_fail("istft", "out of host memory");
    if (ok) {
        const float* in[4] = { reL, imL, reR, imR };
        for (int ch = 0; ch < 2; ++ch)
            for (size_t r = 0; r < rows; ++r) {
                float* dst = hs + ((size_t)ch * rows + r) * SRT_SPEC_LD * 2;
                const float* re = in[2 * ch] + r * FFTSIZE; const float* im = in[2 * ch + 1] + r * FFTSIZE;
                for (int k = 0; k < HALFWNDLEN; ++k) { dst[2 * k] = re[k]; dst[2 * k + 1] = im[k]; }
            }
    }
    ok = ok && hip_ok(hipMalloc((void**)&dS, specFloats * sizeof(float)), "istft");
    ok = ok && hip_ok(hipMalloc((void**)&dO, 2 * n * sizeof(float)), "istft");
    ok = ok && hip_ok(hipMemcpyAsync(dS, hs, specFloats * sizeof(float), hipMemcpyHostToDevice, c->stream), "istft");
    if (ok && srtIstft(c->eng, dS, rows, nullptr, dO)) ok = compat_fail("istft", nullptr);
    ok = ok && hip_ok(hipMemcpyAsync(*outL, dO, n * sizeof(float), hipMemcpyDeviceToHost, c->stream), "istft");
    ok = ok && hip_ok(hipMemcpyAsync(*outR, dO + n, n * sizeof(float), hipMemcpyDeviceToHost, c->stream), "istft");
    ok = ok && hip_ok(hipStreamSynchronize(c->stream), "istft");
    if (!ok) { memset(*outL, 0, n * sizeof(float)); memset(*outR, 0, n * sizeof(float)); }
    if (dS) hipFree(dS);
    if (dO) hipFree(dO);
    free(hs);
    return n;
}
