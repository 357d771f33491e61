// srt_compat.hip — the reference's own C entry points (include/spleeter.h, include/stftFix.h) on top of the engine.
// Host pointers in, host pointers out; one engine per instance; failures are loud (stderr + abort), never a CPU path.
#include "srt_internal.h"
#include "../../include/spleeterrt_amd.h"
#include "../../include/spleeter.h"
#include "../../include/stftFix.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void die(const char* where)
{
    fprintf(stderr, "libspleeterrt_amd: %s failed: %s\n", where, srtLastError());
    abort();
}
#define HIPDIE(x, where) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "libspleeterrt_amd: %s: %s\n", where, hipGetErrorString(_e)); abort(); } } while (0)

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
    float *d_x, *d_y, *h_mask;
    size_t hw2;                 // 2 * height * width
};

size_t getCoeffSize(void) { return sizeof(spleeterCoeff); }
void* allocateSpleeterStr(void) { return calloc(1, sizeof(struct _spleeter)); }

void initSpleeter(struct _spleeter* nn, size_t width, size_t height, int stemMode, void* coeff)
{
    // VST callers pass int arguments (VST/Source/spleeter.h:4): only the low 32 bits are defined for them
    const int F = (int)(width & 0xffffffffu), T = (int)(height & 0xffffffffu);
    srt_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.F = F; cfg.T = T; cfg.n_stems = 1; cfg.stem_mode[0] = stemMode; cfg.oob_weight[0] = 1.0f;
    cfg.variant = env_variant(); cfg.max_tiles = 1; cfg.impl = SRT_IMPL_MFMA; cfg.precision = env_precision();
    if (srtCreate(&cfg, nullptr, &nn->eng)) die("initSpleeter");
    if (srtSetCoeffHost(nn->eng, 0, coeff)) die("initSpleeter(weights)");
    nn->hw2 = 2 * (size_t)F * T;
    HIPDIE(hipMalloc((void**)&nn->d_x, nn->hw2 * sizeof(float)), "initSpleeter");
    HIPDIE(hipMalloc((void**)&nn->d_y, nn->hw2 * sizeof(float)), "initSpleeter");
    nn->h_mask = (float*)malloc(nn->hw2 * sizeof(float));
}

void getMaskPtr(struct _spleeter* nn, float** mask) { *mask = nn->h_mask; }

void processSpleeter(struct _spleeter* nn, float* x, float* y)
{
    HIPDIE(hipMemcpy(nn->d_x, x, nn->hw2 * sizeof(float), hipMemcpyHostToDevice), "processSpleeter");
    if (srtForward(nn->eng, nn->d_x, 1, nn->d_y)) die("processSpleeter");
    HIPDIE(hipMemcpy(y, nn->d_y, nn->hw2 * sizeof(float), hipMemcpyDeviceToHost), "processSpleeter");   // syncs the null stream
}

void freeSpleeter(struct _spleeter* nn)
{
    if (!nn) return;
    srtDestroy(nn->eng); nn->eng = nullptr;
    if (nn->d_x) hipFree(nn->d_x);
    if (nn->d_y) hipFree(nn->d_y);
    free(nn->h_mask);
    nn->d_x = nn->d_y = nn->h_mask = nullptr;
}

// ---------------------------------------------------------------------------------------------- stftFix.h
void InitSTFT(OfflineSTFT* st, size_t targetCore)
{
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
    srt_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.F = 64; cfg.T = 64; cfg.n_stems = 1; cfg.stem_mode[0] = 1; cfg.oob_weight[0] = 1.0f; cfg.max_tiles = 1;
    srt_engine* e = nullptr;
    if (srtCreate(&cfg, nullptr, &e)) die("InitSTFT");
    st->threads = e;
}

void FreeSTFT(OfflineSTFT* st)
{
    if (st && st->threads) { srtDestroy((srt_engine*)st->threads); st->threads = nullptr; }
}

size_t stft(OfflineSTFT* st, const float* dataL, const float* dataR, size_t n, float** reL, float** imL, float** reR, float** imR)
{
    srt_engine* e = (srt_engine*)st->threads;
    const size_t rows = srtStftRows(n);
    float *dL, *dR, *dS;
    HIPDIE(hipMalloc((void**)&dL, n * sizeof(float)), "stft");
    HIPDIE(hipMalloc((void**)&dR, n * sizeof(float)), "stft");
    HIPDIE(hipMalloc((void**)&dS, 2 * rows * SRT_SPEC_LD * 2 * sizeof(float)), "stft");
    HIPDIE(hipMemcpy(dL, dataL, n * sizeof(float), hipMemcpyHostToDevice), "stft");
    HIPDIE(hipMemcpy(dR, dataR, n * sizeof(float), hipMemcpyHostToDevice), "stft");
    if (srtStft(e, dL, dR, n, dS, nullptr)) die("stft");
    float* hs = (float*)malloc(2 * rows * SRT_SPEC_LD * 2 * sizeof(float));
    HIPDIE(hipMemcpy(hs, dS, 2 * rows * SRT_SPEC_LD * 2 * sizeof(float), hipMemcpyDeviceToHost), "stft");
    hipFree(dL); hipFree(dR); hipFree(dS);
    float** out[4] = { reL, imL, reR, imR };
    for (int k = 0; k < 4; ++k) *out[k] = (float*)calloc(rows * FFTSIZE, sizeof(float));     // caller frees (main.c:786-789)
    for (int ch = 0; ch < 2; ++ch)
        for (size_t r = 0; r < rows; ++r) {
            const float* src = hs + ((size_t)ch * rows + r) * SRT_SPEC_LD * 2;
            float* re = *out[2 * ch] + r * FFTSIZE; float* im = *out[2 * ch + 1] + r * FFTSIZE;
            for (int k = 0; k < HALFWNDLEN; ++k) { re[k] = src[2 * k]; im[k] = src[2 * k + 1]; }
        }
    free(hs);
    return rows;
}

size_t istft(OfflineSTFT* st, float* reL, float* imL, float* reR, float* imR, size_t rows, float** outL, float** outR)
{
    srt_engine* e = (srt_engine*)st->threads;
    const size_t specFloats = 2 * rows * SRT_SPEC_LD * 2, n = srtIstftLength(rows);
    float* hs = (float*)calloc(specFloats, sizeof(float));
    const float* in[4] = { reL, imL, reR, imR };
    for (int ch = 0; ch < 2; ++ch)
        for (size_t r = 0; r < rows; ++r) {
            float* dst = hs + ((size_t)ch * rows + r) * SRT_SPEC_LD * 2;
            const float* re = in[2 * ch] + r * FFTSIZE; const float* im = in[2 * ch + 1] + r * FFTSIZE;
            for (int k = 0; k < HALFWNDLEN; ++k) { dst[2 * k] = re[k]; dst[2 * k + 1] = im[k]; }
        }
    float *dS, *dO;
    HIPDIE(hipMalloc((void**)&dS, specFloats * sizeof(float)), "istft");
    HIPDIE(hipMalloc((void**)&dO, 2 * n * sizeof(float)), "istft");
    HIPDIE(hipMemcpy(dS, hs, specFloats * sizeof(float), hipMemcpyHostToDevice), "istft");
    free(hs);
    if (srtIstft(e, dS, rows, nullptr, dO)) die("istft");
    *outL = (float*)calloc(n, sizeof(float)); *outR = (float*)calloc(n, sizeof(float));
    HIPDIE(hipMemcpy(*outL, dO, n * sizeof(float), hipMemcpyDeviceToHost), "istft");
    HIPDIE(hipMemcpy(*outR, dO + n, n * sizeof(float), hipMemcpyDeviceToHost), "istft");
    hipFree(dS); hipFree(dO);
    return n;
}
