/*
 * oracle/spleeter_oracle.c — CPU restatement of the SpleeterRT hot path (see spleeter_oracle.h).
 * TEST INFRASTRUCTURE ONLY — never linked into the product library.
 *
 * Written from the mathematics in SURVEY.md §8a, not from the reference text: direct (non-im2col)
 * convolution, gather-form transposed convolution, generic radix-2 Hartley transform.  The summation
 * ORDER of the reference's naive GEMM path is kept so that the network part is bit-comparable.
 * Citations are file:line under /root/reference.
 */
#include "spleeter_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ synthetic data */
uint32_t orc_lcg_fill(uint32_t s, float *dst, size_t n, float scale)
{
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        dst[i] = scale * ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f);
    }
    return s;
}

static uint16_t f32_to_f16_rne(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int32_t  e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {                               /* half denormal or zero */
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        int shift = 14 - e;
        uint32_t h = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (m >> 13), rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
}

/* main.c:423-434: exponent re-bias, half-denormals -> +-0, no Inf/NaN special case */
void orc_fp16_expand(const uint16_t *in, float *out, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        uint32_t h = in[i], mag = (h & 0x7fffu) << 13, sgn = (h & 0x8000u) << 16;
        mag += 0x38000000u;
        if ((h & 0x7c00u) == 0) mag = 0;
        uint32_t bits = mag | sgn;
        memcpy(&out[i], &bits, 4);
    }
}

/* blob field order = Executable/spleeter.h:5-31 */
static const int ENC_CH[6][2] = { {2,16},{16,32},{32,64},{64,128},{128,256},{256,512} };
static const int DEC_CH[6][2] = { {512,256},{512,128},{256,64},{128,32},{64,16},{32,1} };

void orc_get_layout(orc_layout *lo)
{
    size_t o = 0;
    for (int i = 0; i < 6; ++i) {
        int ci = ENC_CH[i][0], co = ENC_CH[i][1];
        lo->down[i].cin = ci; lo->down[i].cout = co;
        lo->down[i].w = o; o += (size_t)25 * ci * co;
        lo->down[i].b = o; o += co;
        lo->down[i].bn = o; if (i < 5) o += 2 * (size_t)co;      /* down6 has no batchNorm */
    }
    for (int i = 0; i < 6; ++i) {
        int ci = DEC_CH[i][0], co = DEC_CH[i][1];
        lo->up[i].cin = ci; lo->up[i].cout = co;
        lo->up[i].w = o; o += (size_t)25 * ci * co;
        lo->up[i].b = o; o += co;
        lo->up[i].bn = o; o += 2 * (size_t)co;
    }
    lo->head_w = o; o += 32;
    lo->head_b = o; o += 2;
    /* o == ORC_COEFF_FLOATS */
}

static uint32_t synth_field(uint32_t s, uint16_t *dst, size_t n, float scale, float offset)
{
    for (size_t i = 0; i < n; ++i) {
        float u;
        s = orc_lcg_fill(s, &u, 1, 1.0f);
        dst[i] = f32_to_f16_rne(offset + scale * u);
    }
    return s;
}

/* He-uniform weights, small biases, BN scale 1+0.5u / shift 0.2u, all fp16-representable */
void orc_synth_coeff_fp16(uint16_t *h, int stem)
{
    orc_layout lo; orc_get_layout(&lo);
    uint32_t s = 2024u + (uint32_t)stem;
    for (int i = 0; i < 6; ++i) {
        const orc_layer_off *L = &lo.down[i];
        s = synth_field(s, h + L->w, (size_t)25 * L->cin * L->cout, 2.0f * sqrtf(6.0f / (25.0f * L->cin)), 0.0f);
        s = synth_field(s, h + L->b, L->cout, 0.02f, 0.0f);
        if (i < 5) {
            s = synth_field(s, h + L->bn, L->cout, 0.2f, 0.0f);                 /* batchNorm[s]   : shift */
            s = synth_field(s, h + L->bn + L->cout, L->cout, 0.5f, 1.0f);       /* batchNorm[C+s] : scale */
        }
    }
    for (int i = 0; i < 6; ++i) {
        const orc_layer_off *L = &lo.up[i];
        s = synth_field(s, h + L->w, (size_t)25 * L->cin * L->cout, 2.0f * sqrtf(6.0f / (25.0f * L->cin / 4.0f)), 0.0f);
        s = synth_field(s, h + L->b, L->cout, 0.02f, 0.0f);
        s = synth_field(s, h + L->bn, L->cout, 0.2f, 0.0f);
        s = synth_field(s, h + L->bn + L->cout, L->cout, 0.5f, 1.0f);
    }
    s = synth_field(s, h + lo.head_w, 32, 2.0f * sqrtf(6.0f / 16.0f), 0.0f);
    s = synth_field(s, h + lo.head_b, 2, 0.02f, 0.0f);
}

void orc_synth_coeff(float *coeff, int stem)
{
    uint16_t *h = (uint16_t *)malloc(ORC_COEFF_FLOATS * sizeof(uint16_t));
    orc_synth_coeff_fp16(h, stem);
    orc_fp16_expand(h, coeff, ORC_COEFF_FLOATS);
    free(h);
}

/* white stereo +-0.1 (L then R interleaved draw), optionally + 3 sinusoids (220/1760/7040 Hz, amp 0.05) */
void orc_synth_audio(float *L, float *R, size_t n, uint32_t seed, int tones)
{
    uint32_t s = seed;
    for (size_t i = 0; i < n; ++i) {
        float u[2];
        s = orc_lcg_fill(s, u, 2, 0.2f);
        L[i] = u[0]; R[i] = u[1];
        if (tones) {
            double t = (double)i / 44100.0;
            float v = (float)(0.05 * (sin(6.283185307179586 * 220.0 * t) + sin(6.283185307179586 * 1760.0 * t)
                                      + sin(6.283185307179586 * 7040.0 * t)));
            L[i] += v; R[i] += 0.5f * v;
        }
    }
}

/* ------------------------------------------------------------------ activations */
static float g_sig_tbl[1026];
static int   g_sig_init = 0;
static void sig_tbl_init(void)
{
    /* 1025 samples of the logistic on [-7,7] (step 14/1024) + a trailing 1.0, as in Executable/spleeter.c:29.
       Regenerated in closed form (double -> float); differs from the reference's 8-digit table by <= 6e-8. */
    for (int i = 0; i < 1025; ++i) g_sig_tbl[i] = (float)(1.0 / (1.0 + exp(7.0 - 0.013671875 * i)));
    g_sig_tbl[1025] = 1.0f;
    g_sig_init = 1;
}
float orc_sigmoid_lut(float x)      /* Executable/spleeter.c:30-42 */
{
    if (!g_sig_init) sig_tbl_init();
    if (x > 7.0f) return 1.0f;
    if (x < -7.0f) return 0.0f;
    const float step = 0.01367188f;
    short idx = (short)((x + 7.0f) / step);
    float x1 = -7.0f + step * idx;
    return g_sig_tbl[idx] + (g_sig_tbl[idx + 1] - g_sig_tbl[idx]) / (-7.0f + step * (idx + 1) - x1) * (x - x1);
}
float orc_sigmoid_exact(float x)    /* VST/Source/spleeter.c:56-65: two-branch stable form */
{
    if (x >= 0.0f) { float z = expf(-x); return 1.0f / (1.0f + z); }
    float z = expf(x); return z / (1.0f + z);
}
float orc_act(float x, int kind, int variant)
{
    switch (kind) {
    case 0: return x >= 0.0f ? x : 0.2f * x;                                 /* spleeter.c:43-46 */
    case 1: return x >= 0.0f ? x : 0.0f;                                     /* spleeter.c:47-50 */
    default:
        if (variant == ORC_VARIANT_EXE && x < -15.0f) return -1.0f;          /* spleeter.c:51-56 */
        return x >= 0.0f ? x : expf(x) - 1.0f;
    }
}

/* ------------------------------------------------------------------ network primitives */
/* y[co][ho][wo] = sum_{ci,ky,kx} w[co][ci][ky][kx] * x[ci][2ho+ky-1][2wo+kx-1]; k ascending like gemm_nn (gemm.c:6-19) */
void orc_conv5x5_s2(const float *x, int cin, int H, int W, const float *w, int cout, float *y)
{
    int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; ++co) {
        float *yo = y + (size_t)co * Ho * Wo;
        memset(yo, 0, sizeof(float) * (size_t)Ho * Wo);
        for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < 5; ++ky)
                for (int kx = 0; kx < 5; ++kx) {
                    float a = w[((size_t)(co * cin + ci) * 5 + ky) * 5 + kx];
                    for (int ho = 0; ho < Ho; ++ho) {
                        int r = 2 * ho + ky - 1;
                        if (r < 0 || r >= H) continue;
                        const float *xr = x + ((size_t)ci * H + r) * W;
                        float *yr = yo + (size_t)ho * Wo;
                        for (int wo = 0; wo < Wo; ++wo) {
                            int c = 2 * wo + kx - 1;
                            if (c >= 0 && c < W) yr[wo] += a * xr[c];
                        }
                    }
                }
    }
}

/* y[co][2h+ky-1][2w+kx-1] += sum_ci w[ci][co][ky][kx] * x[ci][h][w]  (col = W^T x, then col2im in (ky,kx) order) */
void orc_tconv5x5_s2(const float *x, int cin, int H, int W, const float *w, int cout, float *y)
{
    int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; ++co) {
        float *yo = y + (size_t)co * Ho * Wo;
        float *col = (float *)malloc(sizeof(float) * (size_t)H * W);
        memset(yo, 0, sizeof(float) * (size_t)Ho * Wo);
        for (int ky = 0; ky < 5; ++ky)
            for (int kx = 0; kx < 5; ++kx) {
                memset(col, 0, sizeof(float) * (size_t)H * W);
                for (int ci = 0; ci < cin; ++ci) {                       /* gemm_tn: k = ci ascending (gemm.c:33-45) */
                    float a = w[((size_t)(ci * cout + co) * 5 + ky) * 5 + kx];
                    const float *xc = x + (size_t)ci * H * W;
                    for (int i = 0; i < H * W; ++i) col[i] += a * xc[i];
                }
                for (int h = 0; h < H; ++h) {
                    int r = 2 * h + ky - 1;
                    if (r < 0 || r >= Ho) continue;
                    for (int ww = 0; ww < W; ++ww) {
                        int c = 2 * ww + kx - 1;
                        if (c >= 0 && c < Wo) yo[(size_t)r * Wo + c] += col[h * W + ww];
                    }
                }
            }
        free(col);
    }
}

/* y[s][h][w] = sum_{ky,kx} w[s][ky][kx] * x[h+2ky-3][w+2kx-3] */
void orc_conv4x4_d2(const float *x, int H, int W, const float *w, float *y)
{
#pragma omp parallel for schedule(static)
    for (int s = 0; s < 2; ++s) {
        float *yo = y + (size_t)s * H * W;
        memset(yo, 0, sizeof(float) * (size_t)H * W);
        for (int ky = 0; ky < 4; ++ky)
            for (int kx = 0; kx < 4; ++kx) {
                float a = w[(s * 4 + ky) * 4 + kx];
                for (int h = 0; h < H; ++h) {
                    int r = h + 2 * ky - 3;
                    if (r < 0 || r >= H) continue;
                    for (int ww = 0; ww < W; ++ww) {
                        int c = ww + 2 * kx - 3;
                        if (c >= 0 && c < W) yo[(size_t)h * W + ww] += a * x[(size_t)r * W + c];
                    }
                }
            }
    }
}

/* Executable/spleeter.c:177-301 */
void orc_forward(const float *coeff, int F, int T, int stemMode, int variant, const float *x, float *y, orc_taps *taps)
{
    orc_layout lo; orc_get_layout(&lo);
    const int actE = stemMode ? 2 : 0, actD = stemMode ? 2 : 1;             /* spleeter.c:130-139 */
    float *skip[6];
    int H = T, W = F;
    size_t hw0 = (size_t)T * F;
    float *cat = (float *)malloc(sizeof(float) * 32 * hw0);
    const float *in = x;
    for (int i = 0; i < 6; ++i) {
        const orc_layer_off *L = &lo.down[i];
        int Ho = H / 2, Wo = W / 2; size_t n = (size_t)Ho * Wo;
        skip[i] = (float *)malloc(sizeof(float) * L->cout * n);
        orc_conv5x5_s2(in, L->cin, H, W, coeff + L->w, L->cout, skip[i]);
        for (int s = 0; s < L->cout; ++s)
            for (size_t p = 0; p < n; ++p) {
                float v = skip[i][s * n + p] + coeff[L->b + s];
                skip[i][s * n + p] = v;                                         /* raw skip, pre-BN */
                if (i < 5) cat[s * n + p] = orc_act(coeff[L->bn + L->cout + s] * v + coeff[L->bn + s], actE, variant);
            }
        if (taps && taps->conv[i]) memcpy(taps->conv[i], skip[i], sizeof(float) * L->cout * n);
        if (taps && i < 5 && taps->act[i]) memcpy(taps->act[i], cat, sizeof(float) * L->cout * n);
        in = cat; H = Ho; W = Wo;
    }
    /* decoder: input of up1 is conv6; for k>=2 input = concat(skip (raw), previous decoder output) */
    float *dec_in = (float *)malloc(sizeof(float) * 32 * hw0);
    float *dec_out = (float *)malloc(sizeof(float) * 16 * hw0);
    memcpy(dec_in, skip[5], sizeof(float) * 512 * (size_t)H * W);
    for (int i = 0; i < 6; ++i) {
        const orc_layer_off *L = &lo.up[i];
        int Ho = 2 * H, Wo = 2 * W; size_t n = (size_t)Ho * Wo;
        orc_tconv5x5_s2(dec_in, L->cin, H, W, coeff + L->w, L->cout, dec_out);
        for (int s = 0; s < L->cout; ++s)
            for (size_t p = 0; p < n; ++p) {
                float v = orc_act(dec_out[s * n + p] + coeff[L->b + s], actD, variant);       /* act BEFORE BN */
                dec_out[s * n + p] = coeff[L->bn + L->cout + s] * v + coeff[L->bn + s];
            }
        if (taps && taps->up[i]) memcpy(taps->up[i], dec_out, sizeof(float) * L->cout * n);
        if (i < 5) {
            memcpy(dec_in, skip[4 - i], sizeof(float) * L->cout * n);                           /* low channels = skip */
            memcpy(dec_in + L->cout * n, dec_out, sizeof(float) * L->cout * n);
        }
        H = Ho; W = Wo;
    }
    float *pre = (float *)malloc(sizeof(float) * 2 * hw0);
    orc_conv4x4_d2(dec_out, T, F, coeff + lo.head_w, pre);
    for (int s = 0; s < 2; ++s)
        for (size_t p = 0; p < hw0; ++p) {
            float v = pre[s * hw0 + p] + coeff[lo.head_b + s];
            y[s * hw0 + p] = variant == ORC_VARIANT_EXE ? orc_sigmoid_lut(v) : orc_sigmoid_exact(v);
        }
    free(pre); free(dec_out); free(dec_in); free(cat);
    for (int i = 0; i < 6; ++i) free(skip[i]);
}

/* ------------------------------------------------------------------ DSP */
void orc_stft_init(orc_stft_tables *t)     /* stftFix.c:302-312 with LAP=4 */
{
    const double w0 = 6.283185307179586476925286766559 / ORC_FFT;
    for (unsigned i = 0; i < ORC_FFT; ++i) {
        unsigned r = 0, v = i;
        for (int b = 0; b < 12; ++b) { r = (r << 1) | (v & 1); v >>= 1; }
        t->rev[i] = r;
        float hann_scaled = (float)((1.0 / ORC_FFT) * (0.5 * (1.0 - cos(w0 * (i + 0.5)))));      /* stftFix.c:48-57 */
        t->pre[i] = hann_scaled * (2.0f / 4.0f);                                                  /* :308-309 */
        /* post window: hann/N * (N * (1/2)/(3/8)) * 0.5   (:59-70, :311-313) */
        const float scalefac = (float)ORC_FFT * ((1.0f / 2.0f) / (3.0f / 8.0f));
        t->post[i] = hann_scaled * scalefac * 0.5f;
        t->sine[i] = (float)sin(w0 * i);                                                          /* :310 */
    }
}

/* In-place discrete Hartley transform of bit-reversed input: H[k] = sum a[n] cas(2 pi n k / 4096).
   Generic radix-2 decimation-in-time Hartley butterflies; same stage structure as codelet.c:2-271
   (which unrolls the first three stages), so results agree with DFT4096 to float round-off. */
void orc_fht4096(float *a, const float *sine)
{
    for (int L = 1; L < ORC_FFT; L <<= 1) {            /* L = half block length */
        int step = ORC_FFT / (2 * L);                  /* table stride: angle = 2 pi j / (2L) */
        for (int i = 0; i < ORC_FFT; i += 2 * L) {
            float p = a[i], q = a[i + L];
            a[i] = p + q; a[i + L] = p - q;
            if (L >= 2) {
                int m = i + L / 2;
                p = a[m]; q = a[m + L];
                a[m] = p + q; a[m + L] = p - q;
            }
            for (int j = 1; j < L / 2; ++j) {
                float s = sine[j * step], c = sine[j * step + 1024];
                float al = a[i + j], be = a[i + L - j];
                float u = a[i + L + j], v = a[i + 2 * L - j];
                float b1 = u * c + v * s, b2 = u * s - v * c;
                a[i + j] = al + b1; a[i + L + j] = al - b1;
                a[i + L - j] = be + b2; a[i + 2 * L - j] = be - b2;
            }
        }
    }
}

size_t orc_stft_frames(size_t n) { return (n + ORC_HOP - 1) / ORC_HOP; }

static void stft_one(const orc_stft_tables *t, const float *L, const float *R, size_t pos, size_t n, size_t row,
                     float *reL, float *imL, float *reR, float *imR)
{
    float a[2][ORC_FFT];
    for (int i = 0; i < ORC_FFT; ++i) {
        int ok = pos + i < n;
        a[0][t->rev[i]] = ok ? L[pos + i] * t->pre[i] : 0.0f;
        a[1][t->rev[i]] = ok ? R[pos + i] * t->pre[i] : 0.0f;
    }
    orc_fht4096(a[0], t->sine); orc_fht4096(a[1], t->sine);
    size_t o = row * ORC_FFT;
    reL[o] = a[0][0] * 2.0f; imL[o] = 0.0f; reR[o] = a[1][0] * 2.0f; imR[o] = 0.0f;
    for (int k = 1; k < ORC_HALF; ++k) {
        reL[o + k] = a[0][k] + a[0][ORC_FFT - k]; imL[o + k] = a[0][k] - a[0][ORC_FFT - k];
        reR[o + k] = a[1][k] + a[1][ORC_FFT - k]; imR[o + k] = a[1][k] - a[1][ORC_FFT - k];
    }
}

/* stftFix.c:363-495: rangeM/1024 whole frames + one zero-padded tail frame; remaining rows stay zero */
size_t orc_stft(const orc_stft_tables *t, const float *L, const float *R, size_t n, float *reL, float *imL, float *reR, float *imR)
{
    size_t rows = orc_stft_frames(n);
    size_t rangeM = ((n - ORC_FFT + ORC_HOP / 4) / ORC_HOP) * ORC_HOP;
    size_t nfull = rangeM / ORC_HOP;
#pragma omp parallel for schedule(static)
    for (size_t f = 0; f <= nfull; ++f)
        stft_one(t, L, R, f * ORC_HOP, n, f, reL, imL, reR, imR);
    return rows;
}

/* stftFix.c:554-576 */
size_t orc_istft(const orc_stft_tables *t, const float *reL, const float *imL, const float *reR, const float *imR, size_t frames, float *outL, float *outR)
{
    float *tmp = (float *)malloc(sizeof(float) * 2 * ORC_FFT * frames);
#pragma omp parallel for schedule(static)
    for (size_t f = 0; f < frames; ++f) {
        float *a0 = tmp + (2 * f) * ORC_FFT, *a1 = a0 + ORC_FFT;
        size_t o = f * ORC_FFT;
        a0[0] = reL[o]; a1[0] = reR[o];
        for (int k = 1; k < ORC_HALF; ++k) {
            a0[t->rev[k]] = reL[o + k] + imL[o + k]; a0[t->rev[ORC_FFT - k]] = reL[o + k] - imL[o + k];
            a1[t->rev[k]] = reR[o + k] + imR[o + k]; a1[t->rev[ORC_FFT - k]] = reR[o + k] - imR[o + k];
        }
        orc_fht4096(a0, t->sine); orc_fht4096(a1, t->sine);
    }
    for (size_t f = 0; f < frames; ++f) {                       /* overlap-add in frame order */
        const float *a0 = tmp + (2 * f) * ORC_FFT, *a1 = a0 + ORC_FFT;
        for (int p = 0; p < ORC_FFT; ++p) {
            outL[f * ORC_HOP + p] += a0[p] * t->post[p];
            outR[f * ORC_HOP + p] += a1[p] * t->post[p];
        }
    }
    free(tmp);
    return frames * ORC_HOP + (ORC_FFT - ORC_HOP);
}

void orc_magnitude_tile(const float *reL, const float *imL, const float *reR, const float *imR, size_t frames, size_t row0, int T, int F, float *mag)
{
    for (int t = 0; t < T; ++t)
        for (int f = 0; f < F; ++f) {
            size_t r = row0 + t;
            float l = 0.0f, rr = 0.0f;
            if (r < frames) {
                size_t idx = r * ORC_FFT + f;
                l = hypotf(reL[idx], imL[idx]) * (float)ORC_FFT;
                rr = hypotf(reR[idx], imR[idx]) * (float)ORC_FFT;
            }
            mag[(size_t)t * F + f] = l;
            mag[(size_t)T * F + (size_t)t * F + f] = rr;
        }
}

void orc_mask_apply_tile(float *reL, float *imL, float *reR, float *imR, size_t frames, size_t row0, int T, int F, const float *mask, float unaffected)
{
    for (int t = 0; t < T; ++t) {
        size_t r = row0 + t;
        if (r >= frames) break;
        size_t o = r * ORC_FFT;
        int f = 0;
        for (; f < F; ++f) {
            float mL = mask[(size_t)t * F + f], mR = mask[(size_t)T * F + (size_t)t * F + f];
            reL[o + f] *= mL; imL[o + f] *= mL; reR[o + f] *= mR; imR[o + f] *= mR;
        }
        for (; f < ORC_HALF; ++f) {
            reL[o + f] *= unaffected; imL[o + f] *= unaffected; reR[o + f] *= unaffected; imR[o + f] *= unaffected;
        }
    }
}

void orc_process_spectrogram(const float *coeff, int F, int T, int stemMode, int variant, size_t frames, float *reL, float *imL, float *reR, float *imR, float unaffected)
{
    float *mag = (float *)malloc(sizeof(float) * 2 * (size_t)T * F);
    float *mask = (float *)malloc(sizeof(float) * 2 * (size_t)T * F);
    size_t ntiles = (frames + T - 1) / T;
    for (size_t j = 0; j < ntiles; ++j) {
        orc_magnitude_tile(reL, imL, reR, imR, frames, j * T, T, F, mag);
        orc_forward(coeff, F, T, stemMode, variant, mag, mask, NULL);
        orc_mask_apply_tile(reL, imL, reR, imR, frames, j * T, T, F, mask, unaffected);
    }
    free(mag); free(mask);
}
