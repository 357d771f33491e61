/*
 * oracle/spleeter_oracle.h — CPU restatement of the SpleeterRT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library (spleeterrt_amd/csrc, include/) may include,
 * link or dlopen this.  Importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_vs_ref.py) against the real
 * reference compiled from /root/reference into oracle/_ref/ (oracle/Makefile), and against the golden
 * vectors in tests/golden/ that were generated from that build (tests/golden/gen_golden.py).
 * The network part is bit-exact versus the reference's CPU_GEMM=1 path (same summation order);
 * the Hartley/STFT part agrees to float round-off (different butterfly grouping, see orc_fht4096).
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef SPLEETER_ORACLE_H
#define SPLEETER_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_FFT      4096
#define ORC_HOP      1024
#define ORC_HALF     2049
#define ORC_COEFF_FLOATS 9822725u   /* sizeof(spleeterCoeff)/4, Executable/spleeter.h:5-31 */

/* numerics variants */
enum { ORC_VARIANT_EXE = 0,  /* LUT sigmoid, ELU clamp at -15 : Executable/spleeter.c:29-56 */
       ORC_VARIANT_VST = 1   /* exact sigmoid, plain ELU      : VST/Source/spleeter.c:56-77  */ };

/* ---- deterministic synthetic data ---- */
uint32_t orc_lcg_fill(uint32_t seed, float *dst, size_t n, float scale);      /* returns next state */
void   orc_synth_coeff_fp16(uint16_t *halfs, int stem);                        /* ORC_COEFF_FLOATS halves */
void   orc_fp16_expand(const uint16_t *in, float *out, size_t n);              /* main.c:423-443 */
void   orc_synth_coeff(float *coeff, int stem);                                /* fp16 synth + expand */
void   orc_synth_audio(float *L, float *R, size_t n, uint32_t seed, int tones);

/* ---- weight blob layout: float offsets of every field, in blob order ---- */
typedef struct { size_t w, b, bn; int cin, cout; } orc_layer_off;
typedef struct { orc_layer_off down[6], up[6]; size_t head_w, head_b; } orc_layout;
void orc_get_layout(orc_layout *lo);

/* ---- network primitives (all CHW planar, row = time, contiguous = frequency) ---- */
/* encoder conv 5x5 stride 2, TF-SAME (pad 1 before / 2 after); w is OIHW.   spleeter.c:96-100 + im2col_dilated.c:10-33 */
void orc_conv5x5_s2(const float *x, int cin, int H, int W, const float *w, int cout, float *y);
/* transposed conv 5x5 stride 2 -> exactly 2H x 2W; w is [cin][cout][5][5].   spleeter.c:73-78 + im2col_dilated.c:42-65 */
void orc_tconv5x5_s2(const float *x, int cin, int H, int W, const float *w, int cout, float *y);
/* head conv 4x4 dilation 2 pad 3, 1 -> 2 channels; w is [2][1][4][4].        spleeter.c:156,295 */
void orc_conv4x4_d2(const float *x, int H, int W, const float *w, float *y);
float orc_sigmoid_lut(float x);      /* Executable/spleeter.c:30-42  */
float orc_sigmoid_exact(float x);    /* VST/Source/spleeter.c:56-65  */
float orc_act(float x, int kind, int variant);  /* kind: 0 leakyReLU(0.2) 1 ReLU 2 ELU */

/* optional taps: raw encoder outputs (skip tensors) and post-epilogue decoder outputs */
typedef struct {
    float *conv[6];   /* conv1..conv6 raw (+bias), sizes Cout*H*W of that level, or NULL */
    float *act[5];    /* BN+activation copies of down1..down5, or NULL */
    float *up[6];     /* up1..up6 after bias/act/BN (only the new Cout channels), or NULL */
} orc_taps;
/* whole forward, Executable/spleeter.c:177-301.  x,y: [2][T][F].  stemMode 0: LeakyReLU/ReLU, else ELU/ELU */
void orc_forward(const float *coeff, int F, int T, int stemMode, int variant, const float *x, float *y, orc_taps *taps);

/* ---- DSP ---- */
typedef struct { unsigned rev[ORC_FFT]; float pre[ORC_FFT], post[ORC_FFT], sine[ORC_FFT]; } orc_stft_tables;
void   orc_stft_init(orc_stft_tables *t);                                                     /* stftFix.c:302-312 */
void   orc_fht4096(float *a, const float *sine);                                              /* codelet.c:2-271 (bit-reversed input) */
size_t orc_stft_frames(size_t n);                                                             /* rows = ceil(n/1024), stftFix.c:367 */
/* planes are caller-allocated, zeroed, rows x 4096 floats each.  stftFix.c:363-495 */
size_t orc_stft(const orc_stft_tables *t, const float *L, const float *R, size_t n, float *reL, float *imL, float *reR, float *imR);
/* out buffers caller-allocated+zeroed, frames*1024+3072 floats.  Does NOT clobber inputs. stftFix.c:496-579 */
size_t orc_istft(const orc_stft_tables *t, const float *reL, const float *imL, const float *reR, const float *imR, size_t frames, float *outL, float *outR);
/* magnitude of tile rows [row0,row0+T) (rows >= frames are zero-filled): main.c:462-471,500-514 */
void   orc_magnitude_tile(const float *reL, const float *imL, const float *reR, const float *imR, size_t frames, size_t row0, int T, int F, float *mag);
/* in-place mask apply for the same rows: main.c:473-494,516-537 */
void   orc_mask_apply_tile(float *reL, float *imL, float *reR, float *imR, size_t frames, size_t row0, int T, int F, const float *mask, float unaffected);
/* processMT single-thread restatement over the whole spectrogram (main.c:444-541) */
void   orc_process_spectrogram(const float *coeff, int F, int T, int stemMode, int variant, size_t frames, float *reL, float *imL, float *reR, float *imR, float unaffected);

#ifdef __cplusplus
}
#endif
#endif
