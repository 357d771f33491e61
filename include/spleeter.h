/*
 * spleeter.h — drop-in for the reference's tile API, backed by the MI355X engine (libspleeterrt_amd.so).
 *
 * Same symbols, argument meaning and ownership rules as /root/reference/Executable/spleeter.h:63-69
 * (and VST/Source/spleeter.h:1-7, whose initSpleeter takes int dimensions — both call forms work):
 *   - the caller obtains the instance with allocateSpleeterStr() and free()s it after freeSpleeter() (main.c:538-539);
 *   - `coeff` holds getCoeffSize() bytes laid out as spleeterCoeff; it is copied to HBM at init
 *     (the reference borrows it, spleeter.c:129 — a caller that keeps it alive is still correct);
 *   - x and y are HOST pointers to [2][height][width] floats; y may alias the getMaskPtr() buffer (main.c:453,472);
 *   - all functions return void.  Where the reference has undefined behaviour on failure (no GPU, out of memory, a HIP error),
 *     this library prints the reason to stderr, keeps it readable through srtLastError(), marks the instance failed and from then
 *     on writes ZERO masks — it never abort()s the host and never falls back to a CPU path (SPLEETERRT_ABORT_ON_ERROR=1 aborts instead).
 *   - initSpleeter also pre-warms the device side (workspace allocation, hipGraph capture), so the first processSpleeter costs what
 *     every later one does.
 * One instance is not re-entrant; distinct instances may be used from distinct threads (main.c:296-330).
 * Flavour: environment variable SPLEETERRT_VARIANT = "exe" (default: LUT sigmoid, ELU clamp) or "vst".
 * For throughput use the batched, HBM-resident API in spleeterrt_amd.h; this header is the compatibility surface.
 */
#ifndef SPLEETERRT_AMD_SPLEETER_H
#define SPLEETERRT_AMD_SPLEETER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TBL_SIZE (1025)                       /* reference LUT length, Executable/spleeter.h:3 */
#define TBL_SIZE_MINUS1 (TBL_SIZE - 1)

/* Weight blob of ONE sub-network; field order == memory order (Executable/spleeter.h:5-31).
   Encoder weights OIHW [Cout][Cin][5][5]; decoder weights [Cin][Cout][5][5]; batchNorm[s] = shift, batchNorm[C+s] = scale. */
#define SRT_ENC_(T, n, ci, co)      T n##_convWeight[5 * 5 * (ci) * (co)]; T n##_convBias[co]; T n##_batchNorm[(co) * 2];
#define SRT_ENC_NOBN_(T, n, ci, co) T n##_convWeight[5 * 5 * (ci) * (co)]; T n##_convBias[co];
#define SRT_DEC_(T, n, ci, co)      T n##_transp_convWeight[5 * 5 * (co) * (ci)]; T n##_transp_convBias[co]; T n##_batchNorm[(co) * 2];
#define SRT_COEFF_FIELDS_(T) \
    SRT_ENC_(T, down1, 2, 16) SRT_ENC_(T, down2, 16, 32) SRT_ENC_(T, down3, 32, 64) \
    SRT_ENC_(T, down4, 64, 128) SRT_ENC_(T, down5, 128, 256) SRT_ENC_NOBN_(T, down6, 256, 512) \
    SRT_DEC_(T, up1, 512, 256) SRT_DEC_(T, up2, 512, 128) SRT_DEC_(T, up3, 256, 64) \
    SRT_DEC_(T, up4, 128, 32) SRT_DEC_(T, up5, 64, 16) SRT_DEC_(T, up6, 32, 1) \
    T up7_convWeight[4 * 4 * 1 * 2]; T up7_convBias[2];
typedef struct { SRT_COEFF_FIELDS_(float) } spleeterCoeff;                 /* 39 290 900 bytes */
typedef struct { SRT_COEFF_FIELDS_(uint16_t) } spleeterQuantizedSubNet;    /* IEEE fp16 bit patterns, Executable/spleeter.h:32-58 */
typedef struct { spleeterQuantizedSubNet net[2]; } spleeterQuantized;      /* Executable/spleeter.h:59-62 */

typedef struct _spleeter *spleeter;
#define SPLEETER_API __attribute__((visibility("default")))
SPLEETER_API size_t getCoeffSize(void);
SPLEETER_API void  *allocateSpleeterStr(void);
SPLEETER_API void   initSpleeter(struct _spleeter *nn, size_t width, size_t height, int stemMode, void *coeff);
SPLEETER_API void   getMaskPtr(struct _spleeter *nn, float **mask);
SPLEETER_API void   freeSpleeter(struct _spleeter *nn);
SPLEETER_API void   processSpleeter(struct _spleeter *nn, float *x, float *y);

#ifdef __cplusplus
}
#endif
#endif
