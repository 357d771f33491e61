/*
 * stftFix.h — drop-in for the reference's offline STFT API (Executable/stftFix.h:7-35), backed by the
 * MI355X engine's LDS-resident 4096-point FFT kernels.
 *
 * Same symbols, constants, struct and ownership rules:
 *   - OfflineSTFT is a public, caller-allocated struct (main.c:775); the table members are filled exactly as
 *     InitSTFT does (stftFix.c:302-313) so callers that read them keep working; the void* members carry this
 *     library's private state instead of the reference's pthread pool (`targetCore` is accepted and ignored:
 *     the GPU replaces the worker threads);
 *   - stft() calloc()s four planes of rows*4096 floats (row stride FFTSIZE, bins 2049..4095 zero) that the caller
 *     free()s (main.c:786-789); istft() calloc()s the two output channels.  HOST pointers throughout;
 *   - unlike the reference's multi-threaded istft (stftFix.c:537-538) the input planes are left untouched.
 * Failures never abort() and never fall back to a CPU path: the reason goes to stderr and srtLastError(), the instance is
 * marked failed and from then on returns zeros / silence with the reference's sample accounting (SPLEETERRT_ABORT_ON_ERROR=1 aborts instead).
 */
#ifndef _STFT_H_
#define _STFT_H_
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef SRT_PT_STATE_DEFINED
#define SRT_PT_STATE_DEFINED
enum pt_state { SETUP, IDLE, WORKING, GET_OFF_FROM_WORK };
#endif
#define FFTSIZE 4096
#define LAP 4
#define HOPSIZE (FFTSIZE / LAP)
#define HALFWNDLEN ((FFTSIZE >> 1) + 1)
#define NPTDIV2 (FFTSIZE >> 2)
typedef struct
{
    unsigned int mBitRev[FFTSIZE];
    float mPreWindow[FFTSIZE];
    float mPostWindow[FFTSIZE];
    float mSineTab[FFTSIZE];
    void *threads;           /* -> private engine handle */
    void *stftThreadData;    /* unused */
    void *istftThreadData;   /* unused */
    size_t targetCore;
    float **_data[2];        /* unused */
    void *shared_info;       /* unused */
} OfflineSTFT;
#define STFT_API __attribute__((visibility("default")))
STFT_API void   InitSTFT(OfflineSTFT *st, size_t targetCore);
STFT_API void   FreeSTFT(OfflineSTFT *st);
STFT_API size_t stft(OfflineSTFT *st, const float *dataL, const float *dataR, size_t data_size,
                     float **resultLRe, float **resultLIm, float **resultRRe, float **resultRIm);
STFT_API size_t istft(OfflineSTFT *st, float *dataLRe, float *dataLIm, float *dataRRe, float *dataRIm, size_t data_size,
                      float **resultL, float **resultR);
#ifdef __cplusplus
}
#endif
#endif
