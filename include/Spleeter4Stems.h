/*
 * Spleeter4Stems.h — drop-in for the reference's real-time streaming surface (VST/Source/Spleeter4Stems.h:67-69),
 * the API the JUCE plugin calls from the host's audio callback (VST/Source/PluginProcessor.cpp:115-182).
 *
 * Same three symbols, argument meaning and timing as the reference:
 *   - Spleeter4Stems is caller-allocated (`malloc(sizeof(Spleeter4Stems))`, PluginProcessor.cpp:123); here its body is
 *     opaque storage that holds the engine handle (the reference's members were never part of the calling contract);
 *   - coeffProvider[k] points at 39 290 900 bytes of spleeterCoeff for stem k = drum, bass, accompaniment, vocal
 *     (PluginProcessor.cpp:50-53); copied to HBM at init;
 *   - Spleeter4StemsProcessSamples consumes inSampleCount samples per channel (the plugin passes <= 1024) and writes up
 *     to inSampleCount samples to each of the 8 planar outputs (stem-major L/R pairs).  A 1024-sample segment is produced
 *     per completed hop; masks of batch n are applied while batch n+2 is collected, i.e. the end-to-end delay is
 *     2*timeStep hops + 1024 samples, and the first 2*timeStep hops are silence (Spleeter4Stems.c:257-381,512-582);
 *   - out-of-band bins (>= spectral bin limit) are scaled by 0.25 for stems 0,2,3 and 0 for stem 1 (Spleeter4Stems.c:73,281).
 * Per hop the GPU runs one forward and four masked inverse 4096-point FFTs + 50 % overlap-add on a dedicated stream; every
 * timeStep hops the four U-Nets run on a second stream, overlapped with the following hops exactly like the reference's
 * four network threads, and are joined one batch later (Spleeter4Stems.c:351-371).
 * Failures never abort() and never fall back to a CPU path: the reason goes to stderr and srtLastError(), the instance is
 * marked failed and from then on returns zeros / silence with the reference's sample accounting (SPLEETERRT_ABORT_ON_ERROR=1 aborts instead).
 */
#ifndef SPLEETERRT_AMD_SPLEETER4STEMS_H
#define SPLEETERRT_AMD_SPLEETER4STEMS_H
/* Everything a translation unit gets from the reference header besides the three functions is kept, because the
   plugin relies on it: PluginProcessor.cpp:6-9 includes ONLY this header and then calls getCoeffSize() (:48, declared by
   the nested spleeter.h, Spleeter4Stems.h:23) and min(n - offset, OVPSIZE) (:178, the macro of Spleeter4Stems.h:10-12).
   tests/test_plugin_header.py compiles a caller of that shape against this header (and against the reference's). */
#ifndef FFTSIZE
#define FFTSIZE 4096
#endif
#define ANALYSIS_OVERLAP 4
#define OVPSIZE (FFTSIZE / ANALYSIS_OVERLAP)
#define OUTPUTSEG ((OVPSIZE >> 1) << 1)
#define SAMPLESHIFT (FFTSIZE - (OVPSIZE << 1))
#define MINUSFFTSIZE (FFTSIZE - 1)
#ifndef HALFWNDLEN
#define HALFWNDLEN ((FFTSIZE >> 1) + 1)
#endif
#define MAX_OUTPUT_BUFFERS 2
#define LATENCY ((OVPSIZE << 1) - OUTPUTSEG)
#ifndef min
#define min(a,b) (((a)<(b))?(a):(b))
#endif
#define COMPONENTS 8
#define TASK_NB 5
#ifndef SRT_PT_STATE_DEFINED   /* Spleeter4Stems.h:14-20; stftFix.h declares the same enum */
#define SRT_PT_STATE_DEFINED
enum pt_state { SETUP, IDLE, WORKING, GET_OFF_FROM_WORK };
#endif
#include "spleeter.h"           /* tile API + weight-layout types, as Spleeter4Stems.h:23 */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct
{
    void *impl;                  /* engine + device buffers + host ring/queue state */
    unsigned char reserved[248]; /* keeps the struct a fixed 256 bytes for callers that embed it */
} Spleeter4Stems;
#define S4S_API __attribute__((visibility("default")))
S4S_API void Spleeter4StemsInit(Spleeter4Stems *msr, int initSpectralBinLimit, int initTimeStep, void *coeffProvider[4]);
S4S_API void Spleeter4StemsFree(Spleeter4Stems *msr);
S4S_API void Spleeter4StemsProcessSamples(Spleeter4Stems *msr, const float *inLeft, const float *inRight, int inSampleCount, float **components);
#ifdef __cplusplus
}
#endif
#endif
