/*
 * spleeterrt_amd.h — device-resident C ABI of the MI355X separation engine (libspleeterrt_amd.so).
 *
 * Plain C, no torch / HIP types in the signatures: `stream` is a hipStream_t passed as void*, every `d_*`
 * pointer is a device (HBM) address, every `h_*` pointer is host memory.  All functions return 0 on success
 * and a negative code on failure (srtLastError() gives the text); nothing here ever falls back to a CPU path.
 *
 * This is the batched, HBM-resident form of the reference's per-tile calls.  What each entry point replaces
 * (file:line under the reference tree):
 *   srtCreate / srtDestroy      allocateSpleeterStr + initSpleeter + freeSpleeter   Executable/spleeter.c:111-176,310-320
 *                               and InitSTFT / FreeSTFT                             Executable/stftFix.c:302-362
 *   srtSetCoeff*                the borrowed `coeff` argument of initSpleeter        Executable/spleeter.c:129
 *                               / lib2stem_loadCoefficients (fp16 container)         Executable/main.c:423-443
 *   srtForward                  processSpleeter for nstems x ntiles tiles at once    Executable/spleeter.c:177-301
 *   srtStft                     stft + the magnitude loop of processMT               Executable/stftFix.c:363-495, main.c:462-471
 *   srtIstft                    the mask loop of processMT + istft                   Executable/main.c:473-494, stftFix.c:496-579
 *   srtSeparate                 main()'s stft -> processMT -> istft sequence         Executable/main.c:776-785
 *   srtSeparateCli              main()'s two- and three-output flows incl. residuals  Executable/main.c:776-798, 845-928
 * The drop-in, host-pointer forms with the reference's exact signatures are in spleeter.h / stftFix.h.
 */
#ifndef SPLEETERRT_AMD_H
#define SPLEETERRT_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SRT_API __attribute__((visibility("default")))
#define SRT_MAX_STEMS 8
#define SRT_VARIANT_EXE 0   /* LUT sigmoid + ELU clamp  (Executable/spleeter.c:29-56) */
#define SRT_VARIANT_VST 1   /* exact sigmoid, plain ELU (VST/Source/spleeter.c:56-77) */
#define SRT_IMPL_MFMA   0   /* MFMA implicit-GEMM kernels (product path) */
#define SRT_IMPL_NAIVE  1   /* one-thread-per-output HIP kernels (debug cross-check, still GPU) */
#define SRT_PREC_F32    0   /* v_mfma_f32_32x32x2_f32: exact fp32 products (default; the headline path) */
#define SRT_PREC_F16    1   /* v_mfma_f32_32x32x16_f16, activations rounded to fp16 (BASELINE configs[4]; mask tolerance 2e-2).  Large batches: the network's input
                             * magnitudes are rounded to halves too (saturating at 65504, ~60x what full-scale PCM gives), and srtSeparate keeps its own masks as halves;
                             * STFT / iSTFT and every mask handed to a caller stay fp32 */
#define SRT_PREC_F16X2  2   /* same MFMA, activations split hi+lo: products exact in fp32 BECAUSE the weights are fp16 values - srtSetCoeff* checks
                             * every conv weight of the blob and refuses (-5, srtLastError names the count) one that the fp16 pack would round, e.g. a raw
                             * fp32 .dat blob with 24-bit mantissas: nothing is rounded silently in this mode */

typedef struct srt_engine srt_engine;

typedef struct srt_config {
    int   F;                        /* analyseBinLimit: tile width in bins, multiple of 64, <= 2048   main.c:701 */
    int   T;                        /* timeStep: tile height in frames, multiple of 64                main.c:702 */
    int   n_stems;                  /* sub-networks evaluated per tile, 1..SRT_MAX_STEMS */
    int   stem_mode[SRT_MAX_STEMS]; /* 0: LeakyReLU/ReLU, !=0: ELU/ELU                                spleeter.c:130-139 */
    float oob_weight[SRT_MAX_STEMS];/* weight of bins F..2048 ("unaffectedWeight" = 0.1)              main.c:773 */
    int   variant;                  /* SRT_VARIANT_* */
    int   max_tiles;                /* capacity: tiles per batch */
    int   impl;                     /* SRT_IMPL_* */
    int   precision;                /* SRT_PREC_*: arithmetic of the conv contraction (accumulation and everything else is fp32) */
    int   ratio_mask;               /* 0 (reference behaviour: raw sigmoid masks) | 1: srtSeparate / srtSeparateEx / srtSeparateHostStream normalise
                                     * m_s^2 / sum_j m_j^2 across stems (README.MD:82-85).  The CLI flows (srtSeparateCli*) reject it: their
                                     * sub-networks run one after the other on different inputs, so there is no stem axis to normalise over. */
    int   batch_invariant;          /* 0 (default): the fastest kernel per launch - small launches (<= 16 instances) cut the deep layers' K loops
                                     * into slices (split-K) and keep the direct kernels, larger ones run up1..up5 and down3..down6 in Winograd form, so the
                                     * same tile can differ in the last bits (<= 2e-5 on masks) with the batch it is evaluated in.
                                     * 1: kernel choice by layer geometry only, no split-K: a tile's result is bit-identical whatever the batch
                                     * size, tile slot, stem range or rank partition (what the reference's CPU path guarantees); small batches
                                     * run slower.  The environment variable SPLEETERRT_BATCH_INVARIANT=1 forces it for every engine. */
} srt_config;

SRT_API int  srtCreate(const srt_config *cfg, void *stream, srt_engine **out);
SRT_API void srtDestroy(srt_engine *e);
SRT_API const char *srtLastError(void);
SRT_API size_t srtCoeffBytes(void);                                           /* == getCoeffSize() == 39 290 900 */

/* weights for one sub-network, layout of spleeterCoeff (spleeter.h).  The data is copied; nothing is borrowed. */
SRT_API int  srtSetCoeffHost(srt_engine *e, int stem, const void *h_coeff);
SRT_API int  srtSetCoeffDevice(srt_engine *e, int stem, const void *d_coeff);
SRT_API int  srtSetCoeffFp16Host(srt_engine *e, int stem, const uint16_t *h_halfs);   /* spleeterQuantizedSubNet */
SRT_API int  srtGetCoeffHost(srt_engine *e, int stem, void *h_coeff);                 /* read the stored fp32 blob back (srtCoeffBytes() bytes): tests of the fp16 expansion */

/* d_mag: [ntiles][2][T][F] magnitudes; d_masks: [n_stems][ntiles][2][T][F] */
SRT_API int  srtForward(srt_engine *e, const float *d_mag, int ntiles, float *d_masks);

/* the same for sub-networks [stem0, stem0+nstems) only; d_masks keeps the all-stem layout (stem s at s*ntiles*2*T*F) */
SRT_API int  srtForwardStems(srt_engine *e, const float *d_mag, int ntiles, float *d_masks, int stem0, int nstems);
/* in place on d_masks [n_stems][ntiles][2][T][F]: m_s <- (m_s^2 + 1e-10/S) / (sum_j m_j^2 + 1e-10)  (official-Spleeter ratio mask; not in the reference) */
SRT_API int  srtRatioMask(srt_engine *e, float *d_masks, int ntiles);

/* geometry helpers for an n-sample stereo signal (n >= 4096) */
SRT_API size_t srtStftRows(size_t n);            /* ceil(n/1024): rows the reference allocates          stftFix.c:367 */
SRT_API size_t srtStftFrames(size_t n);          /* rows that actually receive a transform              stftFix.c:378 */
SRT_API size_t srtIstftLength(size_t rows);      /* rows*1024 + 3072                                    stftFix.c:500 */

/* STFT of planar stereo PCM resident in HBM.  d_spec: [2][rows][2052] interleaved (re,im) with the reference's
 * conjugate convention; rows = srtStftRows(n).  d_mag (optional, may be NULL): [ceil(rows/T)][2][T][F]. */
SRT_API int  srtStft(srt_engine *e, const float *d_L, const float *d_R, size_t n, float *d_spec, float *d_mag);
/* d_masks: [n_stems][ntiles][2][T][F] (NULL = all-ones for bins < F; bins F..2048 always get oob_weight[stem]).  d_out: [n_stems][2][srtIstftLength(rows)] */
SRT_API int  srtIstft(srt_engine *e, const float *d_spec, size_t rows, const float *d_masks, float *d_out);
/* whole hot path, everything in HBM: PCM -> STFT -> |.| -> U-Nets -> mask -> iSTFT.  d_out as in srtIstft. */
SRT_API int  srtSeparate(srt_engine *e, const float *d_L, const float *d_R, size_t n, float *d_out);
/* Explicit-geometry forms for streams cut into tile ranges (shards / chunks): transform `frames` frames (frame i starts at
 * sample i*1024, zero padded past n) and emit `rows` >= frames rows (the extra rows are zero, as the reference's calloc). */
SRT_API int  srtStftEx(srt_engine *e, const float *d_L, const float *d_R, size_t n, size_t frames, size_t rows, float *d_spec, float *d_mag);
SRT_API int  srtSeparateEx(srt_engine *e, const float *d_L, const float *d_R, size_t n, size_t frames, size_t rows, float *d_out);

/* A long HOST-resident stream through one GPU: cut into chunks of max_tiles tiles, upload / compute / download overlapped on
 * three HIP streams with double buffers, chunk overlaps (3072 samples) added on the device.  Geometry as srtSeparateEx.
 * h_out: [n_stems][2][srtIstftLength(rows)].  Synchronous; replaces main()'s whole-file stft -> processMT -> istft
 * (Executable/main.c:776-785) for inputs of any length (the reference holds the full 4096-wide spectrogram in RAM). */
SRT_API int  srtSeparateHostStream(srt_engine *e, const float *h_L, const float *h_R, size_t n, size_t frames, size_t rows, float *h_out);
/* The same with flags.  SRT_HOST_PINNED: h_L, h_R and h_out are already page-locked (hipHostMalloc / hipHostRegister), so the
 * call registers nothing (registering GBs of pageable memory per call costs more than the separation itself).  The device
 * double buffers, copy streams and events are kept in the engine between calls either way. */
#define SRT_HOST_PINNED 1u
SRT_API int  srtSeparateHostStreamEx(srt_engine *e, const float *h_L, const float *h_R, size_t n, size_t frames, size_t rows, float *h_out, unsigned flags);

/* The offline CLI's flows (Executable/main.c:776-798 for stems == 2, :845-928 for stems == 3), everything in HBM.
 * Sub-network 0 = the CLI's net[0] (drum, stem_mode 1), sub-network 1 = net[1] (vocal, stem_mode 0)  (main.c:759-760).
 * d_out: [stems][2][srtIstftLength(srtStftRows(n))] in the CLI's output order: Vocal, Accompaniment | Drum, Vocal, Accompaniment. */
SRT_API int  srtSeparateCli(srt_engine *e, const float *d_L, const float *d_R, size_t n, int stems, float *d_out);
/* host buffers, synchronous.  Its device staging (whole-file PCM + outputs when the file fits max_tiles, chunk double buffers otherwise) is
 * grow-only and kept for later calls, like srtSeparateHostStream's; a one-shot caller frees it with srtReleaseStaging. */
SRT_API int  srtSeparateCliHost(srt_engine *e, const float *h_L, const float *h_R, size_t n, int stems, float *h_out);

/* Low-latency callers that repeat the same call (same device pointers and sizes) over and over - the real-time plugin, the tile
 * API on one pair of buffers: with graph mode on, srtForward and srtSeparate / srtSeparateEx capture their launch sequence into
 * a hipGraph the first time an argument tuple is seen and replay it afterwards (one host call instead of ~25 launches; 4 cached
 * tuples, least recently used evicted).  Needs an explicit stream (the legacy null stream cannot be captured: the engine then
 * keeps launching eagerly).  Results are identical either way.  Off by default. */
SRT_API int  srtSetGraphMode(srt_engine *e, int enable);
/* Everything the first srtForward(e, d_mag, ntiles, d_masks) would otherwise do lazily - the split-K workspace allocation and, in
 * graph mode, the capture + instantiation of the launch sequence for exactly this argument tuple - done NOW and synchronised, so
 * that the first real call (on a real-time audio thread) is a plain graph launch.  Runs the networks once: d_masks is overwritten.
 * The drop-in layers call it from their Init functions for every buffer pair they will use. */
SRT_API int  srtPrepareForward(srt_engine *e, const float *d_mag, int ntiles, float *d_masks);
/* free the grow-only device staging of the host-buffer entry points (srtSeparateHostStream*, srtSeparateCliHost); the next such call re-allocates */
SRT_API int  srtReleaseStaging(srt_engine *e);

/* ---- one node, several devices: the reference CLI's tile-range fan-out (Executable/main.c:544-673, `processMT`: spawnNthreads workers, each with
 * its own network instance and a contiguous tile range, one shared read-only weight blob) with a GPU where the reference has a CPU thread.
 * One engine per entry of `devices` (an index may repeat: several engines on one GPU), one host thread per engine while a call runs; weights are
 * uploaded once and distributed with one ncclBroadcast per blob over the devices (RCCL, loaded on first use; hipMemcpyPeer when it is absent or
 * SPLEETERRT_NO_RCCL=1); no data-path collective - neighbouring ranges share 3072 output samples, which the call adds on the host when it joins. */
typedef struct srt_span {           /* one rank's share of an n-sample stream; same arithmetic as spleeterrt_amd/stream.py:rank_span */
    size_t tile0, tile1;            /* tiles [tile0, tile1) */
    size_t sample0, nsamples;       /* PCM samples [sample0, sample0 + nsamples): the range + its 3072-sample halo */
    size_t frames, rows;            /* frames that receive a transform / spectrogram rows of the range (srtSeparateEx geometry) */
    size_t out_offset;              /* where the range's overlap-add contribution (rows*1024 + 3072 samples) starts in the output */
} srt_span;
typedef struct srt_multi srt_multi;
SRT_API int  srtDeviceCount(void);                                               /* HIP devices visible to the process (0: none) */
SRT_API int  srtRankSpan(size_t n, int T, int rank, int world, srt_span *out);   /* pure host arithmetic: works without a device */
SRT_API int  srtMultiCreate(const srt_config *cfg, const int *devices /* NULL: 0..ndev-1 */, int ndev, srt_multi **out);  /* cfg->max_tiles: per engine */
SRT_API void srtMultiDestroy(srt_multi *m);
SRT_API int  srtMultiSetCoeffHost(srt_multi *m, int stem, const void *h_coeff);              /* spleeterCoeff, fp32 */
SRT_API int  srtMultiSetCoeffFp16Host(srt_multi *m, int stem, const uint16_t *h_halfs);      /* spleeterQuantizedSubNet */
/* whole host-resident stream, the engine's n_stems sub-networks: h_out [n_stems][2][srtIstftLength(srtStftRows(n))]; flags as srtSeparateHostStreamEx */
SRT_API int  srtMultiSeparateHost(srt_multi *m, const float *h_L, const float *h_R, size_t n, float *h_out, unsigned flags);
/* the offline CLI's flows (srtSeparateCliHost) over the devices: h_out [stems][2][srtIstftLength(srtStftRows(n))] */
SRT_API int  srtMultiSeparateCliHost(srt_multi *m, const float *h_L, const float *h_R, size_t n, int stems, float *h_out);
/* "engines=2 devices=0,1 distinct=2 weights=rccl broadcasts=2" (tests, logs); returns the number of engines */
SRT_API int  srtMultiInfo(const srt_multi *m, char *text, size_t bytes);
/* engine g of the object (borrowed; for srtSetTiming / srtGetTiming* / srtCopyTensor on it), NULL outside [0, engines) */
SRT_API srt_engine *srtMultiEngine(srt_multi *m, int g);
/* Resident throughput of all engines at once, the measurement `bench.py --host native` reports: every worker thread fills `tiles` (<= max_tiles)
 * tiles of synthetic PCM in its device's HBM, runs `warmup` untimed srtSeparate passes, meets the others at a barrier, runs `steps` passes and
 * synchronises its device.  *seconds = first worker released -> last worker done.  seconds_events (may be NULL): the same K passes once more with
 * per-launch HIP events switched on for engine 0 (read them with srtGetTiming(srtMultiEngine(m, 0), ...); timing stays on). */
SRT_API int  srtMultiBenchResident(srt_multi *m, int tiles, int steps, int warmup, double *seconds, double *seconds_events);
/* (with seconds_events != NULL the per-launch timing window of engine 0 stays OPEN after the call so that srtGetTiming / srtGetTimingKernels on
 * srtMultiEngine(m, 0) can read the K event-timed passes; the caller closes it with srtSetTiming(engine, 0)) */

/* debug / measurement */
SRT_API int  srtCopyTensor(srt_engine *e, const char *name, int stem, int tile, float *h_dst, size_t max_floats); /* "conv1".."conv6","act1".."act5","up1".."up6" */
SRT_API int  srtSetTiming(srt_engine *e, int enable);                    /* record HIP events around every launch of the next calls */
SRT_API int  srtGetTiming(srt_engine *e, char *names, size_t names_bytes, float *ms, int max_entries); /* returns count; syncs the stream */
/* which kernel ran each of those launches (same order), ';'-separated, named as rocprofv3 names kernels ("srt_dec_wino<4, 16, 1, 0>"):
 * the dispatch depends on batch size and geometry, so tests and bench.py read it from here instead of assuming it.  Returns the count. */
SRT_API int  srtGetTimingKernels(srt_engine *e, char *kernels, size_t kernels_bytes);

#ifdef __cplusplus
}
#endif
#endif
