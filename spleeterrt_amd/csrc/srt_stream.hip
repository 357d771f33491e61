// srt_stream.hip — the reference's real-time streaming surface (include/Spleeter4Stems.h) on the GPU engine.
//
// Host side keeps exactly the reference's bookkeeping (input ring, "samples needed" counter, two queued output
// segments interleaved by 8: VST/Source/Spleeter4Stems.c:512-582); every completed hop launches
//   srt_stream_inverse_kernel  x4 stems : delayed spectrum row x mask -> inverse FFT -> synthesis window -> 50 % OLA
//   srt_stream_forward_kernel           : asymmetric-window FFT of the current 4096 samples -> spectrum + magnitude row
// on the hop stream and copies the 1024 x 8 segment back.  Every timeStep hops the four U-Nets are started on the
// engine's own stream (the reference's task_type2 threads, Spleeter4Stems.c:135,351-371) and joined one batch later.
#include "srt_internal.h"
#include "../../include/spleeterrt_amd.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include "../../include/Spleeter4Stems.h"
#undef min   // the header keeps the reference's C macro (Spleeter4Stems.h:10-12) for the plugin; not wanted in this C++ file

// Failure policy (this is the host's real-time audio thread, VST/Source/PluginProcessor.cpp:178-179): never abort(), never a
// CPU path.  The first failure is reported once (stderr + srtLastError()), the instance is marked failed and from then on it
// emits SILENCE with the reference's exact sample accounting, without touching the device again.
static bool stream_fail(const char* where, const char* why)
{
    char buf[400];
    snprintf(buf, sizeof buf, "%s: %s", where, why ? why : srtLastError());
    srt_set_error(-2, "%s", buf);
    fprintf(stderr, "libspleeterrt_amd: %s (no CPU fallback exists; the stream is muted)\n", buf);
    return false;
}
#define HIPTRY(x, where) do { hipError_t _e = (x); if (_e != hipSuccess) { s->failed = true; stream_fail(where, hipGetErrorString(_e)); goto failed; } } while (0)

namespace {
struct Stream {
    srt_engine* eng;
    hipStream_t hop, nn;
    hipEvent_t evMag, evNN;
    int F, T, cursor, ptr;
    bool nnRunning, failed;
    float* d_ring; float2* d_spec; float* d_mag; float* d_tmp; float* d_masks; float* d_overlap; float* d_out;
    float *d_awin, *d_swin; float2* d_tw;
    size_t hw;
    // host state, mirrors Spleeter4Stems.h:35-47
    float ring[2][FFTSIZE];
    unsigned inPos, needed;
    float* outq[2]; float* pinned; float* hostq; // two queued segments of OUTPUTSEG*8 floats (pinned for the D2H copy; plain host memory on a failed instance)
    int outCount, outReadOff;
};

void asymmetric_window(std::vector<float>& an, std::vector<float>& sy)      // Spleeter4Stems.c:383-401 with k=4096, m=1024, p=1
{
    const int k = FFTSIZE, m = OVPSIZE;
    const double PI = 3.141592653589793;
    an.assign(k, 0.f); sy.assign(k, 0.f);
    int n = ((k - m) << 1) + 2;
    for (int i = 0; i < k - m; ++i) an[i] = (float)pow(0.5 * (1.0 - cos(2.0 * PI * (i + 1.0) / (double)n)), 1.0);
    n = (m << 1) + 2;
    for (int i = k - m; i < k; ++i) an[i] = (float)pow(sqrt(0.5 * (1.0 - cos(2.0 * PI * ((m + i - (k - m)) + 1.0) / (double)n))), 1.0);
    n = m << 1;
    for (int i = k - (m << 1); i < k; ++i) sy[i] = (float)(0.5 * (1.0 - cos(2.0 * PI * (double)(i - (k - (m << 1))) / (double)n))) / an[i];
    for (int i = 0; i < k - SAMPLESHIFT; ++i) sy[i] = sy[i + SAMPLESHIFT];   // pre-shift
    for (int i = 0; i < k; ++i) an[i] *= (1.0 / FFTSIZE) * 0.5f;             // Spleeter4Stems.c:414-416 (double product, float store)
}

void process_hop(Stream* s)                                                  // LLPAMSProcessNPR, Spleeter4Stems.c:257-381
{
    const size_t rowF2 = SRT_SPEC_LD, bufF2 = 2 * (size_t)s->T * rowF2;
    if (s->outCount >= 2) { float* t = s->outq[0]; s->outq[0] = s->outq[1]; s->outq[1] = t; s->outCount = 1; s->outReadOff = 0; }   // the reference overruns its 2-slot queue here (caller passed > 1024 samples without draining); drop the oldest segment instead
    float* dst = s->outq[s->outCount];
    s->outCount++;
    s->needed = OUTPUTSEG;
    if (s->failed) goto failed;
    {
        HIPTRY(hipMemcpyAsync(s->d_ring, s->ring, sizeof s->ring, hipMemcpyHostToDevice, s->hop), "stream hop");
        SrtStreamHop p; memset(&p, 0, sizeof p);
        p.ring = s->d_ring; p.inPos = (int)s->inPos;
        p.specRow = s->d_spec + s->ptr * bufF2 + (size_t)s->cursor * rowF2; p.specChStride = (size_t)s->T * rowF2;
        p.magRow = s->d_mag + (size_t)s->cursor * s->F; p.magChStride = s->hw;
        p.maskRow = s->d_masks + (size_t)s->ptr * 4 * 2 * s->hw + (size_t)s->cursor * s->F; p.maskStemStride = 2 * s->hw; p.maskChStride = s->hw;
        p.F = s->F; p.overlap = s->d_overlap; p.out = s->d_out;
        p.analysisWnd = s->d_awin; p.synthesisWnd = s->d_swin; p.twiddle = s->d_tw;
        if (srt_launch_stream_hop(p, s->hop)) { s->failed = true; stream_fail("stream hop", "kernel launch failed"); goto failed; }
        HIPTRY(hipMemcpyAsync(dst, s->d_out, OUTPUTSEG * 8 * sizeof(float), hipMemcpyDeviceToHost, s->hop), "stream hop");
        s->cursor++;
        if (s->cursor >= s->T) {
            // join the networks started one batch ago (their masks land in buffer !ptr), flip, start on this batch's magnitudes
            if (s->nnRunning) HIPTRY(hipStreamWaitEvent(s->hop, s->evNN, 0), "stream join");
            s->ptr = !s->ptr;
            HIPTRY(hipMemcpyAsync(s->d_tmp, s->d_mag, 2 * s->hw * sizeof(float), hipMemcpyDeviceToDevice, s->hop), "stream flip");   // "Prevent race condition" copy (:364-365)
            HIPTRY(hipEventRecord(s->evMag, s->hop), "stream flip");
            HIPTRY(hipStreamWaitEvent(s->nn, s->evMag, 0), "stream flip");
            if (srtForward(s->eng, s->d_tmp, 1, s->d_masks + (size_t)(!s->ptr) * 4 * 2 * s->hw)) { s->failed = true; stream_fail("stream networks", nullptr); goto failed; }
            HIPTRY(hipEventRecord(s->evNN, s->nn), "stream flip");
            s->nnRunning = true;
            s->cursor = 0;
        }
        HIPTRY(hipStreamSynchronize(s->hop), "stream hop");                  // the segment must be in host memory before the callback returns
        return;
    }
failed:
    memset(dst, 0, OUTPUTSEG * 8 * sizeof(float));                           // silence for this hop, same sample accounting
}
}  // namespace

void Spleeter4StemsInit(Spleeter4Stems* msr, int F, int T, void* coeffProvider[4])
{
    if (!msr) return;
    SrtSetupLock setup;                                      // (srt_internal.h: set-up paths are serialised process-wide)
    memset(msr, 0, sizeof *msr);
    Stream* s = new (std::nothrow) Stream();
    if (!s) { stream_fail("Spleeter4StemsInit", "out of host memory"); return; }
    memset(s, 0, sizeof *s);
    msr->impl = s;
    s->F = F; s->T = T; s->hw = (size_t)F * T;
    s->needed = OUTPUTSEG;
    // host-side queue first: a failed instance still accounts for samples (and emits silence) through it
    s->pinned = nullptr;
    s->hostq = (float*)calloc(2 * OUTPUTSEG * 8, sizeof(float));
    s->outq[0] = s->hostq; s->outq[1] = s->hostq ? s->hostq + OUTPUTSEG * 8 : nullptr;
    s->failed = true;                                    // until everything below has succeeded
    if (!s->hostq) { stream_fail("Spleeter4StemsInit", "out of host memory"); return; }
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { stream_fail("Spleeter4StemsInit", "no HIP device (this library has no CPU path)"); return; }
        for (int k = 0; k < 4; ++k) if (!coeffProvider || !coeffProvider[k]) { stream_fail("Spleeter4StemsInit", "null coefficient pointer"); return; }
#define INITTRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { stream_fail("Spleeter4StemsInit", hipGetErrorString(_e)); return; } } while (0)
        // non-blocking streams: no implicit ordering against the legacy null stream, so another instance's (another host thread's) synchronous
        // copies and memsets during ITS Init can neither stall this instance's hops nor invalidate the graph capture of this one's pre-warm
        // Priorities: the per-hop stream (one forward + eight inverse FFTs the audio callback WAITS for) gets the device's highest priority, the
        // network stream (four U-Nets joined only every T hops) the lowest - the reference gives the per-hop iFFT its own thread and joins the
        // network threads every T hops (Spleeter4Stems.c:351-371).  With several plugin instances on one GPU a hop's kernels are then dispatched
        // ahead of every instance's queued network kernels instead of waiting their turn behind them.
        int prLeast = 0, prGreatest = 0;
        INITTRY(hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest));
        INITTRY(hipStreamCreateWithPriority(&s->hop, hipStreamNonBlocking, prGreatest));
        INITTRY(hipStreamCreateWithPriority(&s->nn, hipStreamNonBlocking, prLeast));
        INITTRY(hipEventCreateWithFlags(&s->evMag, hipEventDisableTiming));
        INITTRY(hipEventCreateWithFlags(&s->evNN, hipEventDisableTiming));
        srt_config cfg; memset(&cfg, 0, sizeof cfg);
        cfg.F = F; cfg.T = T; cfg.n_stems = 4; cfg.variant = SRT_VARIANT_VST; cfg.max_tiles = 1; cfg.impl = SRT_IMPL_MFMA;
        for (int k = 0; k < 4; ++k) { cfg.stem_mode[k] = 1; cfg.oob_weight[k] = k == 1 ? 0.0f : 0.25f; }      // Spleeter4Stems.c:444-447
        if (srtCreate(&cfg, s->nn, &s->eng)) { s->eng = nullptr; stream_fail("Spleeter4StemsInit", nullptr); return; }
        for (int k = 0; k < 4; ++k)
            if (srtSetCoeffHost(s->eng, k, coeffProvider[k])) { stream_fail("Spleeter4StemsInit(weights)", nullptr); return; }
        srtSetGraphMode(s->eng, 1);                           // the four U-Nets run on the same buffers every T hops: replay one hipGraph per mask buffer
        const size_t specF = 2 * 2 * (size_t)T * SRT_SPEC_LD * 2;
        INITTRY(hipMalloc((void**)&s->d_ring, sizeof s->ring));
        INITTRY(hipMalloc((void**)&s->d_spec, specF * sizeof(float)));
        INITTRY(hipMalloc((void**)&s->d_mag, 2 * s->hw * sizeof(float)));
        INITTRY(hipMalloc((void**)&s->d_tmp, 2 * s->hw * sizeof(float)));
        INITTRY(hipMalloc((void**)&s->d_masks, 2 * 4 * 2 * s->hw * sizeof(float)));
        INITTRY(hipMalloc((void**)&s->d_overlap, 8 * 1024 * sizeof(float)));
        INITTRY(hipMalloc((void**)&s->d_out, OUTPUTSEG * 8 * sizeof(float)));
        INITTRY(hipMalloc((void**)&s->d_awin, FFTSIZE * sizeof(float)));
        INITTRY(hipMalloc((void**)&s->d_swin, FFTSIZE * sizeof(float)));
        INITTRY(hipMalloc((void**)&s->d_tw, FFTSIZE * sizeof(float2)));
        INITTRY(hipMemset(s->d_spec, 0, specF * sizeof(float)));              // zero spectrum for the first 2T hops (:423-438)
        INITTRY(hipMemset(s->d_mag, 0, 2 * s->hw * sizeof(float)));
        INITTRY(hipMemset(s->d_overlap, 0, 8 * 1024 * sizeof(float)));
        // Pre-warm on THIS thread: the split-K workspace allocation and the capture + instantiation of one hipGraph per mask buffer
        // would otherwise happen inside the host's audio callback at hops T and 2T (an allocation and a graph build there risk a dropout).
        INITTRY(hipMemset(s->d_tmp, 0, 2 * s->hw * sizeof(float)));
        INITTRY(hipStreamSynchronize(nullptr));               // the null-stream memsets are done before the two private (non-blocking) streams touch the buffers
        for (int b = 0; b < 2; ++b)
            if (srtPrepareForward(s->eng, s->d_tmp, 1, s->d_masks + (size_t)b * 4 * 2 * s->hw)) { stream_fail("Spleeter4StemsInit(prepare)", nullptr); return; }
        std::vector<float> ones(2 * 4 * 2 * s->hw, 1.0f), an, sy, tw(2 * FFTSIZE);                 // masks start at 1.0 (:456-467)
        INITTRY(hipMemcpy(s->d_masks, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
        asymmetric_window(an, sy);
        const double w0 = 6.283185307179586476925286766559 / FFTSIZE;
        for (int i = 0; i < FFTSIZE; ++i) { tw[2 * i] = (float)cos(w0 * i); tw[2 * i + 1] = (float)(-sin(w0 * i)); }
        INITTRY(hipMemcpy(s->d_awin, an.data(), FFTSIZE * 4, hipMemcpyHostToDevice));
        INITTRY(hipMemcpy(s->d_swin, sy.data(), FFTSIZE * 4, hipMemcpyHostToDevice));
        INITTRY(hipMemcpy(s->d_tw, tw.data(), 2 * FFTSIZE * 4, hipMemcpyHostToDevice));
        INITTRY(hipHostMalloc((void**)&s->pinned, 2 * OUTPUTSEG * 8 * sizeof(float), hipHostMallocDefault));   // pinned queue for the per-hop D2H copy
        INITTRY(hipStreamSynchronize(nullptr));               // masks / windows / twiddles (null-stream copies) are in place before the first hop
        // Pre-warm the per-hop path too: the first launch of the two hop kernels loads their code, and eight plugin instances making their first call at
        // the same time queued behind each other for it - the slowest call of every instance was its FIRST one, 7.5 ms (round 6, host/rt_latency.c
        // `worst_hop`).  One hop on silence here, on this thread: zero ring, zero spectrum, unit masks - every buffer it writes stays zero.
        INITTRY(hipMemsetAsync(s->d_ring, 0, sizeof s->ring, s->hop));
        {
            SrtStreamHop p; memset(&p, 0, sizeof p);
            const size_t rowF2 = SRT_SPEC_LD;
            p.ring = s->d_ring; p.inPos = 0;
            p.specRow = s->d_spec; p.specChStride = (size_t)s->T * rowF2;
            p.magRow = s->d_mag; p.magChStride = s->hw;
            p.maskRow = s->d_masks; p.maskStemStride = 2 * s->hw; p.maskChStride = s->hw;
            p.F = s->F; p.overlap = s->d_overlap; p.out = s->d_out;
            p.analysisWnd = s->d_awin; p.synthesisWnd = s->d_swin; p.twiddle = s->d_tw;
            if (srt_launch_stream_hop(p, s->hop)) { stream_fail("Spleeter4StemsInit(hop pre-warm)", "kernel launch failed"); return; }
            INITTRY(hipMemcpyAsync(s->pinned, s->d_out, OUTPUTSEG * 8 * sizeof(float), hipMemcpyDeviceToHost, s->hop));
            INITTRY(hipStreamSynchronize(s->hop));
        }
#undef INITTRY
        s->outq[0] = s->pinned; s->outq[1] = s->pinned + OUTPUTSEG * 8;
    }
    s->failed = false;
}

void Spleeter4StemsFree(Spleeter4Stems* msr)
{
    if (!msr || !msr->impl) return;
    Stream* s = (Stream*)msr->impl;
    if (s->hop) hipStreamSynchronize(s->hop);
    if (s->nn) hipStreamSynchronize(s->nn);
    if (s->eng) srtDestroy(s->eng);
    void* d[] = { s->d_ring, s->d_spec, s->d_mag, s->d_tmp, s->d_masks, s->d_overlap, s->d_out, s->d_awin, s->d_swin, s->d_tw };
    for (void* q : d) if (q) hipFree(q);
    if (s->pinned) hipHostFree(s->pinned);
    free(s->hostq);
    if (s->evMag) hipEventDestroy(s->evMag);
    if (s->evNN) hipEventDestroy(s->evNN);
    if (s->hop) hipStreamDestroy(s->hop);
    if (s->nn) hipStreamDestroy(s->nn);
    delete s;
    msr->impl = nullptr;
}

void Spleeter4StemsProcessSamples(Spleeter4Stems* msr, const float* inLeft, const float* inRight, int inSampleCount, float** components)
{
    Stream* s = msr ? (Stream*)msr->impl : nullptr;
    if (!s || !s->outq[0]) return;                                          // Init could not even allocate its host state: nothing is written
    int outSampleCount = 0;
    const int maxOut = inSampleCount;
    while (inSampleCount > 0) {                                             // Spleeter4Stems.c:518-537
        const int c = (int)s->needed < inSampleCount ? (int)s->needed : inSampleCount;
        memcpy(&s->ring[0][s->inPos], inLeft, c * sizeof(float));
        memcpy(&s->ring[1][s->inPos], inRight, c * sizeof(float));
        inLeft += c; inRight += c; inSampleCount -= c;
        s->inPos = (s->inPos + c) & (FFTSIZE - 1);
        s->needed -= c;
        if (s->needed == 0) process_hop(s);
    }
    float* io[COMPONENTS];
    for (int j = 0; j < COMPONENTS; ++j) io[j] = components[j];
    while (s->outCount > 0 && outSampleCount < maxOut) {                    // Spleeter4Stems.c:540-581
        const float* src = s->outq[0] + (size_t)s->outReadOff * COMPONENTS;
        int c = OUTPUTSEG - s->outReadOff;
        if (c > maxOut - outSampleCount) c = maxOut - outSampleCount;
        for (int i = 0; i < c; ++i)
            for (int j = 0; j < COMPONENTS; ++j) *io[j]++ = *src++;
        outSampleCount += c;
        s->outReadOff += c;
        if (s->outReadOff == OUTPUTSEG) {
            s->outCount--;
            s->outReadOff = 0;
            if (s->outCount > 0) { float* t = s->outq[0]; s->outq[0] = s->outq[1]; s->outq[1] = t; }
        }
    }
}
