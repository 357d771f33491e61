// srt_dsp.hip — STFT / magnitude / mask-apply / inverse STFT / overlap-add kernels for gfx950.
//
// Replaces, for the hot path only:
//   stft    : Executable/stftFix.c:363-495 (window, bit-reverse, DFT4096 x2, re/im unpack)      K1
//   mag     : Executable/main.c:462-471, :500-514                                              K2
//   mask    : Executable/main.c:473-494                                                        K8
//   istft   : Executable/stftFix.c:496-579 (Hartley pack, DFT4096, post-window, overlap-add)   K9
//   DFT4096 : Executable/codelet.c:2-271  (4096-point fast Hartley transform)
//
// Design (not a translation of the Hartley codelet): one 256-thread workgroup runs a 4096-point COMPLEX FFT
// entirely in registers + LDS as three radix-16 passes (16 values per thread).  Left and right channels ride in
// the real and imaginary parts of the same transform (z = L + iR), so one FFT serves both channels of a frame;
// the two spectra are separated in the epilogue, where the magnitude tile for the network is emitted as well.
// The inverse uses the same FFT (swap trick) on G = F'_L + i F'_R after the per-stem mask multiply (one workgroup
// column per stem).  Overlap-add is fused and register resident: frames are added in frame order, every output
// sample is written once (see srt_istft_ola_kernel).
#include "srt_internal.h"
#include "srt_device.h"
#include <stdlib.h>

#define FFT_EX1_LD 272      // exchange-1 row stride (cf): 272*2 dwords = 32 (mod 64) -> conflict-free b64 reads
#define FFT_EX2_LD 257      // exchange-2 row stride (cf): odd -> conflict-free strided b64 writes
#define FFT_SMEM_F2 4352    // 16*272 cf scratch (also holds 4096 natural-order points)

// Complex values are 2-wide vectors so the butterflies map onto the packed fp32 VALU (v_pk_add/mul/fma_f32: two lanes per
// instruction).  The two operations the compiler does not fold into one packed instruction by itself — a +/- (-i)b and the
// complex product — are spelled out with their op_sel / neg modifiers (half swaps and sign flips are free on VOP3P);
// with them a radix-4 butterfly is 8 packed adds and a complex multiply is 2 packed ops (2.1x fewer VALU instructions
// per FFT than scalar-component code run through the SLP vectoriser).
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf f2(float x, float y) { cf r = { x, y }; return r; }
__device__ __forceinline__ cf add_mi(cf a, cf b)       // a + (-i) b = (a.x + b.y, a.y - b.x)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf sub_mi(cf a, cf b)       // a - (-i) b = (a.x - b.y, a.y + b.x)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf cmul(cf a, cf w)         // (a.x w.x - a.y w.y, a.y w.x + a.x w.y)
{
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                                              // (a.x w.x, a.y w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));   // (-a.y w.y + t.x, a.x w.y + t.y)
    return r;
}

__device__ __forceinline__ cf herm_hi(cf b, cf a)       // (b.x + a.y, a.x - b.y)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(b), "v"(a));
    return r;
}

// separation of the two real spectra packed in one complex transform (z = L + iR): with zk = Z[k], zm = Z[N-k]
__device__ __forceinline__ cf split_l(cf zk, cf zm)     // (zk.x + zm.x, zm.y - zk.y)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[1,0]" : "=v"(r) : "v"(zk), "v"(zm));
    return r;
}
__device__ __forceinline__ cf split_r(cf zk, cf zm)     // (zk.y + zm.y, zk.x - zm.x)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(zk), "v"(zm));
    return r;
}

// forward 4-point DFT in place (W4 = -i)
__device__ __forceinline__ void dft4(cf& a, cf& b, cf& c, cf& d)
{
    const cf s0 = a + c, s1 = a - c, s2 = b + d, s3 = b - d;
    a = s0 + s2;
    c = s0 - s2;
    b = add_mi(s1, s3);
    d = sub_mi(s1, s3);
}

// forward 16-point DFT, natural-order input v[n]; OUTPUT X[k] is left at v[4*(k&3) + (k>>2)]
#define FFT16_AT(k) (4 * ((k) & 3) + ((k) >> 2))
__device__ __forceinline__ void fft16(cf (&v)[16])
{
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
    // y[n0][k0] sits at v[n0 + 4*k0]; twiddle by W16^(n0*k0)
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, r2 = 0.70710678118654752440f;
    v[1 + 4 * 1] = cmul(v[1 + 4 * 1], f2(c1, -s1));      // W^1
    v[2 + 4 * 1] = cmul(v[2 + 4 * 1], f2(r2, -r2));      // W^2
    v[3 + 4 * 1] = cmul(v[3 + 4 * 1], f2(s1, -c1));      // W^3
    v[1 + 4 * 2] = cmul(v[1 + 4 * 2], f2(r2, -r2));      // W^2
    v[2 + 4 * 2] = add_mi(f2(0.f, 0.f), v[2 + 4 * 2]);   // W^4 = -i
    v[3 + 4 * 2] = cmul(v[3 + 4 * 2], f2(-r2, -r2));    // W^6
    v[1 + 4 * 3] = cmul(v[1 + 4 * 3], f2(s1, -c1));      // W^3
    v[2 + 4 * 3] = cmul(v[2 + 4 * 3], f2(-r2, -r2));     // W^6
    v[3 + 4 * 3] = cmul(v[3 + 4 * 3], f2(-c1, s1));      // W^9
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) dft4(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
}

// Per-thread twiddles are frame independent, so each workgroup re-lays the (double-precision generated) table once
// into the order its threads read it: twA[k0-1][tid] = W^(tid*k0), twB[k1-1][lo] = W^(16*lo*k1).  All twiddle reads are
// then consecutive or broadcast (conflict-free), instead of stride-k gathers into a 4096-entry table.
#define FFT_TW_F2 (15 * 256 + 15 * 16)
__device__ __forceinline__ void fft_load_twiddles(cf* tw, const float2* __restrict__ table_f2, int tid)
{
    const cf* __restrict__ table = reinterpret_cast<const cf*>(table_f2);
#pragma unroll
    for (int k0 = 1; k0 < 16; ++k0) tw[(k0 - 1) * 256 + tid] = table[(tid * k0) & 4095];
    if (tid < 240) tw[15 * 256 + tid] = table[(16 * (tid & 15) * (tid / 16 + 1)) & 4095];
}

// 4096-point forward FFT.  In: v[n2] = x[tid + 256*n2].  Out: v[FFT16_AT(k2)] = X[tid + 256*k2].
// s: FFT_SMEM_F2 cf of LDS scratch, tw: the re-laid twiddles (fft_load_twiddles).
// The caller must __syncthreads() before reusing s after return.
__device__ __forceinline__ void fft4096(cf (&v)[16], cf* s, const cf* tw, int tid)
{
    fft16(v);                                            // over n2 -> k0
#pragma unroll
    for (int k0 = 1; k0 < 16; ++k0) v[FFT16_AT(k0)] = cmul(v[FFT16_AT(k0)], tw[(k0 - 1) * 256 + tid]);
#pragma unroll
    for (int k0 = 0; k0 < 16; ++k0) s[k0 * FFT_EX1_LD + tid] = v[FFT16_AT(k0)];
    __syncthreads();
    const int lo = tid & 15, hi = tid >> 4;              // (n0, k0)
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = s[hi * FFT_EX1_LD + n1 * 16 + lo];
    fft16(v);                                            // over n1 -> k1
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[FFT16_AT(k1)] = cmul(v[FFT16_AT(k1)], tw[15 * 256 + (k1 - 1) * 16 + lo]);
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) s[lo * FFT_EX2_LD + k1 * 16 + hi] = v[FFT16_AT(k1)];
    __syncthreads();
#pragma unroll
    for (int n0 = 0; n0 < 16; ++n0) v[n0] = s[n0 * FFT_EX2_LD + tid];   // tid = k0 + 16*k1
    fft16(v);                                            // over n0 -> k2
}

// The same transform for kernels that walk MANY frames per workgroup, with half the barriers: the 15 pass-1 twiddles of a thread
// are frame independent and live in registers (twa), which shrinks the LDS twiddle table to the 240 pass-2 entries and makes
// room for TWO exchange buffers.  With exchange 1 in one buffer and exchange 2 in the other no barrier is needed between
// reading an exchange and writing the next one:
//   in : v[n2] = x[tid + 256*n2], read by every thread from buffer `e2` (or from registers) AFTER a barrier that follows the
//        last use of buffer `e1` by the previous frame
//   out: v[FFT16_AT(k2)] = X[tid + 256*k2]; `e1` is free again, `e2` still holds exchange 2 until the caller's next barrier
// A frame then costs 3 barriers (staging, exchange 1, exchange 2) instead of 6.
#define FFT_TWB_F2 240
__device__ __forceinline__ void fft_load_twiddles_pp(cf (&twa)[15], cf* twb, const float2* __restrict__ table_f2, int tid)
{
    const cf* __restrict__ table = reinterpret_cast<const cf*>(table_f2);
#pragma unroll
    for (int k0 = 1; k0 < 16; ++k0) twa[k0 - 1] = table[(tid * k0) & 4095];
    if (tid < FFT_TWB_F2) twb[tid] = table[(16 * (tid & 15) * (tid / 16 + 1)) & 4095];
}
__device__ __forceinline__ void fft4096_pp(cf (&v)[16], cf* e1, cf* e2, const cf (&twa)[15], const cf* twb, int tid)
{
    fft16(v);                                            // over n2 -> k0
#pragma unroll
    for (int k0 = 1; k0 < 16; ++k0) v[FFT16_AT(k0)] = cmul(v[FFT16_AT(k0)], twa[k0 - 1]);
#pragma unroll
    for (int k0 = 0; k0 < 16; ++k0) e1[k0 * FFT_EX1_LD + tid] = v[FFT16_AT(k0)];
    __syncthreads();
    const int lo = tid & 15, hi = tid >> 4;              // (n0, k0)
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = e1[hi * FFT_EX1_LD + n1 * 16 + lo];
    fft16(v);                                            // over n1 -> k1
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[FFT16_AT(k1)] = cmul(v[FFT16_AT(k1)], twb[(k1 - 1) * 16 + lo]);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) e2[lo * FFT_EX2_LD + k1 * 16 + hi] = v[FFT16_AT(k1)];
    __syncthreads();
#pragma unroll
    for (int n0 = 0; n0 < 16; ++n0) v[n0] = e2[n0 * FFT_EX2_LD + tid];   // tid = k0 + 16*k1
    fft16(v);                                            // over n0 -> k2
}

#define STFT_FPB 8      // frames per workgroup on long signals (amortises the twiddle-table load; consecutive frames share 75% of input)

// Inverse STFT with the overlap-add fused in (no frame scratch, no second pass, no atomics).
// With the radix-16 FFT's output distribution thread w holds samples w + 256*k2 of a frame, i.e. for each of the four
// 1024-sample quarters the SAME four in-hop offsets q_j = w + 256 j.  Overlap-add across frames is therefore
// thread-local: a workgroup (blockIdx.y = stem) walks a run of consecutive frames and keeps a rolling window of four
// partial output segments in registers; segment s is complete once frame s has been added (frames s-3..s, added in that
// order = the reference's accumulation order, stftFix.c:570-575) and is written out exactly once.  A workgroup that
// owns segments [s0, s1) recomputes the three frames before s0 as warm-up ((G+3)/G extra work).
// The loop body is straight-line: the next frame's rows are prefetched from a CLAMPED frame index (a redundant reload
// at the very end instead of a conditional assignment, which costs a register copy per staged value at every merge).
// Cross-stem ratio mask folded into the inverse transform's prologue (srt_config.ratio_mask): instead of a separate read-modify-write pass over every
// stem's masks (srt_ratio_mask_kernel: 2 x 0.5 GB at the bench shape), the workgroup that applies stem s's mask to a bin reads the OTHER stems' mask values of that bin
// too (its S-1 neighbours on the XCD read the same rows at the same time: L2 hits) and normalises in registers.  Same arithmetic in the same order as
// srt_ratio_mask_kernel (squares summed over stems ascending, (m_s^2 + eps/S) / (sum + eps)), so srtSeparate with ratio_mask equals srtRatioMask + srtIstft bit for bit.
// own: the stem's raw mask value (already loaded); row0: mask row of stem 0 for this (tile, channel, frame); sstride: floats between stems.
#pragma clang fp contract(off)     // (HIP's __fmul_rn / __fadd_rn are plain operators: only this keeps the compiler from fusing square and sum)
__device__ __forceinline__ float srt_ratio_of(float own, const float* __restrict__ row0, size_t sstride, int nstems, int stem, int k)
{
    // (plain operators under contract(off): the square is rounded before it is added, exactly as in srt_ratio_mask_kernel.  HIP's __fmul_rn / __fadd_rn would
    // not do: they are inline `x * y` / `x + y` from a header compiled with contraction on, and fuse after inlining.)
    float sum = 0.0f;
    const float mine = own * own;
    for (int s = 0; s < nstems; ++s) {
        const float v = row0[(size_t)s * sstride + k];
        const float sq = v * v;
        sum = sum + (s == stem ? mine : sq);
    }
    const float eps = 1e-10f, e1 = eps / (float)nstems;
    return (mine + e1) / (sum + eps);
}

#pragma clang fp contract(fast)

template <bool RATIO = false>
__global__ void __launch_bounds__(256, 2) srt_istft_ola_kernel(const SrtIstftParams p, int G)
{
    // 1-D launch in XCD order, stem fastest: the nstems workgroups that walk the SAME run of frames sit next to each other on
    // one XCD, so the spectrum rows the first of them pulls from HBM are L2 hits for the others (each stem re-read them before).
    const int pos = srt_xcd_order(gridDim.x), stem = pos % p.nstems, run = pos / p.nstems;
    __shared__ cf s_mem[2 * FFT_SMEM_F2 + FFT_TWB_F2];   // two staging / exchange buffers + the pass-2 twiddles (one LDS object)
    cf* sx = s_mem;                                      // this frame's staging buffer (and its exchange 2)
    cf* sy = s_mem + FFT_SMEM_F2;                        // this frame's exchange 1; the roles swap every frame
    cf* s_twb = s_mem + 2 * FFT_SMEM_F2;
    const int tid = threadIdx.x;
    cf twa[15];
    fft_load_twiddles_pp(twa, s_twb, p.tab.twiddle, tid);
    const size_t tf = (size_t)p.T * p.F;
    const int nseg = p.frames + 3;
    const int s0 = run * G, s1 = min(s0 + G, nseg);
    const float oob = p.oob[stem];                                           // bins >= F: "unaffectedWeight" (main.c:486-493)
    const cf* spec = reinterpret_cast<const cf*>(p.spec);
    float* oL = p.out + (size_t)(stem * 2 + 0) * p.out_len;
    float* oR = p.out + (size_t)(stem * 2 + 1) * p.out_len;

    cf acc[4][4];                                       // [segment slot][j] = (R, L) pairs: one packed fma per output sample pair
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[h][j] = f2(0.0f, 0.0f);

    // Software pipeline: the spectrum row and the mask rows of the NEXT frame are in flight while the current frame is
    // transformed, so the global-load latency is not exposed once per FFT.
    cf sl[9], sr[9];
    float gl[9], gr[9];
    // fetch only ISSUES loads (no value depends on them until the next iteration): without masks the two mask rows
    // are read from a harmless table instead of branching on the pointer
    const bool has_mask = p.masks != nullptr;
    auto fetch = [&](int f) {                                               // 0 <= f < p.frames
        const int tile = f / p.T, t = f % p.T;
        const cf* specL = spec + (size_t)f * SRT_SPEC_LD;
        const cf* specR = specL + p.spec_ch_stride;
        const float* mL = has_mask ? p.masks + ((size_t)(stem * p.ntiles + tile) * 2) * tf + (size_t)t * p.F : p.tab.postWin;
        const float* mR = has_mask ? mL + tf : p.tab.postWin;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int k = min(tid + 256 * j, 2048), km = min(k, p.F - 1);
            sl[j] = specL[k]; sr[j] = specR[k];
            gl[j] = mL[km]; gr[j] = mR[km];
        }
    };
    float pw[16];                                       // this thread's 16 synthesis-window taps are the same for every frame
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) pw[k2] = p.tab.postWin[tid + 256 * k2];
    const int f0 = max(s0 - 3, 0);                      // frames before 0 do not exist: the window simply starts empty
    if (f0 < p.frames) fetch(f0);
    // Segment f is complete once frame f has been added, but it is WRITTEN in iteration f + 1, right before that iteration requests the rows of
    // frame f + 2: vmcnt counts loads and stores in one queue, so the wait for a frame's rows also waits for every store issued before them -
    // stores issued just ahead of the wait (the end of the previous iteration, where this code stood in rounds 1-2) cost their full
    // acknowledgement time (~1-2 us) once per frame; issued ahead of the NEXT prefetch they have a whole FFT to complete.
    auto emit_and_slide = [&](int seg) {
        if (seg >= s0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { oL[(size_t)seg * SRT_HOP + tid + 256 * j] = acc[0][j].y; oR[(size_t)seg * SRT_HOP + tid + 256 * j] = acc[0][j].x; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                   // slide the window
            acc[0][j] = acc[1][j]; acc[1][j] = acc[2][j]; acc[2][j] = acc[3][j]; acc[3][j] = f2(0.0f, 0.0f);
        }
    };
    for (int f = f0; f < s1; ++f) {
        if (f < p.frames) {                             // workgroup-uniform; false only for the last three segments of the stream
            // staging into sx: its last readers (exchange 1 of the previous frame, when it was `sy`) all passed that frame's
            // second barrier; the first barrier below also orders the twiddle table written before the loop
            if constexpr (RATIO) {
                if (has_mask && p.ratio) {               // normalise this frame's mask values across the stems (they arrived with the spectrum rows)
                    const int tile = f / p.T, t = f % p.T;
                    const float* r0 = p.masks + ((size_t)tile * 2) * tf + (size_t)t * p.F;
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const int km = min(min(tid + 256 * j, 2048), p.F - 1);
                        gl[j] = srt_ratio_of(gl[j], r0, (size_t)p.ntiles * 2 * tf, p.nstems, stem, km);
                        gr[j] = srt_ratio_of(gr[j], r0 + tf, (size_t)p.ntiles * 2 * tf, p.nstems, stem, km);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int k = tid + 256 * j;
                if (k <= 2048) {
                    const float wl = k < p.F ? (has_mask ? gl[j] : 1.0f) : oob, wr = k < p.F ? (has_mask ? gr[j] : 1.0f) : oob;
                    const cf A = sl[j] * wl, B = sr[j] * wr;                      // masked (re, im) of L and R
                    // G = F'_L + i F'_R, F' = re - i im, Hermitian-extended; stored swapped (im,re): inverse-by-forward trick
                    if (k == 0) sx[0] = f2(B.x, A.x);                             // a[0] = re[0]           (stftFix.c:556-557)
                    else if (k == 2048) sx[2048] = f2(B.x - B.y, A.x - A.y);     // rev[2048]: re - im wins (stftFix.c:563-566)
                    else {
                        sx[k] = sub_mi(B, A);                                     // (reR - imL, imR + reL)
                        sx[4096 - k] = herm_hi(B, A);                             // (reR + imL, reL - imR)
                    }
                }
            }
            __syncthreads();
            cf v[16];
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) v[n2] = sx[tid + 256 * n2];
            __builtin_amdgcn_s_waitcnt(0x0F70);          // all of this frame's rows have arrived (also the ones only lane 0 uses): nothing below waits on the stores
            if (f > f0) emit_and_slide(f - 1);
            fetch(min(f + 1, p.frames - 1));            // the staging registers are free again: next frame's rows fly under this FFT
            // exchange 1 in sy (free: its previous contents, the previous frame's exchange 2, were read before the barrier above),
            // exchange 2 back in sx (every thread has read its staged values before the transform's first barrier)
            fft4096_pp(v, sy, sx, twa, s_twb, tid);
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                acc[k2 >> 2][k2 & 3] = __builtin_elementwise_fma(v[FFT16_AT(k2)], f2(pw[k2], pw[k2]), acc[k2 >> 2][k2 & 3]);   // swapped back: L = .y, R = .x
            }
            { cf* t = sx; sx = sy; sy = t; }            // next frame stages where this frame's exchange 1 was (read before the second barrier)
        } else if (f > f0) emit_and_slide(f - 1);
    }
    if (s1 > f0) emit_and_slide(s1 - 1);                // the last segment of the run
}


// ------------------------------------------------------------------------------------------- three workgroups per CU
// Through round 4 both transforms held 2 workgroups per CU (250 / 190 VGPRs, 71.5 KB of LDS each - srt_istft_ola_kernel above is that form, kept for
// F > 1024): with one transform per workgroup in flight the VALU, the LDS pipe and HBM take turns instead of overlapping (r04 counters: active 0.36,
// wait 0.21 of the wave cycles, 0.33 / 0.53 of the HBM roofline).  The forms below fit THREE workgroups per CU (<= 168 VGPRs, <= 53 KB LDS; measured on
// the 64-tile batch: iSTFT 0.587 -> 0.480 ms, STFT 0.212 -> 0.204 ms):
//   * pass-1 twiddles from one register pair (fft_twiddle_powers) and, in the inverse, the synthesis window rebuilt from two registers per frame;
//   * ONE exchange buffer (4 barriers per frame instead of 3: exchange 2 waits until every thread has read exchange 1);
//   * no natural-order staging / result buffer.  What both kernels really need from LDS besides the two FFT exchanges is the HERMITIAN MIRROR:
//     thread t owns indices t + 256 j; for j < 8 these are bins k < 2048, and the partner index 4096 - k = (256 - t) + 256 (15 - j) belongs to thread
//     256 - t, slot 15 - j.  So only the upper half travels through LDS, 2049 values in `mir`: slot (j', t') at j' * 256 + t', read back at
//     (15 - n2) * 256 + 256 - t (thread 0 is its own partner; its "thread 256" column is the extra entry 2048, see each kernel) - 16 KB written and
//     read per frame instead of 32 + 32 KB, reversed-contiguous (conflict-free) on both sides.
// LDS: 4352 + 2049 + 240 complex = 53 128 bytes.
#define FFT_MIR_F2 2049
// pass-1 twiddles W^(tid*k0), k0 = 1..15, from ONE register pair w1 = W^tid: powers by binary splitting (every factor is at most three products
// deep, so the twiddles carry <= ~4 roundings = 2.4e-7 relative - inside the transform's own rounding) instead of 15 resident pairs (30 VGPRs).
__device__ __forceinline__ void fft_twiddle_powers(cf (&v)[16], cf w1)
{
    const cf w2 = cmul(w1, w1), w4 = cmul(w2, w2), w8 = cmul(w4, w4);
    const cf w3 = cmul(w1, w2), w5 = cmul(w1, w4), w6 = cmul(w2, w4), w7 = cmul(w3, w4);
    const cf lowp[8] = { f2(1.f, 0.f), w1, w2, w3, w4, w5, w6, w7 };
#pragma unroll
    for (int k0 = 1; k0 < 8; ++k0) v[FFT16_AT(k0)] = cmul(v[FFT16_AT(k0)], lowp[k0]);
    v[FFT16_AT(8)] = cmul(v[FFT16_AT(8)], w8);
#pragma unroll
    for (int k0 = 9; k0 < 16; ++k0) v[FFT16_AT(k0)] = cmul(cmul(v[FFT16_AT(k0)], w8), lowp[k0 - 8]);
}
__device__ __forceinline__ void fft4096_1b(cf (&v)[16], cf* x, const cf w1, const cf* twb, int tid)
{
    // in : v[n2] = x[tid + 256*n2]; `x` free (every thread is past its last read of the previous frame's exchange 2: the caller's barrier)
    // out: v[FFT16_AT(k2)] = X[tid + 256*k2]; `x` holds exchange 2 until the caller's next barrier
    fft16(v);
    fft_twiddle_powers(v, w1);
#pragma unroll
    for (int k0 = 0; k0 < 16; ++k0) x[k0 * FFT_EX1_LD + tid] = v[FFT16_AT(k0)];
    __syncthreads();
    const int lo = tid & 15, hi = tid >> 4;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = x[hi * FFT_EX1_LD + n1 * 16 + lo];
    fft16(v);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[FFT16_AT(k1)] = cmul(v[FFT16_AT(k1)], twb[(k1 - 1) * 16 + lo]);
    __syncthreads();                                     // every thread has read exchange 1
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) x[lo * FFT_EX2_LD + k1 * 16 + hi] = v[FFT16_AT(k1)];
    __syncthreads();
#pragma unroll
    for (int n0 = 0; n0 < 16; ++n0) v[n0] = x[n0 * FFT_EX2_LD + tid];
    fft16(v);
}

__global__ void __launch_bounds__(256, 3) srt_stft_kernel(const SrtStftParams p, int fpb)
{
    __shared__ cf s_mem[FFT_SMEM_F2 + FFT_MIR_F2 + FFT_TWB_F2];
    cf* sx = s_mem;
    cf* mir = s_mem + FFT_SMEM_F2;
    cf* s_twb = mir + FFT_MIR_F2;
    const int tid = threadIdx.x, blk = blockIdx.x;
    const cf w1 = reinterpret_cast<const cf*>(p.tab.twiddle)[tid];
    if (tid < FFT_TWB_F2) s_twb[tid] = reinterpret_cast<const cf*>(p.tab.twiddle)[(16 * (tid & 15) * (tid / 16 + 1)) & 4095];
    float nxtL[16], nxtR[16];
    float aw[16];
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) aw[n2] = p.tab.preWin[tid + 256 * n2];
    auto fetch = [&](int f) {
        const size_t pos = (size_t)f * SRT_HOP;
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
            const int n = tid + 256 * n2;
            const size_t q = pos + n < p.nsamples ? pos + n : 0;
            nxtL[n2] = p.L[q]; nxtR[n2] = p.R[q];
        }
    };
    // Consecutive frames share three of their four hops, and thread tid's samples tid + 256 n2 of frame f + 1 are its samples n2 + 4 of frame f: the staging
    // registers slide by one hop and only the NEW hop is loaded (4 loads per channel and frame instead of 16).  Round 5's counters had this kernel fetching 0.45 GB
    // for 0.13 GB of PCM: most of the 4x frame overlap came from the fabric again.
    auto fetch_slide = [&](int f) {                      // the registers hold frame f - 1
        const size_t pos = (size_t)f * SRT_HOP;
#pragma unroll
        for (int n2 = 0; n2 < 12; ++n2) { nxtL[n2] = nxtL[n2 + 4]; nxtR[n2] = nxtR[n2 + 4]; }
#pragma unroll
        for (int n2 = 12; n2 < 16; ++n2) {
            const int n = tid + 256 * n2;
            const size_t q = pos + n < p.nsamples ? pos + n : 0;
            nxtL[n2] = p.L[q]; nxtR[n2] = p.R[q];
        }
    };
    const int flast = max(p.frames_computed, 1) - 1;
    fetch(min((int)(blk * fpb), flast));
    __syncthreads();                                     // the pass-2 twiddles are in LDS
    for (int fi = 0; fi < fpb; ++fi) {
        const int f = blk * fpb + fi;
        if (f >= p.rows_total) break;
        const int tile = f / p.T, t = f % p.T;
        float* magL = p.mag ? p.mag + ((size_t)(tile * 2 + 0) * p.T + t) * p.F : nullptr;
        float* magR = p.mag ? p.mag + ((size_t)(tile * 2 + 1) * p.T + t) * p.F : nullptr;
        cf* specL = reinterpret_cast<cf*>(p.spec) + (size_t)f * SRT_SPEC_LD;
        cf* specR = specL + p.spec_ch_stride;
        if (f >= p.frames_computed) {                    // rows the reference leaves calloc'ed (stftFix.c:368-371)
            for (int k = tid; k < SRT_SPEC_LD; k += 256) { specL[k] = f2(0.f, 0.f); specR[k] = f2(0.f, 0.f); }
            if (p.mag) for (int k = tid; k < p.F; k += 256) { magL[k] = 0.f; magR[k] = 0.f; }
            continue;
        }
        cf v[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
            const bool ok = (size_t)f * SRT_HOP + tid + 256 * n2 < p.nsamples;   // tail frame is zero padded (stftFix.c:460-472)
            v[n2] = f2(nxtL[n2], nxtR[n2]) * (ok ? aw[n2] : 0.f);
        }
        if (f + 1 <= flast) fetch_slide(f + 1);          // the next frame's new hop flies under this transform (past the last frame nothing is needed)
        fft4096_1b(v, sx, w1, s_twb, tid);               // v[FFT16_AT(k2)] = Z[tid + 256 k2]
        // upper half into the mirror buffer: slot (k2 - 8, tid); entry 2048 = Z[0] (partner of bin 0, read by thread 0 through its "column 256")
#pragma unroll
        for (int k2 = 8; k2 < 16; ++k2) mir[(k2 - 8) * 256 + tid] = v[FFT16_AT(k2)];
        if (tid == 0) mir[2048] = v[FFT16_AT(0)];
        const cf z2048 = v[FFT16_AT(8)];                 // thread 0: Z[2048], its own partner
        __syncthreads();                                 // also: every thread is done with exchange 2 (sx is free for the next frame)
        // separate the two real spectra; stored spectrum = conj(F) (the reference's re/im convention)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = tid + 256 * j;
            const cf zk = v[FFT16_AT(j)], zm = mir[(7 - j) * 256 + 256 - tid];
            const cf sl = split_l(zk, zm), sr = split_r(zk, zm);
            if (p.mag && k < p.F) {
                magL[k] = hypotf(sl.x, sl.y) * 4096.0f;              // main.c:468-469
                magR[k] = hypotf(sr.x, sr.y) * 4096.0f;
            }
            specL[k] = sl;
            specR[k] = sr;
        }
        if (tid < SRT_SPEC_LD - 2048) {                  // bin 2048 and the zero padding of the row
            const cf sl = tid == 0 ? split_l(z2048, z2048) : f2(0.f, 0.f), sr = tid == 0 ? split_r(z2048, z2048) : f2(0.f, 0.f);
            specL[2048 + tid] = sl;
            specR[2048 + tid] = sr;
            if (p.mag && tid == 0 && 2048 < p.F) { magL[2048] = hypotf(sl.x, sl.y) * 4096.0f; magR[2048] = hypotf(sr.x, sr.y) * 4096.0f; }
        }
    }
}

// NM: mask rows cover bins < F <= 256 NM, so only the first NM of a thread's eight bins can carry a mask value (F = 1024: 4 prefetch registers per channel instead of 8)
// M16: the masks are halves (the engine's own mask buffer in the fp16 mode, written by srt_head_rows_kernel<.., true>; never with RATIO)
template <int NM, bool RATIO = false, bool M16 = false>
__global__ void __launch_bounds__(256, RATIO ? 2 : 3) srt_istft_ola3_kernel(const SrtIstftParams p, int G)
{
    const int pos = srt_xcd_order(gridDim.x), stem = pos % p.nstems, run = pos / p.nstems;       // stem fastest: the stems of a run share the spectrum rows in L2
    __shared__ cf s_mem[FFT_SMEM_F2 + FFT_MIR_F2 + FFT_TWB_F2];
    cf* sx = s_mem;
    cf* mir = s_mem + FFT_SMEM_F2;
    cf* s_twb = mir + FFT_MIR_F2;
    const int tid = threadIdx.x;
    const cf w1 = reinterpret_cast<const cf*>(p.tab.twiddle)[tid];
    if (tid < FFT_TWB_F2) s_twb[tid] = reinterpret_cast<const cf*>(p.tab.twiddle)[(16 * (tid & 15) * (tid / 16 + 1)) & 4095];
    const size_t tf = (size_t)p.T * p.F;
    const int nseg = p.frames + 3;
    const int s0 = run * G, s1 = min(s0 + G, nseg);
    const float oob = p.oob[stem];                                           // bins >= F: "unaffectedWeight" (main.c:486-493)
    const cf* spec = reinterpret_cast<const cf*>(p.spec);
    float* oL = p.out + (size_t)(stem * 2 + 0) * p.out_len;
    float* oR = p.out + (size_t)(stem * 2 + 1) * p.out_len;
    cf acc[4][4];                                       // [segment slot][j] = (R, L) pairs
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[h][j] = f2(0.0f, 0.0f);
    cf sl[8], sr[8];
    float gl[NM], gr[NM];
    cf sl8, sr8;                                        // bin 2048 (only thread 0 uses it; the address is uniform)
    const bool has_mask = p.masks != nullptr;
    auto fetch = [&](int f) {                                               // 0 <= f < p.frames
        const int tile = f / p.T, t = f % p.T;
        const cf* specL = spec + (size_t)f * SRT_SPEC_LD;
        const cf* specR = specL + p.spec_ch_stride;
        const size_t mo = ((size_t)(stem * p.ntiles + tile) * 2) * tf + (size_t)t * p.F;
        const float* mL = has_mask ? p.masks + mo : p.tab.postWin;
        const float* mR = has_mask ? mL + tf : p.tab.postWin;
        const _Float16* hL = reinterpret_cast<const _Float16*>(p.masks) + mo;                   // (M16: the launcher checks that masks are given)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = tid + 256 * j;
            sl[j] = specL[k]; sr[j] = specR[k];
            if (j < NM) {
                const int km = min(k, p.F - 1);
                if constexpr (M16) { gl[j] = (float)hL[km]; gr[j] = (float)hL[tf + km]; }
                else { gl[j] = mL[km]; gr[j] = mR[km]; }
            }
        }
        sl8 = specL[2048]; sr8 = specR[2048];
    };
    // the 16 synthesis-window taps of this thread (samples tid + 256 k2) are rebuilt per frame from two registers: cos / sin of th = 2 pi (tid + 1/2) / 4096, over 3.
    // (A deviation from the table the F > 1024 kernel reads - postWin, which reproduces InitSTFT's rounding: 1/3 - cos/3 in fp32 is off by up to ~3e-8 absolute,
    // which is a large RELATIVE error only on the ~1e-7 taps at the window's edges.  tests/test_gpu_parity.py::test_istft_three_per_cu_kernel_against_the_table_kernel
    // bounds the two kernels against each other on identical input at 5e-7 of the output's peak (measured 3.1e-7).)
    float ws3, wc3;
    sincospif((tid + 0.5f) * (1.0f / 2048.0f), &ws3, &wc3);
    ws3 *= 1.0f / 3.0f; wc3 *= 1.0f / 3.0f;
    constexpr float COS_PI8[16] = { 1.0f, 0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f, 0.0f, -0.38268343236508977173f, -0.70710678118654752440f, -0.92387953251128675613f,
                                    -1.0f, -0.92387953251128675613f, -0.70710678118654752440f, -0.38268343236508977173f, 0.0f, 0.38268343236508977173f, 0.70710678118654752440f, 0.92387953251128675613f };
    constexpr float SIN_PI8[16] = { 0.0f, 0.38268343236508977173f, 0.70710678118654752440f, 0.92387953251128675613f, 1.0f, 0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f,
                                    0.0f, -0.38268343236508977173f, -0.70710678118654752440f, -0.92387953251128675613f, -1.0f, -0.92387953251128675613f, -0.70710678118654752440f, -0.38268343236508977173f };
    const int f0 = max(s0 - 3, 0);
    if (f0 < p.frames) fetch(f0);
    __syncthreads();                                     // the pass-2 twiddles are in LDS
    auto emit_and_slide = [&](int seg) {
        if (seg >= s0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { oL[(size_t)seg * SRT_HOP + tid + 256 * j] = acc[0][j].y; oR[(size_t)seg * SRT_HOP + tid + 256 * j] = acc[0][j].x; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][j] = acc[1][j]; acc[1][j] = acc[2][j]; acc[2][j] = acc[3][j]; acc[3][j] = f2(0.0f, 0.0f);
        }
    };
    for (int f = f0; f < s1; ++f) {
        if (f < p.frames) {
            // G = F'_L + i F'_R, F' = re - i im, Hermitian-extended; kept swapped (im, re): inverse-by-forward trick.  Bins k = tid + 256 j < 2048 are this
            // thread's own transform inputs n2 = j; their partners 4096 - k go through `mir` to thread 256 - tid (slot 15 - j).
            cf v[16];
            if constexpr (RATIO) {
                if (has_mask && p.ratio) {               // (see srt_ratio_of; the extra loads are why this instantiation is built for two workgroups per CU)
                    const int tile = f / p.T, t = f % p.T;
                    const float* r0 = p.masks + ((size_t)tile * 2) * tf + (size_t)t * p.F;
#pragma unroll
                    for (int j = 0; j < NM; ++j) {
                        const int km = min(tid + 256 * j, p.F - 1);
                        gl[j] = srt_ratio_of(gl[j], r0, (size_t)p.ntiles * 2 * tf, p.nstems, stem, km);
                        gr[j] = srt_ratio_of(gr[j], r0 + tf, (size_t)p.ntiles * 2 * tf, p.nstems, stem, km);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = tid + 256 * j;
                float wl = oob, wr = oob;
                if (j < NM) { wl = k < p.F ? (has_mask ? gl[j] : 1.0f) : oob; wr = k < p.F ? (has_mask ? gr[j] : 1.0f) : oob; }
                const cf A = sl[j] * wl, B = sr[j] * wr;                          // masked (re, im) of L and R
                v[j] = sub_mi(B, A);                                              // (reR - imL, imR + reL)
                mir[j * 256 + tid] = herm_hi(B, A);                               // (reR + imL, reL - imR)   (slot (0, 0) is never read)
                if (j == 0 && tid == 0) v[0] = f2(B.x, A.x);                      // a[0] = re[0]           (stftFix.c:556-557)
            }
            if (tid == 0) {                                                       // rev[2048]: re - im wins (stftFix.c:563-566); F <= 2048, so bin 2048 is always out of band
                const cf A = sl8 * oob, B = sr8 * oob;
                mir[2048] = f2(B.x - B.y, A.x - A.y);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);          // all of this frame's rows have arrived: nothing below waits on the stores
            if (f > f0) emit_and_slide(f - 1);
            fetch(min(f + 1, p.frames - 1));             // next frame's rows fly under this transform
            __syncthreads();                             // the mirror values are visible; every thread is past the previous frame's exchange 2
#pragma unroll
            for (int n2 = 8; n2 < 16; ++n2) v[n2] = mir[(15 - n2) * 256 + 256 - tid];
            fft4096_1b(v, sx, w1, s_twb, tid);
            asm volatile("" : "+v"(wc3), "+v"(ws3));     // keeps the 16 taps from being hoisted out of the frame loop (they would come back as 32 resident VGPRs)
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                // synthesis window tap of sample tid + 256 k2: (2/3) hann(i + 1/2) = 1/3 - (cos(th) cos(k2 pi/8) - sin(th) sin(k2 pi/8)) / 3 (stftFix.c:329-341)
                const float pwk = fmaf(-wc3, COS_PI8[k2], fmaf(ws3, SIN_PI8[k2], 1.0f / 3.0f));
                acc[k2 >> 2][k2 & 3] = __builtin_elementwise_fma(v[FFT16_AT(k2)], f2(pwk, pwk), acc[k2 >> 2][k2 & 3]);   // swapped back: L = .y, R = .x
            }
        } else if (f > f0) emit_and_slide(f - 1);
    }
    if (s1 > f0) emit_and_slide(s1 - 1);
}

int srt_launch_stft(const SrtStftParams& p, hipStream_t s)
{
    // short signals (one tile, the real-time regime): fewer frames per workgroup so that the frames spread over the CUs
    int fpb = p.rows_total >= 4096 ? STFT_FPB : (p.rows_total >= 1024 ? 2 : 1);
    if (p.rows_total >= 4096) {                          // whole rounds of the 768 resident workgroups (3 per CU), at most ~24 frames each
        // (round 6, with the sliding staging registers a longer run costs nothing and saves first-frame fetches: cap 6 / 8 / 12 / 24 / 48 -> 0.169 / 0.168 / 0.170 / 0.162 / 0.163 ms)
        constexpr int cap = 24;
        const int slots = 768, rounds = (p.rows_total + slots * cap - 1) / (slots * cap);
        fpb = (p.rows_total + slots * rounds - 1) / (slots * rounds);
    }
    const int blocks = (p.rows_total + fpb - 1) / fpb;
    if (blocks <= 0) return 0;
    SRT_LAUNCH(srt_stft_kernel, dim3(blocks), dim3(256), 0, s, p, fpb);
    return srt_launch_status();
}

int srt_launch_istft(const SrtIstftParams& p, hipStream_t s)
{
    if (p.frames <= 0) return 0;
    const int nseg = p.frames + 3;
    // One launch for all stems (blockIdx.y = stem): ~4 workgroups per CU over the whole grid when the stream is long enough,
    // runs of at least 13 segments so the 3-frame warm-up stays below ~25 % (64-tile batch, 4 stems: G = 65, 4.6 %).
    // three-per-CU form: ONE round of the 768 resident workgroups (64-tile batch, 4 stems: G = 86, warm-up 3.5 %; two rounds measured the same, 1024 / 2304
    // workgroups 6 % / 1 % slower); the two-per-CU form keeps its two rounds of 512
    if (p.masks16 && (!p.masks || p.F > 1024 || (p.ratio && p.nstems > 1))) return -1;      // halves are read by the three-per-CU kernel only
    const int target = p.F > 1024 ? 1024 : 768;
    // (runs x stems must not exceed the resident workgroups: five stems at 768 / 5 = 153.6 runs would leave two workgroups for a second round)
    const int max_runs = target / p.nstems > 0 ? target / p.nstems : 1;
    int G = (nseg + max_runs - 1) / max_runs;
    const int gmin = (size_t)nseg * p.nstems >= 4096 ? 13 : 5;          // short signals: shorter runs (more warm-up, but all CUs busy)
    if (G < gmin) G = gmin;
    const int blocks = (nseg + G - 1) / G;
    // one stem per workgroup: 32 accumulator + 54 prefetch registers + the FFT fit in 256 VGPRs at 2 workgroups per CU
    // F > 1024: eight mask registers per channel do not fit the 168-VGPR budget of the three-per-CU form (22 dwords would spill): the two-per-CU kernel
    if (p.ratio && p.masks && p.nstems > 1) {            // cross-stem ratio mask applied in the prologue (srt_ratio_of)
        if (p.F > 1024) SRT_LAUNCH((srt_istft_ola_kernel<true>), dim3(blocks * p.nstems), dim3(256), 0, s, p, G);
        else SRT_LAUNCH((srt_istft_ola3_kernel<4, true>), dim3(blocks * p.nstems), dim3(256), 0, s, p, G);
    } else if (p.F > 1024) SRT_LAUNCH((srt_istft_ola_kernel<false>), dim3(blocks * p.nstems), dim3(256), 0, s, p, G);
    else if (p.masks16) SRT_LAUNCH((srt_istft_ola3_kernel<4, false, true>), dim3(blocks * p.nstems), dim3(256), 0, s, p, G);
    else SRT_LAUNCH((srt_istft_ola3_kernel<4>), dim3(blocks * p.nstems), dim3(256), 0, s, p, G);
    return srt_launch_status();
}

// ------------------------------------------------------------------------------------------- residual chain / ratio mask
// Complex-domain residual of the three-output CLI flow (main.c:845-866): the first network's masked spectrum is subtracted
// from the original, and the second network sees the magnitude of what is left.  Same two roundings as the reference
// (product, then difference: no fused multiply-add).  One workgroup = one spectrogram row of one channel.  HBM bound:
// 16.4 KB read + 16.4 KB written + 2 x 4 F bytes per row.
__global__ void __launch_bounds__(256) srt_residual_kernel(const SrtResidualParams p)
{
    const int row = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
    const int tile = row / p.T, t = row % p.T;
    const cf* src = reinterpret_cast<const cf*>(p.spec) + (size_t)ch * p.spec_ch_stride + (size_t)row * SRT_SPEC_LD;
    cf* dst = reinterpret_cast<cf*>(p.res) + (size_t)ch * p.spec_ch_stride + (size_t)row * SRT_SPEC_LD;
    const size_t mo = ((size_t)(tile * 2 + ch) * p.T + t) * p.F;
    const float* m = p.mask + mo;
    float* mag = p.mag ? p.mag + mo : nullptr;
    for (int k = tid; k < SRT_SPEC_LD; k += 256) {
        cf r = f2(0.f, 0.f);
        if (k <= 2048) {
            const cf v = src[k];
            const float g = k < p.F ? m[k] : p.oob;
            r.x = __fsub_rn(v.x, __fmul_rn(v.x, g));
            r.y = __fsub_rn(v.y, __fmul_rn(v.y, g));
            if (mag && k < p.F) mag[k] = hypotf(r.x, r.y) * 4096.0f;
        }
        dst[k] = r;
    }
}

int srt_launch_residual(const SrtResidualParams& p, hipStream_t s)
{
    if (p.rows <= 0) return 0;
    SRT_LAUNCH(srt_residual_kernel, dim3(p.rows, 2), dim3(256), 0, s, p);
    return srt_launch_status();
}

// out = a - b on samples [lo, hi) of two planes of nb floats (a: two separate planes of na valid samples, zero beyond)
__global__ void __launch_bounds__(256) srt_time_residual_kernel(const float* aL, const float* aR, size_t na, const float* b, size_t nb, float* out, size_t lo, size_t hi)
{
    const size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= hi) return;
    const float* a = blockIdx.y ? aR : aL;
    const size_t o = (size_t)blockIdx.y * nb + i;
    out[o] = (i < na ? a[i] : 0.0f) - b[o];
}

int srt_launch_time_residual(const float* aL, const float* aR, size_t na, const float* b, size_t nb, float* out, size_t lo, size_t hi, hipStream_t s)
{
    if (hi > nb) hi = nb;
    if (hi <= lo) return 0;
    SRT_LAUNCH(srt_time_residual_kernel, dim3((unsigned)((hi - lo + 255) / 256), 2), dim3(256), 0, s, aL, aR, na, b, nb, out, lo, hi);
    return srt_launch_status();
}

// Overlap-add across the chunks of a long stream: consecutive chunks share 3072 output samples (three hops of the last
// frames' tails, stftFix.c:570-575).  The previous chunk's tail is added to this chunk's head and this chunk's tail is kept.
__global__ void __launch_bounds__(256) srt_carry_kernel(float* out, size_t plane_len, size_t tail, float* carry, int first, int last)
{
    const int i = blockIdx.x * 256 + threadIdx.x, pl = blockIdx.y;         // i < 3072
    float* o = out + (size_t)pl * plane_len;
    float* c = carry + (size_t)pl * (SRT_FFT - SRT_HOP);
    if (!first) o[i] += c[i];
    if (!last) c[i] = o[tail + i];                                         // tail >= 3072 for every chunk but the last
}

int srt_launch_carry(float* out, size_t plane_len, int nplanes, size_t tail, float* carry, int first, int last, hipStream_t s)
{
    if (first && last) return 0;
    SRT_LAUNCH(srt_carry_kernel, dim3((SRT_FFT - SRT_HOP) / 256, nplanes), dim3(256), 0, s, out, plane_len, tail, carry, first, last);
    return srt_launch_status();
}

// Cross-stem ratio mask (what official Spleeter applies and the reference deliberately leaves out, README.MD:82-85):
// every stem's mask is squared and normalised by the sum over stems at the same (tile, channel, frame, bin).
#pragma clang fp contract(off)
__global__ void __launch_bounds__(256) srt_ratio_mask_kernel(float* masks, int nstems, size_t count)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= count) return;                                        // count is a multiple of 4 (F % 64 == 0)
    // every operation individually rounded (plain operators under contract(off), no fused multiply-add): srt_ratio_of, the form folded into the inverse transform's prologue,
    // does the same operations in the same order, and the two must agree bit for bit (tests/test_gpu_parity.py::test_ratio_mask)
    float m[SRT_MAX_STEMS][4];
    float sum[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int s = 0; s < SRT_MAX_STEMS; ++s) {
        if (s < nstems) {
            const float4 v = *reinterpret_cast<const float4*>(masks + (size_t)s * count + i);
            const float x[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int c = 0; c < 4; ++c) { m[s][c] = x[c] * x[c]; sum[c] = sum[c] + m[s][c]; }
        }
    }
    const float eps = 1e-10f, e1 = eps / (float)nstems;
#pragma unroll
    for (int s = 0; s < SRT_MAX_STEMS; ++s) {
        if (s < nstems) {
            float4 v;
            v.x = (m[s][0] + e1) / (sum[0] + eps); v.y = (m[s][1] + e1) / (sum[1] + eps);
            v.z = (m[s][2] + e1) / (sum[2] + eps); v.w = (m[s][3] + e1) / (sum[3] + eps);
            *reinterpret_cast<float4*>(masks + (size_t)s * count + i) = v;
        }
    }
}

#pragma clang fp contract(fast)
int srt_launch_ratio_mask(float* masks, int nstems, size_t count, hipStream_t s)
{
    if (!count || nstems < 1) return 0;
    SRT_LAUNCH(srt_ratio_mask_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, s, masks, nstems, count);
    return srt_launch_status();
}

// ------------------------------------------------------------------------------------------- streaming hop
// Inverse of the DELAYED frame for one stem (blockIdx.x = stem): masked spectrum -> time frame -> synthesis window on the
// last 2048 samples -> 50 % overlap-add with the kept half -> interleaved-by-8 output segment (Spleeter4Stems.c:64-101,272-320).
__global__ void __launch_bounds__(256) srt_stream_inverse_kernel(const SrtStreamHop p)
{
    __shared__ cf s_tw[FFT_TW_F2];
    __shared__ cf s_x[FFT_SMEM_F2];
    const int tid = threadIdx.x, st = blockIdx.x;
    fft_load_twiddles(s_tw, p.twiddle, tid);
    const cf* specL = reinterpret_cast<const cf*>(p.specRow);
    const cf* specR = reinterpret_cast<const cf*>(p.specRow) + p.specChStride;
    const float* mL = p.maskRow + st * p.maskStemStride;
    const float* mR = mL + p.maskChStride;
    const float oob = st == 1 ? 0.0f : 0.25f;                      // Spleeter4Stems.c:73,281
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int k = tid + 256 * j;
        if (k <= 2048) {
            const cf sl = specL[k], sr = specR[k];
            float gl = oob, gr = oob;
            if (k < p.F) { gl = mL[k]; gr = mR[k]; }
            const float reL = sl.x * gl, imL = sl.y * gl, reR = sr.x * gr, imR = sr.y * gr;
            if (k == 0) s_x[0] = f2(reR, reL);
            else if (k == 2048) s_x[2048] = f2(reR - imR, reL - imL);
            else {
                s_x[k] = f2(reR - imL, reL + imR);
                s_x[4096 - k] = f2(reR + imL, reL - imR);
            }
        }
    }
    __syncthreads();
    cf v[16];
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = s_x[tid + 256 * n2];
    __syncthreads();
    fft4096(v, s_x, s_tw, tid);
    float* ovL = p.overlap + (size_t)(2 * st) * 1024;
    float* ovR = ovL + 1024;
    // thread holds samples tid + 256*k2; only the last 2048 are synthesised.  k2 = 8..11 -> output, 12..15 -> kept half.
#pragma unroll
    for (int k2 = 8; k2 < 12; ++k2) {
        const int i = tid + 256 * (k2 - 8);                                   // 0..1023
        const float w0 = p.synthesisWnd[i], w1 = p.synthesisWnd[i + 1024];
        const cf y0 = v[FFT16_AT(k2)], y1 = v[FFT16_AT(k2 + 4)];
        p.out[(size_t)i * 8 + 2 * st + 0] = ovL[i] + y0.y * w0;              // mOverlapStage2dash + timeDomainOut (:313-315)
        p.out[(size_t)i * 8 + 2 * st + 1] = ovR[i] + y0.x * w0;
        ovL[i] = y1.y * w1;                                                   // :317-318
        ovR[i] = y1.x * w1;
    }
}

// Forward transform of the CURRENT 4096 ring samples with the asymmetric analysis window; spectrum + magnitude rows
// of the collecting batch (Spleeter4Stems.c:261-267,322-349).
__global__ void __launch_bounds__(256) srt_stream_forward_kernel(const SrtStreamHop p)
{
    __shared__ cf s_tw[FFT_TW_F2];
    __shared__ cf s_x[FFT_SMEM_F2];
    const int tid = threadIdx.x;
    fft_load_twiddles(s_tw, p.twiddle, tid);
    __syncthreads();
    cf v[16];
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) {
        const int n = tid + 256 * n2, k = (n + p.inPos) & 4095;
        const float w = p.analysisWnd[n];
        v[n2] = f2(p.ring[k] * w, p.ring[4096 + k] * w);
    }
    fft4096(v, s_x, s_tw, tid);
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) s_x[tid + 256 * k2] = v[FFT16_AT(k2)];
    __syncthreads();
    cf* specL = reinterpret_cast<cf*>(p.specRow);
    cf* specR = reinterpret_cast<cf*>(p.specRow) + p.specChStride;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int k = tid + 256 * j;
        if (k <= 2048) {
            const cf zk = s_x[k], zm = s_x[(4096 - k) & 4095];
            const cf sl = f2(zk.x + zm.x, zm.y - zk.y);
            const cf sr = f2(zk.y + zm.y, zk.x - zm.x);
            specL[k] = sl; specR[k] = sr;
            if (k < p.F) {
                p.magRow[k] = hypotf(sl.x, sl.y) * 4096.0f;
                p.magRow[p.magChStride + k] = hypotf(sr.x, sr.y) * 4096.0f;
            }
        }
    }
}

int srt_launch_stream_hop(const SrtStreamHop& p, hipStream_t s)
{
    SRT_LAUNCH(srt_stream_inverse_kernel, dim3(4), dim3(256), 0, s, p);     // reads the delayed row ...
    if (hipGetLastError() != hipSuccess) return -1;
    SRT_LAUNCH(srt_stream_forward_kernel, dim3(1), dim3(256), 0, s, p);     // ... before the current frame overwrites it
    return srt_launch_status();
}
