// srt_internal.h — private declarations shared by the HIP translation units of libspleeterrt_amd.so.
// Nothing here is part of the C ABI (see include/*.h for that).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define SRT_MAX_STEMS 8
#define SRT_FFT 4096
#define SRT_HOP 1024
#define SRT_HALF 2049
#define SRT_SPEC_LD 2052          // float2 elements per spectrum row (2049 padded to a 16-byte multiple)
#define SRT_COEFF_FLOATS 9822725u // sizeof(spleeterCoeff)/4, Executable/spleeter.h:5-31
#define SRT_ENC_MAX_CIN 256       // widest encoder input (down6)
#define SRT_COEFF_STRIDE 9822784u // per-stem stride inside the engine's single weight allocation (64-float aligned)

// error text behind srtLastError() (thread-local), settable from every translation unit of the library; returns `code`
int srt_set_error(int code, const char* fmt, const char* detail);

// Set-up paths (engine creation, weight upload, graph pre-warm, the drop-in Init functions) are serialised process-wide: they use the legacy
// null stream (synchronous copies, memsets, hipMemcpyToSymbol) and stream capture, and HIP refuses a null-stream operation issued by one host
// thread while another thread's stream is capturing ("operation would make the legacy stream depend on a capturing blocking stream") - two
// plugin instances initialised from two threads at once otherwise come up muted now and then.  Steady-state calls never take this lock: they
// use explicit streams only and replay graphs that already exist.
struct SrtSetupLock { SrtSetupLock(); ~SrtSetupLock(); };

enum { SRT_ACT_LEAKY = 0, SRT_ACT_RELU = 1, SRT_ACT_ELU = 2 };

// Every kernel launch of the library goes through SRT_LAUNCH: besides launching, it notes WHICH kernel (host stub pointer + the
// launch expression's text) the calling thread launched first since srt_kernel_note_reset().  The engine's per-launch timers keep
// that with each entry, so srtGetTimingKernels() reports the kernel that actually ran a layer (the dispatch depends on batch
// size and geometry), not a table kept by hand.
void srt_kernel_note(const void* fn, const char* text);
void srt_kernel_note_reset();
// status of the launch just issued on this thread: 0, or -1 with the HIP error kept for the message srt_set_error builds next
void srt_note_hip_error(hipError_t e);
static inline int srt_launch_status() { const hipError_t e = hipGetLastError(); if (e != hipSuccess) srt_note_hip_error(e); return e == hipSuccess ? 0 : -1; }
#define SRT_LAUNCH(kernel, ...) do { srt_kernel_note((const void*)(kernel), #kernel); hipLaunchKernelGGL(kernel, __VA_ARGS__); } while (0)

// One convolution layer evaluated for nstems x ntiles independent instances.
// instance pointer = base + stem * *_stem + tile * *_tile   (strides in floats)
struct SrtConvParams {
    int Cin, Cout;        // channels
    int H, W;             // INPUT spatial size per instance (encoder output = H/2 x W/2, decoder output = 2H x 2W)
    int CA;               // channels [0,CA) come from srcA, [CA,Cin) from srcB (decoder skip concat by pointer)
    int ntiles, nstems;
    const float* srcA; size_t srcA_stem, srcA_tile;
    const float* srcB; size_t srcB_stem, srcB_tile;
    // Encoder layers 2..6 read the RAW (conv + bias) tensor of the previous layer and apply its batch-norm + activation
    // while staging: x = act(inScale[c] * raw + inShift[c]) (spleeter.c:188), per input channel c and per stem
    // (pointer = base + stem * coeff_stem + c).  nullptr: the source is used as it is (down1 reads magnitudes).
    const float* inScale; const float* inShift;
    // per-stem weights: pointer = base + stem * stride (all stems live in one allocation, so no pointer tables)
    const float* wraw;    // reference layout: encoder OIHW [Cout][Cin][5][5], decoder [Cin][Cout][5][5]
    const float* bias; const float* bnShift; const float* bnScale;   // bnScale == nullptr => no batch-norm (down6)
    size_t coeff_stem;    // stride (floats) between stems for wraw/bias/bnShift/bnScale
    const float* wpack;   // GEMM layout [Cin][25][CP], zero padded to CP output channels
    size_t wpack_stem;
    int CP;
    // stacked-M weight layouts (srt_nn2.hip): fill otherwise half-empty 32-row MFMA tiles of the Cout = 16 layers
    const float* wpack2;  // down1: [Cin][25][CP2] rows = stem*16+co over the `stack` stems of this launch (shared input);
                          // up5:   [Cin][15][32] per stem, rows = px*16+co, 15 = (ky, dx) pairs (two x-parity classes per tile)
    size_t wpack2_stem;
    int CP2, stack;
    int rowsplit;         // srt_down1_stream_kernel<.., NW = 2>: runs of intervals a column is cut into (one workgroup each); 0 / 1: whole columns
    // fp16-MFMA variant (srt_nn3.hip): [Cin/16][25][2][CP][8] IEEE halves (k-group of 8 channels innermost)
    const uint16_t* wpack16; size_t wpack16_stem;
    int nsplit;           // 1: activations rounded to fp16; 2: activations split hi+lo (two MFMAs per tap, ~fp32 products)
    // Split-K for small batches (srt_nn2.hip): when a layer's grid would leave most of the 256 CUs idle, its K loop (input
    // channels) is cut into `ksplit` slices that run as separate workgroups; each writes its partial sums to
    // ws[slice][same offsets as the output tensor] and srt_splitk_reduce adds the slices in slice order (bit-stable run to
    // run) and applies the layer's epilogue.  The launcher picks ksplit; ws_floats = capacity of ws (0: never split).
    float* ws; size_t ws_floats; int ksplit; size_t ws_slice;
    // fp16 activation storage (srt_config.precision == SRT_PREC_F16): the tensors behind srcA / srcB (in16) and outRaw / outAct
    // (out16) hold IEEE halves instead of floats - same planar layout, same strides IN ELEMENTS, half the HBM bytes.  The
    // pointers keep their float type in this struct; kernels that honour the flags reinterpret them.
    int in16, out16;
    // Large fp16-storage batches (round 6, srt_nn5.hip): the tensors between down2 and up5 are channel-interleaved by eight ("C8": element (c, y, x) at
    // ((c/8) H W + y W + x) 8 + c % 8 - the B-fragment layout of the fp16 MFMA, so patches go HBM -> LDS by DMA alone).  c8out: srt_enc_f16 (down2: planar
    // input) stores its raw + act outputs in that form.  wpack16cs: up5's class-stacked fp16 weights [Cin/16][15][2][32][8] (srt_pack16_classstack_kernel).
    int c8out;
    int c8srcB;           // up6 (srt_nn.hip): srcB = up5's output holds its 16 channels C8 (two groups of 8) instead of planar
    int c8srcA;           // ... and so does srcA, down1's raw skip tensor (batches whose down1 runs on the streamed kernels: srt_down1_c8_ok)
    const uint16_t* wpack16cs; size_t wpack16cs_stem;
    float* outRaw;        // encoder: conv+bias (the skip tensor AND the next encoder layer's input); decoder: unused
    float* outAct;        // decoder: bn(act(v)); encoder: unused (the BN + activation is applied by the consumer)
    size_t out_stem, out_tile;
    int act;              // activation kind of the stems whose elu_mask bit is clear (encoder: LeakyReLU, decoder: ReLU)
    unsigned elu_mask;    // bit s set: stem s of this launch uses ELU (stemMode != 0, spleeter.c:130-139)
    int variant;
};

struct SrtHeadParams {    // up7: 4x4 dilation-2 conv 1->2 channels + bias + sigmoid  (spleeter.c:156,295-300)
    int H, W, ntiles, nstems;
    const float* src; size_t src_stem, src_tile;
    const float* w; const float* bias; size_t coeff_stem;
    float* out; size_t out_stem, out_tile;    // [2][H][W] per instance
    int variant;
    int out16;            // the two mask planes leave as IEEE halves (same layout, strides in elements): the engine's own mask buffer between the network and the inverse
                          // transform of srtSeparate in the fp16 mode (srt_head_out16_ok); the masks srtForward hands to its caller are always floats
};
int  srt_head_out16_ok(const SrtHeadParams& p);      // the launch runs on srt_head_rows_kernel<.., 4, true>

// launchers (srt_nn.hip)
int  srt_launch_enc(const SrtConvParams& p, int impl, hipStream_t s);
int  srt_launch_dec(const SrtConvParams& p, int impl, hipStream_t s);
int  srt_launch_head(const SrtHeadParams& p, hipStream_t s);
int  srt_launch_up6_head(const SrtConvParams& p, const SrtHeadParams& h, hipStream_t s);   // both layers in one pass; 1: not covered / switched off
int  srt_launch_bn_act(const float* raw, int raw16, float* out, const float* scale, const float* shift, int C, size_t hw, int kind, int variant, hipStream_t s);
int  srt_launch_half_to_float(const void* src, float* dst, size_t n, hipStream_t s);
int  srt_launch_pack_enc(const float* w, float* wp, int Cin, int Cout, int CP, hipStream_t s);
int  srt_launch_pack_dec(const float* w, float* wp, int Cin, int Cout, int CP, hipStream_t s);
// v2 kernels (srt_nn2.hip): return 1 when the layer geometry is not covered (caller falls back to the v1 kernels)
int  srt_launch_enc2(const SrtConvParams& p, hipStream_t s);
int  srt_launch_down1_f16(const SrtConvParams& p, hipStream_t s);   // fp16 mode, C8 outputs: down1 on the fp16 MFMA, p.stack = 1..6 stems in one launch (1: not covered)
int  srt_down1_c8_ok(int H, int W, int ntiles, size_t out_stem);   // fp16 storage: down1 of this batch runs on the streamed kernels, which can write raw1 / act1 C8
int  srt_launch_dec2(const SrtConvParams& p, hipStream_t s);
int  srt_launch_pack_stemstack(const float* coeff_w0, size_t coeff_stem, int nstems, float* wp2, int Cin, int Cout, int CP2, hipStream_t s);
int  srt_launch_pack_classstack(const float* w, float* wp2, int Cin, int Cout, hipStream_t s);
// fp16-MFMA kernels (srt_nn3.hip): return 1 when the layer is not covered (caller uses the fp32 kernels)
int  srt_launch_enc_f16(const SrtConvParams& p, hipStream_t s);
int  srt_launch_dec_f16(const SrtConvParams& p, hipStream_t s);
int  srt_launch_pack16(const float* w, uint16_t* wp16, int Cin, int Cout, int CP, int dec, hipStream_t s);
// C8-form fp16 kernels (srt_nn5.hip): return 1 when the layer is not covered
int  srt_launch_enc_c8(const SrtConvParams& p, hipStream_t s);
int  srt_launch_dec_c8(const SrtConvParams& p, hipStream_t s);
int  srt_launch_pack16_classstack(const float* w, uint16_t* wp, int Cin, int Cout, hipStream_t s);
int  srt_launch_c8_to_float(const void* src, float* dst, int C, size_t hw, hipStream_t s);
int  srt_launch_count_not_fp16(const float* w, size_t n, unsigned* d_count, hipStream_t s);   // weights the fp16 pack would round (SRT_PREC_F16X2 guard)
int  srt_set_sigmoid_table(const float* tbl1026);
void srt_fp16_expand(const uint16_t* d_in, float* d_out, size_t n, hipStream_t s);
// Winograd decoder kernels (srt_nn4.hip): U = transformed weights [Cin/4][Cout/16][4][16][52] per stem; the launcher returns 1
// when the layer is not covered.  srt_wino_mask(): bit i set = up(i+1) runs this form (large batches, fp32 MFMA path).
int  srt_launch_pack_wino(const float* w, float* u, int Cin, int Cout, hipStream_t s);
int  srt_launch_dec_wino(const SrtConvParams& p, const float* U, size_t u_stem, hipStream_t s);
// encoder layers in Winograd form (srt_nn4.hip, srt_enc_wino32): U from the OIHW weights; the layer reads act(BN(raw)) of its input (srcA) and
int srt_enc_producer_copy();          // tuning builds: SRT_TUNE=...,enccopy=0 keeps the separate bn+act pass in front of the first Winograd-form encoder layer
// writes raw (outRaw) + optionally its own act(BN(.)) copy (outAct with bnScale / bnShift).  srt_enc_wino_covers: geometry test of the launcher.
int  srt_launch_pack_wino_enc(const float* w, float* u, int Cin, int Cout, hipStream_t s);
int  srt_launch_enc_wino(const SrtConvParams& p, const float* U, size_t u_stem, hipStream_t s);
int  srt_enc_wino_covers(int Cin, int Cout, int H, int W);
int  srt_launch_bn_act_batch(const float* raw, float* out, const float* scale, const float* shift, size_t coeff_stem, int nstems, int ntiles, int C, size_t hw,
                             int act, unsigned elu_mask, int variant, hipStream_t s);
int  srt_wino_mask();
int  srt_wino_force();

// DSP launchers (srt_dsp.hip)
struct SrtDspTables { const float* preWin; const float* postWin; const float2* twiddle; };
struct SrtStftParams {
    const float* L; const float* R; size_t nsamples;
    int frames_computed;      // frames that get an FFT (tail frame zero padded); rows beyond are zero
    int rows_total;           // rows to write (>= frames_computed): spectrum + magnitude rows (zero filled)
    float2* spec;             // [2][spec_rows][SRT_SPEC_LD]
    size_t spec_ch_stride;    // float2 elements between channels
    float* mag;               // [ntiles][2][T][F] or nullptr
    int T, F;
    SrtDspTables tab;
};
struct SrtIstftParams {
    const float2* spec; size_t spec_ch_stride;
    int frames;               // rows to synthesise
    const float* masks;       // [nstems][ntiles][2][T][F] or nullptr (all-ones)
    int masks16;              // the masks are halves (srt_istft_ola3_kernel<.., M16>: F <= 1024, no ratio); see SrtHeadParams::out16
    int nstems, ntiles, T, F;
    float oob[SRT_MAX_STEMS];
    int ratio;                // 1: masks are normalised across the stems while they are applied, m_s^2 / sum_j m_j^2 (srt_config.ratio_mask; needs masks of ALL nstems)
    float* frames_out;        // [nstems][2][frames][4096] windowed time frames (temp)
    float* out;               // [nstems][2][out_len]
    size_t out_len;           // frames*1024 + 3072
    SrtDspTables tab;
};
int srt_launch_stft(const SrtStftParams& p, hipStream_t s);
int srt_launch_istft(const SrtIstftParams& p, hipStream_t s);

// residual chain of the offline CLI (main.c:845-866): res = spec - spec*mask (bins >= F: spec - spec*oob), |res|*4096 -> mag
struct SrtResidualParams {
    const float2* spec; float2* res; size_t spec_ch_stride;
    int rows, T, F;
    const float* mask;        // ONE stem: [ntiles][2][T][F]
    float oob;
    float* mag;               // [ntiles][2][T][F]
};
int srt_launch_residual(const SrtResidualParams& p, hipStream_t s);
// out[c][i] = (i < na ? a[c][i] : 0) - b[c][i] for lo <= i < min(hi, nb), two channels of plane length nb (time-domain residual, main.c:794-798, 924-928)
int srt_launch_time_residual(const float* aL, const float* aR, size_t na, const float* b, size_t nb, float* out, size_t lo, size_t hi, hipStream_t s);
// chunk stitching on the device: out[p][0:3072] += carry[p] (unless first), then carry[p] = out[p][tail : tail+3072] (unless last)
int srt_launch_carry(float* out, size_t plane_len, int nplanes, size_t tail, float* carry, int first, int last, hipStream_t s);
// cross-stem ratio mask, in place on [nstems][count]: m_s <- (m_s^2 + eps/S) / (sum_j m_j^2 + eps)
int srt_launch_ratio_mask(float* masks, int nstems, size_t count, hipStream_t s);

// streaming (srt_dsp.hip kernels, srt_stream.hip host logic): one hop = 1 forward + 4 masked inverse FFTs + 50 % OLA
struct SrtStreamHop {
    const float* ring;        // [2][4096] device copy of the input ring buffer
    int inPos;                // ring read origin (Spleeter4Stems.c:262)
    float2* specRow;          // [2 ch] rows of the spectrum buffer for this cursor: L at specRow, R at specRow + specChStride
    size_t specChStride;
    float* magRow;            // [2 ch] magnitude rows: L at magRow, R at magRow + magChStride
    size_t magChStride;
    const float* maskRow;     // stem s, channel c at maskRow + s*maskStemStride + c*maskChStride
    size_t maskStemStride, maskChStride;
    int F;
    float* overlap;           // [8][1024]
    float* out;               // [1024][8] interleaved segment
    const float* analysisWnd; const float* synthesisWnd; const float2* twiddle;
};
int srt_launch_stream_hop(const SrtStreamHop& p, hipStream_t s);

// ---- multi-device host driver (srt_multi.hip) over the engine (srt_engine.hip)
struct srt_engine;
// One tile RANGE of a longer host-resident stream through engine `e` (chunked, copies overlapped with compute): h_out points at the range's first
// output sample inside planes of out_stride floats; h_tail (may be null: last range) receives the range's last 3072 samples per plane - the overlap-add
// contribution to the next range's first samples; head: a previous range will add its tail to this range's first 3072 samples.  cli_stems 0 | 2 | 3.
int srt_engine_host_range(srt_engine* e, const float* h_L, const float* h_R, size_t n, size_t frames, size_t rows, float* h_out, size_t out_stride,
                          float* h_tail, bool head, unsigned flags, int cli_stems);
const float* srt_engine_coeff_device(const srt_engine* e, int stem);      // the stem's fp32 spleeterCoeff blob in the engine's HBM (valid after srtSetCoeff*)
int srt_engine_device(const srt_engine* e);
void* srt_engine_stream(const srt_engine* e);
