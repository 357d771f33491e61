// srt_nn.hip — U-Net mask network kernels for gfx950 (CDNA4).
//
// Replaces, for the hot path only, the reference's im2col + SGEMM + col2im stack:
//   encoder conv   : Executable/spleeter.c:96-100  + im2col_dilated.c:10-33 + gemm.c:6-19   (K3/K4 in SURVEY §2.2)
//   decoder tconv  : Executable/spleeter.c:73-78   + gemm.c:33-45 + im2col_dilated.c:42-65   (K5/K6)
//   head + sigmoid : Executable/spleeter.c:295-300, :30-42                                    (K7)
//
// Design (not a translation): no im2col/col buffer is ever materialised.
//   * encoder: implicit GEMM.  A workgroup stages the input patch of KC channels in LDS with the columns
//     de-interleaved by parity (stride-2 taps then read consecutive LDS words), stages the matching slab of
//     K-major packed weights, and issues v_mfma_f32_32x32x2_f32 with the k-pair = two input channels of one tap.
//   * decoder: gather form.  out[2a+py][2b+px] only receives taps with ky = py+1 (mod 2), kx = px+1 (mod 2), so the
//     transposed conv is 4 parity-class convolutions over the SAME input patch: no scatter, no atomics,
//     deterministic, epilogue (bias -> act -> BN) fused, the skip concat is two source pointers.
//   * global->LDS staging is register-prefetched one K-chunk ahead so HBM/L2 latency hides under the MFMAs.
// The naive kernels are a bit-simple cross-check path (SRT_IMPL_NAIVE) and serve layers not yet on MFMA.
#include "srt_device.h"
#include <type_traits>
#include <stdlib.h>

// ------------------------------------------------------------------------------------------- activations
__device__ float g_sigmoid_tbl[1026];

int srt_set_sigmoid_table(const float* tbl1026)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sigmoid_tbl), tbl1026, 1026 * sizeof(float)) == hipSuccess ? 0 : -1;
}

#pragma clang fp contract(off)
__device__ __forceinline__ float srt_sigmoid(float x, int variant)
{
    if (variant == 0) {                                                  // LUT, Executable/spleeter.c:30-42
        if (x > 7.0f) return 1.0f;
        if (x < -7.0f) return 0.0f;
        const float step = 0.01367188f;
        short idx = (short)((x + 7.0f) / step);
        float x1 = -7.0f + step * idx;
        float t0 = g_sigmoid_tbl[idx], t1 = g_sigmoid_tbl[idx + 1];
        return t0 + (t1 - t0) / (-7.0f + step * (idx + 1) - x1) * (x - x1);
    }
    if (x >= 0.0f) { float z = expf(-x); return 1.0f / (1.0f + z); }     // VST/Source/spleeter.c:56-65
    float z = expf(x);
    return z / (1.0f + z);
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------- naive kernels
__global__ void srt_enc_naive(const SrtConvParams p)
{
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int stem = blockIdx.y / p.ntiles, tile = blockIdx.y % p.ntiles;
    const size_t hw = (size_t)p.H * p.W, total = (size_t)p.Cout * Ho * Wo;
    const float* w = p.wraw + stem * p.coeff_stem;
    const size_t cs = stem * p.coeff_stem;
    const int kind = srt_act_kind(p, stem);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ox = e % Wo, oy = (e / Wo) % Ho, co = e / ((size_t)Wo * Ho);
        float acc = 0.0f;
        for (int ci = 0; ci < p.Cin; ++ci) {
            const float* x = srt_src_channel(p, stem, tile, ci, hw);
            const float* wk = w + ((size_t)co * p.Cin + ci) * 25;
            for (int ky = 0; ky < 5; ++ky) {
                const int r = 2 * oy + ky - 1;
                if (r < 0 || r >= p.H) continue;
                for (int kx = 0; kx < 5; ++kx) {
                    const int c = 2 * ox + kx - 1;
                    if (c < 0 || c >= p.W) continue;
                    float xv = x[(size_t)r * p.W + c];
                    if (p.inScale) xv = srt_enc_epilogue(xv, p.inScale[cs + ci], p.inShift[cs + ci], kind, p.variant);   // producer stored conv + bias only
                    acc += wk[ky * 5 + kx] * xv;
                }
            }
        }
        p.outRaw[stem * p.out_stem + tile * p.out_tile + e] = acc + p.bias[cs + co];
    }
}

__global__ void srt_dec_naive(const SrtConvParams p)
{
    const int Ho = p.H << 1, Wo = p.W << 1;
    const int stem = blockIdx.y / p.ntiles, tile = blockIdx.y % p.ntiles;
    const size_t hw = (size_t)p.H * p.W, total = (size_t)p.Cout * Ho * Wo;
    const float* w = p.wraw + stem * p.coeff_stem;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int X = e % Wo, Y = (e / Wo) % Ho, co = e / ((size_t)Wo * Ho);
        float acc = 0.0f;
        for (int ci = 0; ci < p.Cin; ++ci) {
            const float* x = srt_src_channel(p, stem, tile, ci, hw);
            const float* wk = w + ((size_t)ci * p.Cout + co) * 25;
            for (int ky = (Y + 1) & 1; ky < 5; ky += 2) {
                const int h = (Y + 1 - ky) >> 1;                      // 2h + ky - 1 == Y
                if (h < 0 || h >= p.H) continue;
                for (int kx = (X + 1) & 1; kx < 5; kx += 2) {
                    const int ww = (X + 1 - kx) >> 1;
                    if (ww >= 0 && ww < p.W) acc += wk[ky * 5 + kx] * x[(size_t)h * p.W + ww];
                }
            }
        }
        p.outAct[stem * p.out_stem + tile * p.out_tile + e] =
            srt_dec_epilogue(acc, p.bias[stem * p.coeff_stem + co], p.bnScale[stem * p.coeff_stem + co], p.bnShift[stem * p.coeff_stem + co], srt_act_kind(p, stem), p.variant);
    }
}

// act(bn(raw)) of one instance into a scratch tensor: what the next encoder layer applies while staging, materialised only
// for srtCopyTensor("actN") (debug / parity taps)
__global__ void srt_bn_act_kernel(const float* __restrict__ raw, int raw16, float* __restrict__ out, const float* scale, const float* shift, int C, size_t hw, int kind, int variant)
{
    const size_t total = (size_t)C * hw;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / hw);
        const float v = raw16 ? (float)reinterpret_cast<const _Float16*>(raw)[e] : raw[e];
        out[e] = srt_enc_epilogue(v, scale[c], shift[c], kind, variant);
    }
}
int srt_launch_bn_act(const float* raw, int raw16, float* out, const float* scale, const float* shift, int C, size_t hw, int kind, int variant, hipStream_t s)
{
    SRT_LAUNCH(srt_bn_act_kernel, dim3(1024), dim3(256), 0, s, raw, raw16, out, scale, shift, C, hw, kind, variant);
    return srt_launch_status();
}
// The same for a whole batch, [stem][tile][C][hw] -> [stem][tile][C][hw]: the act(BN(raw)) copy a Winograd-form encoder layer reads (srt_nn4.hip:
// the non-linearity cannot ride through the input transform) when its producer was a direct kernel, which stores the raw tensor only.
__global__ void __launch_bounds__(256) srt_bn_act_batch_kernel(const float* __restrict__ raw, float* __restrict__ out, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 size_t coeff_stem, int ntiles, int C, size_t hw4, int act, unsigned elu_mask, int variant)
{
    const int sc = blockIdx.y, stem = sc / C, c = sc % C;                    // one (stem, channel) per grid row: its BN constants are scalars
    const SrtAct actp = srt_act_params(((elu_mask >> stem) & 1u) ? SRT_ACT_ELU : act, variant);
    const float a = scale[stem * coeff_stem + c], b = shift[stem * coeff_stem + c];
    const size_t per = (size_t)ntiles * hw4;                                 // float4s of this (stem, channel) over the tiles
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < per; e += (size_t)gridDim.x * 256) {
        const size_t tile = e / hw4, i = e % hw4;
        const size_t off = (((size_t)stem * ntiles + tile) * C + c) * hw4 + i;
        const float4 v = reinterpret_cast<const float4*>(raw)[off];
        reinterpret_cast<float4*>(out)[off] = srt_enc_input4(v, a, b, actp);
    }
}
int srt_launch_bn_act_batch(const float* raw, float* out, const float* scale, const float* shift, size_t coeff_stem, int nstems, int ntiles, int C, size_t hw,
                            int act, unsigned elu_mask, int variant, hipStream_t s)
{
    if (hw % 4) return -1;
    const size_t per = (size_t)ntiles * (hw / 4);
    const unsigned bx = (unsigned)((per + 255) / 256 > 64 ? 64 : (per + 255) / 256);
    SRT_LAUNCH(srt_bn_act_batch_kernel, dim3(bx, nstems * C), dim3(256), 0, s, raw, out, scale, shift, coeff_stem, ntiles, C, hw / 4, act, elu_mask, variant);
    return srt_launch_status();
}
__global__ void srt_half_to_float_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, size_t n)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) dst[e] = (float)src[e];
}
int srt_launch_half_to_float(const void* src, float* dst, size_t n, hipStream_t s)
{
    SRT_LAUNCH(srt_half_to_float_kernel, dim3(1024), dim3(256), 0, s, (const _Float16*)src, dst, n);
    return srt_launch_status();
}

// up7 head: direct 16-tap stencil, both output channels per thread (bandwidth kernel; 1 MiB in, 2 MiB out per instance)
__global__ void srt_head_kernel(const SrtHeadParams p)
{
    const int stem = blockIdx.y / p.ntiles, tile = blockIdx.y % p.ntiles;
    const size_t hw = (size_t)p.H * p.W;
    const float* x = p.src + stem * p.src_stem + tile * p.src_tile;
    float* y = p.out + stem * p.out_stem + tile * p.out_tile;
    float wk[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) wk[i] = p.w[stem * p.coeff_stem + i];
    const float b0 = p.bias[stem * p.coeff_stem], b1 = p.bias[stem * p.coeff_stem + 1];
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < hw; e += (size_t)gridDim.x * blockDim.x) {
        const int w = e % p.W, h = e / p.W;
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int r = h + 2 * ky - 3;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int c = w + 2 * kx - 3;
                const float v = (r >= 0 && r < p.H && c >= 0 && c < p.W) ? x[(size_t)r * p.W + c] : 0.0f;
                a0 += wk[ky * 4 + kx] * v;
                a1 += wk[16 + ky * 4 + kx] * v;
            }
        }
        y[e] = srt_sigmoid(a0 + b0, p.variant);
        y[hw + e] = srt_sigmoid(a1 + b1, p.variant);
    }
}


// Branch-free logistic for the product head kernel: one v_exp_f32 and one v_rcp_f32 (|error| < 3e-7; the parity
// tolerance on masks is 2e-4).  The debug kernels above keep the libm form.
__device__ __forceinline__ float srt_sigmoid_fast(float x)
{
    const float z = __expf(-fabsf(x));
    const float r = __builtin_amdgcn_rcpf(1.0f + z);
    return x >= 0.0f ? r : z * r;
}

// 4 pixels per thread along frequency: 12 aligned float4 loads feed 64 packed FMAs (the two output channels of a tap ride
// in one v_pk_fma_f32), outputs leave as float4 (W % 4 == 0).  LUT selects the Executable's table sigmoid.
typedef float srt_v2f __attribute__((ext_vector_type(2)));
template <bool LUT>
__global__ void __launch_bounds__(256) srt_head_kernel4(const SrtHeadParams p)
{
    // 1-D launch in XCD order: the four workgroups that read one input row (output rows h-3, h-1, h+1, h+3) run on the same
    // L2 instead of four different XCDs (measured before: 4x the input bytes from HBM)
    const int nbx = gridDim.x / (p.nstems * p.ntiles);
    const int pos = srt_xcd_order(gridDim.x), bx = pos % nbx, inst = pos / nbx;
    const int stem = inst / p.ntiles, tile = inst % p.ntiles;
    const size_t hw = (size_t)p.H * p.W;
    const float* x = p.src + stem * p.src_stem + tile * p.src_tile;
    float* y = p.out + stem * p.out_stem + tile * p.out_tile;
    srt_v2f wk[16];                                                          // (channel 0, channel 1) weight of each tap
#pragma unroll
    for (int i = 0; i < 16; ++i) { wk[i].x = p.w[stem * p.coeff_stem + i]; wk[i].y = p.w[stem * p.coeff_stem + 16 + i]; }
    const float b0 = p.bias[stem * p.coeff_stem], b1 = p.bias[stem * p.coeff_stem + 1];
    const int W4 = p.W >> 2;
    const size_t nq = (size_t)p.H * W4;
    for (size_t e = (size_t)bx * blockDim.x + threadIdx.x; e < nq; e += (size_t)nbx * blockDim.x) {
        const int w0 = (int)(e % W4) * 4, h = (int)(e / W4);
        srt_v2f a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i].x = 0.f; a[i].y = 0.f; }
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int r = h + 2 * ky - 3;
            const bool rok = r >= 0 && r < p.H;
            float win[12];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = w0 - 4 + 4 * j;
                const bool ok = rok && c >= 0 && c < p.W;
                const float4 v = *reinterpret_cast<const float4*>(x + (ok ? (size_t)r * p.W + c : 0));
                win[4 * j + 0] = ok ? v.x : 0.f; win[4 * j + 1] = ok ? v.y : 0.f; win[4 * j + 2] = ok ? v.z : 0.f; win[4 * j + 3] = ok ? v.w : 0.f;
            }
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = win[i + 2 * kx + 1];                     // column w0 + i + 2kx - 3
                    const srt_v2f vv = { v, v };
                    a[i] = __builtin_elementwise_fma(wk[ky * 4 + kx], vv, a[i]);
                }
        }
        float4 o0, o1;
        if (LUT) {
            o0.x = srt_sigmoid(a[0].x + b0, 0); o0.y = srt_sigmoid(a[1].x + b0, 0); o0.z = srt_sigmoid(a[2].x + b0, 0); o0.w = srt_sigmoid(a[3].x + b0, 0);
            o1.x = srt_sigmoid(a[0].y + b1, 0); o1.y = srt_sigmoid(a[1].y + b1, 0); o1.z = srt_sigmoid(a[2].y + b1, 0); o1.w = srt_sigmoid(a[3].y + b1, 0);
        } else {
            o0.x = srt_sigmoid_fast(a[0].x + b0); o0.y = srt_sigmoid_fast(a[1].x + b0); o0.z = srt_sigmoid_fast(a[2].x + b0); o0.w = srt_sigmoid_fast(a[3].x + b0);
            o1.x = srt_sigmoid_fast(a[0].y + b1); o1.y = srt_sigmoid_fast(a[1].y + b1); o1.z = srt_sigmoid_fast(a[2].y + b1); o1.w = srt_sigmoid_fast(a[3].y + b1);
        }
        *reinterpret_cast<float4*>(y + (size_t)h * p.W + w0) = o0;
        *reinterpret_cast<float4*>(y + hw + (size_t)h * p.W + w0) = o1;
    }
}

// ------------------------------------------------------------------------------------------- weight packing
// encoder OIHW [Cout][Cin][25] -> [Cin][25][CP];  decoder [Cin][Cout][25] -> [Cin][25][CP]
__global__ void srt_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int CP, int dec)
{
    const size_t total = (size_t)Cin * 25 * CP;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int co = e % CP, tap = (e / CP) % 25, ci = e / ((size_t)CP * 25);
        float v = 0.0f;
        if (co < Cout) v = dec ? w[((size_t)ci * Cout + co) * 25 + tap] : w[((size_t)co * Cin + ci) * 25 + tap];
        wp[e] = v;
    }
}
int srt_launch_pack_enc(const float* w, float* wp, int Cin, int Cout, int CP, hipStream_t s)
{
    SRT_LAUNCH(srt_pack_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CP, 0);
    return srt_launch_status();
}
int srt_launch_pack_dec(const float* w, float* wp, int Cin, int Cout, int CP, hipStream_t s)
{
    SRT_LAUNCH(srt_pack_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CP, 1);
    return srt_launch_status();
}

// fp16 container -> fp32, half-denormals flushed to zero, no Inf/NaN special case (main.c:423-434)
__global__ void srt_fp16_expand_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t h = in[i];
        uint32_t mag = ((h & 0x7fffu) << 13) + 0x38000000u;
        if ((h & 0x7c00u) == 0) mag = 0;
        out[i] = __uint_as_float(mag | ((h & 0x8000u) << 16));
    }
}
void srt_fp16_expand(const uint16_t* d_in, float* d_out, size_t n, hipStream_t s)
{
    SRT_LAUNCH(srt_fp16_expand_kernel, dim3(2048), dim3(256), 0, s, d_in, d_out, n);
}

// ------------------------------------------------------------------------------------------- MFMA encoder
// Workgroup = 256 threads = 4 waves arranged WM x WN.  Output tile = BM channels x (NI instances x TH x TW pixels),
// split into 32-pixel sub-tiles of SH x SW (SH*SW == 32) so that one v_mfma_f32_32x32x2_f32 covers
// 32 output channels x one sub-tile.  k-pair of an MFMA = input channels (2cp, 2cp+1) at one tap.
template <int TW, int SW> struct EncPad {
    // half-plane width: >= TW+2, chosen so that the SH rows of a sub-tile land on disjoint LDS banks
    static constexpr int base = TW + 2;
    static constexpr int value = SW == 32 ? base : (SW == 16 ? ((base + 3) / 8 * 8 + 4) : ((base + 5) / 8 * 8 + 2));
};

template <int BM, int WM, int SW, int NSX, int NSY, int NI, int KC>
__global__ void __launch_bounds__(256, 2) srt_enc_mfma(const SrtConvParams p)
{
    constexpr int SH = 32 / SW, TW = NSX * SW, TH = NSY * SH;
    constexpr int NS = NSX * NSY * NI, WN = 4 / WM, MR = BM / (32 * WM), NR = NS / WN;
    static_assert(SH * SW == 32 && WM * WN == 4 && MR * 32 * WM == BM && NR * WN == NS, "bad tile");
    constexpr int PH = 2 * TH + 3, PCOLS = 2 * TW + 3;
    constexpr int PWH = EncPad<TW, SW>::value;
    static_assert(PWH >= TW + 2, "pad");
    constexpr int ROWS = 2 * PWH, INS = PH * ROWS, CHS = NI * INS;
    constexpr int NIN = KC * NI * PH * PCOLS, NLD = (NIN + 255) / 256;
    constexpr int NW4 = KC * 25 * BM / 4, NWL = (NW4 + 255) / 256;

    __shared__ float s_in[KC * CHS];
    __shared__ __attribute__((aligned(16))) float s_w[KC * 25 * BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave % WM, wn = wave / WM;
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int tilesX = (Wo + TW - 1) / TW;
    const int tx0 = (blockIdx.x % tilesX) * TW, ty0 = (blockIdx.x / tilesX) * TH;
    const int m0 = blockIdx.y * BM;
    const int groups = (p.ntiles + NI - 1) / NI;
    const int stem = blockIdx.z / groups, tile0 = (blockIdx.z % groups) * NI;
    const SrtAct actp = srt_act_params(srt_act_kind(p, stem), p.variant);
    const size_t hw = (size_t)p.H * p.W;
    const float* wp = p.wpack + stem * p.wpack_stem;

    float pin[NLD];
    float4 pw[NWL];

    // Branch-free staging: every lane always issues its loads (address clamped to a valid element) and the
    // out-of-range / padding lanes are zeroed by a select, so the loads pipeline instead of serialising.
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = min(tid + i * 256, NIN - 1);
            const int col = e % PCOLS, ru = e / PCOLS, r = ru % PH, il = (ru / PH) % NI, c = ru / (PH * NI);
            const int gy = 2 * ty0 + r - 1, gx = 2 * tx0 + col - 1, tile = tile0 + il;
            const bool ok = tile < p.ntiles && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const float* src = srt_src_channel(p, stem, ok ? tile : tile0, c0 + c, hw);
            float v = src[ok ? (size_t)gy * p.W + gx : 0];
            // fallback kernel (widths that are not multiples of 4): the producer's BN + activation is applied right at the load
            if (p.inScale) v = srt_enc_input1(v, p.inScale[stem * p.coeff_stem + c0 + c], p.inShift[stem * p.coeff_stem + c0 + c], actp);
            pin[i] = ok ? v : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int e = min(tid + i * 256, NW4 - 1);
            const int m4 = e % (BM / 4), row = e / (BM / 4);
            pw[i] = *reinterpret_cast<const float4*>(wp + ((size_t)c0 * 25 + row) * p.CP + m0 + m4 * 4);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * 256;
            if (e < NIN) {
                const int col = e % PCOLS, r = (e / PCOLS) % PH, il = (e / (PCOLS * PH)) % NI, c = e / (PCOLS * PH * NI);
                s_in[c * CHS + il * INS + r * ROWS + (col & 1) * PWH + (col >> 1)] = pin[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int e = tid + i * 256;
            if (e < NW4) *reinterpret_cast<float4*>(&s_w[e * 4]) = pw[i];
        }
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    int boff[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int s = wn * NR + nr;
        const int il = s / (NSX * NSY), sy = (s / NSX) % NSY, sx = s % NSX;
        const int oy = sy * SH + l31 / SW, ox = sx * SW + l31 % SW;
        boff[nr] = half * CHS + il * INS + 2 * oy * ROWS + ox;
    }
    const int aoff = half * 25 * BM + wm * MR * 32 + l31;

    const int nchunks = p.Cin / KC;
    load_chunk(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk((ch + 1) * KC);
#pragma unroll
        for (int cp = 0; cp < KC / 2; ++cp) {
#pragma unroll
            for (int tap = 0; tap < 25; ++tap) {
                const int ky = tap / 5, kx = tap % 5;
                float a[MR], b[NR];
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) a[mr] = s_w[aoff + (2 * cp * 25 + tap) * BM + mr * 32];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) b[nr] = s_in[boff[nr] + 2 * cp * CHS + ky * ROWS + (kx & 1) * PWH + (kx >> 1)];
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mr], b[nr], acc[mr][nr], 0, 0, 0);
            }
        }
    }

    // epilogue: lane holds pixel l31 of each sub-tile and 16 output channels per accumulator; conv + bias is stored once
    const float* bias = p.bias + stem * p.coeff_stem;
    const size_t ohw = (size_t)Ho * Wo;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = bias[min(m0 + (wm * MR + mr) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.Cout - 1)];
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int s = wn * NR + nr;
            const int il = s / (NSX * NSY), sy = (s / NSX) % NSY, sx = s % NSX;
            const int oy = ty0 + sy * SH + l31 / SW, ox = tx0 + sx * SW + l31 % SW, tile = tile0 + il;
            const bool pix_ok = tile < p.ntiles && oy < Ho && ox < Wo;
            const size_t obase = stem * p.out_stem + (pix_ok ? tile : 0) * p.out_tile + (pix_ok ? (size_t)oy * Wo + ox : 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * MR + mr) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pix_ok && m < p.Cout) p.outRaw[obase + (size_t)m * ohw] = acc[mr][nr][r] + bi[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- MFMA decoder
// Tile is expressed in INPUT-resolution pixels (a,b); the workgroup produces the 2TH x 2TW output patch as four
// parity-class accumulators.  tap (ky,kx) -> class (py,px) = ((ky+1)&1, (kx+1)&1), input shift dy = (py+1-ky)/2.
template <int BM, int WM, int SW, int NSX, int NSY, int NI, int KC>
__global__ void __launch_bounds__(256, 2) srt_dec_mfma(const SrtConvParams p)
{
    constexpr int SH = 32 / SW, TW = NSX * SW, TH = NSY * SH;
    constexpr int NS = NSX * NSY * NI, WN = 4 / WM, MR = BM / (32 * WM), NR = NS / WN;
    static_assert(SH * SW == 32 && WM * WN == 4 && MR * 32 * WM == BM && NR * WN == NS, "bad tile");
    constexpr int PH = TH + 2, PC = TW + 2;
    constexpr int ROWS = PC + 1, INS = PH * ROWS, CHS = NI * INS;
    constexpr int NIN = KC * NI * PH * PC, NLD = (NIN + 255) / 256;
    constexpr int NW4 = KC * 25 * BM / 4, NWL = (NW4 + 255) / 256;

    __shared__ float s_in[KC * CHS];
    __shared__ __attribute__((aligned(16))) float s_w[KC * 25 * BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave % WM, wn = wave / WM;
    const int tilesX = (p.W + TW - 1) / TW;
    const int tx0 = (blockIdx.x % tilesX) * TW, ty0 = (blockIdx.x / tilesX) * TH;
    const int m0 = blockIdx.y * BM;
    const int groups = (p.ntiles + NI - 1) / NI;
    const int stem = blockIdx.z / groups, tile0 = (blockIdx.z % groups) * NI;
    const SrtAct actp = srt_act_params(srt_act_kind(p, stem), p.variant);
    const size_t hw = (size_t)p.H * p.W;
    const float* wp = p.wpack + stem * p.wpack_stem;

    float pin[NLD];
    float4 pw[NWL];

    auto load_chunk = [&](int c0) {              // branch-free, see srt_enc_mfma
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = min(tid + i * 256, NIN - 1);
            const int col = e % PC, ru = e / PC, r = ru % PH, il = (ru / PH) % NI, c = ru / (PH * NI);
            const int gy = ty0 + r - 1, gx = tx0 + col - 1, tile = tile0 + il;
            const bool ok = tile < p.ntiles && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const float* src = srt_src_channel(p, stem, ok ? tile : tile0, c0 + c, hw);
            const float v = src[ok ? (size_t)gy * p.W + gx : 0];
            pin[i] = ok ? v : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int e = min(tid + i * 256, NW4 - 1);
            const int m4 = e % (BM / 4), row = e / (BM / 4);
            pw[i] = *reinterpret_cast<const float4*>(wp + ((size_t)c0 * 25 + row) * p.CP + m0 + m4 * 4);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * 256;
            if (e < NIN) {
                const int col = e % PC, r = (e / PC) % PH, il = (e / (PC * PH)) % NI, c = e / (PC * PH * NI);
                s_in[c * CHS + il * INS + r * ROWS + col] = pin[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int e = tid + i * 256;
            if (e < NW4) *reinterpret_cast<float4*>(&s_w[e * 4]) = pw[i];
        }
    };

    f32x16 acc[4][MR][NR];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.0f;

    int boff[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int s = wn * NR + nr;
        const int il = s / (NSX * NSY), sy = (s / NSX) % NSY, sx = s % NSX;
        const int a = sy * SH + l31 / SW, b = sx * SW + l31 % SW;
        boff[nr] = half * CHS + il * INS + a * ROWS + b;
    }
    const int aoff = half * 25 * BM + wm * MR * 32 + l31;

    const int nchunks = p.Cin / KC;
    load_chunk(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk((ch + 1) * KC);
#pragma unroll
        for (int cp = 0; cp < KC / 2; ++cp) {
            float b[9][NR];
#pragma unroll
            for (int sh = 0; sh < 9; ++sh)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    b[sh][nr] = s_in[boff[nr] + 2 * cp * CHS + (sh / 3) * ROWS + (sh % 3)];   // (1+dy)*ROWS + (1+dx)
#pragma unroll
            for (int tap = 0; tap < 25; ++tap) {
                const int ky = tap / 5, kx = tap % 5;
                const int py = (ky + 1) & 1, px = (kx + 1) & 1;
                const int dy = (py + 1 - ky) / 2, dx = (px + 1 - kx) / 2;     // exact: numerators are even
                const int cls = py * 2 + px, sh = (dy + 1) * 3 + (dx + 1);
                float a[MR];
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) a[mr] = s_w[aoff + (2 * cp * 25 + tap) * BM + mr * 32];
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        acc[cls][mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mr], b[sh][nr], acc[cls][mr][nr], 0, 0, 0);
            }
        }
    }

    const float* bias = p.bias + stem * p.coeff_stem;
    const float* scale = p.bnScale + stem * p.coeff_stem;
    const float* shift = p.bnShift + stem * p.coeff_stem;
    const int Ho = p.H << 1, Wo = p.W << 1;
    const size_t ohw = (size_t)Ho * Wo;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        float bi[16], sc[16], sf[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + (wm * MR + mr) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.Cout - 1);
            bi[r] = bias[m]; sc[r] = scale[m]; sf[r] = shift[m];
        }
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int s = wn * NR + nr;
            const int il = s / (NSX * NSY), sy = (s / NSX) % NSY, sx = s % NSX;
            const int a = ty0 + sy * SH + l31 / SW, b = tx0 + sx * SW + l31 % SW, tile = tile0 + il;
            const bool pix_ok = tile < p.ntiles && a < p.H && b < p.W;
            const size_t obase = stem * p.out_stem + (pix_ok ? tile : 0) * p.out_tile + (pix_ok ? (size_t)(2 * a) * Wo + 2 * b : 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * MR + mr) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pix_ok && m < p.Cout) {
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        float2 v;
                        v.x = srt_dec_epilogue(acc[py * 2 + 0][mr][nr][r], bi[r], sc[r], sf[r], actp);
                        v.y = srt_dec_epilogue(acc[py * 2 + 1][mr][nr][r], bi[r], sc[r], sf[r], actp);
                        *reinterpret_cast<float2*>(p.outAct + obase + (size_t)m * ohw + (size_t)py * Wo) = v;
                    }
                }
            }
        }
    }
}


// The same stencil with NRO output rows per thread: output rows h0, h0 + 2, ... (one parity) read the input rows h0 - 3, h0 - 1, ... of the
// other parity, so NRO of them share all but three of their input rows: 3 (NRO + 3) float4 loads per NRO output quads instead of 12 NRO - for
// NRO = 4, 5.25 instead of 12 per quad.  The layer is 0.8 GB of HBM traffic; what it was short of is load issue, not bytes.  Per output the taps
// accumulate in the order of the kernel above (ky, then kx): identical results.
// O16: the masks leave as halves (8-byte stores): the engine's own mask buffer in the fp16 mode (SrtHeadParams::out16); same values, rounded once
template <bool LUT, int NRO, bool O16 = false>
__global__ void __launch_bounds__(256) srt_head_rows_kernel(const SrtHeadParams p)
{
    const int W4 = p.W >> 2, nsets = 2 * ((p.H + 2 * NRO - 1) / (2 * NRO));
    const int nbx = gridDim.x / (p.nstems * p.ntiles);
    const int pos = srt_xcd_order(gridDim.x), bx = pos % nbx, inst = pos / nbx;
    const int stem = inst / p.ntiles, tile = inst % p.ntiles;
    const size_t hw = (size_t)p.H * p.W;
    const float* x = p.src + stem * p.src_stem + tile * p.src_tile;
    float* y = p.out + stem * p.out_stem + tile * p.out_tile;
    srt_v2f wk[16];                                                          // (channel 0, channel 1) weight of each tap
#pragma unroll
    for (int i = 0; i < 16; ++i) { wk[i].x = p.w[stem * p.coeff_stem + i]; wk[i].y = p.w[stem * p.coeff_stem + 16 + i]; }
    const float b0 = p.bias[stem * p.coeff_stem], b1 = p.bias[stem * p.coeff_stem + 1];
    const size_t nq = (size_t)nsets * W4;
    for (size_t e = (size_t)bx * blockDim.x + threadIdx.x; e < nq; e += (size_t)nbx * blockDim.x) {
        const int w0 = (int)(e % W4) * 4, rs = (int)(e / W4), h0 = (rs & 1) + 2 * NRO * (rs >> 1);
        srt_v2f a[NRO][4];
#pragma unroll
        for (int j = 0; j < NRO; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[j][i].x = 0.f; a[j][i].y = 0.f; }
#pragma unroll
        for (int m = 0; m < NRO + 3; ++m) {                                  // input row h0 - 3 + 2 m = row of tap ky = m - j for output row h0 + 2 j
            const int r = h0 - 3 + 2 * m;
            const bool rok = r >= 0 && r < p.H;
            float win[12];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = w0 - 4 + 4 * j;
                const bool ok = rok && c >= 0 && c < p.W;
                const float4 v = *reinterpret_cast<const float4*>(x + (ok ? (size_t)r * p.W + c : 0));
                win[4 * j + 0] = ok ? v.x : 0.f; win[4 * j + 1] = ok ? v.y : 0.f; win[4 * j + 2] = ok ? v.z : 0.f; win[4 * j + 3] = ok ? v.w : 0.f;
            }
#pragma unroll
            for (int j = 0; j < NRO; ++j) {
                const int ky = m - j;
                if (ky < 0 || ky > 3) continue;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = win[i + 2 * kx + 1];                 // column w0 + i + 2kx - 3
                        const srt_v2f vv = { v, v };
                        a[j][i] = __builtin_elementwise_fma(wk[ky * 4 + kx], vv, a[j][i]);
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < NRO; ++j) {
            const int h = h0 + 2 * j;
            if (h >= p.H) continue;
            float4 o0, o1;
            if (LUT) {
                o0.x = srt_sigmoid(a[j][0].x + b0, 0); o0.y = srt_sigmoid(a[j][1].x + b0, 0); o0.z = srt_sigmoid(a[j][2].x + b0, 0); o0.w = srt_sigmoid(a[j][3].x + b0, 0);
                o1.x = srt_sigmoid(a[j][0].y + b1, 0); o1.y = srt_sigmoid(a[j][1].y + b1, 0); o1.z = srt_sigmoid(a[j][2].y + b1, 0); o1.w = srt_sigmoid(a[j][3].y + b1, 0);
            } else {
                o0.x = srt_sigmoid_fast(a[j][0].x + b0); o0.y = srt_sigmoid_fast(a[j][1].x + b0); o0.z = srt_sigmoid_fast(a[j][2].x + b0); o0.w = srt_sigmoid_fast(a[j][3].x + b0);
                o1.x = srt_sigmoid_fast(a[j][0].y + b1); o1.y = srt_sigmoid_fast(a[j][1].y + b1); o1.z = srt_sigmoid_fast(a[j][2].y + b1); o1.w = srt_sigmoid_fast(a[j][3].y + b1);
            }
            if constexpr (O16) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                _Float16* yh = reinterpret_cast<_Float16*>(p.out) + stem * p.out_stem + tile * p.out_tile;
                const h4 q0 = { (_Float16)o0.x, (_Float16)o0.y, (_Float16)o0.z, (_Float16)o0.w }, q1 = { (_Float16)o1.x, (_Float16)o1.y, (_Float16)o1.z, (_Float16)o1.w };
                *reinterpret_cast<h4*>(yh + (size_t)h * p.W + w0) = q0;
                *reinterpret_cast<h4*>(yh + hw + (size_t)h * p.W + w0) = q1;
            } else {
                *reinterpret_cast<float4*>(y + (size_t)h * p.W + w0) = o0;
                *reinterpret_cast<float4*>(y + hw + (size_t)h * p.W + w0) = o1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- up6 (Cout = 1)
// A 1-channel transposed conv has no M dimension for the parity-class form, so this layer uses the GEMM form
// on the matrix cores: col[tap][pix] = sum_ci w[ci][tap] * x[ci][pix]  (M = 25 taps padded to 32, K = Cin, N = pixels
// of the input tile + 1-pixel halo), the 25 x N result goes to LDS only, and each input pixel's 2x2 output quad then
// GATHERS its taps from LDS (no scatter, no atomics) with bias -> act -> BN fused.  B operands come straight from
// global memory (every element feeds exactly one MFMA, so LDS staging would buy nothing).
// IN16: the two source tensors hold halves (fp16 activation storage).  The contraction then runs on v_mfma_f32_32x32x16_f16
// (two MFMAs per 32-pixel sub-tile instead of sixteen fp32 ones: the 0.30 ms of matrix-pipe time this layer does not overlap
// with its loads drops to 0.02 ms); the fp16 values are used as they are, the weights are rounded to fp16 (exact for the
// reference's fp16 model container), accumulation and the epilogue stay fp32.
typedef _Float16 srt_h8 __attribute__((ext_vector_type(8)));
template <int TH, int TW, int CIN, bool IN16 = false, int OCC = 2>     // OCC: workgroups per CU the register budget is held to (the LDS tile must fit as often)
__global__ void __launch_bounds__(256, OCC) srt_up6_kernel(const SrtConvParams p)
{
    static_assert(!IN16 || CIN == 32, "the fp16 form is written for two 16-channel k-groups");
    constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW, NSUB = (NPIX + 31) / 32, NPAD = NSUB * 32;
    __shared__ float s_col[25 * NPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    // 1-D launch in XCD order (srt_device.h): an XCD walks a contiguous run of (instance, tile row, tile column), so the
    // tiles that share halo rows / the cache lines straddling a tile edge are in flight on the SAME L2.  With the plain
    // 3-D grid every neighbour sat on another XCD and the halo was fetched from HBM once per tile: 2.16x the input bytes.
    const int tilesX = (p.W + TW - 1) / TW, nsp = tilesX * ((p.H + TH - 1) / TH);
    const int pos = srt_xcd_order(nsp * p.nstems * p.ntiles), sp = pos % nsp, inst = pos / nsp;
    const int tx0 = (sp % tilesX) * TW, ty0 = (sp / tilesX) * TH;
    const int stem = inst / p.ntiles, tile = inst % p.ntiles;
    const SrtAct actp = srt_act_params(srt_act_kind(p, stem), p.variant);
    const size_t hw = (size_t)p.H * p.W;
    const float* w = p.wraw + stem * p.coeff_stem;               // [Cin][1][25]
    // fp32 form: k-pair = channels (2cp, 2cp+1); fp16 form: k-group kg = channels 16kg..16kg+15, lane half g holds 8 of them
    float a[IN16 ? 1 : CIN / 2];
    srt_h8 a16[IN16 ? 2 : 1];
    if (IN16) {
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float v = w[(kg * 16 + half * 8 + q) * 25 + min(l31, 24)];
                a16[kg][q] = (_Float16)(l31 < 25 ? v : 0.0f);
            }
    } else {
#pragma unroll
        for (int cp = 0; cp < CIN / 2; ++cp) {
            const float v = w[(2 * cp + half) * 25 + min(l31, 24)];
            a[cp] = l31 < 25 ? v : 0.0f;
        }
    }
    // A wave owns sub-tiles wave, wave+4, ...: the B fragments of ALL of them are requested up front (NW x 16 registers), so
    // the HBM latency is paid once per workgroup instead of once per sub-tile.
    constexpr int NW = (NSUB + 3) / 4;
    float b[IN16 ? 1 : NW][CIN / 2];
    srt_h8 b16[IN16 ? NW : 1][2];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int sub = wave + 4 * i;
        const int pix = sub * 32 + l31, pr = pix / PW, pc = pix % PW;
        const int gy = ty0 + pr - 1, gx = tx0 + pc - 1;
        const bool ok = sub < NSUB && pix < NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const size_t off = ok ? (size_t)gy * p.W + gx : 0;
        if (IN16) {
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) {
                if (kg == 1 ? p.c8srcB : p.c8srcA) {          // the tensor is stored C8 (srt_nn5.hip): the lane's eight channels of the pixel are one 16-byte slot of group `half`
                    const _Float16* cb = kg == 1 ? reinterpret_cast<const _Float16*>(p.srcB) + stem * p.srcB_stem + tile * p.srcB_tile : reinterpret_cast<const _Float16*>(p.srcA) + stem * p.srcA_stem + tile * p.srcA_tile;
                    const srt_h8 v = *reinterpret_cast<const srt_h8*>(cb + ((size_t)half * hw + off) * 8);
#pragma unroll
                    for (int q = 0; q < 8; ++q) b16[i][kg][q] = ok ? v[q] : (_Float16)0.0f;
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const _Float16 v = srt_src_channel_t<_Float16>(p, stem, tile, kg * 16 + half * 8 + q, hw)[off];
                    b16[i][kg][q] = ok ? v : (_Float16)0.0f;
                }
            }
        } else {
#pragma unroll
            for (int cp = 0; cp < CIN / 2; ++cp) {
                const float v = srt_src_channel(p, stem, tile, 2 * cp + half, hw)[off];
                b[IN16 ? 0 : i][cp] = ok ? v : 0.0f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int sub = wave + 4 * i;
        if (sub < NSUB) {                                         // wave-uniform
            const int pix = sub * 32 + l31;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            if (IN16) {
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[kg], b16[IN16 ? i : 0][kg], acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int cp = 0; cp < CIN / 2; ++cp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp], b[IN16 ? 0 : i][cp], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tap = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (tap < 25) s_col[tap * NPAD + pix] = acc[r];
            }
        }
    }
    __syncthreads();
    const float bi = p.bias[stem * p.coeff_stem], sc = p.bnScale[stem * p.coeff_stem], sf = p.bnShift[stem * p.coeff_stem];
    const int Wo = p.W << 1;
    float* out = p.outAct + stem * p.out_stem + tile * p.out_tile;
    for (int q = tid; q < TH * TW; q += 256) {
        const int a0 = q / TW, b0 = q % TW;
        float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {                      // ascending (ky,kx): the reference's col2im order
            const int ky = tap / 5, kx = tap % 5;
            const int py = (ky + 1) & 1, px = (kx + 1) & 1;
            const int dy = (py + 1 - ky) / 2, dx = (px + 1 - kx) / 2;
            o[py * 2 + px] += s_col[tap * NPAD + (a0 + 1 + dy) * PW + (b0 + 1 + dx)];
        }
        const int ga = ty0 + a0, gb = tx0 + b0;
        if (ga < p.H && gb < p.W) {
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                float2 v;
                v.x = srt_dec_epilogue(o[py * 2 + 0], bi, sc, sf, actp);
                v.y = srt_dec_epilogue(o[py * 2 + 1], bi, sc, sf, actp);
                *reinterpret_cast<float2*>(out + (size_t)(2 * ga + py) * Wo + 2 * gb) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- up6, streamed down a tile column
// The kernel above asks HBM for its input one dword per lane, 320 row pieces of 264 bytes per workgroup: those loads alone take 0.62 ms of its
// 0.74 (scripts/ubench/tile_reads.hip); the same bytes as aligned 16-byte row segments take 0.43.  This form moves the input by LDS-DMA only
// (buffer_load_dwordx4 ... lds: 16 bytes per lane, zeros outside the image, no registers in flight) and overlaps it with the arithmetic by
// construction: a workgroup owns a TW-pixel wide column of one instance and walks down it CR input rows at a time,
//   interval i:   DMA   rows of chunk i+1 (32 channels x CR rows x (TW+8) floats)      -> patch buffer (i+1) & 1
//                 MFMA  chunk i: col[tap][pixel] = sum_ci w[ci][tap] x[ci][pixel]      -> ring of 3 CR rows of tap planes in LDS
//                 gather the 2x2 output quads of input rows CR(i-1)-1 .. CR i-2 from the ring (their three tap rows are complete), epilogue, store
// with ONE barrier per interval.  No input row is fetched twice by a workgroup (no y halo), the x halo is the 16 bytes either side that sit in
// the neighbour column's lines, read by the neighbouring workgroup at the same time on the same XCD.  Sums in the order of the kernel above
// (channels ascending in the MFMA chain, taps ascending in the gather): bit-identical results.
typedef int srt_i32x4 __attribute__((ext_vector_type(4)));
// H16 (fp16 activation storage): the two source tensors hold halves - 16-byte DMA pieces are 8 pixels, a patch row is tx0 - 8 .. tx0 + 71 - and the
// contraction is the tiled fp16 kernel's (srt_up6_kernel<.., true>: two v_mfma_f32_32x32x16_f16 per 32 pixels, weights rounded to fp16, the same
// operands in the same order): bit-identical to it, half the input bytes.
// C8M (H16 only, round 6; bit 0: srcA - down1's raw skip tensor, bit 1: srcB - up5's output): the tensor is stored C8 (SrtConvParams::c8srcA / c8srcB): its half of a
// chunk lands as 16-byte pixel slots and its MFMA's B fragment is one ds_read_b128 instead of eight 2-byte reads and their packing.  Same values into the same MFMAs:
// bit-identical to the planar form.
template <int TW, int CR, int ABL = 0, bool H16 = false, int C8M = 0>       // ABL (tuning builds, wrong results): 1 no DMA, 2 DMA only, 3 no gather
__global__ void __launch_bounds__(512, 2) srt_up6_stream_kernel(const SrtConvParams p)
{
    static_assert(C8M == 0 || H16, "C8 tensors hold halves");
    constexpr bool AC8 = (C8M & 1) != 0, BC8 = (C8M & 2) != 0;
    constexpr int CIN = 32, CAH = 16;
    constexpr int EPP = H16 ? 8 : 4, ESZ = H16 ? 2 : 4, LEAD = EPP - 1;      // elements per 16-byte DMA piece; bytes per element; patch column of pixel tx0 - 1
    constexpr int PW = TW + 2, SEG = TW / EPP + 2, PROW = SEG * EPP;         // pixels of a chunk row incl. halo; 16-byte segments per row (from tx0 - EPP)
    constexpr int CHF4 = CR * SEG, NF4 = CIN * CHF4, NPIECE = NF4 / 64, NWAVE = 8, NDW = 4, PPW = (NPIECE + NDW - 1) / NDW;
    static_assert(NF4 % 64 == 0 && (CAH * CHF4) % 64 == 0, "a DMA piece (64 lanes x 16 B) must not straddle the two source tensors");
    static_assert(TW == 64 && 2 * CR <= NWAVE, "gather: a wave per (row of the chunk, output row parity), a lane per pixel");
    constexpr int PBUF = CIN * CR * PROW;                                     // elements
    constexpr int PBUF_F = PBUF * ESZ / 4;                                    // ... in floats of the LDS array
    constexpr int RR = 2 * CR + 2, RWP = 72;                                  // ring rows; row pitch of a tap plane (4 * RWP = 32 mod 64 banks: the two lane halves write disjoint banks)
    static_assert(RWP >= PW, "ring pitch");
    constexpr int NPIX = CR * PW, NG = (NPIX + 31) / 32;
    static_assert(NG <= NWAVE && 2 * CR == NWAVE - NDW, "one pixel group per wave; waves NDW.. gather");
    __shared__ __attribute__((aligned(16))) float s_all[2 * PBUF_F + RR * 25 * RWP];
    float* s_p = s_all;
    float* s_r = s_all + 2 * PBUF_F;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesX = (p.W + TW - 1) / TW;
    const int pos = srt_xcd_order(tilesX * p.nstems * p.ntiles), tx0 = (pos % tilesX) * TW, inst = pos / tilesX;
    const int stem = inst / p.ntiles, tile = inst % p.ntiles;
    const SrtAct actp = srt_act_params(srt_act_kind(p, stem), p.variant);
    const size_t hw = (size_t)p.H * p.W;
    const float* w = p.wraw + stem * p.coeff_stem;                            // [Cin][1][25]
    float a[H16 ? 1 : CIN / 2];
    srt_h8 a16[H16 ? 2 : 1];
    if constexpr (H16) {
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float v = w[(kg * 16 + half * 8 + q) * 25 + min(l31, 24)];
                a16[kg][q] = (_Float16)(l31 < 25 ? v : 0.0f);
            }
    } else {
#pragma unroll
        for (int cp = 0; cp < CIN / 2; ++cp) {
            const float v = w[(2 * cp + half) * 25 + min(l31, 24)];
            a[cp] = l31 < 25 ? v : 0.0f;
        }
    }
    // ---- DMA: float4 e = (ch * CR + row) * SEG + j of a chunk <- channel ch, image row CR i + row, columns tx0 - 4 + 4 j .. + 3; wave w moves pieces w, w + 4, ...
    constexpr unsigned OOR = 0x80000000u;
    unsigned c0[PPW]; int prow[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int piece = min(wave + NDW * q, NPIECE - 1), e = piece * 64 + lane;
        const int j = e % SEG, row = (e / SEG) % CR, chl = (e / CHF4) % CAH, gx = tx0 - EPP + EPP * j;
        prow[q] = row;
        c0[q] = (gx >= 0 && gx + EPP - 1 < p.W) ? (unsigned)ESZ * (unsigned)((size_t)chl * hw + (size_t)row * p.W + gx) : OOR;
        if constexpr (C8M != 0) {
            // a source tensor in C8: its half of a chunk is [channel group 2][row CR][pixel PROW] slots of 16 bytes (the same 5 pieces), slot <- 8 channels of one pixel
            if (piece >= NPIECE / 2 ? BC8 : AC8) {
                const int e2 = e % ((NPIECE / 2) * 64), px = e2 % PROW, row2 = (e2 / PROW) % CR, gq = e2 / (PROW * CR), gxp = tx0 - EPP + px;
                prow[q] = row2;
                c0[q] = (gxp >= 0 && gxp < p.W) ? 16u * (unsigned)((size_t)gq * hw + (size_t)row2 * p.W + gxp) : OOR;
            }
        }
    }
    const size_t ba_ = (size_t)p.srcA + (size_t)ESZ * (stem * p.srcA_stem + tile * p.srcA_tile), bb_ = (size_t)p.srcB + (size_t)ESZ * (stem * p.srcB_stem + tile * p.srcB_tile);
    srt_i32x4 rsA, rsB;
    rsA.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ba_); rsA.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba_ >> 32) & 0xffffu));
    rsB.x = __builtin_amdgcn_readfirstlane((int)(unsigned)bb_); rsB.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(bb_ >> 32) & 0xffffu));
    rsA.z = rsB.z = (int)(unsigned)min((size_t)0x7fffffff, (size_t)ESZ * CAH * hw); rsA.w = rsB.w = 0x00020000;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)s_all;
    auto dma_chunk = [&](int i) {
        const unsigned adv = (unsigned)ESZ * (unsigned)(i * CR * p.W), base = lds0 + (unsigned)((i & 1) * PBUF * ESZ);
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int piece = min(wave + NDW * q, NPIECE - 1);                // wave-uniform (a piece past the last one repeats it)
            const unsigned advq = (piece >= NPIECE / 2 ? BC8 : AC8) ? 16u * (unsigned)(i * CR * p.W) : adv;      // C8 rows are 16 bytes per pixel
            const unsigned voff = (c0[q] != OOR && i * CR + prow[q] < p.H) ? c0[q] + advq : OOR;
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)(piece * 1024));
            if (piece < NPIECE / 2) asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsA), "s"(dst) : "memory");
            else asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsB), "s"(dst) : "memory");
        }
    };
    for (int e = tid; e < RR * 25 * RWP; e += 64 * NWAVE) s_r[e] = 0.0f;          // rows above the image (and every slot before its first use)
    const int nchunks = (p.H + CR - 1) / CR + 1;                             // the last chunk lies below the image: zeros, the bottom halo row
    if (ABL != 1 && wave < NDW) dma_chunk(0);
    const float bi = p.bias[stem * p.coeff_stem], sc = p.bnScale[stem * p.coeff_stem], sf = p.bnShift[stem * p.coeff_stem];
    const int Wo = p.W << 1;
    float* out = p.outAct + stem * p.out_stem + tile * p.out_tile;
    int slot0 = 0;                                                           // ring slot of image row CR i
    for (int i = 0; i <= nchunks; ++i) {
        if (wave < NDW) __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's pieces of chunk i.  (The gather waves never wait for their stores.)
        __syncthreads();
        if (ABL != 1 && wave < NDW && i + 1 < nchunks) dma_chunk(i + 1);
        const int grp = wave < NDW ? wave : NDW + ((wave - i) & (NWAVE - NDW - 1));   // groups NDW.. go round the gather waves (one SIMD each) interval by interval
        if (ABL != 2 && i < nchunks && grp < NG) {                            // wave-uniform
            const int pix = min(grp * 32 + l31, NPIX - 1), row = pix / PW, col = pix % PW;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            if constexpr (H16) {
                // k-group kg = channels 16 kg .. 16 kg + 15; lane half holds 8 of them for its pixel (channel-major rows in LDS: 8 single reads)
                const _Float16* bsrc = reinterpret_cast<const _Float16*>(s_p) + (i & 1) * PBUF + (half * 8 * CR + row) * PROW + col + LEAD;
                srt_h8 b16[2];
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) {
                    if (kg == 0 ? AC8 : BC8) {                                  // the tensor's eight channels of the pixel: ONE aligned 16-byte slot of channel group `half`
                        b16[kg] = *reinterpret_cast<const srt_h8*>(reinterpret_cast<const _Float16*>(s_p) + (i & 1) * PBUF + kg * (PBUF / 2) + ((half * CR + row) * PROW + col + LEAD) * 8);
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) b16[kg][q] = bsrc[(kg * 16 + q) * CR * PROW];
                    }
                }
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[kg], b16[kg], acc, 0, 0, 0);
            } else {
                const float* bsrc = s_p + (i & 1) * PBUF + (half * CR + row) * PROW + col + LEAD;
                float b[CIN / 2];
#pragma unroll
                for (int cp = 0; cp < CIN / 2; ++cp) b[cp] = bsrc[cp * 2 * CR * PROW];
#pragma unroll
                for (int cp = 0; cp < CIN / 2; ++cp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp], b[cp], acc, 0, 0, 0);
            }
            int slot = slot0 + row; slot = slot >= RR ? slot - RR : slot;
            if (grp * 32 + l31 < NPIX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tap = (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (tap < 25) s_r[(slot * 25 + tap) * RWP + col] = acc[r];
                }
            }
        }
        // gather: a wave per (row of the chunk, output row parity), a lane per pixel
        const int gw = wave - NDW;
        const int b0 = lane, a0 = gw % CR, py = gw / CR;
        const int g = CR * (i - 1) - 1 + a0, gb = tx0 + b0;                  // input row whose output row 2 g + py this thread emits
        auto gather = [&](auto pyc) __attribute__((always_inline)) {
            constexpr int PY = decltype(pyc)::value;
            int sm = slot0 - CR - 2 + a0; sm = sm < 0 ? sm + RR : sm;        // ring slots of rows g - 1, g, g + 1
            const int s1 = sm + 1 >= RR ? sm + 1 - RR : sm + 1, s2 = s1 + 1 >= RR ? s1 + 1 - RR : s1 + 1;
            const float* r0 = s_r + sm * 25 * RWP + b0 + 1;
            const float* r1 = s_r + s1 * 25 * RWP + b0 + 1;
            const float* r2 = s_r + s2 * 25 * RWP + b0 + 1;
            float o[2] = {0.0f, 0.0f};
#pragma unroll
            for (int tap = 0; tap < 25; ++tap) {                              // ascending (ky, kx): the reference's col2im order
                const int ky = tap / 5, kx = tap % 5;
                if (((ky + 1) & 1) != PY) continue;
                const int px = (kx + 1) & 1, dy = (PY + 1 - ky) / 2, dx = (px + 1 - kx) / 2;
                o[px] += (dy < 0 ? r0 : dy == 0 ? r1 : r2)[tap * RWP + dx];
            }
            float2 v;
            v.x = srt_dec_epilogue(o[0], bi, sc, sf, actp);
            v.y = srt_dec_epilogue(o[1], bi, sc, sf, actp);
            *reinterpret_cast<float2*>(out + (size_t)(2 * g + PY) * Wo + 2 * gb) = v;
        };
        if (ABL < 2 && gw >= 0 && g >= 0 && g < p.H && gb < p.W) {
            if (py) gather(std::integral_constant<int, 1>{}); else gather(std::integral_constant<int, 0>{});
        }
        slot0 += CR; slot0 = slot0 >= RR ? slot0 - RR : slot0;
    }
}

// ------------------------------------------------------------------------------------------- up6 + head in one pass down a tile column
// The reference runs up6 and the head in place over one buffer (spleeter.c:285-300); srt_up6_stream_kernel + srt_head_rows_kernel write the 1-channel plane
// (1 MiB per instance) and read it back.  Here the column workgroup of the streamed up6 keeps its output rows in an LDS ring and emits the two mask planes itself:
//   * the head's taps reach three output columns either side of the column, i.e. the up6 outputs of TWO more input pixels per side: the tap planes are computed for
//     TW + 6 patch columns instead of TW + 2 - columns the 16-byte DMA segments already bring (tx0 - EPP .. tx0 + TW + EPP - 1) and the fifth 32-pixel MFMA group
//     already has room for (2 x 70 = 140 <= 160 pixels), so the halo costs no load and no MFMA, only a second gather pass of four lanes;
//   * the gather waves store their quads into a ring of 16 up6 output rows (136 columns) instead of global memory, and each of the four DMA / MFMA waves then evaluates
//     the head for 64 columns of two same-parity output rows (they share three of their five input rows: 20 LDS reads for 4 mask values), lagging the gather by one
//     interval: interval i gathers rows 4i-6..4i-3 and emits head rows 4i-13..4i-10 from rows 4i-16..4i-7 - fourteen live rows, one barrier per interval as before.
// Rows / columns outside the image are never read from the ring: the head predicates on coordinates (and, like srt_head_rows_kernel, still issues the FMA with a zero).
// Same MFMA chains, same gather order, same FMA chain per mask value ((ky, kx) ascending on both channels at once) as the two kernels it replaces: bit-identical masks.
// LDS: fp32 88.8 KB (ONE workgroup per CU), fp16 storage 72.4 KB (two).  The up6 plane is not written; srtCopyTensor("up6") re-runs the plain up6 launch on demand.
// MEASURED SLOWER than the two kernels in both modes (profiles/r06_fuse_pmc.json) and therefore OFF unless SPLEETERRT_FUSE_HEAD=1: the head's 181 M
// VALU instructions are free in a kernel of its own (bandwidth-bound, idle vector ALU) and are not inside a column workgroup whose interval is a chain of dependent
// LDS round trips behind one barrier; and with the intervals twice as long the columns of an XCD drift apart, so the halo lines neighbours share stop hitting in L2
// (fp16 storage: 2.23 GB fetched against 1.45).
template <int TW, int CR, bool H16, bool LUT>
__global__ void __launch_bounds__(512, H16 ? 4 : 2) srt_up6_head_kernel(const SrtConvParams p, const SrtHeadParams hp)
{
    constexpr int CIN = 32, CAH = 16, HALO = 2;
    constexpr int EPP = H16 ? 8 : 4, ESZ = H16 ? 2 : 4, LEAD = EPP - 1 - HALO;   // patch column of pixel tx0 - 1 - HALO = tap-plane column 0
    constexpr int PW = TW + 2 + 2 * HALO, SEG = TW / EPP + 2, PROW = SEG * EPP;
    constexpr int CHF4 = CR * SEG, NF4 = CIN * CHF4, NPIECE = NF4 / 64, NWAVE = 8, NDW = 4, PPW = (NPIECE + NDW - 1) / NDW;
    static_assert(NF4 % 64 == 0 && (CAH * CHF4) % 64 == 0, "a DMA piece (64 lanes x 16 B) must not straddle the two source tensors");
    static_assert(TW == 64 && CR == 2 && 2 * CR == NWAVE - NDW, "gather: a wave per (row of the chunk, output row parity); head: a wave per (64-column half, row parity)");
    static_assert(LEAD >= 0 && LEAD + PW <= PROW, "the halo columns lie inside the DMA'd row");
    constexpr int PBUF = CIN * CR * PROW, PBUF_F = PBUF * ESZ / 4;
    constexpr int RR = 2 * CR + 2, RWP = 72;
    static_assert(RWP >= PW, "ring pitch");
    constexpr int NPIX = CR * PW, NG = (NPIX + 31) / 32;
    static_assert(NG <= NWAVE, "one pixel group per wave");
    constexpr int OR = 16, OP = 2 * (TW + 2 * HALO);                         // ring of up6 output rows: slot = row & 15; column 0 = output column 2 (tx0 - HALO)
    __shared__ __attribute__((aligned(16))) float s_all[2 * PBUF_F + RR * 25 * RWP + OR * OP];
    float* s_p = s_all;
    float* s_r = s_all + 2 * PBUF_F;
    float* s_o = s_r + RR * 25 * RWP;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesX = (p.W + TW - 1) / TW;
    const int pos = srt_xcd_order(tilesX * p.nstems * p.ntiles), tx0 = (pos % tilesX) * TW, inst = pos / tilesX;
    const int stem = inst / p.ntiles, tile = inst % p.ntiles;
    const SrtAct actp = srt_act_params(srt_act_kind(p, stem), p.variant);
    const size_t hw = (size_t)p.H * p.W;
    const float* w = p.wraw + stem * p.coeff_stem;                            // [Cin][1][25]
    float a[H16 ? 1 : CIN / 2];
    srt_h8 a16[H16 ? 2 : 1];
    if constexpr (H16) {                                                      // (every wave may run an MFMA group: group NDW goes round the gather waves)
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float v = w[(kg * 16 + half * 8 + q) * 25 + min(l31, 24)];
                a16[kg][q] = (_Float16)(l31 < 25 ? v : 0.0f);
            }
    } else {
#pragma unroll
        for (int cp = 0; cp < CIN / 2; ++cp) {
            const float v = w[(2 * cp + half) * 25 + min(l31, 24)];
            a[cp] = l31 < 25 ? v : 0.0f;
        }
    }
    constexpr unsigned OOR = 0x80000000u;
    unsigned c0[PPW]; int prow[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int piece = min(wave + NDW * q, NPIECE - 1), e = piece * 64 + lane;
        const int j = e % SEG, row = (e / SEG) % CR, chl = (e / CHF4) % CAH, gx = tx0 - EPP + EPP * j;
        prow[q] = row;
        c0[q] = (gx >= 0 && gx + EPP - 1 < p.W) ? (unsigned)ESZ * (unsigned)((size_t)chl * hw + (size_t)row * p.W + gx) : OOR;
    }
    const size_t ba_ = (size_t)p.srcA + (size_t)ESZ * (stem * p.srcA_stem + tile * p.srcA_tile), bb_ = (size_t)p.srcB + (size_t)ESZ * (stem * p.srcB_stem + tile * p.srcB_tile);
    srt_i32x4 rsA, rsB;
    rsA.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ba_); rsA.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba_ >> 32) & 0xffffu));
    rsB.x = __builtin_amdgcn_readfirstlane((int)(unsigned)bb_); rsB.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(bb_ >> 32) & 0xffffu));
    rsA.z = rsB.z = (int)(unsigned)min((size_t)0x7fffffff, (size_t)ESZ * CAH * hw); rsA.w = rsB.w = 0x00020000;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)s_all;
    auto dma_chunk = [&](int i) {
        const unsigned adv = (unsigned)ESZ * (unsigned)(i * CR * p.W), base = lds0 + (unsigned)((i & 1) * PBUF * ESZ);
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int piece = min(wave + NDW * q, NPIECE - 1);
            const unsigned voff = (c0[q] != OOR && i * CR + prow[q] < p.H) ? c0[q] + adv : OOR;
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)(piece * 1024));
            if (piece < NPIECE / 2) asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsA), "s"(dst) : "memory");
            else asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsB), "s"(dst) : "memory");
        }
    };
    for (int e = tid; e < RR * 25 * RWP; e += 64 * NWAVE) s_r[e] = 0.0f;
    const int nchunks = (p.H + CR - 1) / CR + 1;
    if (wave < NDW) dma_chunk(0);
    const float bi = p.bias[stem * p.coeff_stem], sc = p.bnScale[stem * p.coeff_stem], sf = p.bnShift[stem * p.coeff_stem];
    // ---- head constants (gather waves): (channel 0, channel 1) weight of each tap
    const int Ho = p.H << 1, Wo = p.W << 1;
    const size_t ohw = (size_t)Ho * Wo;
    srt_v2f wk[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { wk[t].x = hp.w[stem * hp.coeff_stem + t]; wk[t].y = hp.w[stem * hp.coeff_stem + 16 + t]; }
    const float hb0 = hp.bias[stem * hp.coeff_stem], hb1 = hp.bias[stem * hp.coeff_stem + 1];
    float* y = hp.out + stem * hp.out_stem + tile * hp.out_tile;
    int slot0 = 0;
    for (int i = 0; i <= nchunks + 2; ++i) {                                 // two more intervals than the plain kernel: the head lags the gather
        if (wave < NDW) __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (wave < NDW && i + 1 < nchunks) dma_chunk(i + 1);
        const int grp = wave < NDW ? wave : NDW + ((wave - i) & (NWAVE - NDW - 1));
        if (i < nchunks && grp < NG) {
            const int pix = min(grp * 32 + l31, NPIX - 1), row = pix / PW, col = pix % PW;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            if constexpr (H16) {
                const _Float16* bsrc = reinterpret_cast<const _Float16*>(s_p) + (i & 1) * PBUF + (half * 8 * CR + row) * PROW + col + LEAD;
                srt_h8 b16[2];
#pragma unroll
                for (int kg = 0; kg < 2; ++kg)
#pragma unroll
                    for (int q = 0; q < 8; ++q) b16[kg][q] = bsrc[(kg * 16 + q) * CR * PROW];
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[kg], b16[kg], acc, 0, 0, 0);
            } else {
                const float* bsrc = s_p + (i & 1) * PBUF + (half * CR + row) * PROW + col + LEAD;
                float b[CIN / 2];
#pragma unroll
                for (int cp = 0; cp < CIN / 2; ++cp) b[cp] = bsrc[cp * 2 * CR * PROW];
#pragma unroll
                for (int cp = 0; cp < CIN / 2; ++cp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp], b[cp], acc, 0, 0, 0);
            }
            int slot = slot0 + row; slot = slot >= RR ? slot - RR : slot;
            if (grp * 32 + l31 < NPIX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tap = (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (tap < 25) s_r[(slot * 25 + tap) * RWP + col] = acc[r];
                }
            }
        }
        const int gw = wave - NDW;
        if (gw >= 0) {                                                        // wave-uniform
            // ---- gather: input row g = CR (i - 1) - 1 + a0, output row 2 g + py, pixel tx0 - HALO + q -> ring of up6 output rows
            auto gather = [&](auto pyc, int a0, int q) __attribute__((always_inline)) {
                constexpr int PY = decltype(pyc)::value;
                const int g = CR * (i - 1) - 1 + a0;
                int sm = slot0 - CR - 2 + a0; sm = sm < 0 ? sm + RR : sm;
                const int s1 = sm + 1 >= RR ? sm + 1 - RR : sm + 1, s2 = s1 + 1 >= RR ? s1 + 1 - RR : s1 + 1;
                const float* r0 = s_r + sm * 25 * RWP + q + 1;
                const float* r1 = s_r + s1 * 25 * RWP + q + 1;
                const float* r2 = s_r + s2 * 25 * RWP + q + 1;
                float o[2] = {0.0f, 0.0f};
#pragma unroll
                for (int tap = 0; tap < 25; ++tap) {                          // ascending (ky, kx): the reference's col2im order
                    const int ky = tap / 5, kx = tap % 5;
                    if (((ky + 1) & 1) != PY) continue;
                    const int px = (kx + 1) & 1, dy = (PY + 1 - ky) / 2, dx = (px + 1 - kx) / 2;
                    o[px] += (dy < 0 ? r0 : dy == 0 ? r1 : r2)[tap * RWP + dx];
                }
                float2 v;
                v.x = srt_dec_epilogue(o[0], bi, sc, sf, actp);
                v.y = srt_dec_epilogue(o[1], bi, sc, sf, actp);
                *reinterpret_cast<float2*>(s_o + ((2 * g + PY) & (OR - 1)) * OP + 2 * q) = v;
            };
            {                                                                 // the wave's (a0, py), pixels tx0 - HALO + lane
                const int a0 = gw % CR, py = gw / CR, g = CR * (i - 1) - 1 + a0, gb = tx0 - HALO + lane;
                if (g >= 0 && g < p.H && gb >= 0 && gb < p.W) {
                    if (py) gather(std::integral_constant<int, 1>{}, a0, lane); else gather(std::integral_constant<int, 0>{}, a0, lane);
                }
            }
            if (gw == ((i + 2) & (NWAVE - NDW - 1))) {                        // the last 2 HALO pixels of all four (a0, py): sixteen lanes of ONE wave (not the one that runs MFMA group NDW this interval)
                const int q = TW + (lane & 3), a0 = (lane >> 3) & 1, py = (lane >> 2) & 1, g = CR * (i - 1) - 1 + a0, gb = tx0 - HALO + q;
                const bool ok = lane < 16 && g >= 0 && g < p.H && gb >= 0 && gb < p.W;
                if (ok && py == 0) gather(std::integral_constant<int, 0>{}, a0, q);
                if (ok && py == 1) gather(std::integral_constant<int, 1>{}, a0, q);
            }
        } else {
            // ---- head (the DMA / MFMA waves: two instructions' worth of MFMA work in the fp16 form): output rows h, h + 2 (h = 4 i - 13 + parity) of columns
            //      2 tx0 + 64 hh + lane, from ring rows h - 3, h - 1, .. h + 5 (written in earlier intervals)
            const int hh = wave & 1, h = 4 * i - 13 + (wave >> 1);
            const int wcol = 2 * tx0 + 64 * hh + lane;
            if (h + 2 >= 0 && h < Ho && wcol < Wo) {
                srt_v2f ac[2];
                ac[0].x = ac[0].y = ac[1].x = ac[1].y = 0.0f;
                const float* oc = s_o + 64 * hh + lane + 2 * HALO - 3;        // ring column of output column wcol - 3
                const bool inside = h - 3 >= 0 && h + 5 < Ho && 2 * tx0 + 64 * hh - 3 >= 0 && 2 * tx0 + 64 * hh + 66 < Wo;   // wave-uniform: no tap of the wave leaves the image
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    const int r = h - 3 + 2 * m;
                    const bool rok = r >= 0 && r < Ho;
                    const float* orow = oc + (r & (OR - 1)) * OP;
                    float win[4];
                    if (inside) {
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) win[kx] = orow[2 * kx];
                    } else {
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) {
                            const int c = wcol - 3 + 2 * kx;
                            const float v = orow[2 * kx];
                            win[kx] = (rok && c >= 0 && c < Wo) ? v : 0.0f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ky = m - j;
                        if (ky < 0 || ky > 3) continue;
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) {
                            const srt_v2f vv = { win[kx], win[kx] };
                            ac[j] = __builtin_elementwise_fma(wk[ky * 4 + kx], vv, ac[j]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int hr = h + 2 * j;
                    if (hr < 0 || hr >= Ho) continue;
                    float m0, m1;
                    if (LUT) { m0 = srt_sigmoid(ac[j].x + hb0, 0); m1 = srt_sigmoid(ac[j].y + hb1, 0); }
                    else { m0 = srt_sigmoid_fast(ac[j].x + hb0); m1 = srt_sigmoid_fast(ac[j].y + hb1); }
                    y[(size_t)hr * Wo + wcol] = m0;
                    y[ohw + (size_t)hr * Wo + wcol] = m1;
                }
            }
        }
        slot0 += CR; slot0 = slot0 >= RR ? slot0 - RR : slot0;
    }
}

// ------------------------------------------------------------------------------------------- dispatch
template <int BM, int WM, int SW, int NSX, int NSY, int NI, int KC>
static int launch_enc_cfg(const SrtConvParams& p, hipStream_t s)
{
    constexpr int SH = 32 / SW, TW = NSX * SW, TH = NSY * SH;
    const int Ho = p.H / 2, Wo = p.W / 2;
    dim3 grid(((Wo + TW - 1) / TW) * ((Ho + TH - 1) / TH), (p.Cout + BM - 1) / BM, p.nstems * ((p.ntiles + NI - 1) / NI));
    SRT_LAUNCH((srt_enc_mfma<BM, WM, SW, NSX, NSY, NI, KC>), grid, dim3(256), 0, s, p);
    return srt_launch_status();
}
template <int BM, int WM, int SW, int NSX, int NSY, int NI, int KC>
static int launch_dec_cfg(const SrtConvParams& p, hipStream_t s)
{
    constexpr int SH = 32 / SW, TW = NSX * SW, TH = NSY * SH;
    dim3 grid(((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH), (p.Cout + BM - 1) / BM, p.nstems * ((p.ntiles + NI - 1) / NI));
    SRT_LAUNCH((srt_dec_mfma<BM, WM, SW, NSX, NSY, NI, KC>), grid, dim3(256), 0, s, p);
    return srt_launch_status();
}

static int launch_naive(void (*k)(const SrtConvParams), const SrtConvParams& p, size_t total, hipStream_t s)
{
    size_t bx = (total + 255) / 256;
    if (bx > 65535) bx = 65535;
    SRT_LAUNCH(k, dim3((unsigned)bx, p.nstems * p.ntiles), dim3(256), 0, s, p);
    return srt_launch_status();
}

int srt_launch_enc(const SrtConvParams& p, int impl, hipStream_t s)
{
    if (p.in16 || p.out16) return -1;                     // the fallback kernels only know fp32 tensors (the engine keeps fp32 storage where they may run)
    const int Ho = p.H / 2, Wo = p.W / 2;
    if (impl != 0) return launch_naive(srt_enc_naive, p, (size_t)p.Cout * Ho * Wo, s);
    const int KC2 = 2;
    (void)KC2;
    if (p.Cin == 2) return launch_enc_cfg<32, 1, 32, 2, 4, 1, 2>(p, s);                     // down1: K = 50 in one chunk
    if (p.Cout <= 32) return launch_enc_cfg<32, 1, 32, 2, 4, 1, 4>(p, s);                    // down2
    if (Wo >= 64) return launch_enc_cfg<64, 2, 32, 2, 4, 1, 4>(p, s);                        // down3/down4 class
    if (Wo >= 32) return launch_enc_cfg<64, 2, 32, 1, 8, 1, 4>(p, s);                        // down5 class (one instance = 8x32)
    return launch_enc_cfg<64, 2, 16, 1, 2, 4, 4>(p, s);                                      // down6 class (4 instances of 4x16)
}

#ifndef SRT_UP6_STREAM_DEFAULT
#define SRT_UP6_STREAM_DEFAULT 1
#endif
#ifndef SRT_FUSE_HEAD_DEFAULT
#define SRT_FUSE_HEAD_DEFAULT(in16) 0       // measured slower in both storage modes (round 6: fp32 1.24 vs 0.58 + 0.16 ms, fp16 storage at five stems 0.71 vs 0.39 + 0.21)
#endif
int srt_launch_dec(const SrtConvParams& p, int impl, hipStream_t s)
{
    if ((p.in16 || p.out16) && !(impl == 0 && p.Cout == 1 && p.Cin == 32 && !p.out16)) return -1;   // only up6 reads fp16 tensors here
    if (impl != 0) return launch_naive(srt_dec_naive, p, (size_t)p.Cout * p.H * p.W * 4, s);
    if ((p.c8srcB || p.c8srcA) && !p.in16) return -1;
    if (p.Cout == 1 && p.Cin == 32) {                                                         // up6
#define UP6_LAUNCH(TH, TW) SRT_LAUNCH((srt_up6_kernel<TH, TW, 32>), dim3(((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.nstems * p.ntiles), dim3(256), 0, s, p)
        int v = 0;
#ifdef SRT_TUNING
        const char* tv = getenv("SRT_TUNE_UP6");
        v = tv ? atoi(tv) : 0;
#endif
        // fp32 tensors, batches that fill the chip with column workgroups (2 per CU): the streamed form
        const long cols = (long)((p.W + 63) / 64) * p.nstems * p.ntiles;
        if (p.CA == 16 && p.W % (p.in16 ? 8 : 4) == 0 && p.srcA && p.srcB && (size_t)64 * p.H * p.W < 0x7fffffffu && ((cols >= 512 && v == 0 && SRT_UP6_STREAM_DEFAULT) || v == 11)) {
            if (p.in16 && p.c8srcB && p.c8srcA) SRT_LAUNCH((srt_up6_stream_kernel<64, 2, 0, true, 3>), dim3((unsigned)cols), dim3(512), 0, s, p);   // ... both tensors in C8
            else if (p.in16 && p.c8srcB) SRT_LAUNCH((srt_up6_stream_kernel<64, 2, 0, true, 2>), dim3((unsigned)cols), dim3(512), 0, s, p);          // ... up5's output in C8
            else if (p.c8srcA) return -1;
            else if (p.in16) SRT_LAUNCH((srt_up6_stream_kernel<64, 2, 0, true>), dim3((unsigned)cols), dim3(512), 0, s, p);       // halves in, fp32 out (fp16 activation storage)
            else SRT_LAUNCH((srt_up6_stream_kernel<64, 2>), dim3((unsigned)cols), dim3(512), 0, s, p);
            return srt_launch_status();
        }
#ifdef SRT_TUNING
        if (v >= 12 && v <= 14) {
            if (v == 12) SRT_LAUNCH((srt_up6_stream_kernel<64, 2, 1>), dim3((unsigned)cols), dim3(512), 0, s, p);
            if (v == 13) SRT_LAUNCH((srt_up6_stream_kernel<64, 2, 2>), dim3((unsigned)cols), dim3(512), 0, s, p);
            if (v == 14) SRT_LAUNCH((srt_up6_stream_kernel<64, 2, 3>), dim3((unsigned)cols), dim3(512), 0, s, p);
            return srt_launch_status();
        }
#endif
#ifdef SRT_TUNING
        if (v == 1) UP6_LAUNCH(16, 32);
        else if (v == 5) UP6_LAUNCH(8, 32);
        else if (v == 3) UP6_LAUNCH(4, 64);
        else if (v == 4) UP6_LAUNCH(4, 128);
#define UP6_LAUNCH_OCC(TH, TW, OCC) SRT_LAUNCH((srt_up6_kernel<TH, TW, 32, false, OCC>), dim3(((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.nstems * p.ntiles), dim3(256), 0, s, p)
        else if (v == 6) UP6_LAUNCH_OCC(8, 32, 4);          // 35 KB of LDS per workgroup: four per CU
        else if (v == 7) UP6_LAUNCH_OCC(4, 64, 3);          // 42 KB: three per CU
        else if (v == 8) UP6_LAUNCH_OCC(8, 32, 3);
        else if (v == 9) UP6_LAUNCH_OCC(16, 16, 4);         // 18 x 18 = 324 pixels: 11 sub-tiles, 35 KB
#undef UP6_LAUNCH_OCC
#endif
        if (v > 5 && v < 10) {}
        else if (v == 0 && p.in16) SRT_LAUNCH((srt_up6_kernel<8, 64, 32, true>), dim3(((p.W + 63) / 64) * ((p.H + 7) / 8) * p.nstems * p.ntiles), dim3(256), 0, s, p);
        else if (v == 0 || v >= 10) UP6_LAUNCH(8, 64);                // measured (XCD order, loads up front): 16x32 0.73 ms, 8x64 0.74, 8x32 0.75, 4x128 0.84, 4x64 0.86
#undef UP6_LAUNCH
        return srt_launch_status();
    }
    if (p.Cout < 16) return launch_naive(srt_dec_naive, p, (size_t)p.Cout * p.H * p.W * 4, s);
    if (p.Cout <= 32) return launch_dec_cfg<32, 1, 32, 2, 4, 1, 4>(p, s);                    // up4/up5: 4 rows x 64 cols
    if (p.W >= 32) return launch_dec_cfg<64, 2, 32, 1, 4, 1, 4>(p, s);                       // up2/up3: 4 rows x 32 cols
    return launch_dec_cfg<64, 2, 16, 1, 2, 2, 4>(p, s);                                      // up1: 2 instances of 4x16
}

// up6 + head in one pass (srt_up6_head_kernel).  Returns 1 when the launch is not covered or the form is switched off: the caller then launches the two layers
// separately.  SPLEETERRT_FUSE_HEAD: 0 never, 1 whenever covered, unset: where it measured faster.
int srt_launch_up6_head(const SrtConvParams& p, const SrtHeadParams& h, hipStream_t s)
{
    const char* fv = getenv("SPLEETERRT_FUSE_HEAD");                          // (read per launch: the parity tests switch it inside one process)
    const int mode = fv && fv[0] ? atoi(fv) : -1;
    if (mode == 0 || (mode < 0 && !SRT_FUSE_HEAD_DEFAULT(p.in16))) return 1;
    if (p.Cout != 1 || p.Cin != 32 || p.CA != 16 || p.out16 || h.out16 || !p.srcA || !p.srcB || (p.H & 1) || p.W % (p.in16 ? 8 : 4)) return 1;
    // up5's output in C8 (large fp16-storage batches, srt_nn5.hip): not covered.  A BC8 instantiation of this kernel (the same three edits as in srt_up6_stream_kernel) was
    // built and dropped: its masks differed from run to run in 16-lane pieces of single rows (first / last interval of the head, channel 0) while the planar form and the
    // BC8 form of the two-kernel path are stable - unexplained, and this form is slower anyway.
    if (p.c8srcB || p.c8srcA) return 1;
    if ((size_t)64 * p.H * p.W >= 0x7fffffffu || h.H != 2 * p.H || h.W != 2 * p.W || h.ntiles != p.ntiles || h.nstems != p.nstems) return 1;
    const long cols = (long)((p.W + 63) / 64) * p.nstems * p.ntiles;
    if (cols < 512) return 1;
    if (p.in16) { if (h.variant == 0) SRT_LAUNCH((srt_up6_head_kernel<64, 2, true, true>), dim3((unsigned)cols), dim3(512), 0, s, p, h); else SRT_LAUNCH((srt_up6_head_kernel<64, 2, true, false>), dim3((unsigned)cols), dim3(512), 0, s, p, h); }
    else { if (h.variant == 0) SRT_LAUNCH((srt_up6_head_kernel<64, 2, false, true>), dim3((unsigned)cols), dim3(512), 0, s, p, h); else SRT_LAUNCH((srt_up6_head_kernel<64, 2, false, false>), dim3((unsigned)cols), dim3(512), 0, s, p, h); }
    return srt_launch_status();
}

#ifndef SRT_HEAD_ROWS_DEFAULT
#define SRT_HEAD_ROWS_DEFAULT 4
#endif
int srt_head_out16_ok(const SrtHeadParams& p)
{
    return SRT_HEAD_ROWS_DEFAULT == 4 && p.W % 4 == 0 && (size_t)p.H * (p.W / 4) * p.nstems * p.ntiles >= (size_t)256 * 1024 * 4;
}
int srt_launch_head(const SrtHeadParams& p, hipStream_t s)
{
    int nro = SRT_HEAD_ROWS_DEFAULT;
    if (p.out16) {
        if (!srt_head_out16_ok(p)) return -1;
        const size_t nsets = 2 * (size_t)((p.H + 7) / 8);
        size_t bx = (nsets * (p.W / 4) + 255) / 256;
        if (bx > 65535) bx = 65535;
        const unsigned grid = (unsigned)bx * p.nstems * p.ntiles;
        if (p.variant == 0) SRT_LAUNCH((srt_head_rows_kernel<true, 4, true>), dim3(grid), dim3(256), 0, s, p); else SRT_LAUNCH((srt_head_rows_kernel<false, 4, true>), dim3(grid), dim3(256), 0, s, p);
        return srt_launch_status();
    }
#ifdef SRT_TUNING
    if (const char* tv = getenv("SRT_TUNE_HEADROWS")) nro = atoi(tv);         // 0: one output row per thread (srt_head_kernel4), 2, 4
#endif
    if (p.W % 4 == 0 && (nro == 2 || nro == 4) && (size_t)p.H * (p.W / 4) * p.nstems * p.ntiles >= (size_t)256 * 1024 * nro) {   // at least 1024 workgroups: four per CU
        const size_t nsets = 2 * (size_t)((p.H + 2 * nro - 1) / (2 * nro));
        size_t bx = (nsets * (p.W / 4) + 255) / 256;
        if (bx > 65535) bx = 65535;
        const unsigned grid = (unsigned)bx * p.nstems * p.ntiles;
        if (nro == 4) { if (p.variant == 0) SRT_LAUNCH((srt_head_rows_kernel<true, 4>), dim3(grid), dim3(256), 0, s, p); else SRT_LAUNCH((srt_head_rows_kernel<false, 4>), dim3(grid), dim3(256), 0, s, p); }
        else { if (p.variant == 0) SRT_LAUNCH((srt_head_rows_kernel<true, 2>), dim3(grid), dim3(256), 0, s, p); else SRT_LAUNCH((srt_head_rows_kernel<false, 2>), dim3(grid), dim3(256), 0, s, p); }
        return srt_launch_status();
    }
    if (p.W % 4 == 0) {
        size_t bx4 = ((size_t)p.H * (p.W / 4) + 255) / 256;
#ifdef SRT_TUNING
        if (const char* tv = getenv("SRT_TUNE_HEAD")) { const int it = atoi(tv); if (it > 1) bx4 = (bx4 + it - 1) / it; }   // pixels-per-thread sweep
#endif
        if (bx4 > 65535) bx4 = 65535;
        const unsigned grid = (unsigned)bx4 * p.nstems * p.ntiles;
        if (p.variant == 0) SRT_LAUNCH(srt_head_kernel4<true>, dim3(grid), dim3(256), 0, s, p);
        else SRT_LAUNCH(srt_head_kernel4<false>, dim3(grid), dim3(256), 0, s, p);
        return srt_launch_status();
    }
    size_t total = (size_t)p.H * p.W, bx = (total + 255) / 256;
    if (bx > 65535) bx = 65535;
    SRT_LAUNCH(srt_head_kernel, dim3((unsigned)bx, p.nstems * p.ntiles), dim3(256), 0, s, p);
    return srt_launch_status();
}
