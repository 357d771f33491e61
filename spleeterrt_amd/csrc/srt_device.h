// srt_device.h — device helpers shared by the network kernels (activations, fused epilogues, source-channel select).
#pragma once
#include "srt_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The epilogue math keeps the reference's operation order with contraction off, so the only difference
// from the CPU path is the summation order inside the dot products.
#pragma clang fp contract(off)
__device__ __forceinline__ float srt_act(float x, int kind, int variant)
{
    if (kind == SRT_ACT_LEAKY) return x >= 0.0f ? x : 0.2f * x;          // Executable/spleeter.c:43-46
    if (kind == SRT_ACT_RELU) return x >= 0.0f ? x : 0.0f;               // :47-50
    if (variant == 0 && x < -15.0f) return -1.0f;                        // :51-56 (VST flavour has no clamp)
    // __expf = v_exp_f32(x * log2e): |abs error| <= e^x * |x| * 6e-8 <= 2.2e-8 on x < 0, far below the fp32 noise of the
    // dot product feeding it, at ~1/5 of the instructions of expf (the epilogue evaluates it for every output element)
    return x >= 0.0f ? x : __expf(x) - 1.0f;
}
// Branch-free form for the MFMA epilogues (which evaluate it for every accumulator element): the activation kind is
// launch-uniform, so it is folded once into three scalars and every element costs one v_exp_f32 + a few VALU ops instead
// of a chain of scalar branches (the branchy form was 88 % of the decoder kernel's instructions).
//   x >= 0 ? x : lin*x + ue*((x < thr) ? -1 : exp(x) - 1)      leaky: (0.2, 0, -inf)  relu: (0, 0, -inf)  elu: (0, 1, -15 | -inf)
struct SrtAct { float lin, ue, thr; };
__device__ __forceinline__ int srt_act_kind(const SrtConvParams& p, int stem) { return ((p.elu_mask >> stem) & 1u) ? SRT_ACT_ELU : p.act; }
__device__ __forceinline__ SrtAct srt_act_params(int kind, int variant)
{
    SrtAct a;
    a.lin = kind == SRT_ACT_LEAKY ? 0.2f : 0.0f;
    a.ue = kind == SRT_ACT_ELU ? 1.0f : 0.0f;
    a.thr = (kind == SRT_ACT_ELU && variant == 0) ? -15.0f : -__builtin_huge_valf();
    return a;
}
__device__ __forceinline__ float srt_act_apply(float x, const SrtAct& a)
{
    const float e = x < a.thr ? -1.0f : __expf(fminf(x, 0.0f)) - 1.0f;
    const float neg = a.lin * x + a.ue * e;
    return x >= 0.0f ? x : neg;
}
// The same values with fewer instructions, for code that takes the activation kind as a workgroup-uniform branch (the fp32 MFMA
// shares the vector ALU with every other VALU instruction, so instruction count is what these paths cost):
//   LeakyReLU / ReLU   max(x, lin*x)                      lin in [0, 1): x >= 0 picks x, x < 0 picks lin*x       (2 instructions, was 3)
//   ELU, no clamp      max(x, 0) + (exp(min(x, 0)) - 1)   x >= 0: x + 0, x < 0: 0 + (exp(x) - 1)                 (6, was 11 with the selects)
// Both give bit-identical results to srt_act_apply except for the sign of a zero.
__device__ __forceinline__ float srt_act_linear(float x, float lin) { return fmaxf(x, lin * x); }
__device__ __forceinline__ float srt_act_elu_noclamp(float x) { return fmaxf(x, 0.0f) + (__expf(fminf(x, 0.0f)) - 1.0f); }
__device__ __forceinline__ bool srt_act_is_plain_elu(const SrtAct& a) { return a.ue != 0.0f && a.thr == -__builtin_huge_valf(); }
__device__ __forceinline__ float srt_enc_epilogue(float v, float scale, float shift, const SrtAct& a)
{
    return srt_act_apply(scale * v + shift, a);                          // spleeter.c:188: act(bn[C+s]*v + bn[s])
}
__device__ __forceinline__ float srt_dec_epilogue(float acc, float bias, float scale, float shift, const SrtAct& a)
{
    const float v = srt_act_apply(acc + bias, a);                        // spleeter.c:244-245: activation BEFORE BN
    return scale * v + shift;
}
// decoder epilogue with the activation kind resolved by the caller's uniform branch (same values as srt_dec_epilogue)
__device__ __forceinline__ float srt_dec_epilogue_elu(float acc, float bias, float scale, float shift)
{
    const float v = srt_act_elu_noclamp(acc + bias);
    return scale * v + shift;
}
__device__ __forceinline__ float srt_dec_epilogue_lin(float acc, float bias, float scale, float shift, float lin)
{
    const float v = srt_act_linear(acc + bias, lin);
    return scale * v + shift;
}
// Consumer-side form of the encoder's batch-norm + activation (spleeter.c:188): the producing layer stores conv + bias once
// (the skip tensor) and the next encoder layer applies act(scale * v + shift) to the four staged values of one channel.
// Padding must stay exactly zero: the CALLER passes scale = shift = 0 for padded elements (their staged value is 0 or any finite
// number), which gives act(0 * v + 0) = 0 for every activation kind without a per-element select - hipcc turns a select around
// this much arithmetic into a divergent branch per element.  The ELU / non-ELU choice is workgroup-uniform (a scalar branch),
// so LeakyReLU stems do not pay for a v_exp_f32 per staged value.  Same operations, same order as srt_enc_epilogue.
__device__ __forceinline__ float srt_bn(float v, float scale, float shift) { return scale * v + shift; }      // contraction is off here
__device__ __forceinline__ float srt_act_lin(float x, const SrtAct& a) { return fmaxf(x, a.lin * x); }       // LeakyReLU / ReLU stems (lin in [0, 1))
__device__ __forceinline__ float srt_enc_input1(float v, float scale, float shift, const SrtAct& a)
{
    const float x = scale * v + shift;
    if (a.ue != 0.0f) return srt_act_is_plain_elu(a) ? srt_act_elu_noclamp(x) : srt_act_apply(x, a);
    return srt_act_linear(x, a.lin);
}
__device__ __forceinline__ float4 srt_enc_input4(float4 v, float scale, float shift, const SrtAct& a)
{
    float4 o;
    const float x0 = scale * v.x + shift, x1 = scale * v.y + shift, x2 = scale * v.z + shift, x3 = scale * v.w + shift;
    if (a.ue != 0.0f) {
        if (srt_act_is_plain_elu(a)) {                                       // VST flavour (no -15 clamp): the common case
            o.x = srt_act_elu_noclamp(x0); o.y = srt_act_elu_noclamp(x1); o.z = srt_act_elu_noclamp(x2); o.w = srt_act_elu_noclamp(x3);
        } else {
            o.x = srt_act_apply(x0, a); o.y = srt_act_apply(x1, a); o.z = srt_act_apply(x2, a); o.w = srt_act_apply(x3, a);
        }
    } else {
        o.x = srt_act_linear(x0, a.lin); o.y = srt_act_linear(x1, a.lin); o.z = srt_act_linear(x2, a.lin); o.w = srt_act_linear(x3, a.lin);
    }
    return o;
}
// branchy forms (naive cross-check kernels)
__device__ __forceinline__ float srt_enc_epilogue(float v, float scale, float shift, int act, int variant)
{
    return srt_act(scale * v + shift, act, variant);
}
__device__ __forceinline__ float srt_dec_epilogue(float acc, float bias, float scale, float shift, int act, int variant)
{
    float v = srt_act(acc + bias, act, variant);
    return scale * v + shift;
}
#pragma clang fp contract(fast)

// element-typed view of the same selection (T = _Float16 when the source tensors are stored as halves: strides are in elements)
template <class T>
__device__ __forceinline__ const T* srt_src_channel_t(const SrtConvParams& p, int stem, int tile, int ch, size_t hw)
{
    const bool a = ch < p.CA;
    const T* base = reinterpret_cast<const T*>(a ? p.srcA : p.srcB);
    const size_t ss = a ? p.srcA_stem : p.srcB_stem, ts = a ? p.srcA_tile : p.srcB_tile;
    return base + stem * ss + tile * ts + (size_t)(a ? ch : ch - p.CA) * hw;
}
__device__ __forceinline__ const float* srt_src_channel(const SrtConvParams& p, int stem, int tile, int ch, size_t hw)
{
    const bool a = ch < p.CA;
    const float* base = a ? p.srcA : p.srcB;
    const size_t ss = a ? p.srcA_stem : p.srcB_stem, ts = a ? p.srcA_tile : p.srcB_tile;
    return base + stem * ss + tile * ts + (size_t)(a ? ch : ch - p.CA) * hw;
}


// ------------------------------------------------------------------------------------------- XCD-aware block order
// The hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with a private 4 MiB L2.  With the plain
// (spatial, M-block, instance) order every XCD sees the weight slabs of several (stem, M-block) pairs at once (13 MB for
// up2) and streams them from MALL/HBM for every workgroup (ablation: +8 % when the weight/patch traffic is removed).
// Here the launch is 1-D and XCD x walks the x-th CONTIGUOUS chunk of the (stem, M-block)-major order, so at any time an
// XCD works on one weight slab (1.6-3.3 MB, L2 resident) and on neighbouring pixel tiles.  Speed only: any placement
// gives the same result.  Returns the position in (stem, M-block)-major order.
__device__ __forceinline__ int srt_xcd_order(int total)
{
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int q = total >> 3, r = total & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
struct SrtBlockCoord { int sp, mblk, stem, grp, ks; };
// order: w = (stem, mblk) slowest, then instance group, then spatial tile (fastest: neighbours share halo rows)
// ksplit > 1 (split-K launches): the K slice is the fastest index, so the workgroups that share one input patch are neighbours
// dual > 1 (two-tile workgroups): workgroup w owns the neighbouring tiles dual*w + sub of the same order
__device__ __forceinline__ SrtBlockCoord srt_block_coord(int nsp, int nmb, int nstem, int ngrp, int ksplit = 1, int dual = 1, int sub = 0)
{
    const int pos0 = srt_xcd_order(nsp * nmb * nstem * ngrp * ksplit / dual) * dual + sub;
    SrtBlockCoord c;
    c.ks = pos0 % ksplit;
    const int pos = pos0 / ksplit;
    c.sp = pos % nsp;
    const int t = pos / nsp;
    c.grp = t % ngrp;
    const int w = t / ngrp;
    c.mblk = w % nmb;
    c.stem = w / nmb;
    return c;
}

