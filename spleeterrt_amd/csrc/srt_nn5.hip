// srt_nn5.hip — the fp16-storage layers of large batches in "C8" form (round 6): down3..down6 (spleeter.c:96-100, im2col_dilated.c:10-33) and
// up1..up5 (spleeter.c:73-78, im2col_dilated.c:34-65) on v_mfma_f32_32x32x16_f16, fed by LDS-DMA only.
//
// Why a second fp16 form.  The kernels of srt_nn3.hip read PLANAR half tensors ([C][H][W]): the B operand of the fp16 MFMA wants the 8 input channels of a
// k-group contiguous per pixel, so every staged value goes global -> VGPR -> register transpose -> ds_write, 16 eight-byte loads and ~100 VALU operations per
// thread and K chunk beside 25 MFMAs, one tile per workgroup, one chunk of prefetch.  At the 5-stem batch those layers sat under NEITHER roof (round-6 counters,
// profiles/r06a_f16_pmc.json: down3 0.34 ms at 0.18 matrix-pipe busy and 0.46 wait; 10 us per workgroup for 0.8 us of MFMAs).  Here the activation tensors between
// down2 and up5 are stored channel-interleaved by eight,
//     C8:  element (c, y, x) of an instance at ((c / 8) * H * W + y * W + x) * 8 + c % 8          (16 B = the 8 channels of one k-group at one pixel)
// which IS the B-fragment layout: a patch goes HBM -> LDS by buffer_load_dwordx4 ... lds (16 B per lane, out-of-image lanes land as zeros through the buffer range
// check), no register staging, no transposes, and the MFMA epilogue writes 8-byte pieces that tile whole 64-B lines (the planar form wrote 2-byte pieces).
// A workgroup is 8 waves, one 32-pixel sub-tile each, sharing one 32-row weight slab per 16-channel K chunk; it walks `tpw` consecutive (instance group, tile)
// units of one (stem, M block) as ONE stream of K steps through an LDS ring (patch + slab per stage): the DMA of step s+1 (s+2 in the decoder) is in flight under
// the MFMAs of step s across unit boundaries, and a unit's epilogue is issued right after the next step's DMA, in its shadow.  One barrier per step.
// Tensors in C8: raw2..raw6, act2..act5, up1..up5 (srt_engine.hip: forward_range, `c8`); down1's outputs (down2's input, up6's skip) stay planar, so
// srt_down1_stream_kernel is untouched and down2 (srt_enc_f16) only switches its epilogue; up5 runs here class-stacked and the up6 kernels (srt_nn.hip) take
// its 16 channels as one ds_read_b128 / one 16-byte load per B fragment (SrtConvParams::c8srcB).
#include "srt_device.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <stdio.h>
#include <map>
#include <mutex>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// s_waitcnt immediate for "at most n vector-memory operations outstanding" (gfx9 encoding, see srt_nn4.hip)
constexpr int c8_vmcnt(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }
constexpr unsigned C8_OOR = 0x80000000u;                       // >= num_records: the DMA lands zeros

// All DMA from inline assembly (the compiler's own LDS-DMA bookkeeping would wait vmcnt(0) before every later LDS read, srt_nn4.hip)
__device__ __forceinline__ void c8_dma_buffer(unsigned voff, i32x4 rs, unsigned soff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(rs), "s"(lds_dst), "s"(soff) : "memory");
}
__device__ __forceinline__ void c8_dma_global(unsigned voff, const void* sbase, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ i32x4 c8_rsrc(const void* base, unsigned nrec)
{
    const size_t b = (size_t)base;
    i32x4 rs;
    // (readfirstlane: the words are wave-uniform by construction; inside the loader-wave branch the compiler no longer proves it and would hand the asm VGPRs)
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(b & 0xffffffffu)); rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffu));
    rs.z = __builtin_amdgcn_readfirstlane((int)nrec); rs.w = 0x00020000;
    return rs;
}

// ------------------------------------------------------------------------------------------- epilogue arithmetic, two values per instruction
// The generic per-value forms (srt_device.h: srt_enc_epilogue / srt_dec_epilogue through srt_act_apply) are ~15 VALU instructions + one v_exp_f32 per value:
// with the 16x faster fp16 MFMA that was 30-45 % of these kernels (ablation, round 6).  Here a PAIR of values goes through v_pk_add/mul_f32 (the compiler forms
// them from the 2-vector arithmetic below; max / min / exp stay scalar), with the activation kind a wave-uniform branch.  Same operations in the same order as
// the generic forms (contraction off), so the values are the same up to the sign of a zero.
typedef float f2 __attribute__((ext_vector_type(2)));
#define F2(a, b) (f2){ (a), (b) }
#pragma clang fp contract(off)
__device__ __forceinline__ f2 c8_act_pair(f2 x, const SrtAct& a)
{
    if (a.ue != 0.0f) {
        if (srt_act_is_plain_elu(a)) {                         // max(x, 0) + (exp(min(x, 0)) - 1)
            const f2 m = { fminf(x.x, 0.0f), fminf(x.y, 0.0f) };
            const f2 e = { __expf(m.x), __expf(m.y) };
            const f2 p = { fmaxf(x.x, 0.0f), fmaxf(x.y, 0.0f) };
            return p + (e - 1.0f);
        }
        return F2(srt_act_apply(x.x, a), srt_act_apply(x.y, a));
    }
    const f2 l = x * a.lin;                                    // LeakyReLU / ReLU: max(x, lin x)
    return F2(fmaxf(x.x, l.x), fmaxf(x.y, l.y));
}
__device__ __forceinline__ f2 c8_dec_pair(f2 acc, f2 bias, f2 scale, f2 shift, const SrtAct& a)
{
    const f2 v = c8_act_pair(acc + bias, a);                   // spleeter.c:244-245: activation BEFORE BN
    return scale * v + shift;
}
__device__ __forceinline__ void c8_enc_pair(f2 acc, f2 bias, f2 scale, f2 shift, const SrtAct& a, f2& raw, f2& act)
{
    raw = acc + bias;                                          // conv + bias: the skip tensor
    act = c8_act_pair(scale * raw + shift, a);                 // spleeter.c:188
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------- 16-byte stores from the MFMA accumulator layout
// A lane of the 32x32 MFMA result holds FOUR consecutive channels (rows 8 q + 4 g + 0..3) of its pixel: 8 bytes of a C8 pixel slot, the other 8 in lane + 32.
// Stored as they are, an epilogue is 8-byte stores - and the store path, not the arithmetic, was what the epilogue cost (round 6: halving the VALU count moved
// nothing; MI355X_MICROARCH.md: store tails are issue-bound, dwordx4 halves them).  One v_permlane32_swap per dword trades halves between lanes l and l + 32 so
// that each lane ends up with ALL eight channels of ONE slot (the low lane: slot A, the high lane: slot B): one 16-byte store per lane where there were two 8-byte
// ones, a wave writing 1 KiB of whole lines per instruction.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// a = this lane's channels of slot A, b = its channels of slot B.  Returns the 16 bytes of the slot the lane stores: slot A for lanes 0-31, slot B for lanes 32-63.
__device__ __forceinline__ u32x4 c8_pair16(h4 a, h4 b)
{
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    const auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);     // r[0]: low lanes keep a, high lanes get the low partner's b; r[1]: low lanes get the high partner's a, high lanes keep b
    const auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
    return (u32x4){ r0[0], r1[0], r0[1], r1[1] };
}

// ------------------------------------------------------------------------------------------- packing / unpacking
// up5 (Cout = 16): the two x-parity classes of an output row share an input shift, so their 16 + 16 output channels fill one 32-row MFMA tile:
// wp[((cg*15 + ky*3 + (dx+1))*2 + g)*32 + px*16 + co][q] = w[ci = cg*16 + g*8 + q][co][ky][kx],  kx = px + 1 - 2 dx (zero outside 0..4) - 15 MFMAs per
// K chunk and sub-tile instead of 25 half-empty ones (the fp32 form of the same idea: srt_pack_classstack_kernel, srt_nn2.hip)
__global__ void srt_pack16_classstack_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int Cin, int Cout)
{
    const int total = (Cin / 16) * 15 * 2 * 32 * 8;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int q = e % 8, row = (e / 8) % 32, g = (e / 256) % 2, t = (e / 512) % 15, cg = e / (512 * 15);
        const int px = row / 16, co = row % 16, ky = t / 3, dx = t % 3 - 1, kx = px + 1 - 2 * dx, ci = cg * 16 + g * 8 + q;
        wp[e] = (_Float16)((kx >= 0 && kx < 5 && co < Cout) ? w[((size_t)ci * Cout + co) * 25 + ky * 5 + kx] : 0.0f);
    }
}
int srt_launch_pack16_classstack(const float* w, uint16_t* wp, int Cin, int Cout, hipStream_t s)
{
    SRT_LAUNCH(srt_pack16_classstack_kernel, dim3(64), dim3(256), 0, s, w, (_Float16*)wp, Cin, Cout);
    return srt_launch_status();
}
// test taps (srtCopyTensor): one instance, C8 halves -> planar floats
__global__ void srt_c8_to_float_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int C, size_t hw)
{
    const size_t total = (size_t)C * hw;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / hw); const size_t pix = e % hw;
        dst[e] = (float)src[((size_t)(c / 8) * hw + pix) * 8 + c % 8];
    }
}
int srt_launch_c8_to_float(const void* src, float* dst, int C, size_t hw, hipStream_t s)
{
    SRT_LAUNCH(srt_c8_to_float_kernel, dim3(1024), dim3(256), 0, s, (const _Float16*)src, dst, C, hw);
    return srt_launch_status();
}

// ------------------------------------------------------------------------------------------- operand prefetch distance of the tap loops
// SRT_C8_PDE: taps the encoder's A / B fragment reads run ahead of their MFMAs; SRT_C8_PDA / SRT_C8_PDB: the decoder's A fragments (taps ahead) and B fragments
// (input shifts ahead).  0 = the compiler's own order.  The group barriers take literal sizes, hence the switches (their argument is a constant once the tap loop is unrolled).
#ifndef SRT_C8_PDE
#define SRT_C8_PDE 3
#endif
#ifndef SRT_C8_PDA
#define SRT_C8_PDA 3
#endif
#ifndef SRT_C8_PDB
#define SRT_C8_PDB 1
#endif
__device__ __forceinline__ void c8_sgb_reads(int n)
{
    switch (n) {
    case 1: __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); break;
    case 2: __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); break;
    case 3: __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); break;
    case 4: __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); break;
    default: break;
    }
}
__device__ __forceinline__ void c8_sgb_mfma(int n)
{
    switch (n) {
    case 1: __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); break;
    case 2: __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); break;
    default: break;
    }
}

// the decoder's taps in shift-major order (the order of srt_dec_c8's reference loop): A-fragment row block, accumulator, input shift, first tap of its shift
template <bool CS>
struct C8DecOrder {
    int aidx[25], acc[25], sh[25], first[25], n;
    constexpr C8DecOrder() : aidx{}, acc{}, sh{}, first{}, n(0)
    {
        for (int s = 0; s < 9; ++s) {
            const int dy = s / 3 - 1, dx = s % 3 - 1;
            int f = 1;
            for (int ky = 0; ky < 5; ++ky) {
                const int py = (ky + 1) & 1;
                if ((py + 1 - ky) / 2 != dy) continue;
                if (CS) { aidx[n] = ky * 3 + dx + 1; acc[n] = py; sh[n] = s; first[n] = f; f = 0; ++n; }
                else {
                    for (int kx = 0; kx < 5; ++kx) {
                        const int px = (kx + 1) & 1;
                        if ((px + 1 - kx) / 2 != dx) continue;
                        aidx[n] = ky * 5 + kx; acc[n] = py * 2 + px; sh[n] = s; first[n] = f; f = 0; ++n;
                    }
                }
            }
        }
    }
};

// ------------------------------------------------------------------------------------------- encoder, C8 in -> C8 out (raw + act)
// Tile = TH x TW outputs of NI instances = 8 sub-tiles of 32 pixels (SW wide); M block = 32 output channels.
// LDS patch of a stage: [k-group 2][NI][PH = 2 TH + 3 rows][even columns PWH | odd columns PWH] pixel slots of 16 B: the stride-2 taps of 32 neighbouring
// outputs read 32 neighbouring slots (conflict-free ds_read_b128), the split is done by the DMA's per-lane source addresses.
// LW (both kernels): 0 (shipped) - every one of the 8 waves moves its share of the DMA pieces AND computes one sub-tile.  1 (tuning library) - waves 0-3 compute
// two sub-tiles each (one A fragment per two MFMAs), waves 4-7 only issue DMA, so that the 60-190 cycles an LDS-DMA instruction holds its wave
// (MI355X_MICROARCH.md) are not taken from a computing wave.  Measured slower (see the dispatch section).
// ABL (SRT_TUNING builds only; wrong results): timing ablations - 1 no patch DMA after the ring is primed, 2 no weight DMA, 4 no MFMAs (and no LDS reads), 8 no
// epilogue, 16 MFMAs on constant operands (no LDS reads), 32 no barrier / no DMA wait
template <int SW, int NSY, int NI, int LW, int ABL = 0>
__global__ void __launch_bounds__(512, 1) srt_enc_c8(const SrtConvParams p, int tpw_)
{
    const int tpw = tpw_ & 0xffff;
    const bool c8_weights_stay = (tpw_ >> 16) != 0;             // (launcher: SPLEETERRT_C8_WRES, default on)
    static_assert(NSY * NI == 8 && 32 % SW == 0, "eight 32-pixel sub-tiles");
    constexpr int SH = 32 / SW, TH = NSY * SH, TW = SW;
    constexpr int NLW = LW ? 4 : 8, NR = LW ? 2 : 1;            // waves that issue DMA; sub-tiles per computing wave
    constexpr int PH = 2 * TH + 3, PWH = TW + 3, ROWS = 2 * PWH;
    constexpr int PLANE = NI * PH * ROWS;                      // 16-B slots per k-group plane
    constexpr int PITEMS = 2 * PLANE, NPP = (PITEMS + 63) / 64, PPW = (NPP + NLW - 1) / NLW;
    constexpr int PATCH_H = NPP * 512, WSLAB_H = 25 * 512, WPW = (25 + NLW - 1) / NLW;   // halves; the slab = 25 pieces of 1 KiB (two (tap, k-group) rows of 32 x 16 B each)
    constexpr int STAGE_H = PATCH_H + WSLAB_H;
    __shared__ __attribute__((aligned(16))) _Float16 s_mem[2 * STAGE_H];
    static_assert(sizeof(s_mem) <= 160 * 1024, "LDS");

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = !LW || wave >= 4, worker = !LW || wave < 4;
    const int lw = LW ? (wave & 3) : wave;
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int tilesX = (Wo + TW - 1) / TW, tilesY = (Ho + TH - 1) / TH, nsp = tilesX * tilesY;
    const int groups = (p.ntiles + NI - 1) / NI, nunits = nsp * groups;
    const int upw = (nunits + tpw - 1) / tpw, MBK = p.Cout / 32;
    const int pos = srt_xcd_order(upw * MBK * p.nstems);
    const int wsel = pos / upw, mblk = wsel % MBK, stem = wsel / MBK, m0 = mblk * 32;
    const int unit0 = (pos % upw) * tpw, unit1 = min(unit0 + tpw, nunits);
    const int nch = p.Cin / 16, nsteps = (unit1 - unit0) * nch;
    if (nsteps <= 0) return;
    const SrtAct actp = srt_act_params(srt_act_kind(p, stem), p.variant);
    const size_t hw = (size_t)p.H * p.W;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)s_mem;

    // ---- weight slab DMA: loader lw moves pieces lw, lw + NLW, ... (past the last piece: the last piece again - same bytes, no branch)
    const _Float16* wp = (const _Float16*)(p.wpack16 + stem * p.wpack16_stem) + (size_t)m0 * 8;
    const size_t cgStride = (size_t)50 * p.CP * 8;             // halves per 16-channel chunk
    unsigned wvoff[WPW], wm0[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int piece = min(lw + NLW * i, 24);
        wvoff[i] = (unsigned)(((2 * piece + g) * p.CP + l31) * 16);
        wm0[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(PATCH_H * 2 + piece * 1024));
    }
    // ---- patch DMA: slot e = ((gg * NI + il) * PH + r) * ROWS + slot of the stage <- k-group gg of the chunk, instance tile0 + il, input row 2 ty0 - 1 + r,
    // input column 2 tx0 - 4 + 2 idx + par (slot = par * PWH + idx)
    unsigned pvoff[PPW], pm0[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) pm0[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(min(lw + NLW * i, NPP - 1) * 1024));
    const unsigned nrec = (unsigned)min((size_t)0x7fffffff, (size_t)2 * NI * p.srcA_tile);
    const _Float16* pa = nullptr;                              // wave-uniform: plane 0 of the DMA unit's first instance
    auto set_dma_unit = [&](int unit) {
        const int sp = unit % nsp, tile0 = (unit / nsp) * NI, tx0 = (sp % tilesX) * TW, ty0 = (sp / tilesX) * TH;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int e = min(lw + NLW * i, NPP - 1) * 64 + lane;
            const int gg = e / PLANE, rem = e % PLANE, slot = rem % ROWS, r = (rem / ROWS) % PH, il = rem / (ROWS * PH);
            const int par = slot >= PWH ? 1 : 0, idx = slot - par * PWH;
            const int gy = 2 * ty0 - 1 + r, gx = 2 * tx0 - 4 + 2 * idx + par;
            const bool ok = e < PITEMS && tile0 + il < p.ntiles && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            pvoff[i] = ok ? (unsigned)(2 * (size_t)il * p.srcA_tile + 16 * ((size_t)gg * hw + (size_t)gy * p.W + gx)) : C8_OOR;
        }
        pa = reinterpret_cast<const _Float16*>(p.srcA) + stem * p.srcA_stem + tile0 * p.srcA_tile;
    };
    const unsigned chunk_bytes = (unsigned)(32 * hw);          // two C8 planes
    // One K chunk per unit (down2, Cin = 16): every step of the workgroup multiplies by the SAME 25 KiB slab, so it is moved once per stage and stays - 25 of the
    // 66 KiB a step used to move (the waits below count stores only, so a step with fewer DMA instructions needs no other bookkeeping).
    const bool wres = nch == 1 && c8_weights_stay;
    auto issue_dma = [&](int ch, int stage, bool primed) {
        const unsigned sb = (unsigned)stage * (unsigned)(STAGE_H * 2);
        const i32x4 rs = c8_rsrc(pa, nrec);
        const _Float16* ws = wp + (size_t)ch * cgStride;
        if (!(((ABL & 2) || wres) && primed)) {
#pragma unroll
            for (int i = 0; i < WPW; ++i) c8_dma_global(wvoff[i], ws, wm0[i] + sb);
        }
        if (!((ABL & 1) && primed)) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) c8_dma_buffer(pvoff[i], rs, (unsigned)ch * chunk_bytes, pm0[i] + sb);
        }
    };

    // ---- this wave's sub-tiles: NR * (wave & (8 / NR - 1)) + n
    int boff[NR], il_w[NR], oy_l[NR];
    const int ox_l = l31 % SW;
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int st = NR * (LW ? (wave & 3) : wave) + n;
        il_w[n] = st / NSY; oy_l[n] = (st % NSY) * SH + l31 / SW;
        boff[n] = (g * PLANE + (il_w[n] * PH + 2 * oy_l[n]) * ROWS + ox_l) * 8;
    }
    const int aoff = (g * 32 + l31) * 8;
    const size_t ohw = (size_t)Ho * Wo;
    const bool twoOut = p.outAct != nullptr && p.bnScale != nullptr;
    float bi[16], sc[16], sf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const size_t ci = stem * p.coeff_stem + m0 + (r & 3) + 8 * (r >> 2) + 4 * g;
        bi[r] = p.bias[ci]; sc[r] = twoOut ? p.bnScale[ci] : 0.0f; sf[r] = twoOut ? p.bnShift[ci] : 0.0f;
    }
    _Float16* rawh = reinterpret_cast<_Float16*>(p.outRaw);
    _Float16* acth = reinterpret_cast<_Float16*>(p.outAct);
    size_t obase[NR]; bool pix_ok[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) { obase[n] = 0; pix_ok[n] = false; }
    auto set_out_unit = [&](int unit) {
        const int sp = unit % nsp;
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int tile = (unit / nsp) * NI + il_w[n], oy = (sp / tilesX) * TH + oy_l[n], ox = (sp % tilesX) * TW + ox_l;
            pix_ok[n] = tile < p.ntiles && oy < Ho && ox < Wo;
            // the lane stores channel group m0/8 + 2 k + g of its pixel (c8_pair16: low lanes the even group of a pair, high lanes the odd one)
            obase[n] = stem * p.out_stem + (pix_ok[n] ? tile : 0) * p.out_tile + ((size_t)(m0 / 8 + g) * ohw + (pix_ok[n] ? (size_t)oy * Wo + ox : 0)) * 8;
        }
    };
    // (Round 6 also tried the epilogue cut into its four channel groups and spread over the K steps of the next unit, the two waves of a SIMD at different
    // points of the step: 5-25 % SLOWER per layer - the slice's branch splits the unrolled MFMA stream and every step then carries stores - so it is whole.)
    f32x16 acc[NR];
    auto epilogue = [&]() {
        // (no early return for lanes outside the image: the lane exchange below needs every lane; their stores are masked)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                h4 rv[2], av[2];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const int r = 4 * (2 * k + qq) + j;
                        f2 v, a;
                        c8_enc_pair(F2(acc[n][r], acc[n][r + 1]), F2(bi[r], bi[r + 1]), F2(sc[r], sc[r + 1]), F2(sf[r], sf[r + 1]), actp, v, a);
                        rv[qq][j] = (_Float16)v.x; rv[qq][j + 1] = (_Float16)v.y; av[qq][j] = (_Float16)a.x; av[qq][j + 1] = (_Float16)a.y;
                    }
                const u32x4 r16 = c8_pair16(rv[0], rv[1]);
                if (pix_ok[n]) *reinterpret_cast<u32x4*>(rawh + obase[n] + (size_t)(2 * k) * ohw * 8) = r16;
                if (twoOut) {
                    const u32x4 a16 = c8_pair16(av[0], av[1]);
                    if (pix_ok[n]) *reinterpret_cast<u32x4*>(acth + obase[n] + (size_t)(2 * k) * ohw * 8) = a16;
                }
            }
    };

    int du = unit0, dch = 0, cu = unit0, ch = 0;
    // LW = 0: a unit's epilogue stores are issued AFTER the DMA of the next step, and vmcnt is one in-order queue (loads, LDS-DMA and stores retire in issue order -
    // the property srt_down1_stream_kernel is built on): "my pieces of step s have landed" is "at most the NST stores behind them are outstanding".
    int pending = 0;                                           // store instructions this wave issued in the previous step (wave-uniform); stays 0 on a loader wave
    if (loader) { set_dma_unit(unit0); issue_dma(0, 0, false); }
    for (int s = 0; s < nsteps; ++s) {
        if (!(ABL & 32) || s == 0) {
            if (loader) {                                      // this wave's pieces of step s have landed
                if (pending == 4) __builtin_amdgcn_s_waitcnt(c8_vmcnt(4));
                else if (pending == 2) __builtin_amdgcn_s_waitcnt(c8_vmcnt(2));
                else __builtin_amdgcn_s_waitcnt(c8_vmcnt(0));
            }
            __syncthreads();                                   // everybody's have; everybody is past step s-1 (its stage is free)
        }
        pending = 0;
        if (loader && s + 1 < nsteps) {
            if (++dch == nch) { dch = 0; set_dma_unit(++du); }
            issue_dma(dch, (s + 1) & 1, s >= 1);
        }
        if (worker) {
            if (ch == 0) {
                if (s > 0 && !(ABL & 8)) {                     // the previous unit's stores go out in the shadow of the DMA just issued
                    epilogue();
                    if (!LW) pending = __builtin_amdgcn_ballot_w64(pix_ok[0]) != 0 ? (twoOut ? 4 : 2) : 0;   // (a wave with no lane inside the image skips the store instructions)
                }
                set_out_unit(cu);
#pragma unroll
                for (int n = 0; n < NR; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
            }
            const _Float16* spatch = s_mem + (s & 1) * STAGE_H;
            const _Float16* sw = spatch + PATCH_H;
            if constexpr (ABL == 0 && SRT_C8_PDE > 0) {
                // Operand reads run PD taps ahead of the MFMAs that use them.  Left to itself the compiler issues a tap's two ds_read_b128 right in front of its MFMA
                // (s_waitcnt lgkmcnt(1..2) before every one of the 25): a 32-cycle MFMA behind a ~130-cycle LDS round trip, with one other wave per SIMD to fill it.
                // The sched_group_barrier pairs pin the order "reads of tap t + PD, then the MFMAs of tap t"; same MFMA chain, same results.
                constexpr int PD = SRT_C8_PDE;
                h8 af[PD + 1], bfr[PD + 1][NR];
                auto ld = [&](int tap) __attribute__((always_inline)) {
                    const int ky = tap / 5, kx = tap % 5, sl = tap % (PD + 1);
                    const int koff = (ky * ROWS + ((kx + 1) & 1) * PWH + ((kx + 3) >> 1)) * 8;
                    af[sl] = *reinterpret_cast<const h8*>(sw + tap * 512 + aoff);
#pragma unroll
                    for (int n = 0; n < NR; ++n) bfr[sl][n] = *reinterpret_cast<const h8*>(spatch + boff[n] + koff);
                };
#pragma unroll
                for (int t = 0; t < PD; ++t) { ld(t); c8_sgb_reads(1 + NR); }
#pragma unroll
                for (int tap = 0; tap < 25; ++tap) {
                    if (tap + PD < 25) { ld(tap + PD); c8_sgb_reads(1 + NR); }
#pragma unroll
                    for (int n = 0; n < NR; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap % (PD + 1)], bfr[tap % (PD + 1)][n], acc[n], 0, 0, 0);
                    c8_sgb_mfma(NR);
                }
            } else if (!(ABL & 4)) {
#pragma unroll
                for (int tap = 0; tap < 25; ++tap) {
                    const int ky = tap / 5, kx = tap % 5;
                    const int koff = (ky * ROWS + ((kx + 1) & 1) * PWH + ((kx + 3) >> 1)) * 8;
                    h8 a;
                    if (ABL & 16) { for (int q = 0; q < 8; ++q) a[q] = (_Float16)(float)(tap + lane); }
                    else a = *reinterpret_cast<const h8*>(sw + tap * 512 + aoff);
#pragma unroll
                    for (int n = 0; n < NR; ++n) {
                        h8 b;
                        if (ABL & 16) { for (int q = 0; q < 8; ++q) b[q] = (_Float16)(float)(s + q + n); }
                        else b = *reinterpret_cast<const h8*>(spatch + boff[n] + koff);
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
                    }
                }
            }
            if (++ch == nch) { ch = 0; ++cu; }
        }
    }
    if (loader) __builtin_amdgcn_s_waitcnt(c8_vmcnt(0));
    if (worker) epilogue();
}

// ------------------------------------------------------------------------------------------- decoder, C8 in -> C8 out
// Transposed 5x5 stride-2 convolution as four parity classes of the output (srt_nn3.hip: srt_dec_f16); 32-pixel INPUT sub-tiles, all four classes accumulated
// from one B fragment per input shift.  CS (up5, Cout = 16): class-stacked weights - 32 rows = 2 x-classes x 16 channels, 15 (ky, dx)
// products per chunk, two accumulators (one per row class).  LW: see srt_enc_c8.
// WPE (waves per SIMD the kernel is built for): 1 - one workgroup per CU's worth of registers.  4 - TWO workgroups per CU (<= 128 VGPRs: the epilogue constants
// come from LDS instead of 48 registers): two independent barrier domains on a CU, so one workgroup's epilogue / DMA issue runs under the other's MFMAs - the overlap
// the eight lock-stepped waves of one workgroup cannot give each other (ablation matrix in DESIGN.md 3.2).
// NRW (LW = 0): sub-tiles per wave.  2 - a unit is SIXTEEN 32-pixel sub-tiles (16 x 32 input pixels): every A fragment read from LDS feeds two MFMAs, the 25 KiB weight slab of a
// K step - 70 % of the bytes a step moves into LDS - is amortised over twice the outputs, and there is one barrier per 50 MFMAs of a wave instead of per 25.
template <int SW, int NSY, int NI, int ST, bool CS, int LW, int ABL = 0, int WPE = 1, int NRW = 1>
__global__ void __launch_bounds__(512, WPE) srt_dec_c8(const SrtConvParams p, int tpw)
{
    constexpr bool EPI_LDS = WPE >= 4;
    constexpr int NLW = LW ? 4 : 8, NR = LW ? 2 : NRW;
    static_assert(NSY * NI == (LW ? 4 : 8) * NR && 32 % SW == 0 && ST >= 2 && ST <= 4 && (NRW == 1 || (NRW == 2 && !LW)), "eight (NRW = 2: sixteen) 32-pixel sub-tiles");
    constexpr int SH = 32 / SW, TH = NSY * SH, TW = SW;
    constexpr int PH = TH + 2, PC = TW + 2;
    constexpr int PLANE = NI * PH * PC;
    constexpr int PITEMS = 2 * PLANE, NPP = (PITEMS + 63) / 64, PPW = (NPP + NLW - 1) / NLW;
    constexpr int NT = CS ? 15 : 25, WPW = (NT + NLW - 1) / NLW;   // NT pieces of 1 KiB: the two k-group rows of one tap (or (ky, dx) pair)
    constexpr int PATCH_H = NPP * 512, WSLAB_H = NT * 512, STAGE_H = PATCH_H + WSLAB_H;
    constexpr int DPW = PPW + WPW;                             // DMA instructions per loader wave and step
    constexpr int NACC = CS ? 2 : 4;
    __shared__ __attribute__((aligned(16))) _Float16 s_mem[ST * STAGE_H + (EPI_LDS ? 192 : 0)];
    static_assert(sizeof(s_mem) * (EPI_LDS ? 2 : 1) <= 160 * 1024, "LDS");
    float* s_epi = reinterpret_cast<float*>(s_mem + ST * STAGE_H);          // EPI_LDS: bias | BN scale | BN shift of the workgroup's 32 rows

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = !LW || wave >= 4, worker = !LW || wave < 4;
    const int lw = LW ? (wave & 3) : wave;
    const int tilesX = (p.W + TW - 1) / TW, tilesY = (p.H + TH - 1) / TH, nsp = tilesX * tilesY;
    const int groups = (p.ntiles + NI - 1) / NI, nunits = nsp * groups;
    const int upw = (nunits + tpw - 1) / tpw, MBK = CS ? 1 : p.Cout / 32;
    const int pos = srt_xcd_order(upw * MBK * p.nstems);
    const int wsel = pos / upw, mblk = wsel % MBK, stem = wsel / MBK, m0 = mblk * 32;
    const int unit0 = (pos % upw) * tpw, unit1 = min(unit0 + tpw, nunits);
    const int nch = p.Cin / 16, nsteps = (unit1 - unit0) * nch;
    if (nsteps <= 0) return;
    const SrtAct actp = srt_act_params(srt_act_kind(p, stem), p.variant);
    const size_t hw = (size_t)p.H * p.W;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)s_mem;

    const _Float16* wp = CS ? (const _Float16*)(p.wpack16cs + stem * p.wpack16cs_stem)
                            : (const _Float16*)(p.wpack16 + stem * p.wpack16_stem) + (size_t)m0 * 8;
    const int CPW = CS ? 32 : p.CP;
    const size_t cgStride = (size_t)(2 * NT) * CPW * 8;
    unsigned wvoff[WPW], wm0[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int piece = min(lw + NLW * i, NT - 1);
        wvoff[i] = (unsigned)(((2 * piece + g) * CPW + l31) * 16);
        wm0[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(PATCH_H * 2 + piece * 1024));
    }
    // patch slot e = ((gg * NI + il) * PH + r) * PC + col <- k-group gg of the chunk, instance tile0 + il, input row ty0 - 1 + r, column tx0 - 1 + col
    unsigned pvoff[PPW], pm0[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) pm0[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(min(lw + NLW * i, NPP - 1) * 1024));
    const unsigned nrec = (unsigned)min((size_t)0x7fffffff, (size_t)2 * NI * p.srcA_tile);     // srcA_tile == srcB_tile (launcher)
    const _Float16* pa = nullptr; const _Float16* pb = nullptr;
    auto set_dma_unit = [&](int unit) {
        const int sp = unit % nsp, tile0 = (unit / nsp) * NI, tx0 = (sp % tilesX) * TW, ty0 = (sp / tilesX) * TH;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int e = min(lw + NLW * i, NPP - 1) * 64 + lane;
            const int gg = e / PLANE, rem = e % PLANE, col = rem % PC, r = (rem / PC) % PH, il = rem / (PC * PH);
            const int gy = ty0 - 1 + r, gx = tx0 - 1 + col;
            const bool ok = e < PITEMS && tile0 + il < p.ntiles && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            pvoff[i] = ok ? (unsigned)(2 * (size_t)il * p.srcA_tile + 16 * ((size_t)gg * hw + (size_t)gy * p.W + gx)) : C8_OOR;
        }
        pa = reinterpret_cast<const _Float16*>(p.srcA) + stem * p.srcA_stem + tile0 * p.srcA_tile;
        pb = reinterpret_cast<const _Float16*>(p.srcB) + stem * p.srcB_stem + tile0 * p.srcB_tile;
    };
    const int chA = p.CA / 16;                                 // chunks [0, chA) read the skip tensor, the rest the previous decoder output
    const unsigned chunk_bytes = (unsigned)(32 * hw);
    auto issue_dma = [&](int ch, int stage, bool primed) {
        const unsigned sb = (unsigned)stage * (unsigned)(STAGE_H * 2);
        const bool fromA = ch < chA;
        const i32x4 rs = c8_rsrc(fromA ? pa : pb, nrec);
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(fromA ? ch : ch - chA) * chunk_bytes));
        const _Float16* ws = wp + (size_t)ch * cgStride;
        if (!((ABL & 2) && primed)) {
#pragma unroll
            for (int i = 0; i < WPW; ++i) c8_dma_global(wvoff[i], ws, wm0[i] + sb);
        }
        if (!((ABL & 1) && primed)) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) c8_dma_buffer(pvoff[i], rs, soff, pm0[i] + sb);
        }
    };

    int boff[NR], il_w[NR], a_l[NR];
    const int b_l = l31 % SW;
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int st = NR * (LW ? (wave & 3) : wave) + n;
        il_w[n] = st / NSY; a_l[n] = (st % NSY) * SH + l31 / SW;
        boff[n] = (g * PLANE + (il_w[n] * PH + a_l[n] + 1) * PC + b_l + 1) * 8;
    }
    const int aoff = (g * 32 + l31) * 8;
    const int Wo = p.W << 1;
    const size_t ohw = (size_t)(p.H << 1) * Wo;
    float bi[16], sc[16], sf[16];
    if constexpr (EPI_LDS) {
        if (tid < 32) {                                        // (visible after the K loop's first barrier; first read in the first epilogue)
            const size_t ci = stem * p.coeff_stem + (CS ? tid % 16 : m0 + tid);
            s_epi[tid] = p.bias[ci]; s_epi[32 + tid] = p.bnScale[ci]; s_epi[64 + tid] = p.bnShift[ci];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
            const size_t ci = stem * p.coeff_stem + (CS ? row % 16 : m0 + row);
            bi[r] = p.bias[ci]; sc[r] = p.bnScale[ci]; sf[r] = p.bnShift[ci];
        }
    }
    auto load_epi = [&]() __attribute__((always_inline)) {    // EPI_LDS: the lane's 16 rows are four runs of four floats (rows 8 q + 4 g + 0..3)
        if constexpr (EPI_LDS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(s_epi + 8 * q + 4 * g), s4 = *reinterpret_cast<const float4*>(s_epi + 32 + 8 * q + 4 * g),
                             f4 = *reinterpret_cast<const float4*>(s_epi + 64 + 8 * q + 4 * g);
                bi[4 * q] = b4.x; bi[4 * q + 1] = b4.y; bi[4 * q + 2] = b4.z; bi[4 * q + 3] = b4.w;
                sc[4 * q] = s4.x; sc[4 * q + 1] = s4.y; sc[4 * q + 2] = s4.z; sc[4 * q + 3] = s4.w;
                sf[4 * q] = f4.x; sf[4 * q + 1] = f4.y; sf[4 * q + 2] = f4.z; sf[4 * q + 3] = f4.w;
            }
        }
    };
    _Float16* outh = reinterpret_cast<_Float16*>(p.outAct);
    size_t obase[NR]; bool pix_ok[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) { obase[n] = 0; pix_ok[n] = false; }
    auto set_out_unit = [&](int unit) {
        const int sp = unit % nsp;
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int tile = (unit / nsp) * NI + il_w[n], a = (sp / tilesX) * TH + a_l[n], b = (sp % tilesX) * TW + b_l;
            pix_ok[n] = tile < p.ntiles && a < p.H && b < p.W;
            const size_t pix = pix_ok[n] ? (size_t)(2 * a) * Wo + 2 * b : 0;
            // C8 out: the lane stores output column 2 b + g of its pixel pair (c8_pair16); CS: the layer's 16 channels are channel groups 0 and 1
            obase[n] = stem * p.out_stem + (pix_ok[n] ? tile : 0) * p.out_tile + ((size_t)(CS ? 0 : m0 / 8) * ohw + pix + (pix_ok[n] ? g : 0)) * 8;
        }
    };
    f32x16 acc[NR][NACC];
    auto epilogue = [&]() {
        // (no early return for lanes outside the image: the lane exchanges need every lane; their stores are masked)
        load_epi();
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            if constexpr (CS) {
                // rows of the MFMA tile: m = 8 q + 4 g + j = px * 16 + co  ->  px = q >> 1, co = 8 (q & 1) + 4 g + j: the lane's four values of (q, py) are half of the
                // C8 slot (channel group q & 1, output column 2 b + px), lane + 32 holds the other half - the same exchange as below with the two x-classes as the
                // slot pair: 16-byte stores, 1 KiB of whole lines per wave instruction (round 6: up5's output is C8 too; srt_up6_* read it as B fragments)
#pragma unroll
                for (int cgp = 0; cgp < 2; ++cgp)
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        h4 v[2];
#pragma unroll
                        for (int px = 0; px < 2; ++px)
#pragma unroll
                            for (int j = 0; j < 4; j += 2) {
                                const int r = 4 * (2 * px + cgp) + j;
                                const f2 o = c8_dec_pair(F2(acc[n][py][r], acc[n][py][r + 1]), F2(bi[r], bi[r + 1]), F2(sc[r], sc[r + 1]), F2(sf[r], sf[r + 1]), actp);
                                v[px][j] = (_Float16)o.x; v[px][j + 1] = (_Float16)o.y;
                            }
                        const u32x4 o16 = c8_pair16(v[0], v[1]);
                        if (pix_ok[n]) *reinterpret_cast<u32x4*>(outh + obase[n] + ((size_t)cgp * ohw + (size_t)py * Wo) * 8) = o16;
                    }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        h4 v[2];
#pragma unroll
                        for (int px = 0; px < 2; ++px)
#pragma unroll
                            for (int j = 0; j < 4; j += 2) {
                                const int r = 4 * q + j;
                                const f2 o = c8_dec_pair(F2(acc[n][py * 2 + px][r], acc[n][py * 2 + px][r + 1]), F2(bi[r], bi[r + 1]), F2(sc[r], sc[r + 1]), F2(sf[r], sf[r + 1]), actp);
                                v[px][j] = (_Float16)o.x; v[px][j + 1] = (_Float16)o.y;
                            }
                        const u32x4 o16 = c8_pair16(v[0], v[1]);
                        if (pix_ok[n]) *reinterpret_cast<u32x4*>(outh + obase[n] + ((size_t)q * ohw + (size_t)py * Wo) * 8) = o16;
                    }
            }
        }
    };

    int du = unit0, dch = 0, cu = unit0, ch = 0, issued = 0;
    constexpr int NST = CS ? 4 : 8;                            // LW = 0: store instructions of one sub-tile's epilogue (4 channel groups x 2 rows of 16 B; class-stacked: 2 groups x 2 rows); pinned by tests/test_abi.py
    int pending = 0;                                           // sub-tiles whose stores this wave issued in the previous step (wave-uniform; never on a loader-only wave)
    auto issue_next = [&]() {                                  // the step after the last one issued (always exactly DPW instructions: past the end, the last step again - harmless, its stage is free)
        issue_dma(dch, issued % ST, issued >= ST);
        ++issued;
        if (issued < nsteps && ++dch == nch) { dch = 0; set_dma_unit(++du); }
    };
    if (loader) {
        set_dma_unit(unit0);
        for (int i = 0; i < ST - 1; ++i) issue_next();
    }
    for (int s = 0; s < nsteps; ++s) {
        if (!(ABL & 32) || s == 0) {
            if (loader) {
                // everything older than this wave's pieces of the ST-2 newest steps has landed: step s (+ the NST epilogue stores of the previous step, issued
                // behind its DMA: vmcnt retires in issue order, see srt_enc_c8)
                if (NR == 2 && pending == 2) __builtin_amdgcn_s_waitcnt(c8_vmcnt((ABL & 3) ? 0 : (ST - 2) * DPW + 2 * NST));
                else if (pending) __builtin_amdgcn_s_waitcnt(c8_vmcnt((ABL & 3) ? 0 : (ST - 2) * DPW + NST));
                else __builtin_amdgcn_s_waitcnt(c8_vmcnt((ABL & 3) ? 0 : (ST - 2) * DPW));
            }
            __syncthreads();
        }
        pending = 0;
        if (loader) issue_next();                              // step s + ST - 1 into the stage step s - 1 just left
        if (worker) {
            if (ch == 0) {
                if (s > 0 && !(ABL & 8)) {
                    epilogue();
                    if (!LW) {                                 // (a sub-tile with no pixel inside the image may or may not have issued its masked stores: counting it out errs on the waiting side)
#pragma unroll
                        for (int n = 0; n < NR; ++n) pending += __builtin_amdgcn_ballot_w64(pix_ok[n]) != 0 ? 1 : 0;
                    }
                }
                set_out_unit(cu);
#pragma unroll
                for (int n = 0; n < NR; ++n)
#pragma unroll
                    for (int c = 0; c < NACC; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[n][c][r] = 0.0f;
            }
            const _Float16* spatch = s_mem + (s % ST) * STAGE_H;
            const _Float16* sw = spatch + PATCH_H;
            if constexpr (ABL == 0 && SRT_C8_PDA > 0) {
                // operand reads ahead of their MFMAs (see srt_enc_c8): A fragments PDA taps ahead, a shift's B fragments when the shift PDB before it starts; the flat tap
                // order is the shift-major order of the loop below (C8DecOrder), so every accumulator sees the same MFMA chain
                constexpr C8DecOrder<CS> O{};
                static_assert(O.n == NT, "tap count");
                constexpr int PDA = SRT_C8_PDA, PDB = SRT_C8_PDB > 0 ? SRT_C8_PDB : 1;
                h8 af[PDA + 1], bfr[PDB + 1][NR];
                auto lda = [&](int t) __attribute__((always_inline)) { af[t % (PDA + 1)] = *reinterpret_cast<const h8*>(sw + O.aidx[t] * 512 + aoff); };
                auto ldb = [&](int sft) __attribute__((always_inline)) {
#pragma unroll
                    for (int n = 0; n < NR; ++n) bfr[sft % (PDB + 1)][n] = *reinterpret_cast<const h8*>(spatch + boff[n] + ((sft / 3 - 1) * PC + (sft % 3 - 1)) * 8);
                };
#pragma unroll
                for (int sft = 0; sft < PDB; ++sft) { ldb(sft); c8_sgb_reads(NR); }
#pragma unroll
                for (int t = 0; t < PDA; ++t) { lda(t); c8_sgb_reads(1); }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (O.first[t] && O.sh[t] + PDB < 9) { ldb(O.sh[t] + PDB); c8_sgb_reads(NR); }
                    if (t + PDA < NT) { lda(t + PDA); c8_sgb_reads(1); }
#pragma unroll
                    for (int n = 0; n < NR; ++n)
                        acc[n][O.acc[t]] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t % (PDA + 1)], bfr[O.sh[t] % (PDB + 1)][n], acc[n][O.acc[t]], 0, 0, 0);
                    c8_sgb_mfma(NR);
                }
            } else
#pragma unroll
            for (int sh = 0; sh < ((ABL & 4) ? 0 : 9); ++sh) { // shift-major: one B fragment per sub-tile and input shift
                const int dy = sh / 3 - 1, dx = sh % 3 - 1;
                h8 b[NR];
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    if (ABL & 16) { for (int q = 0; q < 8; ++q) b[n][q] = (_Float16)(float)(s + q + lane + n); }
                    else b[n] = *reinterpret_cast<const h8*>(spatch + boff[n] + (dy * PC + dx) * 8);
                }
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    const int py = (ky + 1) & 1;
                    if ((py + 1 - ky) / 2 != dy) continue;
                    if constexpr (CS) {
                        h8 a;
                        if (ABL & 16) { for (int q = 0; q < 8; ++q) a[q] = (_Float16)(float)(ky + lane); }
                        else a = *reinterpret_cast<const h8*>(sw + (ky * 3 + dx + 1) * 512 + aoff);
#pragma unroll
                        for (int n = 0; n < NR; ++n) acc[n][py] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[n], acc[n][py], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int kx = 0; kx < 5; ++kx) {
                            const int px = (kx + 1) & 1;
                            if ((px + 1 - kx) / 2 != dx) continue;
                            h8 a;
                            if (ABL & 16) { for (int q = 0; q < 8; ++q) a[q] = (_Float16)(float)(ky * 5 + kx + lane); }
                            else a = *reinterpret_cast<const h8*>(sw + (ky * 5 + kx) * 512 + aoff);
#pragma unroll
                            for (int n = 0; n < NR; ++n) acc[n][py * 2 + px] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[n], acc[n][py * 2 + px], 0, 0, 0);
                        }
                    }
                }
            }
            if (++ch == nch) { ch = 0; ++cu; }
        }
    }
    if (loader) __builtin_amdgcn_s_waitcnt(c8_vmcnt(0));       // the tail's harmless extra DMA must not outlive the workgroup's LDS
    if (worker) epilogue();
}

// ------------------------------------------------------------------------------------------- dispatch
#ifdef SRT_TUNING
static int c8_abl() { const char* t = getenv("SRT_TUNE_C8"); return t ? atoi(t) : 0; }
#define C8_ABL_CASES(X) X(3) X(4) X(8) X(16) X(12) X(32)
#endif
static int c8_env(const char* name, int dflt)
{
    const char* t = getenv(name);
    return t ? atoi(t) : dflt;
}
// Two decoder workgroups per CU (WPE = 4) measured EQUAL to one (round 6, same box, 5 stems: step 5.006 vs 4.997 ms; up3 0.301 vs 0.322, up5 0.499 vs 0.514, the rest
// within 2 % either way): the second barrier domain does not buy the overlap the ablation matrix prices.  Tuning library only (SRT_TUNE_C8WPE=4).
#ifdef SRT_TUNING
static bool c8_wpe4() { static const bool v = c8_env("SRT_TUNE_C8WPE", 1) == 4; return v; }
#endif
// Workgroups per launch (each walks its run of units as one stream of K steps).  Measured per layer at 64 tiles x 5 stems (round 6, SPLEETERRT_C8_WGS = 512 .. 4096 on one
// box, profiles/r06_c8_wgs_sweep.json): the short-K full-resolution layers want long runs (down2 512: 0.357 ms against 0.375 at 1024; down3 / down4 768), the deep encoder layers
// and the two-sub-tile decoder layers want many short ones (down5 / down6 1536: 0.181 against 0.196 / 0.199; up2..up4 1536: 0.270 / 0.270 / 0.313 against 0.298 / 0.298 / 0.347),
// the class-stacked up5 1280.  SPLEETERRT_C8_WGS=<n> overrides them all (the sweep).
// None of that transfers from five stems to four (there 1024 is as good as anything for most layers and 512 the best for some): what decides is how the runs fall
// on the 256 CUs.  So the choice is MEASURED: the first launch of a layer shape times the candidates (c8_tuned below) and the process keeps the winner.
static int c8_target_wgs(int dflt) { const int v = c8_env("SPLEETERRT_C8_WGS", 0); return v > 0 ? v : dflt; }
// The loader-wave form (LW = 1) measured SLOWER on every layer but down5 / down6 (round 6, same box: 5-stem step 5.26 vs 5.09 ms; up4 0.434 vs 0.385): with one
// computing wave per SIMD the MFMA stream loses more to its own LDS-read latency than the other waves lose to the DMA issue.  It is compiled into the tuning
// library only (SRT_TUNE_C8LW=1); the product instantiates LW = 0.
#ifdef SRT_TUNING
static bool c8_lw() { static const bool v = c8_env("SRT_TUNE_C8LW", 0) != 0; return v; }
#define C8_LW(then_, else_) do { if (c8_lw()) { then_; } else { else_; } } while (0)
#else
#define C8_LW(then_, else_) do { else_; } while (0)
#endif
// units per workgroup: about `target` workgroups per launch, never across a (stem, M block) boundary
static int c8_tpw(int nunits, int pairs, int target)
{
    const int upw = target / pairs > 0 ? target / pairs : 1;
    const int tpw = (nunits + upw - 1) / upw;
    return tpw < 1 ? 1 : tpw;
}

// target: workgroups per launch to aim for; 0 = SPLEETERRT_C8_WGS or the layer's table value
static int enc_c8_launch(const SrtConvParams& p, hipStream_t s, int target)
{
    if (!p.wpack16 || p.Cin % 16 || p.Cout % 32 || !p.in16 || !p.out16 || p.nsplit == 2 || p.inScale) return 1;
    const int Ho = p.H / 2, Wo = p.W / 2, pairs = (p.Cout / 32) * p.nstems;
    if (Wo > 16) {
        constexpr int TH = 8, TW = 32, NI = 1;
        if (target <= 0) target = c8_target_wgs(p.Cin <= 64 ? 768 : 1536);
        const int nunits = ((Wo + TW - 1) / TW) * ((Ho + TH - 1) / TH) * ((p.ntiles + NI - 1) / NI), tpw = c8_tpw(nunits, pairs, target);
        const dim3 grid((unsigned)(((nunits + tpw - 1) / tpw) * pairs));
#ifdef SRT_TUNING
#define C8_ENC_CASE(A) if (c8_abl() == A) { SRT_LAUNCH((srt_enc_c8<32, 8, 1, 0, A>), grid, dim3(512), 0, s, p, tpw); return srt_launch_status(); }
        C8_ABL_CASES(C8_ENC_CASE)
#endif
        const int tpwf = tpw | (c8_env("SPLEETERRT_C8_WRES", 1) != 0 ? 0x10000 : 0);         // (=0: the weight slab is moved every step even when a unit is one K chunk - A/B runs)
        C8_LW(SRT_LAUNCH((srt_enc_c8<32, 8, 1, 1>), grid, dim3(512), 0, s, p, tpwf), SRT_LAUNCH((srt_enc_c8<32, 8, 1, 0>), grid, dim3(512), 0, s, p, tpwf));
    } else {
        constexpr int TH = 4, TW = 16, NI = 4;
        if (target <= 0) target = c8_target_wgs(1536);
        const int nunits = ((Wo + TW - 1) / TW) * ((Ho + TH - 1) / TH) * ((p.ntiles + NI - 1) / NI), tpw = c8_tpw(nunits, pairs, target);
        const dim3 grid((unsigned)(((nunits + tpw - 1) / tpw) * pairs));
        C8_LW(SRT_LAUNCH((srt_enc_c8<16, 2, 4, 1>), grid, dim3(512), 0, s, p, tpw), SRT_LAUNCH((srt_enc_c8<16, 2, 4, 0>), grid, dim3(512), 0, s, p, tpw));
    }
    return srt_launch_status();
}

static int dec_c8_launch(const SrtConvParams& p, hipStream_t s, int target)
{
    const bool cs = p.Cout == 16;
    if (p.Cin % 16 || p.CA % 16 || !p.in16 || !p.out16 || p.nsplit == 2 || (cs ? !p.wpack16cs : (!p.wpack16 || p.Cout % 32 != 0))) return 1;
    if (p.CA < p.Cin && p.srcA_tile != p.srcB_tile) return -1;
    const int pairs = (cs ? 1 : p.Cout / 32) * p.nstems;
    if (p.W > 16) {
        constexpr int TH = 8, TW = 32, NI = 1;
        if (target <= 0) target = c8_target_wgs(cs ? 1280 : 1536);
        const int nunits = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * ((p.ntiles + NI - 1) / NI), tpw = c8_tpw(nunits, pairs, target);
        const dim3 grid((unsigned)(((nunits + tpw - 1) / tpw) * pairs));
#ifdef SRT_TUNING
#define C8_DEC_CASE(A) if (c8_abl() == A) { if (cs) SRT_LAUNCH((srt_dec_c8<32, 8, 1, 3, true, 0, A>), grid, dim3(512), 0, s, p, tpw); else SRT_LAUNCH((srt_dec_c8<32, 8, 1, 3, false, 0, A>), grid, dim3(512), 0, s, p, tpw); return srt_launch_status(); }
        C8_ABL_CASES(C8_DEC_CASE)
#endif
#ifdef SRT_TUNING
        if (c8_wpe4()) {
            if (cs) SRT_LAUNCH((srt_dec_c8<32, 8, 1, 3, true, 0, 0, 4>), grid, dim3(512), 0, s, p, tpw);
            else SRT_LAUNCH((srt_dec_c8<32, 8, 1, 2, false, 0, 0, 4>), grid, dim3(512), 0, s, p, tpw);
            return srt_launch_status();
        }
#endif
        // Two sub-tiles per wave (NRW = 2: units of 16 x 32 input pixels), round 6, same box, 5 stems: up3 0.327 -> 0.297 ms, up4 0.380 -> 0.346, up2 (one tile row per
        // instance) 0.297 -> 0.297.  The class-stacked up5 measured SLOWER in this form (0.445 -> 0.487: its 15 KiB slab is the smaller part of a step and the unit's
        // epilogue doubles), so it keeps one sub-tile per wave; its NRW = 2 instantiation is in the tuning library (SPLEETERRT_C8_NR2=3).  =0: one sub-tile everywhere (A/B runs).
        const int nr2 = c8_env("SPLEETERRT_C8_NR2", 1);
#ifndef SRT_TUNING
        const bool cs2 = false;
#else
        const bool cs2 = cs && (nr2 & 2);
#endif
        if (p.H % 16 == 0 && (cs ? cs2 : (nr2 & 1) != 0)) {
            const int nunits2 = ((p.W + TW - 1) / TW) * (p.H / 16) * p.ntiles, tpw2 = c8_tpw(nunits2, pairs, target);
            const dim3 grid2((unsigned)(((nunits2 + tpw2 - 1) / tpw2) * pairs));
#ifdef SRT_TUNING
            if (cs) { SRT_LAUNCH((srt_dec_c8<32, 16, 1, 3, true, 0, 0, 1, 2>), grid2, dim3(512), 0, s, p, tpw2); return srt_launch_status(); }
#endif
            SRT_LAUNCH((srt_dec_c8<32, 16, 1, 3, false, 0, 0, 1, 2>), grid2, dim3(512), 0, s, p, tpw2);
            return srt_launch_status();
        }
        if (cs) C8_LW(SRT_LAUNCH((srt_dec_c8<32, 8, 1, 3, true, 1>), grid, dim3(512), 0, s, p, tpw), SRT_LAUNCH((srt_dec_c8<32, 8, 1, 3, true, 0>), grid, dim3(512), 0, s, p, tpw));
        else C8_LW(SRT_LAUNCH((srt_dec_c8<32, 8, 1, 3, false, 1>), grid, dim3(512), 0, s, p, tpw), SRT_LAUNCH((srt_dec_c8<32, 8, 1, 3, false, 0>), grid, dim3(512), 0, s, p, tpw));
    } else {
        constexpr int TH = 4, TW = 16, NI = 4;
        if (target <= 0) target = c8_target_wgs(1024);
        const int nunits = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * ((p.ntiles + NI - 1) / NI), tpw = c8_tpw(nunits, pairs, target);
        const dim3 grid((unsigned)(((nunits + tpw - 1) / tpw) * pairs));
#ifdef SRT_TUNING
        if (c8_wpe4()) {
            if (cs) SRT_LAUNCH((srt_dec_c8<16, 2, 4, 2, true, 0, 0, 4>), grid, dim3(512), 0, s, p, tpw);
            else SRT_LAUNCH((srt_dec_c8<16, 2, 4, 2, false, 0, 0, 4>), grid, dim3(512), 0, s, p, tpw);
            return srt_launch_status();
        }
#endif
        if (cs) C8_LW(SRT_LAUNCH((srt_dec_c8<16, 2, 4, 3, true, 1>), grid, dim3(512), 0, s, p, tpw), SRT_LAUNCH((srt_dec_c8<16, 2, 4, 3, true, 0>), grid, dim3(512), 0, s, p, tpw));
        else C8_LW(SRT_LAUNCH((srt_dec_c8<16, 2, 4, 3, false, 1>), grid, dim3(512), 0, s, p, tpw), SRT_LAUNCH((srt_dec_c8<16, 2, 4, 3, false, 0>), grid, dim3(512), 0, s, p, tpw));
    }
    return srt_launch_status();
}

// ------------------------------------------------------------------------------------------- measured workgroup counts
// How many workgroups a C8 launch is cut into changes NOTHING in its results (a workgroup's run of units is only a partition of the same work) and up to 10 % of its
// time, in a way that follows from how the runs fall on the 256 CUs and differs between four and five stems (see c8_target_wgs).  The first launch of a layer shape
// therefore times the candidates on the caller's own tensors - one untimed launch each, then the best of three - and the process keeps the winner (about 5 ms per
// layer shape, once).  No measuring inside a stream capture (the table value is used), none under SPLEETERRT_C8_WGS=<n> or SPLEETERRT_C8_TUNE=0.
struct C8Key {
    int v[10];
    bool operator<(const C8Key& o) const { for (int i = 0; i < 10; ++i) if (v[i] != o.v[i]) return v[i] < o.v[i]; return false; }
};
static std::map<C8Key, int> g_c8_best;
static std::mutex g_c8_mu;
static int c8_tuned(int kind, const SrtConvParams& p, hipStream_t s)
{
    auto go = [&](int target) { return kind ? dec_c8_launch(p, s, target) : enc_c8_launch(p, s, target); };
    // The bandwidth-heavy full-resolution layers (down2, down3, the class-stacked up5) are NOT measured: timed back to back on warm inputs they rank the candidates
    // differently from how they run behind their producer in a step (round 6, 4 stems: the isolated winner was 8 % / 4 % / 3 % slower in the step); they keep table values
    // that are within 2 % of the best at both four and five stems (768 / 768 / 1280).
    const bool measured = kind ? p.Cout != 16 : p.Cin >= 64;
    if (!measured || c8_env("SPLEETERRT_C8_WGS", 0) > 0 || c8_env("SPLEETERRT_C8_TUNE", 1) == 0) return go(0);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const C8Key key = {{ kind, p.Cin, p.Cout, p.H, p.W, p.ntiles, p.nstems, (p.outAct && p.bnScale) ? 1 : 0, c8_env("SPLEETERRT_C8_NR2", 1) * 2 + (c8_env("SPLEETERRT_C8_WRES", 1) != 0), dev }};
    {
        std::lock_guard<std::mutex> lk(g_c8_mu);
        const auto it = g_c8_best.find(key);
        if (it != g_c8_best.end()) return go(it->second);
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s && (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) { (void)hipGetLastError(); return go(0); }
    const int rc0 = go(0);                                       // (not covered / a launch error: nothing to tune)
    if (rc0) return rc0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) hipEventDestroy(e0); (void)hipGetLastError(); return 0; }
    static const int cands[] = { 512, 768, 1024, 1280, 1536, 2048 };
    int best = 0; float best_ms = 1e30f;
    for (int c : cands) {
        if (go(c)) { best = 0; break; }
        float lo = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            float ms = 0.0f;
            if (hipEventRecord(e0, s) != hipSuccess || go(c) || hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { lo = 1e30f; break; }
            lo = ms < lo ? ms : lo;
        }
        if (lo < best_ms) { best_ms = lo; best = c; }
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    (void)hipGetLastError();
    {
        std::lock_guard<std::mutex> lk(g_c8_mu);
        g_c8_best[key] = best;                                   // (0 if the timing failed: the table value from now on)
    }
    if (c8_env("SPLEETERRT_C8_TUNE", 1) >= 2) fprintf(stderr, "[spleeterrt_amd] C8 %s Cin %d Cout %d %dx%d x%d x%d: %d workgroups (%.3f ms)\n", kind ? "dec" : "enc", p.Cin, p.Cout, p.H, p.W, p.ntiles, p.nstems, best, best_ms);
    return 0;                                                    // (the layer's outputs are in place: every candidate wrote the same values)
}
int srt_launch_enc_c8(const SrtConvParams& p, hipStream_t s) { return c8_tuned(0, p, s); }
int srt_launch_dec_c8(const SrtConvParams& p, hipStream_t s) { return c8_tuned(1, p, s); }
