// srt_nn4.hip — Winograd form of the decoder's transposed convolutions (spleeter.c:239-294, 5x5 stride 2) on
// v_mfma_f32_16x16x4_f32.
//
// A stride-2 transposed 5x5 convolution is four ordinary convolutions, one per output parity class (py, px): along an axis
// the odd outputs see 3 taps (k = 4, 2, 0 at input shifts -1, 0, +1) and the even outputs 2 taps (k = 3, 1 at shifts -1, 0).
// The direct kernels (srt_nn2.hip) spend 9 + 6 + 6 + 4 = 25 multiply-adds per input pixel, channel pair and output channel.
// Winograd's minimal filtering over blocks of 2 x 2 input pixels (4 x 4 output pixels) needs F(2,3): 4 products per axis
// for the 3-tap classes and F(2,2): 3 products for the 2-tap ones, i.e. 16 + 12 + 12 + 9 = 49 products per block against
// 100: 2.04x fewer MFMA cycles.  With xi = 0..48 the transform point (class-major, then row-major inside the class):
//
//      M[xi][co][block] = sum over ci of  U[xi][ci][co] * V[xi][ci][block]            49 independent GEMMs, K = Cin
//
//   U = G_y g G_x^T   the transformed weights: computed ONCE per srtSetCoeff by srt_pack_wino_kernel (exact halves and sums
//                     of the taps), stored [Cin/4][Cout/16][4][16][52] so that a workgroup's slab of a 4-channel K step is
//                     13 KiB of contiguous memory, moved by 13 LDS-DMA instructions;
//   V = B_y X B_x^T   the transformed 4 x 4 input patch (rows a0-1..a0+2, columns b0-1..b0+2) of the block: additions and
//                     subtractions only, computed in registers by the lane that needs it as its MFMA B operand (see the kernel);
//   Y = A_y M A_x^T   the 2 x 2 outputs of the class, then bias -> activation -> batch-norm (spleeter.c:244-245).
//
// Workgroup = 16 output channels x 64 blocks, 8 waves: 4 groups of 16 blocks x 2 output-row parities (the kernel's comment has
// the details and DESIGN.md section 3.2a the measurements that led there).  Used for up2..up5 when the batch has more than 16
// instances; smaller batches and up1 stay on the direct kernels of srt_nn2.hip.
//
// Numerics: U is exact up to one rounding per point (sums of at most 9 weights, scaled by 1/2 or 1/4); V and Y add one or two
// roundings of additions.  The F(2,3)/F(2,2) matrices have entries 0, +-1, 1/2 only, so there is none of the cancellation
// that larger Winograd tiles are known for; the layer outputs agree with the CPU reference to ~1e-6 relative (tests/test_gpu_parity.py).
#include "srt_device.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WINO_LD 52                 // 49 points padded to 13 float4
#define WINO_C11 0                 // class (py=1, px=1): 4 x 4 points
#define WINO_C10 16                // class (1, 0): 4 x 3
#define WINO_C01 28                // class (0, 1): 3 x 4
#define WINO_C00 40                // class (0, 0): 3 x 3

// ------------------------------------------------------------------------------------------- weight transform
// 1-D: p = 1 taps (k = 4, 2, 0) -> [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2];  p = 0 taps (k = 3, 1) -> [g0, g0+g1, g1]
// ENC: the taps in ascending order (stride-2 convolution of the encoder: odd input plane k = 0, 2, 4, even plane k = 1, 3) instead of the
// transposed convolution's descending one
template <bool ENC = false>
__device__ __forceinline__ int wino_w1d(int p, const float* g, int stride, float* o)      // returns the number of points
{
    if (p) {
        const float g0 = g[(ENC ? 0 : 4) * stride], g1 = g[2 * stride], g2 = g[(ENC ? 4 : 0) * stride];
        o[0] = g0; o[1] = 0.5f * ((g0 + g2) + g1); o[2] = 0.5f * ((g0 + g2) - g1); o[3] = g2;
        return 4;
    }
    const float g0 = g[(ENC ? 1 : 3) * stride], g1 = g[(ENC ? 3 : 1) * stride];
    o[0] = g0; o[1] = g0 + g1; o[2] = g1;
    return 3;
}
// ENC: w is the encoder's OIHW [Cout][Cin][5][5]; otherwise the decoder's [Cin][Cout][5][5].  Same packed layout either way.
template <bool ENC>
__global__ void srt_pack_wino_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx / Cout, co = idx % Cout;
    const float* g = w + (ENC ? (size_t)co * Cin + ci : (size_t)ci * Cout + co) * 25;      // [ky][kx]
    float out[WINO_LD];
    int n = 0;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        const int py = cls < 2, px = !(cls & 1);                             // (1,1) (1,0) (0,1) (0,0)
        float rowt[5][4];                                                    // x transform of each kernel row ky
        int nx = 0;
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) nx = wino_w1d<ENC>(px, g + ky * 5, 1, rowt[ky]);
        const int ny = py ? 4 : 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= nx) continue;
            float col[5], t[4];
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) col[ky] = rowt[ky][j];
            wino_w1d<ENC>(py, col, 1, t);
#pragma unroll
            for (int i = 0; i < 4; ++i) if (i < ny) out[n + i * nx + j] = t[i];
        }
        n += ny * nx;
    }
    out[49] = out[50] = out[51] = 0.0f;
    const int MB = Cout / 16;
    float* dst = u + ((((size_t)(ci / 4) * MB + co / 16) * 4 + (ci & 3)) * 16 + (co & 15)) * WINO_LD;
#pragma unroll
    for (int q = 0; q < WINO_LD / 4; ++q) *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
}
int srt_launch_pack_wino(const float* w, float* u, int Cin, int Cout, hipStream_t s)
{
    if (Cin % 4 || Cout % 16) return -1;
    SRT_LAUNCH(srt_pack_wino_kernel<false>, dim3((Cin * Cout + 255) / 256), dim3(256), 0, s, w, u, Cin, Cout);
    return srt_launch_status();
}
int srt_launch_pack_wino_enc(const float* w, float* u, int Cin, int Cout, hipStream_t s)
{
    if (Cin % 4 || Cout % 16) return -1;
    SRT_LAUNCH(srt_pack_wino_kernel<true>, dim3((Cin * Cout + 255) / 256), dim3(256), 0, s, w, u, Cin, Cout);
    return srt_launch_status();
}

// ------------------------------------------------------------------------------------------- transforms (registers)
// input: B3 = [d0-d2, d1+d2, d2-d1, d1-d3] (4 points), B2 = [d0-d1, d1, d1-d2] (3 points, d3 unused)
__device__ __forceinline__ void wino_in3(float d0, float d1, float d2, float d3, float* o) { o[0] = d0 - d2; o[1] = d1 + d2; o[2] = d2 - d1; o[3] = d1 - d3; }
// (the middle point is d1 itself; it gets its own register - an opaque v_mov - so that the loaded patch registers die with the row
//  transforms and the next patch can be loaded straight into them: otherwise the value lives on as t2[r][1] until the last quad, the
//  loads land elsewhere and a copy at the loop end waits for them)
__device__ __forceinline__ void wino_in2(float d0, float d1, float d2, float* o)
{
    float c; asm("v_mov_b32 %0, %1" : "=v"(c) : "v"(d1));
    o[0] = d0 - d1; o[1] = c; o[2] = d1 - d2;
}
// output: A3: y0 = m0+m1+m2, y1 = m1-m2-m3;  A2: y0 = m0+m1, y1 = m1-m2
template <int N> __device__ __forceinline__ void wino_out1d(const float* m, int stride, float& y0, float& y1)
{
    if (N == 4) { y0 = (m[0] + m[stride]) + m[2 * stride]; y1 = (m[stride] - m[2 * stride]) - m[3 * stride]; }
    else { y0 = m[0] + m[stride]; y1 = m[stride] - m[2 * stride]; }
}
template <int NY, int NX> __device__ __forceinline__ void wino_out2d(const float* m, float (&y)[2][2])      // m[NY][NX]
{
    float z[4][2];
#pragma unroll
    for (int i = 0; i < NY; ++i) wino_out1d<NX>(m + i * NX, 1, z[i][0], z[i][1]);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float c[4] = { z[0][d], z[1][d], z[2][d], NY == 4 ? z[3][d] : 0.0f };
        wino_out1d<NY>(c, 1, y[0][d], y[1][d]);
    }
}

// transform point X (compile-time) of the next K step from the row-transformed patch: class-major, then [i][j] inside the class
template <int X> __device__ __forceinline__ float wino_point(const float (&t3)[4][4], const float (&t2)[4][3])
{
    constexpr int cls = X < WINO_C10 ? 0 : X < WINO_C01 ? 1 : X < WINO_C00 ? 2 : 3;
    constexpr int e = X - (cls == 0 ? WINO_C11 : cls == 1 ? WINO_C10 : cls == 2 ? WINO_C01 : WINO_C00);
    constexpr bool y3 = cls < 2, x3 = !(cls & 1);                            // 3-tap (4-point) transform along y / x
    constexpr int nx = x3 ? 4 : 3, i = e / nx, j = e % nx;
    auto c = [&](int r) { if constexpr (x3) return t3[r][j]; else return t2[r][j]; };
    if constexpr (y3) return i == 0 ? c(0) - c(2) : i == 1 ? c(1) + c(2) : i == 2 ? c(2) - c(1) : c(1) - c(3);
    else return i == 0 ? c(0) - c(1) : i == 1 ? c(1) : c(1) - c(2);      // (patch row 3 is not used by the even output rows)
}
// C operand of a K step's MFMAs: the accumulator, or - first K step of a unit - the constant 0 (an inline operand: the unit's accumulators are never
// zeroed with 64-128 v_mov per wave; same bits, 0 + a b either way)
template <bool FIRST> __device__ __forceinline__ f32x4 wino_c(const f32x4& acc) { if constexpr (FIRST) return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; else return acc; }
template <int I, int N> struct WinoFor {                                    // compile-time loop: f(integral_constant<I>) ... f(integral_constant<N-1>)
    template <class F> static __device__ __forceinline__ void run(F&& f) { f(std::integral_constant<int, I>{}); WinoFor<I + 1, N>::run(f); }
};
template <int N> struct WinoFor<N, N> { template <class F> static __device__ __forceinline__ void run(F&&) {} };

// Which spatial tile a workgroup's unit is.  A workgroup walks `tpw` consecutive units; with the column walk (bit 9 of the launch's tpw argument,
// when tpw divides the tile rows) they are `tpw` tiles of ONE tile column, top to bottom, and workgroup p + 1 - next to it on the same XCD, in step
// with it - walks the column to its right.  A patch row is 128-byte lines of which the tile owns the middle ones; the first and last 16 bytes sit
// in the x neighbours' lines.  Walking along x, a workgroup needs those lines again one unit (~20 us) later, when the L2 (4 MiB per XCD, ~10 us
// of traffic) has dropped them: up5 fetched 2.8x its input from HBM.  Walking along y the x neighbours read the shared lines at the same time.
__device__ __forceinline__ void wino_sp_xy(int sp, int tilesX, int colrun, int& tx, int& ty)
{
    if (colrun > 1) { const int run = sp / colrun, step = sp - run * colrun; tx = run % tilesX; ty = (run / tilesX) * colrun + step; }
    else { tx = sp % tilesX; ty = sp / tilesX; }
}
// ------------------------------------------------------------------------------------------- the layer kernel
// tile = BA x BB blocks (2 BA x 2 BB input pixels) of NI instances; BA * BB * NI == 64, BA * BB a multiple of 16.
// Requires H even, W % 4 == 0, Cin % 4 == 0, CA % 4 == 0, Cout % 16 == 0 (the launcher checks).
//
// Workgroup = 8 waves = 4 block groups x 2 output-row parities.  Wave (g, h) owns blocks 16 g .. 16 g + 15 and the transform points
// of the output rows with parity py = h: h = 1 the classes (1,1) + (1,0) = 28 points (xi 0..27), h = 0 the classes (0,1) + (0,0)
// = 21 points (xi 28..48); waves g and g + 4 share a SIMD, so each SIMD still issues 49 MFMAs per K step.  Two waves per SIMD is
// the point of the split: measured with one wave per SIMD holding all 49 accumulators (196 registers), every instruction issued
// between two MFMAs of that wave cost 6-7 cycles on top of the MFMA's 32 - transform arithmetic, LDS reads, address SALU all
// added up: 46 cycles per MFMA.  A second wave covers one wave's waits (LDS operands, the barrier, DMA issue) with the other's MFMAs
// and halves the per-workgroup prologue / epilogue.  The transform arithmetic itself stays visible: the fp32 MFMA runs on the vector
// ALU it shares with every other VALU instruction (which is why its peak equals the vector peak), so the ~100 transform instructions
// per SIMD and K step cost ~20 % of the layer, and why this file is compiled without SLP vectorisation (v_pk_add_f32 beside MFMAs
// measured 11 % slower than the scalar adds it replaces).
//
// Lane (kq = lane / 16, l15 = lane % 16) of a wave owns block 16 g + l15 and, in K step k, input channel 4 k + kq: exactly the
// (k, n) element the 16x16x4 MFMA wants from this lane as its B operand.  So the lane transforms that block's 4 x 4 patch of that
// channel itself, in registers, and the results ARE its B operands of the K step: V never touches LDS.
//
// Global memory reaches the CU by LDS-DMA only (no load has a register destination, so nothing in flight pins registers and the
// prefetch distance is free to choose):
//   U slab k+2   13 pieces of 1 KiB (global_load_lds), three buffers: lands two steps ahead (one step is enough while the slab sits in the
//                XCD's L2; the deep layers with few workgroups per (stem, M block) keep several slabs in flight per XCD and miss)
//   patch k+3    the tile's 4-channel input patch, rows ty0-1..ty0+TH, columns tx0-4..tx0+TW+3 as aligned float4s
//                (buffer_load_dwordx4 ... lds: a float4 outside the image has an out-of-range offset and lands as zeros), three
//                buffers: lands two steps ahead - each XCD walks its own (stem, M-block) slice of the launch (the weights stay in
//                its L2), so every patch is an L2 miss.
// All DMA is issued from inline assembly: the compiler's own bookkeeping of LDS-DMA makes every later LDS read wait for vmcnt(0),
// i.e. for the pieces just issued.  Arrival is synchronised by hand: s_waitcnt vmcnt(n) + the K step's one barrier.
typedef int i32x4 __attribute__((ext_vector_type(4)));
// s_waitcnt immediate for "at most n vector-memory operations outstanding" (gfx9 encoding: vmcnt = bits [15:14] : [3:0]; expcnt / lgkmcnt not waited for)
constexpr int wino_vmcnt(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }
// UR = depth of the two LDS rings: U slab k lives in buffer k % UR and is issued UR - 1 K steps before its use, patch k in slot k % UR,
// issued UR - 1 steps before the step that reads it (UR = 3 was the round-2 kernel).
// CS 1: the units of a workgroup run as ONE stream of K steps (see srt_dec_wino32): no per-unit DMA drain, barrier or separate first transform.
// RB 1 (round 5): conflict-free patch reads - channel pitch padded to 32 mod 64 floats, a row read as three aligned b64 pairs (see srt_dec_wino32)
template <int BA, int BB, int NI, int ABL = 0, int UR = 3, int SB = 0, int CS = 0, int RB = 1>   // SB 1: a quad's VALU work fenced behind its MFMAs (see srt_dec_wino32); ABL (SRT_TUNING builds): timing ablations with wrong results - 1 no barrier, 2 no patch DMA, 3 no U DMA, 4 no transform, 5 no A reads
__global__ void __launch_bounds__(512, 1) srt_dec_wino(const SrtConvParams p, const float* __restrict__ U, size_t u_stem, int tpw)
{
    static_assert(UR >= 3 && UR <= 6, "ring depth");
    static_assert(BA * BB * NI == 64 && (BA * BB) % 16 == 0, "tile");
    constexpr int UBUF = 4 * 16 * WINO_LD;                                   // 3328 floats = 13 KiB = 13 DMA pieces
    constexpr int TH = 2 * BA, TW = 2 * BB;
    constexpr int PH = TH + 2, PROW = TW + 8, PR4 = PROW / 4;                // patch: PH rows of PROW floats per (channel, instance)
    constexpr int PCH0 = NI * PH * PROW;                                     // floats of one channel's patch
    constexpr int PCH = RB ? (PCH0 + 31) / 64 * 64 + 32 : PCH0;              // channel pitch in LDS (RB: the smallest value >= PCH0 that is 32 mod 64)
    constexpr int NF4 = PCH, NPP = (NF4 + 63) / 64, PBUF = NPP * 256;        // float4s (4 channels), DMA pieces and floats per patch buffer
    constexpr int PPW = (NPP + 7) / 8;                                       // patch pieces per wave
    static_assert(PPW == ((PCH0 + 63) / 64 + 7) / 8, "padding the pitch adds no DMA instruction");
    __shared__ __attribute__((aligned(16))) float s_all[UR * UBUF + UR * PBUF];
    float* s_u = s_all;
    float* s_p = s_all + UR * UBUF;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = wave & 3, h = wave >> 2;
    const int tilesX = (p.W + TW - 1) / TW, tilesY = (p.H + TH - 1) / TH;
    const int groups = (p.ntiles + NI - 1) / NI, MB = p.Cout / 16;
    const int walk = (tpw >> 9) & 1;                                         // column walk (wino_sp_xy)
    const bool mfast = ((tpw >> 10) & 1) != 0;                               // M-block pair fastest in the launch order (see srt_dec_wino32)
    tpw &= 255;
    const int colrun = !walk ? 1 : tilesY % tpw == 0 ? tpw : tpw % tilesY == 0 ? tilesY : 1;
    // A workgroup walks `tpw` consecutive (instance group, spatial tile) units of one (stem, M block): same U slabs, and the first
    // DMA of the next unit is issued BEFORE the epilogue of the current one, so its latency hides under the output transform and
    // the stores (at one workgroup per CU nothing else would cover it).  Launch order as srt_block_coord: (stem, M block) slowest.
    const int nsp = tilesX * tilesY, upw = nsp * groups / tpw;               // workgroups per (stem, M block)
    const int pos = srt_xcd_order(upw * MB * p.nstems);
    const int wsel = pos / upw, mblk = wsel % MB, stem = wsel / MB, unit0 = (pos % upw) * tpw;
    const int m0 = mblk * 16;
    const SrtAct actp = srt_act_params(srt_act_kind(p, stem), p.variant);
    const size_t hw = (size_t)p.H * p.W;
    const float* up = U + stem * u_stem + (size_t)mblk * UBUF;               // K step k at + k * MB * UBUF

    const int blk = g * 16 + l15;
    const int il = blk / (BA * BB), ba = (blk / BB) % BA, bb = blk % BB;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)s_all;

    // ---- U slab DMA: wave w moves pieces w and min(w + 8, 12) (three waves repeat piece 12: same bytes, and no branch)
    unsigned dvoff[2], dm0[2];                                               // byte offset inside the slab (+ lane*16), LDS byte address in buffer 0
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = min(wave + 8 * i, 12);
        dvoff[i] = (unsigned)(piece * 1024 + lane * 16);
        dm0[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(piece * 1024));
    }
    auto dma_u = [&](int k, int buf, int i) {
        const float* src = up + (size_t)k * MB * UBUF;                       // wave-uniform: SGPR base + VGPR offset
        const unsigned dst = dm0[i] + (unsigned)buf * (unsigned)(UBUF * 4);
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(dvoff[i]), "s"(src), "s"(dst) : "memory");
    };
    // ---- patch DMA: float4 e = ((c * NI + il) * PH + row) * PR4 + j of the patch buffer <- channel 4k+c, instance tile0+il,
    // image row ty0-1+row, columns tx0-4+4j..+3.  Wave w moves pieces w, w+8 (a piece past the last one repeats it).
    constexpr unsigned OOR = 0x80000000u;                                    // >= num_records: lands as zeros
    unsigned pvoff[PPW], pm0[PPW];
    const unsigned nrec = (unsigned)min((size_t)0x7fffffff, (size_t)4 * NI * p.srcA_tile);
    const float* pa; const float* pb;                                        // wave-uniform: channel 0 of the unit's first instance
    unsigned n_pvoff[PPW]; const float* n_pa; const float* n_pb;             // CS: the same for the workgroup's next unit
#pragma unroll
    for (int i = 0; i < PPW; ++i) pm0[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(UR * UBUF * 4 + min(wave + 8 * i, NPP - 1) * 1024));
    auto set_dma_unit = [&](int unit) {
        int sx_, sy_; wino_sp_xy(unit % nsp, tilesX, colrun, sx_, sy_);
        const int tile0 = (unit / nsp) * NI, tx0 = sx_ * TW, ty0 = sy_ * TH;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = min(wave + 8 * i, NPP - 1), e = piece * 64 + lane;
            const int c = e / (PCH / 4), rem = e % (PCH / 4);                // float4 `rem` of channel c's patch (rem >= PCH0 / 4: padding of the pitch)
            const int j = rem % PR4, row = (rem / PR4) % PH, ii = rem / (PR4 * PH);
            const int gy = ty0 - 1 + row, gx = tx0 - 4 + 4 * j;
            const bool ok = e < NF4 && rem < PCH0 / 4 && tile0 + ii < p.ntiles && gy >= 0 && gy < p.H && gx >= 0 && gx + 3 < p.W;
            pvoff[i] = ok ? 4u * (unsigned)((size_t)ii * p.srcA_tile + (size_t)c * hw + (size_t)gy * p.W + gx) : OOR;  // srcA_tile == srcB_tile (launcher)
        }
        pa = p.srcA + stem * p.srcA_stem + tile0 * p.srcA_tile;
        pb = p.srcB + stem * p.srcB_stem + tile0 * p.srcB_tile;
    };
    auto dma_keep_as_next = [&]() {
#pragma unroll
        for (int i = 0; i < PPW; ++i) n_pvoff[i] = pvoff[i];
        n_pa = pa; n_pb = pb;
    };
    auto dma_advance = [&]() {
#pragma unroll
        for (int i = 0; i < PPW; ++i) pvoff[i] = n_pvoff[i];
        pa = n_pa; pb = n_pb;
    };
    // the DMA state of unit `unit` into the "next" slot (the current one is saved around the computation)
    auto set_dma_next = [&](int unit) {
        unsigned keep[PPW]; const float* ka = pa; const float* kb = pb;
#pragma unroll
        for (int i = 0; i < PPW; ++i) keep[i] = pvoff[i];
        set_dma_unit(unit);
        dma_keep_as_next();
#pragma unroll
        for (int i = 0; i < PPW; ++i) pvoff[i] = keep[i];
        pa = ka; pb = kb;
    };
    const int kA = p.CA / 4;                                                 // K steps [0, kA) read srcA, the rest srcB (CA % 4 == 0)
    const unsigned kstep_bytes = (unsigned)(16 * hw);                        // 4 channels
    auto dma_patch = [&](int k, int slot, int i) {
        const bool fromA = k < kA;
        const size_t base = (size_t)(fromA ? pa : pb);
        i32x4 rs;
        rs.x = (int)(unsigned)(base & 0xffffffffu); rs.y = (int)(unsigned)((base >> 32) & 0xffffu); rs.z = (int)nrec; rs.w = 0x00020000;
        const unsigned soff = (unsigned)(fromA ? k : k - kA) * kstep_bytes;
        const unsigned dst = pm0[i] + (unsigned)slot * (unsigned)(PBUF * 4);
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(pvoff[i]), "s"(rs), "s"(dst), "s"(soff) : "memory");
    };
    const int poff = kq * PCH + (il * PH + 2 * ba) * PROW + 2 * bb + 3;      // this lane's patch: rows +0..3, columns +0..3 (b0-1..b0+2)
    const int aoff = (kq * 16 + l15) * WINO_LD;
    const int nk = p.Cin / 4;
    const int Wo = p.W << 1;
    const size_t ohw = (size_t)(p.H << 1) * Wo;
    float* obase; bool blk_ok;
    auto set_out_unit = [&](int unit) {
        int sx_, sy_; wino_sp_xy(unit % nsp, tilesX, colrun, sx_, sy_);
        const int tile = (unit / nsp) * NI + il, a0 = sy_ * TH + 2 * ba, b0 = sx_ * TW + 2 * bb;
        blk_ok = tile < p.ntiles && a0 < p.H && b0 < p.W;
        obase = p.outAct + stem * p.out_stem + (blk_ok ? tile : 0) * p.out_tile + (size_t)(blk_ok ? 2 * a0 : 0) * Wo + (blk_ok ? 2 * b0 : 0);
    };
    // epilogue constants before the K loop (see srt_dec16_kernel)
    float bi[4], sc[4], sf[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t ci = stem * p.coeff_stem + m0 + 4 * kq + r;
        bi[r] = p.bias[ci]; sc[r] = p.bnScale[ci]; sf[r] = p.bnShift[ci];
    }

    // ---- everything below is specialised on the wave's row parity H (a wave-uniform branch; both paths run the same barriers)
    auto body = [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        constexpr int X0 = H ? 0 : 28, NP = H ? 28 : 21, NQ = H ? 7 : 6, NROW = H ? 4 : 3;   // points xi = X0 .. X0+NP-1; patch rows used
        f32x4 acc[NP];
        // the transform in two stages: rows (x direction) of the patch -> t3 / t2, then one transform point at a time
        float t3[4][4], t2[4][3];
        float2 xm[4], xo[4];                                                 // patch row r: columns (b0, b0+1) | (b0-1, b0+2)
        auto read_row = [&](const float* pbuf, int r) {
            const float* q = pbuf + poff + r * PROW;
            xm[r] = *reinterpret_cast<const float2*>(q + 1);
            if constexpr (RB) xo[r] = make_float2(reinterpret_cast<const float2*>(q - 1)->y, reinterpret_cast<const float2*>(q + 3)->x);
            else xo[r] = make_float2(q[0], q[3]);
        };
        auto rows = [&](int r) {
            wino_in3(xo[r].x, xm[r].x, xm[r].y, xo[r].y, t3[r]);
            wino_in2(xo[r].x, xm[r].x, xm[r].y, t2[r]);
        };
        // K step k, in MFMA quads q = 0..NQ-1 (4 transform points each; the last quad of H = 0 has one).  Quad q issues its MFMAs on this
        // step's points; ONE QUAD LATER those registers are refilled with the points of step k+1, computed from the row transforms of
        // patch k+1 (a VALU write to a register that an MFMA issued just before still reads as its B operand has to wait for it).
        // Quad 0 first refills the last quad's points from the OLD row transforms, then reads its patch (k+1) and transforms the rows.
        float v[NP];
        auto issue_first = [&]() {                                           // U slabs 0..UR-2 -> buffers 0..UR-2; patches 0..UR-2 -> slots 0..UR-2 of the unit set by set_dma_unit
#pragma unroll
            for (int j = 0; j < UR - 1; ++j) {
#pragma unroll
                for (int i = 0; i < 2; ++i) dma_u(min(j, nk - 1), j, i);
#pragma unroll
                for (int i = 0; i < PPW; ++i) dma_patch(min(j, nk - 1), j, i);
            }
        };
        set_dma_unit(unit0);
        if (CS) { if (tpw > 1) set_dma_next(unit0 + 1); else dma_keep_as_next(); }
        issue_first();
        int slot = 0;                                                        // k % UR (CS: of the running step count): U slab k is in buffer `slot`, patch k+1 in slot+1; patch k+UR goes to `slot`
        auto unit_prologue = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0): the unit's first slab and patches (and whatever the previous unit left in flight)
            __syncthreads();
#pragma unroll
            for (int r = 0; r < NROW; ++r) { read_row(s_p, r); rows(r); }
            WinoFor<0, NP>::run([&](auto xc) { constexpr int x = decltype(xc)::value; v[x] = wino_point<X0 + x>(t3, t2); });
#pragma unroll
            for (int i = 0; i < PPW; ++i) dma_patch(min(UR - 1, nk - 1), UR - 1, i);
            slot = 0;
        };
        if (CS) unit_prologue();
        for (int t = 0; t < tpw; ++t) {
        if (!CS) unit_prologue();
        auto kstep = [&](int k, auto fc) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(fc)::value;                      // first K step of a unit: C = 0
            // vmcnt((UR-2)(2 + PPW)): everything older than this wave's pieces of the last UR-2 steps has landed - its pieces of U slab k
            // and of patch k+1 (both issued in step k+1-UR).  After the barrier so have everyone's, and every wave is done with step k-1:
            // U buffer (k-1)%UR and patch slot k%UR are free.
            if (ABL != 1) { __builtin_amdgcn_s_waitcnt(wino_vmcnt((UR - 2) * (2 + PPW))); __syncthreads(); }
            // past the end of the unit: CS - the first slabs / patches of the workgroup's next unit (the DMA state switches to it at the first step
            // whose patch belongs to it; U does not depend on the unit); otherwise the last slab / patch again, unused
            if (CS && k + UR == nk) dma_advance();
            const int kd = CS ? (k + UR - 1 >= nk ? k + UR - 1 - nk : k + UR - 1) : min(k + UR - 1, nk - 1);
            const int kp = CS ? (k + UR >= nk ? k + UR - nk : k + UR) : min(k + UR, nk - 1);
            const int ubn = slot == 0 ? UR - 1 : slot - 1;                   // (k + UR - 1) % UR
            const float* ub = s_u + slot * UBUF + aoff + X0;
            const float* pbuf = s_p + (slot == UR - 1 ? 0 : slot + 1) * PBUF;
            float4 a0 = *reinterpret_cast<const float4*>(ub), a1 = *reinterpret_cast<const float4*>(ub + 4);
            __builtin_amdgcn_sched_barrier(0);
            WinoFor<0, NQ>::run([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int nm = (4 * q + 4 <= NP) ? 4 : NP - 4 * q;       // MFMAs of this quad
                const float4 a = a0;
                a0 = a1;
                if constexpr (q + 2 < NQ && ABL != 5) a1 = *reinterpret_cast<const float4*>(ub + 4 * (q + 2));      // two quads ahead
                if constexpr (q == 1 && ABL != 3) { dma_u(kd, ubn, 0); dma_u(kd, ubn, 1); }
                if constexpr (q == 2 && ABL != 2) {
#pragma unroll
                    for (int i = 0; i < PPW; ++i) dma_patch(kp, slot, i);
                }
                acc[4 * q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, v[4 * q], wino_c<FIRST>(acc[4 * q]), 0, 0, 0);
                if constexpr (nm > 1) {
                    acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, v[4 * q + 1], wino_c<FIRST>(acc[4 * q + 1]), 0, 0, 0);
                    acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, v[4 * q + 2], wino_c<FIRST>(acc[4 * q + 2]), 0, 0, 0);
                    acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, v[4 * q + 3], wino_c<FIRST>(acc[4 * q + 3]), 0, 0, 0);
                }
                if constexpr (SB == 1) __builtin_amdgcn_sched_barrier(0);
                if constexpr (ABL != 4) {
                    if constexpr (q == 0) {
                        constexpr int l0 = 4 * (NQ - 1);                     // the last quad's points, from the old row transforms
                        WinoFor<l0, NP>::run([&](auto xc) { constexpr int x = decltype(xc)::value; v[x] = wino_point<X0 + x>(t3, t2); });
#pragma unroll
                        for (int r = 0; r < NROW; ++r) read_row(pbuf, r);
                        // row transforms where the refills first need them (H = 1: quad 1 refills i = 0 of class (1,1): rows 0, 2; quad 2
                        // i = 1: row 1; quad 4 i = 3: row 3.  H = 0: quad 1 refills i = 0 of class (0,1): rows 0, 1; quad 3 i = 2: row 2)
                        rows(0); rows(H ? 2 : 1);
                    } else {
                        if constexpr (q == 1) rows(H ? 1 : 2);
                        if constexpr (q == 2 && H) rows(3);
                        WinoFor<4 * q - 4, 4 * q>::run([&](auto xc) { constexpr int x = decltype(xc)::value; v[x] = wino_point<X0 + x>(t3, t2); });
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                           // quads stay in order: bounded live ranges, no accumulator copies
            });
            slot = slot == UR - 1 ? 0 : slot + 1;
        };
        kstep(0, std::true_type{});
        for (int k = 1; k < nk; ++k) kstep(k, std::false_type{});

        // ---- the next unit's first DMA goes out before this unit's epilogue.  Barrier: every wave is out of the K loop, so the U
        // buffers and patch slots are free (the refills the last K steps issued past the end target the same pieces from the same
        // waves, earlier in each wave's DMA order, so they land first).
        if (!CS && t + 1 < tpw) {
            __syncthreads();
            set_dma_unit(unit0 + t + 1);
            issue_first();
        }
        // ---- output transform + bias -> activation -> batch-norm: this lane's block, channels m0 + 4*kq + r, output rows of parity H
        set_out_unit(unit0 + t);
        auto emit = [&](auto act) {                                          // act(y, r): bias -> activation -> batch-norm of channel r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m[NP];
#pragma unroll
                for (int x = 0; x < NP; ++x) m[x] = acc[x][r];
                float yA[2][2], yB[2][2];                                    // px = 1 and px = 0 classes of this row parity
                if constexpr (H) { wino_out2d<4, 4>(m + WINO_C11 - X0, yA); wino_out2d<4, 3>(m + WINO_C10 - X0, yB); }
                else { wino_out2d<3, 4>(m + WINO_C01 - X0, yA); wino_out2d<3, 3>(m + WINO_C00 - X0, yB); }
                float* oc = obase + (size_t)(m0 + 4 * kq + r) * ohw + (size_t)H * Wo;
#pragma unroll
                for (int da = 0; da < 2; ++da) {
                    // output row 2(a0+da)+H, columns 2(b0+db)+px: one float4 = (px0 db0, px1 db0, px0 db1, px1 db1)
                    float4 e;
                    e.x = act(yB[da][0], r); e.y = act(yA[da][0], r); e.z = act(yB[da][1], r); e.w = act(yA[da][1], r);
                    *reinterpret_cast<float4*>(oc + (size_t)(2 * da) * Wo) = e;
                }
            }
        };
        if (blk_ok) {                                                        // the activation kind is workgroup-uniform: one branch around the whole epilogue
            if (srt_act_is_plain_elu(actp)) emit([&](float y, int r) { return srt_dec_epilogue_elu(y, bi[r], sc[r], sf[r]); });
            else if (actp.ue != 0.0f) emit([&](float y, int r) { return srt_dec_epilogue(y, bi[r], sc[r], sf[r], actp); });
            else emit([&](float y, int r) { return srt_dec_epilogue_lin(y, bi[r], sc[r], sf[r], actp.lin); });
        }
        if (CS && t + 2 < tpw) set_dma_next(unit0 + t + 2);                  // (the DMA state itself moved on UR steps before the unit ended)
        }                                                                    // units
    };
    if (h) body(std::integral_constant<int, 1>{});
    else body(std::integral_constant<int, 0>{});
}

// ------------------------------------------------------------------------------------------- 32 output channels per workgroup
// Round 3.  The fp32 MFMA and ordinary VALU instructions share one ALU on gfx950 (scripts/ubench/mfma_valu.hip: beside
// v_mfma_f32_16x16x4_f32 every v_add costs its full 4 cycles and the first one after an MFMA ~10 more, with one or two waves per SIMD;
// beside a bf16 MFMA the same instructions are free), so the kernel above pays its ~116 transform instructions per 49 MFMAs in
// matrix time.  Here a lane's transformed patch V feeds TWO MFMAs - the same block against two 16-channel M blocks - which halves the
// transform work per MFMA, and the accumulators are split over the waves by parity CLASS instead of row parity, so no row transform is
// computed twice:
//
//   workgroup = 32 output channels x 32 blocks (BA x BB blocks = 2 BA x 2 BB input pixels of ONE instance), 8 waves
//   wave (g, CLS): block group g (16 blocks) x class CLS x both M blocks: 2 x {16, 12, 12, 9} accumulator tiles (128 / 96 / 96 / 72 registers)
//   SIMD s holds waves s and s + 4: (g, C11) with (g, C00) on SIMDs 0-1 (50 MFMAs per K step), (g, C10) with (g, C01) on SIMDs 2-3 (48)
//
// Per K step a wave issues its points in quads: 8 MFMAs (4 points x 2 M blocks), then ONE burst of VALU work (4 points of the next
// step, in quad 1 also the row transforms of the next patch): a transition between matrix and vector work per 8 MFMAs instead of per 4.
// Global memory reaches the CU by LDS-DMA as above: per K step the 26 KiB U slab of the two M blocks (26 pieces) and the 4-channel
// patch (4 pieces for 4 x 32 input pixels), four DMA instructions per wave; the input patch is now fetched once per 32 output channels.
// The two x-parity classes of an output row live in different waves, so a lane stores single floats (each class owns every second
// column); the four stores that complete a 16-byte segment are issued within the same epilogue and merge in L2.
// Rings: U slab k lives in buffer k % UR, patch k in slot k % UR; both are issued D K steps before the step that uses them (slab k+D and
// patch k+1+D during step k) and the workgroup meets at a barrier every BPS K steps (UR >= D + BPS keeps a DMA from overwriting what a
// slower wave may still read).  Round-2 arrangement: UR = 3, D = 2, BPS = 1.
// SB 1: the VALU work of a quad is fenced behind its MFMAs (clean bursts).  EA 1: the A operands of the first two quads of step k+1 are read
// during the last two quads of step k, ahead of the barrier (slab k+1 is then waited for one barrier earlier), so no wave starts a step by
// waiting for LDS.
// ST 1: the two waves of a SIMD are staggered - the second one (classes C01 / C00) reads its patch rows right after the barrier and transforms them
// behind its FIRST quad, the first one (C11 / C10) behind its second quad, so that one wave's long VALU burst meets the other's MFMAs
// instead of the other's burst (the barrier starts both at the same point of the step every time).
// CS 1: the units of a workgroup form ONE stream of K steps: the last D steps of a unit already fetch the first slabs / patches of the next
// unit (instead of refetching the last ones as filler), and the next unit's first transformed patch comes out of the ordinary refills of the
// last step, so a unit boundary is an epilogue and nothing else: no drained DMA queue, no extra barrier, no separate first transform.
// NI: instances per workgroup (BA * BB * NI == 32 blocks): 1 for inputs of at least 4 x 32 pixels, 2 for 4 x 16 (up1: one instance is 16 blocks).
// RB 1 (round 5): LDS bank discipline of the patch reads.  A lane's block columns are 2 floats apart, the four channels of a K step (kq) one channel pitch: with
// the natural pitch of 240 floats (48 mod 64) the b64 of the middle columns and the two b32 of the outer ones are all 2-way conflicted (12 LDS cycles per row of a
// 3-tap class).  RB 1 pads the pitch to 32 mod 64 floats (288: one more DMA piece per K step, still four DMA instructions per wave) and reads a row as THREE aligned
// b64 pairs (columns b0-2..b0-1, b0..b0+1, b0+2..b0+3): within each 32-lane group kq 0 covers one half of the banks and kq 1 the other - 6 cycles, conflict-free.
template <int BA, int BB, int ABL = 0, int UR = 3, int D = UR - 1, int BPS = 1, int SB = 0, int EA = 0, int ST = 0, int CS = 0, int NI = 1, int PEEL = 0, int RB = 1>   // PEEL 1: a unit's first K step starts from C = 0 instead of zeroed accumulators (measured slower here, faster in srt_dec_wino)
__global__ void __launch_bounds__(512, 1) srt_dec_wino32(const SrtConvParams p, const float* __restrict__ U, size_t u_stem, int tpw)
{
    static_assert(UR >= 3 && UR <= 5 && D >= BPS + EA && UR >= D + BPS && !(CS && EA), "rings (5 x 30 KiB = 150 KiB of LDS)");
    static_assert(BA * BB * NI == 32 && (BA * BB) % 16 == 0, "tile");
    constexpr int UB1 = 4 * 16 * WINO_LD;                                    // one M block: 3328 floats = 13 pieces
    constexpr int UBUF = 2 * UB1, NUP = 26;                                  // two M blocks (consecutive in the packed layout)
    constexpr int TH = 2 * BA, TW = 2 * BB;
    constexpr int PH = TH + 2, PROW = TW + 8, PR4 = PROW / 4;
    constexpr int PCH0 = NI * PH * PROW;                                     // floats of one channel's patch
    constexpr int PCH = RB ? (PCH0 + 31) / 64 * 64 + 32 : PCH0;              // channel pitch in LDS (RB: the smallest value >= PCH0 that is 32 mod 64)
    constexpr int NF4 = PCH, NPP = (NF4 + 63) / 64, PBUF = NPP * 256;
    constexpr int NPIECE = NUP + NPP, DPW = (NPIECE + 7) / 8;                // DMA pieces per K step, per wave (the tail repeats the last piece)
    static_assert(PCH >= PCH0 && PCH % 4 == 0 && NPIECE <= 8 * DPW, "patch pitch / piece map");
    __shared__ __attribute__((aligned(16))) float s_all[UR * UBUF + UR * PBUF];
    __shared__ __attribute__((aligned(16))) float s_epi[96];                   // bias | BN scale | BN shift of the workgroup's 32 channels
    float* s_u = s_all;
    float* s_p = s_all + UR * UBUF;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int flags = tpw >> 8;                                              // bit 0 (tuning aid): static priority 1 for waves 4-7, the younger half of each SIMD pair; bit 1: column walk
    tpw &= 255;
    if ((flags & 1) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int simd = wave & 3, hi = wave >> 2, g = simd & 1;                 // block group; class: SIMDs 0-1 C11 | C00, SIMDs 2-3 C10 | C01
    const int cls = (simd >> 1) == 0 ? (hi ? 3 : 0) : (hi ? 2 : 1);
    const int tilesX = (p.W + TW - 1) / TW, tilesY = (p.H + TH - 1) / TH;
    const int MB2 = p.Cout / 32, MB = p.Cout / 16;
    const int colrun = !(flags & 2) ? 1 : tilesY % tpw == 0 ? tpw : tpw % tilesY == 0 ? tilesY : 1;
    const int nsp = tilesX * tilesY, groups = (p.ntiles + NI - 1) / NI, upw = nsp * groups / tpw;      // workgroups per (stem, M-block pair)
    const int pos = srt_xcd_order(upw * MB2 * p.nstems);
    // launch order: (stem, M-block pair) slowest - an XCD works on one weight slab at a time (L2 resident) and re-reads the input from HBM once per pair -
    // or, flag bit 2, the M-block pair FASTEST: the MB2 workgroups that read the same input patches are neighbours on one XCD and run in step (the
    // patch is an L2 hit for all but the first), while the U slabs of all pairs of a stem share the L2 (layers whose weights are small)
    const bool mfast = (flags & 4) != 0;
    const int wsel = pos / upw, mblk2 = mfast ? pos % MB2 : wsel % MB2, stem = mfast ? pos / (MB2 * upw) : wsel / MB2, unit0 = (mfast ? (pos / MB2) % upw : pos % upw) * tpw;
    const int m0 = mblk2 * 32;
    if (tid < 32) {                                                          // (visible after the first barrier of the K stream)
        const size_t ci = stem * p.coeff_stem + m0 + tid;
        s_epi[tid] = p.bias[ci]; s_epi[32 + tid] = p.bnScale[ci]; s_epi[64 + tid] = p.bnShift[ci];
    }
    const SrtAct actp = srt_act_params(srt_act_kind(p, stem), p.variant);
    const size_t hw = (size_t)p.H * p.W;
    const float* up = U + stem * u_stem + (size_t)(2 * mblk2) * UB1;         // K step k at + k * MB * UB1

    const int blk = g * 16 + l15, il = blk / (BA * BB), ba = (blk / BB) % BA, bb = blk % BB;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)s_all;

    // ---- DMA pieces of a K step: 0..25 the U slab, 26.. the patch.  Wave w moves pieces w, w + 8, ... (DPW of them; past the end: the last one
    // again).  All but a wave's LAST piece are U pieces for every wave (global_load_lds); the last one is a U piece for waves 0-1 and a patch
    // piece for the rest: ONE buffer_load ... lds whose descriptor, offsets and destination are picked by scalar selects on the wave-uniform
    // `last_patch`, so the K loop has no branch around its DMA (a branchy version of the same loop measured 8 % slower).
    static_assert(NUP >= 8 * (DPW - 1) && NUP < 8 * DPW, "piece map");
    const bool last_patch = wave >= NUP - 8 * (DPW - 1);                     // wave-uniform
    unsigned dvoff[DPW - 1], dm0[DPW - 1];
#pragma unroll
    for (int i = 0; i < DPW - 1; ++i) {
        dvoff[i] = (unsigned)((wave + 8 * i) * 1024 + lane * 16);            // byte offset inside the slab
        dm0[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((wave + 8 * i) * 1024));
    }
    const int lpiece = min(wave + 8 * (DPW - 1), NPIECE - 1);                // the last ("flex") piece
    const unsigned fm0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(last_patch ? UR * UBUF * 4 + (lpiece - NUP) * 1024 : lpiece * 1024));
    constexpr unsigned OOR = 0x80000000u;
    const unsigned nrec = last_patch ? (unsigned)min((size_t)0x7fffffff, (size_t)4 * NI * p.srcA_tile) : 0x7fffffffu;
    // per-unit DMA state of the flex piece: descriptor base halves (wave-uniform) and the lane's offset.  For waves 0-1 (U piece) the base is the
    // U slab and the offset the lane's place in it, whatever the unit.  (64-bit selects on a uniform condition come out of the compiler as
    // vector selects, which an "s" asm operand cannot take: the halves are selected and pinned to SGPRs.)
    unsigned c_alo, c_ahi, c_blo, c_bhi, c_voff, n_alo, n_ahi, n_blo, n_bhi, n_voff;       // current / next unit
    struct UnitBase { unsigned alo, ahi, blo, bhi; };                        // (uniform members only: with the per-lane offset in the same struct the compiler treats all of it as divergent)
    auto unit_base = [&](int unit) {
        const int tile0 = (unit / nsp) * NI;
        const size_t ba_ = (size_t)(p.srcA + stem * p.srcA_stem + tile0 * p.srcA_tile), bb_ = (size_t)(p.srcB + stem * p.srcB_stem + tile0 * p.srcB_tile), bu_ = (size_t)up;
        UnitBase u;
        u.alo = __builtin_amdgcn_readfirstlane(last_patch ? (unsigned)ba_ : (unsigned)bu_); u.ahi = __builtin_amdgcn_readfirstlane(last_patch ? (unsigned)(ba_ >> 32) : (unsigned)(bu_ >> 32));
        u.blo = __builtin_amdgcn_readfirstlane(last_patch ? (unsigned)bb_ : (unsigned)bu_); u.bhi = __builtin_amdgcn_readfirstlane(last_patch ? (unsigned)(bb_ >> 32) : (unsigned)(bu_ >> 32));
        return u;
    };
    auto unit_voff = [&](int unit) {
        int sx_, sy_; wino_sp_xy(unit % nsp, tilesX, colrun, sx_, sy_);
        const int tile0 = (unit / nsp) * NI, tx0 = sx_ * TW, ty0 = sy_ * TH;
        const int e = (lpiece - NUP) * 64 + lane;
        const int c = e / (PCH / 4), rem = e % (PCH / 4);                    // float4 `rem` of channel c's patch (rem >= PCH0 / 4: padding of the pitch)
        const int j = rem % PR4, row = (rem / PR4) % PH, ii = rem / (PR4 * PH);
        const int gy = ty0 - 1 + row, gx = tx0 - 4 + 4 * j;
        const bool ok = e >= 0 && e < NF4 && rem < PCH0 / 4 && tile0 + ii < p.ntiles && gy >= 0 && gy < p.H && gx >= 0 && gx + 3 < p.W;
        const unsigned pvoff = ok ? 4u * (unsigned)((size_t)ii * p.srcA_tile + (size_t)c * hw + (size_t)gy * p.W + gx) : OOR;   // srcA_tile == srcB_tile (launcher)
        return last_patch ? pvoff : (unsigned)(lpiece * 1024 + lane * 16);
    };
    auto unit_cur = [&](int unit) { const UnitBase u = unit_base(unit); c_alo = u.alo; c_ahi = u.ahi; c_blo = u.blo; c_bhi = u.bhi; c_voff = unit_voff(unit); };
    auto unit_nxt = [&](int unit) { const UnitBase u = unit_base(unit); n_alo = u.alo; n_ahi = u.ahi; n_blo = u.blo; n_bhi = u.bhi; n_voff = unit_voff(unit); };
    auto unit_nxt_is_cur = [&]() { n_alo = c_alo; n_ahi = c_ahi; n_blo = c_blo; n_bhi = c_bhi; n_voff = c_voff; };
    auto unit_advance = [&]() { c_alo = n_alo; c_ahi = n_ahi; c_blo = n_blo; c_bhi = n_bhi; c_voff = n_voff; };
    const int kA = p.CA / 4;
    const unsigned kstep_bytes = (unsigned)(16 * hw), ustep_bytes = (unsigned)((size_t)MB * UB1 * 4);
    auto dma_u = [&](int i, int ku, int ubuf) {
        const float* src = up + (size_t)ku * MB * UB1;
        const unsigned dst = dm0[i] + (unsigned)ubuf * (unsigned)(UBUF * 4);
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(dvoff[i]), "s"(src), "s"(dst) : "memory");
    };
    // U slab ku -> buffer ubuf (waves 0-1) | patch kp of the DMA state's unit -> slot pslot
    auto dma_flex = [&](int ku, int ubuf, int kp, int pslot) {
        const bool fromA = kp < kA;
        const unsigned lo = fromA ? c_alo : c_blo, hi = fromA ? c_ahi : c_bhi;       // (both are the U slab for waves 0-1)
        i32x4 rs;
        rs.x = (int)lo; rs.y = (int)(hi & 0xffffu); rs.z = (int)nrec; rs.w = 0x00020000;
        const unsigned soff = last_patch ? (unsigned)(fromA ? kp : kp - kA) * kstep_bytes : (unsigned)ku * ustep_bytes;
        const unsigned dst = fm0 + (last_patch ? (unsigned)pslot * (unsigned)(PBUF * 4) : (unsigned)ubuf * (unsigned)(UBUF * 4));
        const unsigned voff = c_voff;
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(rs), "s"(dst), "s"(soff) : "memory");
    };
    // piece i of this wave
    auto dma = [&](int i, int ku, int ubuf, int kp, int pslot) {
        if (i == DPW - 1) dma_flex(ku, ubuf, kp, pslot);
        else dma_u(i, ku, ubuf);
    };
    const int poff = kq * PCH + (il * PH + 2 * ba) * PROW + 2 * bb + 3;      // this lane's patch: rows +0..3, columns +0..3 (b0-1..b0+2)
    const int aoff = (kq * 16 + l15) * WINO_LD;
    const int nk = p.Cin / 4;
    const int Wo = p.W << 1;
    const size_t ohw = (size_t)(p.H << 1) * Wo;
    float* obase; bool blk_ok;
    auto set_out_unit = [&](int unit) {
        int sx_, sy_; wino_sp_xy(unit % nsp, tilesX, colrun, sx_, sy_);
        const int tile = (unit / nsp) * NI + il, a0 = sy_ * TH + 2 * ba, b0 = sx_ * TW + 2 * bb;
        blk_ok = tile < p.ntiles && a0 < p.H && b0 < p.W;
        obase = p.outAct + stem * p.out_stem + (blk_ok ? tile : 0) * p.out_tile + (size_t)(blk_ok ? 2 * a0 : 0) * Wo + (blk_ok ? 2 * b0 : 0);
    };

    auto body = [&](auto cc) __attribute__((always_inline)) {
        constexpr int CLS = decltype(cc)::value;
        constexpr int X0 = CLS == 0 ? WINO_C11 : CLS == 1 ? WINO_C10 : CLS == 2 ? WINO_C01 : WINO_C00;
        constexpr bool Y3 = CLS < 2, X3 = !(CLS & 1);
        constexpr int NY = Y3 ? 4 : 3, NX = X3 ? 4 : 3, NP = NY * NX, NQ = (NP + 3) / 4, NROW = Y3 ? 4 : 3;
        constexpr int PY = CLS < 2 ? 1 : 0, PX = X3 ? 1 : 0;
        constexpr int RQ = ((ST == 1 && CLS >= 2) || (ST == 2 && CLS < 2)) ? 0 : 1;                         // quad whose burst carries the row transforms
        f32x4 acc[2][NP];
        float t3[4][4], t2[4][3];
        float2 xm[4]; float xa[4], xb[4];                                    // patch row r: columns (b0, b0+1) | b0-1 | b0+2 (3-tap classes only)
        auto read_row = [&](const float* pbuf, int r) {
            const float* q = pbuf + poff + r * PROW;
            xm[r] = *reinterpret_cast<const float2*>(q + 1);
            if constexpr (RB) {                                              // the outer columns as halves of aligned b64 pairs (see RB above)
                xa[r] = reinterpret_cast<const float2*>(q - 1)->y;
                if constexpr (X3) xb[r] = reinterpret_cast<const float2*>(q + 3)->x;
            } else {
                xa[r] = q[0];
                if constexpr (X3) xb[r] = q[3];
            }
        };
        auto rows = [&](int r) {
            if constexpr (X3) wino_in3(xa[r], xm[r].x, xm[r].y, xb[r], t3[r]);
            else wino_in2(xa[r], xm[r].x, xm[r].y, t2[r]);
        };
        float v[NP];
        auto issue_first = [&]() {                                           // U slabs 0..D-1 -> buffers 0..D-1; patches 0..D-1 -> slots 0..D-1
#pragma unroll
            for (int j = 0; j < D; ++j)
#pragma unroll
                for (int i = 0; i < DPW; ++i) dma(i, min(j, nk - 1), j, min(j, nk - 1), j);
        };
        // K step k: U slab k in buffer su (= k % UR), patch k+1 in slot sp1; issues U slab k+D -> buffer sd and patch k+1+D -> slot sd1
        float4 a0[2], a1[2];                                                 // A operands of the next two quads (EA: carried across K steps)
        auto kstep = [&](int k, int su, int sp1, int sd, int sd1, auto fc) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(fc)::value;                      // first K step of a unit: C = 0 (wino_c)
            // past the end of the unit: CS - the first slabs / patches of the workgroup's next unit (U does not depend on the unit; after the last unit the
            // "next" state equals the current one: valid addresses, unused data); otherwise the last slab / patch again, unused
            // (the DMA state - c_* - switches to the next unit at the first step whose patch belongs to it: a 4-way select between the two
            //  states inside dma_flex turns into a dynamically indexed load from the lambda's closure, which then stays in scratch memory and
            //  makes every wave-uniform value of the kernel a vector value)
            if (CS && k + 1 + D == nk) unit_advance();
            const int kd = CS ? (k + D >= nk ? k + D - nk : k + D) : min(k + D, nk - 1), kp = CS ? (k + 1 + D >= nk ? k + 1 + D - nk : k + 1 + D) : min(k + 1 + D, nk - 1);
            const float* ub = s_u + su * UBUF + aoff + X0;
            const float* ubx = s_u + (su == UR - 1 ? 0 : su + 1) * UBUF + aoff + X0;       // slab k+1 (EA)
            const float* pbuf = s_p + sp1 * PBUF;
            if (!EA || k == 0) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) { a0[mb] = *reinterpret_cast<const float4*>(ub + mb * UB1); a1[mb] = *reinterpret_cast<const float4*>(ub + mb * UB1 + 4); }
            }
            if constexpr (RQ == 0 && ABL != 4) {
#pragma unroll
                for (int r = 0; r < NROW; ++r) read_row(pbuf, r);            // patch k+1 (landed before this step's barrier): in registers by the end of quad 0
            }
            __builtin_amdgcn_sched_barrier(0);
            WinoFor<0, NQ>::run([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int nm = (4 * q + 4 <= NP) ? 4 : NP - 4 * q;       // points of this quad
                float4 a[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    a[mb] = a0[mb];
                    if constexpr (EA && q == NQ - 1) a0[mb] = *reinterpret_cast<const float4*>(ubx + mb * UB1);            // next step, quad 0
                    else a0[mb] = a1[mb];
                    if constexpr (q + 2 < NQ && ABL != 5) a1[mb] = *reinterpret_cast<const float4*>(ub + mb * UB1 + 4 * (q + 2));
                    else if constexpr (EA && q == NQ - 2) a1[mb] = *reinterpret_cast<const float4*>(ubx + mb * UB1 + 4);  // next step, quad 1
                }
                if constexpr (ABL != 3) {                                    // DMA spread over the quads (NQ is 3 or 4, DPW 4)
#pragma unroll
                    for (int i = 0; i < DPW; ++i) if (i * NQ / DPW == q) dma(i, kd, sd, kp, sd1);
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    acc[mb][4 * q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].x, v[4 * q], wino_c<FIRST>(acc[mb][4 * q]), 0, 0, 0);
                    if constexpr (nm > 1) acc[mb][4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].y, v[4 * q + 1], wino_c<FIRST>(acc[mb][4 * q + 1]), 0, 0, 0);
                    if constexpr (nm > 2) acc[mb][4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].z, v[4 * q + 2], wino_c<FIRST>(acc[mb][4 * q + 2]), 0, 0, 0);
                    if constexpr (nm > 3) acc[mb][4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].w, v[4 * q + 3], wino_c<FIRST>(acc[mb][4 * q + 3]), 0, 0, 0);
                }
                // one VALU instruction between two MFMAs of a wave costs about twice what it costs inside a burst (scripts/ubench/mfma_valu.hip:
                // +26 % against +13 % at one VALU per MFMA): keep the quad's vector work behind its eight MFMAs instead of letting the
                // scheduler interleave the two
                if constexpr (SB == 1) __builtin_amdgcn_sched_barrier(0);
                if constexpr (ABL != 4) {
                    if constexpr (q == 0) {
                        constexpr int l0 = 4 * (NQ - 1);                     // the last quad's points of THIS step, from the old row transforms
                        WinoFor<l0, NP>::run([&](auto xc) { constexpr int x = decltype(xc)::value; v[x] = wino_point<X0 + x>(t3, t2); });
                        if constexpr (RQ == 0) {
#pragma unroll
                            for (int r = 0; r < NROW; ++r) rows(r);
                        } else {
#pragma unroll
                            for (int r = 0; r < NROW; ++r) read_row(pbuf, r);    // patch k+1: lands in registers under quad 1's MFMAs
                        }
                    } else {
                        if constexpr (q == 1 && RQ == 1) {
#pragma unroll
                            for (int r = 0; r < NROW; ++r) rows(r);
                        }
                        WinoFor<4 * q - 4, 4 * q>::run([&](auto xc) { constexpr int x = decltype(xc)::value; v[x] = wino_point<X0 + x>(t3, t2); });
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        unit_cur(unit0);
        if (CS && tpw > 1) unit_nxt(unit0 + 1); else unit_nxt_is_cur();
        issue_first();
        int su = 0, sp1 = 1 % UR, sd = D % UR, sd1 = (D + 1) % UR;           // k % UR, (k+1) % UR, (k+D) % UR, (k+1+D) % UR  (CS: of the running step count)
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int x = 0; x < NP; ++x)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mb][x][r] = 0.0f;
        };
        if constexpr (!PEEL) zero_acc();
        auto unit_prologue = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
            __syncthreads();
#pragma unroll
            for (int r = 0; r < NROW; ++r) { read_row(s_p, r); rows(r); }
            WinoFor<0, NP>::run([&](auto xc) { constexpr int x = decltype(xc)::value; v[x] = wino_point<X0 + x>(t3, t2); });
            // patch D -> slot D % UR: the patch piece only (waves whose flex piece is one).  From here on every wave issues exactly DPW DMA
            // instructions per K step, which is what the vmcnt arithmetic below counts on.
            if (last_patch) dma_flex(0, 0, min(D, nk - 1), D % UR);
            su = 0; sp1 = 1 % UR; sd = D % UR; sd1 = (D + 1) % UR;
        };
        if (CS) unit_prologue();
        for (int t = 0; t < tpw; ++t) {
        if (!CS) unit_prologue();
        // (the first group of K steps is peeled: its first step starts the unit's accumulators from C = 0)
        auto kgroup = [&](int k, auto fc) __attribute__((always_inline)) {
            // vmcnt((D-BPS) DPW): everything older than the pieces of this wave's last D-BPS steps has landed: U slabs up to k+BPS-1 and patches up
            // to k+BPS (issued in step k+BPS-1-D).  After the barrier so have everyone's, and every wave has finished step k-1.
            if (ABL != 1) { __builtin_amdgcn_s_waitcnt(wino_vmcnt((D - BPS - EA) * DPW)); __syncthreads(); }
#pragma unroll
            for (int b = 0; b < BPS; ++b) {
                if (b == 0) kstep(k, su, sp1, sd, sd1, fc);
                else if (k + b < nk) kstep(k + b, su, sp1, sd, sd1, std::false_type{});
                su = su == UR - 1 ? 0 : su + 1; sp1 = sp1 == UR - 1 ? 0 : sp1 + 1; sd = sd == UR - 1 ? 0 : sd + 1; sd1 = sd1 == UR - 1 ? 0 : sd1 + 1;
            }
        };
        if constexpr (PEEL) { kgroup(0, std::true_type{}); for (int k = BPS; k < nk; k += BPS) kgroup(k, std::false_type{}); }
        else for (int k = 0; k < nk; k += BPS) kgroup(k, std::false_type{});
        if (!CS && t + 1 < tpw) {
            __syncthreads();
            unit_cur(unit0 + t + 1);
            unit_nxt_is_cur();
            issue_first();
        }
        // ---- output transform + bias -> activation -> batch-norm: this lane's block, class (PY, PX), channels m0 + 16 mb + 4 kq + r
        set_out_unit(unit0 + t);
        auto emit = [&](auto act) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                // (constants out of LDS: global loads here would sit between the stores of the two M blocks, and a wait for a load also waits for
                //  every store issued before it - one store round trip per unit)
                const float4 bi4 = *reinterpret_cast<const float4*>(&s_epi[16 * mb + 4 * kq]), sc4 = *reinterpret_cast<const float4*>(&s_epi[32 + 16 * mb + 4 * kq]),
                             sf4 = *reinterpret_cast<const float4*>(&s_epi[64 + 16 * mb + 4 * kq]);
                const float bi[4] = { bi4.x, bi4.y, bi4.z, bi4.w }, sc[4] = { sc4.x, sc4.y, sc4.z, sc4.w }, sf[4] = { sf4.x, sf4.y, sf4.z, sf4.w };
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float m[NP];
#pragma unroll
                    for (int x = 0; x < NP; ++x) m[x] = acc[mb][x][r];
                    float y[2][2];
                    wino_out2d<NY, NX>(m, y);
                    float* oc = obase + (size_t)(m0 + 16 * mb + 4 * kq + r) * ohw + (size_t)PY * Wo + PX;
#pragma unroll
                    for (int da = 0; da < 2; ++da)
#pragma unroll
                        for (int db = 0; db < 2; ++db) oc[(size_t)(2 * da) * Wo + 2 * db] = act(y[da][db], bi[r], sc[r], sf[r]);
                }
            }
        };
        if (blk_ok) {
            if (srt_act_is_plain_elu(actp)) emit([&](float y, float b, float s, float f) { return srt_dec_epilogue_elu(y, b, s, f); });
            else if (actp.ue != 0.0f) emit([&](float y, float b, float s, float f) { return srt_dec_epilogue(y, b, s, f, actp); });
            else emit([&](float y, float b, float s, float f) { return srt_dec_epilogue_lin(y, b, s, f, actp.lin); });
        }
        if (t + 1 < tpw) {
            if constexpr (!PEEL) zero_acc();
            if (CS && t + 2 < tpw) unit_nxt(unit0 + t + 2);                  // (the DMA state itself moved on D+1 steps before the unit ended; after the last unit it keeps valid addresses)
        }
        }                                                                    // units
    };
    if (cls == 0) body(std::integral_constant<int, 0>{});
    else if (cls == 1) body(std::integral_constant<int, 1>{});
    else if (cls == 2) body(std::integral_constant<int, 2>{});
    else body(std::integral_constant<int, 3>{});
}

// ------------------------------------------------------------------------------------------- encoder layers in Winograd form
// The stride-2 5x5 convolution of the encoder (spleeter.c:182-238; TF-SAME: pad 1 before, 2 after) seen from the input's four parity planes:
// output row oy reads the ODD input rows 2(oy-1)+1, 2oy+1, 2(oy+1)+1 with taps ky = 0, 2, 4 and the EVEN rows 2oy, 2(oy+1) with taps ky = 1, 3.
// Over blocks of 2 x 2 OUTPUT pixels that is F(2,3) on an odd plane and F(2,2) on an even one, per axis: the same 16 + 12 + 12 + 9 = 49
// products per block and (ci, co) against 100, with the same B / G / A matrices as the decoder (tests/test_wino_algebra.py holds the algebra).
// Differences from srt_dec_wino32, whose structure this kernel shares (class-split waves, two M blocks per wave, fenced bursts, staggered
// pairs, units as one continuous K stream):
//   * a block's 49 inputs are 49 DIFFERENT pixels (rows 4ya-1 .. 4ya+5, columns 4xb-1 .. 4xb+5; a class takes every second row / column of
//     the patch), so the workgroup's patch is (4 BA + 3) x (4 BB + 8) pixels per channel: 13-14 DMA pieces per K step beside the 26 of the U slab;
//   * the input must already be act(BN(raw)) - the non-linearity cannot ride through the transform - so the PRODUCER of the tensor supplies
//     that copy (the engine keeps one for the inputs of the layers that run here) and this kernel writes its own when p.outAct is set;
//   * the four classes ADD into the same 2 x 2 outputs, and they live in four different waves: at the end of a unit the waves exchange their
//     classes' outputs through LDS (24 KiB per M block; the second M block borrows the U ring slot the unit has just finished with) and each
//     lane finishes one of its four channels (+ bias -> raw, and act(BN(.)) -> the copy for the next layer): two barriers per unit.
// RB 1 (round 5): LDS bank discipline of the patch reads.  A lane's block columns are 4 floats apart and the four channels of a K step (kq) sit one channel
// pitch apart, so (MI355X_MICROARCH.md, LDS lane groups) a ds_read_b32 of the wave lands on 8 of 32 banks (4-way conflict: 8 LDS cycles for ONE value), and with
// the natural pitch of 792 floats (24 mod 64) even the aligned ds_read_b128 of the middle columns is 2-way conflicted (its 16-lane groups mix kq 0 / kq 1).
// RB 1 pads the channel pitch to a multiple of 64 floats - the DMA pieces per K step stay 13 (14): the padding lanes of the last piece fetch nothing - which
// makes the b128 conflict-free, and takes each outer column as the half of an aligned ds_read_b64 (2-way: 4 cycles): 12 instead of 24 LDS cycles per patch row
// of a 3-tap class, 8 instead of 16 for a 2-tap class.  MEASURED (r05b_tune): down3 / down4 / down5 0.751 / 0.638 / 0.572 ms against 0.738 / 0.631 / 0.562 with the
// round-4 reads - the LDS pipe is not what these layers wait for, and the extra result registers cost more than the conflicts - so RB 0 stays the default here
// (RB 2: the padded pitch alone, RB 3: the pairs alone; tuning builds, SRT_TUNE=encrb=1|2|3).  The same change is a 1 % gain in srt_dec_wino32, where it ships.
template <int BA, int BB, int NI, int ABL = 0, int PEEL = 0, int PR = 3, int RB = 0>     // PEEL: as srt_dec_wino32.  PR: depth of the PATCH ring (patches PR - 1 K steps ahead; the U slabs stay two ahead in a ring of three)
__global__ void __launch_bounds__(512, 1) srt_enc_wino32(const SrtConvParams p, const float* __restrict__ U, size_t u_stem, int tpw)
{
    static_assert(BA * BB * NI == 32 && (BA * BB) % 16 == 0, "tile");
    constexpr int UR = 3, D = 2, PD = PR - 1;                                // U ring of three, slabs two K steps ahead, one barrier per K step; patches PD steps ahead
    static_assert(PR >= 3 && PR <= 4, "patch ring");
    constexpr int UB1 = 4 * 16 * WINO_LD, UBUF = 2 * UB1, NUP = 26;
    constexpr int TH = 2 * BA, TW = 2 * BB;                                  // OUTPUT pixels per instance
    constexpr int PH = 4 * BA + 3, PROW = 4 * BB + 8, PR4 = PROW / 4;        // input patch
    constexpr int PCH0 = NI * PH * PROW;                                     // floats of one channel's patch
    constexpr int PCH = (RB == 1 || RB == 2) ? (PCH0 + 63) / 64 * 64 : PCH0; // channel pitch in LDS (RB 1 / 2: a multiple of 64 floats)
    constexpr int NF4 = PCH, NPP = (NF4 + 63) / 64, PBUF = NPP * 256;        // float4 of a K step's four channels; DMA pieces
    constexpr int NPIECE = NUP + NPP, DPW = (NPIECE + 7) / 8;
    static_assert(DPW == 5 && NPP >= 7 && NPP <= 14 && NPP == (PCH0 + 63) / 64, "piece map: per wave three U pieces, one U-or-patch piece, one patch piece (padding the pitch adds no piece)");
    constexpr int XBUF = 4 * 2 * 4 * 3 * 16 * 4;                             // class exchange of one M block: [writer class][group][kq][3 published channels][block][2 x 2 outputs] = 24 KiB
    static_assert(XBUF <= UBUF, "the second M block's exchange lives in a U ring slot");
    __shared__ __attribute__((aligned(16))) float s_all[UR * UBUF + PR * PBUF + XBUF];
    float* s_u = s_all;
    float* s_p = s_all + UR * UBUF;
    float* s_x = s_all + UR * UBUF + PR * PBUF;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int simd = wave & 3, hi = wave >> 2, g = simd & 1;
    const int cls = (simd >> 1) == 0 ? (hi ? 3 : 0) : (hi ? 2 : 1);
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int tilesX = (Wo + TW - 1) / TW, tilesY = (Ho + TH - 1) / TH;
    const int MB2 = p.Cout / 32, MB = p.Cout / 16;
    const int walk = (tpw >> 9) & 1;                                         // column walk (wino_sp_xy)
    const bool mfast = ((tpw >> 10) & 1) != 0;                               // M-block pair fastest in the launch order (see srt_dec_wino32)
    tpw &= 255;
    const int colrun = !walk ? 1 : tilesY % tpw == 0 ? tpw : tpw % tilesY == 0 ? tilesY : 1;
    const int nsp = tilesX * tilesY, groups = (p.ntiles + NI - 1) / NI, upw = nsp * groups / tpw;
    const int pos = srt_xcd_order(upw * MB2 * p.nstems);
    const int wsel = pos / upw, mblk2 = mfast ? pos % MB2 : wsel % MB2, stem = mfast ? pos / (MB2 * upw) : wsel / MB2, unit0 = (mfast ? (pos / MB2) % upw : pos % upw) * tpw;
    const int m0 = mblk2 * 32;
    const SrtAct actp = srt_act_params(srt_act_kind(p, stem), p.variant);
    const size_t hw = (size_t)p.H * p.W;
    const float* up = U + stem * u_stem + (size_t)(2 * mblk2) * UB1;

    const int blk = g * 16 + l15, il = blk / (BA * BB), ba = (blk / BB) % BA, bb = blk % BB;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)s_all;

    // ---- DMA pieces of a K step: 0..25 the U slab, 26.. the patch.  Wave w: pieces w, w+8, w+16 (U), w+24 (U for waves 0-1, patch piece w-2
    // otherwise: the "flex" piece, one buffer_load ... lds with scalar-selected operands), w+32 (patch piece w+6, the last one again past the end)
    const bool flex_patch = wave >= 2;
    unsigned dvoff[3], dm0[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dvoff[i] = (unsigned)((wave + 8 * i) * 1024 + lane * 16);
        dm0[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((wave + 8 * i) * 1024));
    }
    const int fpiece = wave + 24, ppiece = min(wave + 6, NPP - 1);           // flex: slab piece (waves 0-1) or patch piece fpiece - 26; pure patch piece
    const unsigned fm0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(flex_patch ? UR * UBUF * 4 + (fpiece - NUP) * 1024 : fpiece * 1024));
    const unsigned pm0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(UR * UBUF * 4 + ppiece * 1024));
    constexpr unsigned OOR = 0x80000000u;
    const unsigned prec = (unsigned)min((size_t)0x7fffffff, (size_t)4 * NI * p.srcA_tile), frec = flex_patch ? prec : 0x7fffffffu;
    unsigned c_flo, c_fhi, c_plo, c_phi, c_fvoff, c_pvoff, n_flo, n_fhi, n_plo, n_phi, n_fvoff, n_pvoff;      // DMA state of the current / next unit
    struct UnitBase { unsigned flo, fhi, plo, phi; };
    auto unit_base = [&](int unit) {
        const int tile0 = ABL == 10 ? 0 : (unit / nsp) * NI;
        const size_t bi_ = (size_t)(p.srcA + stem * p.srcA_stem + tile0 * p.srcA_tile), bu_ = (size_t)up;
        UnitBase u;
        u.plo = __builtin_amdgcn_readfirstlane((unsigned)bi_); u.phi = __builtin_amdgcn_readfirstlane((unsigned)(bi_ >> 32));
        u.flo = __builtin_amdgcn_readfirstlane(flex_patch ? (unsigned)bi_ : (unsigned)bu_); u.fhi = __builtin_amdgcn_readfirstlane(flex_patch ? (unsigned)(bi_ >> 32) : (unsigned)(bu_ >> 32));
        return u;
    };
    // patch float4 e = ((c * NI + ii) * PH + row) * PR4 + j  <-  channel 4k+c, instance tile0+ii, input row 4 ty0 - 1 + row, columns 4 tx0 - 4 + 4j .. +3
    auto patch_voff = [&](int unit, int piece) {
        if (ABL == 10) unit = 0;                                             // (tuning builds: every patch from one place - L2 hits - to see what the patch traffic costs)
        int sx_, sy_; wino_sp_xy(unit % nsp, tilesX, colrun, sx_, sy_);
        const int tile0 = (unit / nsp) * NI, tx0 = sx_ * BB, ty0 = sy_ * BA;     // tile origin in BLOCKS
        const int e = piece * 64 + lane;
        const int c = e / (PCH / 4), rem = e % (PCH / 4);                    // float4 `rem` of channel c's patch (rem >= PCH0 / 4: padding of the pitch)
        const int j = rem % PR4, row = (rem / PR4) % PH, ii = rem / (PR4 * PH);
        const int gy = 4 * ty0 - 1 + row, gx = 4 * tx0 - 4 + 4 * j;
        const bool ok = piece >= 0 && e < NF4 && rem < PCH0 / 4 && tile0 + ii < p.ntiles && gy >= 0 && gy < p.H && gx >= 0 && gx + 3 < p.W;
        return ok ? 4u * (unsigned)((size_t)ii * p.srcA_tile + (size_t)c * hw + (size_t)gy * p.W + gx) : OOR;
    };
    auto unit_cur = [&](int unit) {
        const UnitBase u = unit_base(unit); c_flo = u.flo; c_fhi = u.fhi; c_plo = u.plo; c_phi = u.phi;
        c_fvoff = flex_patch ? patch_voff(unit, fpiece - NUP) : (unsigned)(fpiece * 1024 + lane * 16); c_pvoff = patch_voff(unit, ppiece);
    };
    auto unit_nxt = [&](int unit) {
        const UnitBase u = unit_base(unit); n_flo = u.flo; n_fhi = u.fhi; n_plo = u.plo; n_phi = u.phi;
        n_fvoff = flex_patch ? patch_voff(unit, fpiece - NUP) : (unsigned)(fpiece * 1024 + lane * 16); n_pvoff = patch_voff(unit, ppiece);
    };
    auto unit_nxt_is_cur = [&]() { n_flo = c_flo; n_fhi = c_fhi; n_plo = c_plo; n_phi = c_phi; n_fvoff = c_fvoff; n_pvoff = c_pvoff; };
    auto unit_advance = [&]() { c_flo = n_flo; c_fhi = n_fhi; c_plo = n_plo; c_phi = n_phi; c_fvoff = n_fvoff; c_pvoff = n_pvoff; };
    const unsigned kstep_bytes = (unsigned)(16 * hw), ustep_bytes = (unsigned)((size_t)MB * UB1 * 4);
    auto dma_u = [&](int i, int ku, int ubuf) {
        const float* src = up + (size_t)ku * MB * UB1;
        const unsigned dst = dm0[i] + (unsigned)ubuf * (unsigned)(UBUF * 4);
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(dvoff[i]), "s"(src), "s"(dst) : "memory");
    };
    auto dma_flex = [&](int ku, int ubuf, int kp, int pslot) {
        i32x4 rs;
        rs.x = (int)c_flo; rs.y = (int)(c_fhi & 0xffffu); rs.z = (int)frec; rs.w = 0x00020000;
        const unsigned soff = flex_patch ? (unsigned)kp * kstep_bytes : (unsigned)ku * ustep_bytes;
        const unsigned dst = fm0 + (flex_patch ? (unsigned)pslot * (unsigned)(PBUF * 4) : (unsigned)ubuf * (unsigned)(UBUF * 4));
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(c_fvoff), "s"(rs), "s"(dst), "s"(soff) : "memory");
    };
    auto dma_patch = [&](int kp, int pslot) {
        i32x4 rs;
        rs.x = (int)c_plo; rs.y = (int)(c_phi & 0xffffu); rs.z = (int)prec; rs.w = 0x00020000;
        const unsigned soff = (unsigned)kp * kstep_bytes;
        const unsigned dst = pm0 + (unsigned)pslot * (unsigned)(PBUF * 4);
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(c_pvoff), "s"(rs), "s"(dst), "s"(soff) : "memory");
    };
    auto dma = [&](int i, int ku, int ubuf, int kp, int pslot) {             // piece i of this wave
        if (i < 3) dma_u(i, ku, ubuf);
        else if (i == 3) dma_flex(ku, ubuf, kp, pslot);
        else dma_patch(kp, pslot);
    };
    const int poff = kq * PCH + (il * PH + 4 * ba) * PROW + 4 * bb + 3;      // this lane's block: patch rows +0..6, columns +0..6 (input column 4xb-1 first)
    const int aoff = (kq * 16 + l15) * WINO_LD;
    const int nk = p.Cin / 4;
    const size_t ohw = (size_t)Ho * Wo;
    size_t obase; bool blk_ok;
    auto set_out_unit = [&](int unit) {
        int sx_, sy_; wino_sp_xy(unit % nsp, tilesX, colrun, sx_, sy_);
        const int tile = (unit / nsp) * NI + il, oy0 = sy_ * TH + 2 * ba, ox0 = sx_ * TW + 2 * bb;
        blk_ok = tile < p.ntiles && oy0 < Ho && ox0 < Wo;                    // (Ho, Wo even: a block is inside the image or outside it)
        obase = stem * p.out_stem + (blk_ok ? tile : 0) * p.out_tile + (size_t)(blk_ok ? oy0 : 0) * Wo + (blk_ok ? ox0 : 0);
    };

    auto body = [&](auto cc) __attribute__((always_inline)) {
        constexpr int CLS = decltype(cc)::value;
        constexpr int X0 = CLS == 0 ? WINO_C11 : CLS == 1 ? WINO_C10 : CLS == 2 ? WINO_C01 : WINO_C00;
        constexpr bool Y3 = CLS < 2, X3 = !(CLS & 1);                        // odd input plane (3 taps, 4 points) along y / x
        constexpr int NY = Y3 ? 4 : 3, NX = X3 ? 4 : 3, NP = NY * NX, NQ = (NP + 3) / 4, NROW = Y3 ? 4 : 3;
        constexpr int RQ = CLS >= 2 ? 0 : 1;                                 // quad whose burst carries the row transforms (staggered pairs)
        f32x4 acc[2][NP];
        float t3[4][4], t2[4][3];
        float xr[4][4];                                                      // patch row r of the class: its plane's 4 (3) pixels
        // a block's columns are 4 floats apart for every lane and the channel pitch is a multiple of 4 floats (16-byte DMA pieces), so a ds_read_b32 of the
        // wave touches 16 of the 64 banks: a 4-way conflict, as long in the LDS pipe as a conflict-free ds_read_b128.  The middle pair of a row therefore
        // comes from ONE aligned b128 (columns 4xb .. 4xb+3: the odd plane takes .y / .w, the even plane .x / .z) and only the outer one or two values
        // are single reads: 3 (2) LDS instructions per row instead of 4 (3), half the conflicted ones.
        auto read_row = [&](const float* pbuf, int r) {
            const float* b = pbuf + poff - 3 + (Y3 ? 2 * r : 2 * r + 1) * PROW;      // input column 4xb - 4: 16-byte aligned
            const float4 m = *reinterpret_cast<const float4*>(b + 4);
            if constexpr (RB == 1 || RB == 3) {                              // outer columns as halves of aligned b64 pairs (see RB above)
                const float2 hi2 = *reinterpret_cast<const float2*>(b + 8);
                if constexpr (X3) { const float2 lo2 = *reinterpret_cast<const float2*>(b + 2); xr[r][0] = lo2.y; xr[r][1] = m.y; xr[r][2] = m.w; xr[r][3] = hi2.y; }
                else { xr[r][0] = m.x; xr[r][1] = m.z; xr[r][2] = hi2.x; }
            } else if constexpr (X3) { xr[r][0] = b[3]; xr[r][1] = m.y; xr[r][2] = m.w; xr[r][3] = b[9]; }
            else { xr[r][0] = m.x; xr[r][1] = m.z; xr[r][2] = b[8]; }
        };
        auto rows = [&](int r) {
            if constexpr (X3) wino_in3(xr[r][0], xr[r][1], xr[r][2], xr[r][3], t3[r]);
            else wino_in2(xr[r][0], xr[r][1], xr[r][2], t2[r]);
        };
        float v[NP];
        auto issue_first = [&]() {
#pragma unroll
            for (int j = 0; j < D; ++j)
#pragma unroll
                for (int i = 0; i < DPW; ++i) dma(i, min(j, nk - 1), j, min(j, nk - 1), j);
#pragma unroll
            for (int j = D; j < PD; ++j) {                                   // deeper patch ring: the patches between the slab lead and the patch lead
                if (flex_patch) dma_flex(0, 0, min(j, nk - 1), j);
                dma_patch(min(j, nk - 1), j);
            }
        };
        float4 a0[2], a1[2];
        auto kstep = [&](int k, int su, int sp1, int sd, int sd1, auto fc) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(fc)::value;                      // first K step of a unit: C = 0 (wino_c)
            if (k + 1 + PD == nk) unit_advance();                            // the DMA state moves to the next unit with the first patch that belongs to it
            const int kd = k + D >= nk ? k + D - nk : k + D, kp = k + 1 + PD >= nk ? k + 1 + PD - nk : k + 1 + PD;
            const float* ub = s_u + su * UBUF + aoff + X0;
            const float* pbuf = s_p + sp1 * PBUF;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) { a0[mb] = *reinterpret_cast<const float4*>(ub + mb * UB1); a1[mb] = *reinterpret_cast<const float4*>(ub + mb * UB1 + 4); }
            if constexpr (RQ == 0 && ABL != 4) {
#pragma unroll
                for (int r = 0; r < NROW; ++r) read_row(pbuf, r);
            }
            __builtin_amdgcn_sched_barrier(0);
            WinoFor<0, NQ>::run([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int nm = (4 * q + 4 <= NP) ? 4 : NP - 4 * q;
                float4 a[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    a[mb] = a0[mb]; a0[mb] = a1[mb];
                    if constexpr (q + 2 < NQ && ABL != 5) a1[mb] = *reinterpret_cast<const float4*>(ub + mb * UB1 + 4 * (q + 2));
                }
                if constexpr (ABL != 3) {
#pragma unroll
                    for (int i = 0; i < DPW; ++i) if (i * NQ / DPW == q) dma(i, kd, sd, kp, sd1);
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    acc[mb][4 * q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].x, v[4 * q], wino_c<FIRST>(acc[mb][4 * q]), 0, 0, 0);
                    if constexpr (nm > 1) acc[mb][4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].y, v[4 * q + 1], wino_c<FIRST>(acc[mb][4 * q + 1]), 0, 0, 0);
                    if constexpr (nm > 2) acc[mb][4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].z, v[4 * q + 2], wino_c<FIRST>(acc[mb][4 * q + 2]), 0, 0, 0);
                    if constexpr (nm > 3) acc[mb][4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].w, v[4 * q + 3], wino_c<FIRST>(acc[mb][4 * q + 3]), 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);                           // the quad's vector work stays behind its eight MFMAs
                if constexpr (ABL != 4) {
                    if constexpr (q == 0) {
                        constexpr int l0 = 4 * (NQ - 1);
                        WinoFor<l0, NP>::run([&](auto xc) { constexpr int x = decltype(xc)::value; v[x] = wino_point<X0 + x>(t3, t2); });
                        if constexpr (RQ == 0) {
#pragma unroll
                            for (int r = 0; r < NROW; ++r) rows(r);
                        } else {
#pragma unroll
                            for (int r = 0; r < NROW; ++r) read_row(pbuf, r);
                        }
                    } else {
                        if constexpr (q == 1 && RQ == 1) {
#pragma unroll
                            for (int r = 0; r < NROW; ++r) rows(r);
                        }
                        WinoFor<4 * q - 4, 4 * q>::run([&](auto xc) { constexpr int x = decltype(xc)::value; v[x] = wino_point<X0 + x>(t3, t2); });
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        unit_cur(unit0);
        if (tpw > 1) unit_nxt(unit0 + 1); else unit_nxt_is_cur();
        issue_first();
        if constexpr (!PEEL) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int x = 0; x < NP; ++x)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mb][x][r] = 0.0f;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0)
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NROW; ++r) { read_row(s_p, r); rows(r); }
        WinoFor<0, NP>::run([&](auto xc) { constexpr int x = decltype(xc)::value; v[x] = wino_point<X0 + x>(t3, t2); });
        // patch D -> slot D: the wave's patch pieces only (its U pieces of this round went out above); from here on exactly DPW DMA instructions per step
        if (flex_patch) dma_flex(0, 0, min(PD, nk - 1), PD % PR);
        dma_patch(min(PD, nk - 1), PD % PR);
        int su = 0, sp1 = 1 % PR, sd = D % UR, sd1 = (PD + 1) % PR;
        // epilogue constants of this lane's two output channels, before the K stream: loaded in the epilogue they sat between its stores, and a wait
        // for a load also waits for every store issued before it (three store round trips per unit)
        float ebias[2], esc[2], esf[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const size_t ci = stem * p.coeff_stem + m0 + 16 * mb + 4 * kq + CLS;
            ebias[mb] = p.bias[ci]; esc[mb] = p.outAct ? p.bnScale[ci] : 0.0f; esf[mb] = p.outAct ? p.bnShift[ci] : 0.0f;
        }
        for (int t = 0; t < tpw; ++t) {
        if constexpr (PEEL) {                                                // (the unit's accumulators start from C = 0: no zeroing pass)
            if (ABL != 1) { __builtin_amdgcn_s_waitcnt(wino_vmcnt((D - 1) * DPW)); __syncthreads(); }
            kstep(0, su, sp1, sd, sd1, std::true_type{});
            su = su == UR - 1 ? 0 : su + 1; sp1 = sp1 == PR - 1 ? 0 : sp1 + 1; sd = sd == UR - 1 ? 0 : sd + 1; sd1 = sd1 == PR - 1 ? 0 : sd1 + 1;
        }
        for (int k = PEEL ? 1 : 0; k < nk; ++k) {
            // vmcnt: everything older than this wave's last step of pieces has landed (U slab k, patch k+1); the prologue's extra patch pieces are older still
            if (ABL != 1) { __builtin_amdgcn_s_waitcnt(wino_vmcnt((D - 1) * DPW)); __syncthreads(); }
            kstep(k, su, sp1, sd, sd1, std::false_type{});
            su = su == UR - 1 ? 0 : su + 1; sp1 = sp1 == PR - 1 ? 0 : sp1 + 1; sd = sd == UR - 1 ? 0 : sd + 1; sd1 = sd1 == PR - 1 ? 0 : sd1 + 1;
        }
        // ---- unit epilogue: the four classes' 2 x 2 outputs of a (channel, block) live in four waves and meet in LDS.  Lane (kq, l15) of the class-c
        // wave finishes channel 4 kq + c of each M block for its block: sum in the fixed order (C11 + C10) + (C01 + C00), + bias -> raw; act(BN(.)) ->
        // the copy.  It keeps its OWN class's term of that channel in registers and publishes only the three channels the other classes' lanes
        // finish: X[writer class][g][kq][pr][l15] float4 (pr = rank of the channel among those three), 24 KiB per M block.  M block 0 goes to the
        // dedicated buffer; M block 1 to the U ring slot the unit's last K step has just finished with (slot sd after the loop's increment: the next
        // DMA into it is issued behind the NEXT K step's barrier, after every wave has read its terms).  Two barriers per unit (round 3: four, one
        // M block at a time through one 32 KiB buffer):
        //   transform both M blocks, write X0 | barrier (everyone is past the last K step: the ring slot is free; X0 complete) |
        //   write X1, read X0, finish M block 0 | barrier (X1 complete) | read X1, finish M block 1
        set_out_unit(unit0 + t);
        // ABL (tuning builds, wrong results): 6 no epilogue (the accumulators stay alive through a store that never executes), 7 epilogue without its two
        // barriers, 8 without its global stores, 9 without the LDS exchange
        if constexpr (ABL == 6) {
            if (p.ntiles < 0) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int x = 0; x < NP; ++x) *reinterpret_cast<f32x4*>(p.outRaw + (size_t)(mb * NP + x) * 4) = acc[mb][x];
            }
        } else {
            float4 own[2], pub1[3];
            float* x1 = s_u + sd * UBUF;
            const int xo = (((CLS * 2 + g) * 4 + kq) * 3) * 64 + l15 * 4;          // this lane's three float4 slots (writer view), + pr * 64
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float m[NP];
#pragma unroll
                    for (int x = 0; x < NP; ++x) m[x] = acc[mb][x][r];
                    float y[2][2];
                    wino_out2d<NY, NX>(m, y);
                    const float4 y4 = make_float4(y[0][0], y[0][1], y[1][0], y[1][1]);
                    if (r == CLS) own[mb] = y4;
                    else if (mb == 0) { if constexpr (ABL != 9) *reinterpret_cast<float4*>(&s_x[xo + (r < CLS ? r : r - 1) * 64]) = y4; else own[0].x += y4.y; }
                    else pub1[r < CLS ? r : r - 1] = y4;
                }
            }
            if constexpr (ABL != 7 && ABL != 9) __syncthreads();
#pragma unroll
            for (int j = 0; j < 3; ++j) { if constexpr (ABL != 9) *reinterpret_cast<float4*>(&x1[xo + j * 64]) = pub1[j]; else own[1].x += pub1[j].y; }
            auto finish = [&](int mb, const float* xb) __attribute__((always_inline)) {
                float4 c4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)                                     // class c's term of channel 4 kq + CLS: its lane (g, kq, l15), slot pr(CLS)
                    if (c != CLS) { if constexpr (ABL != 9) c4[c] = *reinterpret_cast<const float4*>(&xb[((((c * 2 + g) * 4 + kq) * 3) + (CLS < c ? CLS : CLS - 1)) * 64 + l15 * 4]); else c4[c] = own[mb]; }
                c4[CLS] = own[mb];
                const int co = m0 + 16 * mb + 4 * kq + CLS;
                const float bias = ebias[mb];
                float o[4];
                o[0] = ((c4[0].x + c4[1].x) + (c4[2].x + c4[3].x)) + bias; o[1] = ((c4[0].y + c4[1].y) + (c4[2].y + c4[3].y)) + bias;
                o[2] = ((c4[0].z + c4[1].z) + (c4[2].z + c4[3].z)) + bias; o[3] = ((c4[0].w + c4[1].w) + (c4[2].w + c4[3].w)) + bias;
                if (ABL == 8 ? (blk_ok && p.ntiles < 0) : blk_ok) {
                    float* dst = p.outRaw + obase + (size_t)co * ohw;
                    *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
                    *reinterpret_cast<float2*>(dst + Wo) = make_float2(o[2], o[3]);
                    if (p.outAct) {
                        const float sc = esc[mb], sf = esf[mb];
                        float* da = p.outAct + obase + (size_t)co * ohw;
                        *reinterpret_cast<float2*>(da) = make_float2(srt_enc_epilogue(o[0], sc, sf, actp), srt_enc_epilogue(o[1], sc, sf, actp));
                        *reinterpret_cast<float2*>(da + Wo) = make_float2(srt_enc_epilogue(o[2], sc, sf, actp), srt_enc_epilogue(o[3], sc, sf, actp));
                    }
                }
            };
            finish(0, s_x);
            if constexpr (ABL != 7 && ABL != 9) __syncthreads();
            finish(1, x1);
        }
        if (t + 1 < tpw) {
            if constexpr (!PEEL) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int x = 0; x < NP; ++x)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[mb][x][r] = 0.0f;
            }
            if (t + 2 < tpw) unit_nxt(unit0 + t + 2);
        }
        }                                                                    // units
    };
    if (cls == 0) body(std::integral_constant<int, 0>{});
    else if (cls == 1) body(std::integral_constant<int, 1>{});
    else if (cls == 2) body(std::integral_constant<int, 2>{});
    else body(std::integral_constant<int, 3>{});
}

// ------------------------------------------------------------------------------------------- launcher
// Which decoder layers (bit i = up(i+1)) run this form.  The default is the set measured faster than the direct kernels on
// MI355X at 64 tiles x 4 stems; a -DSRT_TUNING build overrides it with SRT_TUNE=wino=<mask>.
#ifndef SRT_WINO_DEFAULT_MASK
#define SRT_WINO_DEFAULT_MASK 31      // up1..up5 (round 2 kept up1 direct: 0.80 vs 0.79 ms with the 16-channel kernel; the 32-channel one is measured in DESIGN.md section 3.2b)
#endif
#ifdef SRT_TUNING
static int wino_tune(const char* key)                                        // key includes the '='
{
    const char* e = getenv("SRT_TUNE");
    const char* q = e ? strstr(e, key) : nullptr;
    return (q && (q == e || q[-1] == ',')) ? atoi(q + strlen(key)) : -1;
}
#endif
int srt_wino_mask()
{
#ifdef SRT_TUNING
    const int m = wino_tune("wino=");
    if (m >= 0) return m;
#endif
    return SRT_WINO_DEFAULT_MASK;
}
int srt_wino_force()               // tuning builds: SRT_TUNE=...,winoforce=1 runs this form for small batches too (parity tests at CPU-reference sizes)
{
#ifdef SRT_TUNING
    return wino_tune("winoforce=") > 0;
#else
    return 0;
#endif
}
// workgroups that walk `tpw` units each: up to 8 (measured: 1 -> 4 takes up5 from 1.43 to 1.34 ms, up4 1.24 -> 1.20, up2 1.12 -> 1.10; 8 another
// 0.3-0.5 %), as long as it divides the units of a (stem, M block) and leaves two workgroups per CU
static int wino_tpw(long wgs, long units)
{
#ifdef SRT_TUNING
    const int t = wino_tune("winotpw=");
    if (t > 0 && units % t == 0) return t;
#endif
    int tpw = 1;
    while (tpw < 8 && wgs / (2 * tpw) >= 512 && units % (2 * tpw) == 0) tpw *= 2;
    return tpw;
}
// bit 9 of the kernels' tpw argument: the units of a workgroup walk down a tile column (wino_sp_xy).  SRT_TUNE=winowalk=0|1 in tuning builds.
#ifndef SRT_WINO_WALK_DEFAULT
#define SRT_WINO_WALK_DEFAULT 1
#endif
static int wino_walk_bit()
{
#ifdef SRT_TUNING
    const int v = wino_tune("winowalk=");
    if (v >= 0) return v ? 512 : 0;
#endif
    return SRT_WINO_WALK_DEFAULT ? 512 : 0;
}
// bit 10 of the kernels' tpw argument: the M-block pair is the fastest index of the launch order (srt_dec_wino32 / srt_enc_wino32) for layers whose
// transformed weights are small enough for all pairs of a stem to share an XCD's L2 (4 MiB) beside the patches: `ubytes` per stem at most 2 MiB.
// SRT_TUNE=winomf=0|1 forces it off / on in tuning builds.
#ifndef SRT_WINO_MFAST_DEFAULT
#define SRT_WINO_MFAST_DEFAULT 0
#endif
static int wino_mfast_bit(size_t ubytes, int mb2)
{
    if (mb2 < 2) return 0;
#ifdef SRT_TUNING
    const int v = wino_tune("winomf=");
    if (v >= 0) return v ? 1024 : 0;
#endif
    return SRT_WINO_MFAST_DEFAULT && ubytes <= ((size_t)2 << 20) ? 1024 : 0;
}
// Layers with a multiple of 32 output channels (up2..up4) run the 32-channel workgroup, srt_dec_wino32, in the arrangement measured
// fastest at 64 tiles x 4 stems: rings of three, fenced VALU bursts, staggered wave pairs, the units of a
// workgroup as one continuous K stream.  SRT_TUNE=wino32=0|1 and winocfg=<n> select the other measured arrangements in tuning builds.
#ifndef SRT_WINO32_DEFAULT
#define SRT_WINO32_DEFAULT 1
#endif
#define SRT_WINO32_SHIPPED 2, 16, 0, 3, 2, 1, 1, 0, 1, 1
#define SRT_WINO_RING 3
static int wino32_on()
{
#ifdef SRT_TUNING
    const int v = wino_tune("wino32=");
    if (v >= 0) return v;
#endif
    return SRT_WINO32_DEFAULT;
}
int srt_launch_dec_wino(const SrtConvParams& p, const float* U, size_t u_stem, hipStream_t s)
{
    if (!U || p.in16 || p.out16 || p.srcA_tile != p.srcB_tile || (size_t)16 * p.srcA_tile > 0x7fffffffu || p.Cout % 16 || p.Cin % 4 || p.CA % 4 || (p.H & 1) || (p.W & 3)) return 1;
    const int MB = p.Cout / 16;
    if (p.Cout % 32 == 0 && p.Cin >= 32 && p.H >= 4 && p.W >= 16 && p.W < 32 && wino32_on()) {     // 4 x 16 .. 28 inputs (up1 of 256 x 1024 tiles): two instances per workgroup
        const long units = (long)((p.W + 15) / 16) * ((p.H + 3) / 4) * ((p.ntiles + 1) / 2), wgs = units * (p.Cout / 32) * p.nstems;
        const int tpw = wino_tpw(wgs, units);
        SRT_LAUNCH((srt_dec_wino32<2, 8, 0, 3, 2, 1, 1, 0, 1, 1, 2>), dim3((unsigned)(wgs / tpw)), dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit());
        return srt_launch_status();
    }
    if (p.Cout % 32 == 0 && p.Cin >= 32 && p.H >= 4 && p.W >= 32 && wino32_on()) {      // (Cin >= 32: at least 8 K steps, the continuous stream looks D + 1 = 3 steps ahead)
        const long units = (long)((p.W + 31) / 32) * ((p.H + 3) / 4) * p.ntiles, wgs = units * (p.Cout / 32) * p.nstems;
        int tpw = wino_tpw(wgs, units);
        const dim3 grid((unsigned)(wgs / tpw));
#ifdef SRT_TUNING
        if (wino_tune("winoprio=") > 0) tpw |= 256;
#define W32(...) do { SRT_LAUNCH((srt_dec_wino32<2, 16, __VA_ARGS__>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0; } while (0)
        switch (wino_tune("winoabl=")) {                                     // ablations of the shipped arrangement (wrong results, timing only)
        case 1: W32(1, 3, 2, 1, 1, 0, 1, 1);
        case 3: W32(3, 3, 2, 1, 1, 0, 1, 1);
        case 4: W32(4, 3, 2, 1, 1, 0, 1, 1);
        case 5: W32(5, 3, 2, 1, 1, 0, 1, 1);
        }
        switch (wino_tune("winocfg=")) {                                     // <ABL, UR, D, BPS, SB, EA, ST, CS>: arrangements measured on the way
        case 1: W32(0, 3, 2, 1, 0, 0, 0);                                    // first version: compiler-interleaved VALU
        case 2: W32(0, 3, 2, 1, 1, 0, 0);                                    // + fenced bursts
        case 3: W32(0, 3, 2, 1, 1, 0, 1);                                    // + staggered pairs
        case 4: W32(0, 5, 3, 2, 1, 0, 0);                                    // fenced bursts, barrier per two steps
        case 5: W32(0, 4, 3, 1, 1, 1, 0);                                    // early A-operand reads
        case 6: W32(0, 3, 2, 1, 1, 0, 2);                                    // stagger the other way round
        case 7: W32(0, 5, 4, 1, 1, 0, 1);                                    // slabs four steps ahead
        case 8: W32(0, 5, 3, 2, 1, 0, 1, 1);                                 // rings of 5, barrier per two steps, continuous K stream across units
        case 10: W32(0, 5, 3, 2, 1, 0, 1, 0);                                // the same without the continuous stream
        }
        if (wino_tune("winopeel=") == 1) W32(0, 3, 2, 1, 1, 0, 1, 1, 1, 1);            // the shipped arrangement with the first K step peeled (C = 0)
        if (wino_tune("decrb=") == 0) { SRT_LAUNCH((srt_dec_wino32<2, 16, 0, 3, 2, 1, 1, 0, 1, 1, 1, 0, 0>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit() | wino_mfast_bit(u_stem * 4, p.Cout / 32)); return 0; }      // round-4 patch reads (natural pitch, b32 outer columns)
#undef W32
#endif
        SRT_LAUNCH((srt_dec_wino32<SRT_WINO32_SHIPPED>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit() | wino_mfast_bit(u_stem * 4, p.Cout / 32));
        return srt_launch_status();
    }
    if (p.H >= 8 && p.W >= 32) {
        const long units = (long)((p.W + 31) / 32) * ((p.H + 7) / 8) * p.ntiles, wgs = units * MB * p.nstems;
        const int tpw = wino_tpw(wgs, units);
        const dim3 grid((unsigned)(wgs / tpw));
#ifdef SRT_TUNING
        switch (wino_tune("winoabl=")) {
        case 1: SRT_LAUNCH((srt_dec_wino<4, 16, 1, 1>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 2: SRT_LAUNCH((srt_dec_wino<4, 16, 1, 2>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 3: SRT_LAUNCH((srt_dec_wino<4, 16, 1, 3>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 4: SRT_LAUNCH((srt_dec_wino<4, 16, 1, 4>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 5: SRT_LAUNCH((srt_dec_wino<4, 16, 1, 5>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        }
        if (wino_tune("dec16rb=") == 0) { SRT_LAUNCH((srt_dec_wino<4, 16, 1, 0, 3, 0, 0, 0>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0; }      // round-4 patch reads
        if (wino_tune("winosb=") == 1) { SRT_LAUNCH((srt_dec_wino<4, 16, 1, 0, 3, 1>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0; }
        if (wino_tune("winocs=") == 1) { SRT_LAUNCH((srt_dec_wino<4, 16, 1, 0, 3, 0, 1>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0; }
        if (wino_tune("winocs=") == 2) { SRT_LAUNCH((srt_dec_wino<4, 16, 1, 0, 4, 0, 1>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0; }
        switch (wino_tune("winoring=")) {
        case 4: SRT_LAUNCH((srt_dec_wino<4, 16, 1, 0, 4>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 5: SRT_LAUNCH((srt_dec_wino<4, 16, 1, 0, 5>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 6: SRT_LAUNCH((srt_dec_wino<4, 16, 1, 0, 6>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        }
#endif
        SRT_LAUNCH((srt_dec_wino<4, 16, 1, 0, SRT_WINO_RING>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit());
    } else if (p.H >= 4 && p.W >= 16) {
        const long units = (long)((p.W + 15) / 16) * ((p.H + 3) / 4) * ((p.ntiles + 3) / 4), wgs = units * MB * p.nstems;
        const int tpw = wino_tpw(wgs, units);
        SRT_LAUNCH((srt_dec_wino<2, 8, 4>), dim3((unsigned)(wgs / tpw)), dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit());
    } else return 1;
    return srt_launch_status();
}

// Encoder layers in Winograd form (srt_enc_wino32): the input must be the producer's act(BN(raw)) copy.  Returns 1 when the layer is not covered.
// Which layers: down3..down6 (Cin >= 32: at least 8 K steps per unit, the continuous stream looks D + 1 = 3 steps ahead).  down3 pays since the
// direct layer in front writes the act(BN(.)) copy itself (srt_enc_mfma2's second output) and the epilogue no longer loads its constants between
// its stores: down2 + down3 1.87 -> 1.76 ms.  down2 (Cin = 16, 4 K steps) would be all epilogue, and down1 would have to write a 1 GB copy.
// SRT_TUNE=encwino=0 keeps the direct kernels, encwino=<min Cin> moves the threshold (tuning builds).
int srt_enc_producer_copy()
{
#ifdef SRT_TUNING
    const int v = wino_tune("enccopy=");
    if (v >= 0) return v;
#endif
    return 1;
}
int srt_enc_wino_covers(int Cin, int Cout, int H, int W)
{
    int min_cin = 32;
#ifdef SRT_TUNING
    const int v = wino_tune("encwino=");
    if (v == 0) return 0;
    if (v > 0) min_cin = v;
#endif
    if (Cout % 32 || Cin % 4 || Cin < min_cin || Cin < 32 || (H & 3) || (W & 3)) return 0;
    return H / 2 >= 4 && W / 2 >= 16;
}
int srt_launch_enc_wino(const SrtConvParams& p, const float* U, size_t u_stem, hipStream_t s)
{
    if (!U || p.in16 || p.out16 || p.inScale || (size_t)16 * p.srcA_tile > 0x7fffffffu || !srt_enc_wino_covers(p.Cin, p.Cout, p.H, p.W)) return 1;
#ifdef SRT_TUNING
    if (wino_tune("encnoact=") == 1 && p.outAct) {                           // timing ablation (wrong results): no act(BN(.)) second output for the next Winograd-form layer
        SrtConvParams q = p; q.outAct = nullptr; q.bnScale = q.bnShift = nullptr;
        static thread_local int depth = 0;
        if (!depth) { ++depth; const int rc = srt_launch_enc_wino(q, U, u_stem, s); --depth; return rc; }
    }
#endif
    const int Ho = p.H / 2, Wo = p.W / 2;
    if (Ho >= 4 && Wo >= 32) {
        const long units = (long)((Wo + 31) / 32) * ((Ho + 3) / 4) * p.ntiles, wgs = units * (p.Cout / 32) * p.nstems;
        const int tpw = wino_tpw(wgs, units);
#ifdef SRT_TUNING
        const dim3 grid((unsigned)(wgs / tpw));
        switch (wino_tune("encabl=")) {                                      // timing ablations (wrong results): 1 no barrier, 3 no U DMA, 4 no transform, 5 no A reads, 6 no epilogue, 7 / 8 / 9 epilogue without barriers / stores / exchange
        case 1: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 1>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 3: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 3>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 4: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 4>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 5: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 5>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 6: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 6>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 7: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 7>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 8: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 8>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 9: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 9>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        case 10: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 10>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0;
        }
        if (wino_tune("winopeel=") == 1) { SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 0, 1>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0; }
        if (wino_tune("winopr=") == 4) { SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 0, 0, 4>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit()); return 0; }
        switch (wino_tune("encrb=")) {                                       // patch-read forms (RB): 1 padded pitch + b64 pairs, 2 padded pitch only, 3 b64 pairs only
        case 1: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 0, 0, 3, 1>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit() | wino_mfast_bit(u_stem * 4, p.Cout / 32)); return 0;
        case 2: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 0, 0, 3, 2>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit() | wino_mfast_bit(u_stem * 4, p.Cout / 32)); return 0;
        case 3: SRT_LAUNCH((srt_enc_wino32<2, 16, 1, 0, 0, 3, 3>), grid, dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit() | wino_mfast_bit(u_stem * 4, p.Cout / 32)); return 0;
        }
#endif
        SRT_LAUNCH((srt_enc_wino32<2, 16, 1>), dim3((unsigned)(wgs / tpw)), dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit() | wino_mfast_bit(u_stem * 4, p.Cout / 32));
    } else if (Ho >= 4 && Wo >= 16) {
        const long units = (long)((Wo + 15) / 16) * ((Ho + 3) / 4) * ((p.ntiles + 1) / 2), wgs = units * (p.Cout / 32) * p.nstems;
        const int tpw = wino_tpw(wgs, units);
        SRT_LAUNCH((srt_enc_wino32<2, 8, 2>), dim3((unsigned)(wgs / tpw)), dim3(512), 0, s, p, U, u_stem, tpw | wino_walk_bit());
    } else return 1;
    return srt_launch_status();
}
