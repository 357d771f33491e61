// scripts/ubench/mfma_valu.hip — does ordinary VALU work hide under fp32 MFMAs on gfx950, and for whom?
//
// Measurement aid for DESIGN.md section 3.2a (the Winograd decoder's input transform is ~2 VALU instructions per MFMA).
//   mode 0  every wave: per MFMA (v_mfma_f32_16x16x4_f32, 16 independent accumulators) NV independent v_add_f32 / v_fma_f32
//   mode 1  waves 0-3 of a 512-thread workgroup MFMA only, waves 4-7 (their SIMD partners) the same number of VALU only
//   mode 2  as mode 0 with v_mfma_f32_32x32x16_bf16 (the guide's reference point: <= 5 fillers hide in its 32-cycle gap)
// Prints shader cycles (s_memtime) per MFMA for 1 and 2 waves per SIMD.      hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int MODE, int FMA>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE != 1 || wave < 4, do_valu = MODE != 1 || wave >= 4;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    const float c = out[0];
    float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f + 1.0f;
    long long t0, t1;
    if (MODE == 2) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 av, bv;
#pragma unroll
        for (int i = 0; i < 8; ++i) { av[i] = (__bf16)a; bv[i] = (__bf16)b; }
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m & 3], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(m * NV + v) & 7]) : "v"(c));
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(m * NV + v) & 7]) : "v"(c));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = __builtin_readcyclecounter();
        float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        x[0] += s;
    } else {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (do_mfma) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
                if (do_valu) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        if (FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(m * NV + v) & 7]) : "v"(c));
                        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(m * NV + v) & 7]) : "v"(c));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = __builtin_readcyclecounter();
        float s = 0; for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        x[0] += s;
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456f) out[1] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// G MFMAs back to back, then G*NV fillers in one burst.  KIND 0: v_add_f32, 1: ds_read_b128 (result unused), 2: LDS-DMA piece
// (global_load_lds_dwordx4 of 1 KiB from an L2-resident buffer), 3: v_mov_b32, 4: v_pk_add_f32 (two adds per instruction), 5: v_pk_fma_f32,
// 6: v_pk_add_f32 with op_sel / neg modifiers (the form a register-pair Winograd transform would use)
template <int G, int NV, int KIND>
__global__ void __launch_bounds__(512) kb(float* out, long long* cyc, const float* src, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float x[8];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 xp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; xp[i] = f32x2{x[i], x[i] + 0.5f}; }
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    const float c = out[0];
    const f32x2 cp = {c, c};
    float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f + 1.0f;
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + wv * 4096);
    const unsigned voff = lane * 16;
    const float* sp = src + wv * 1024;
    f32x4 sink = {0, 0, 0, 0};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m0 = 0; m0 < 16; m0 += G) {
#pragma unroll
            for (int m = 0; m < G; ++m) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m0 + m], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < (KIND == 2 ? NV : G * NV); ++v) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(c));
                else if (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(x[v & 7]) : "v"(c));
                else if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[v & 7]) : "v"(cp));
                else if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(xp[v & 7]) : "v"(cp));
                else if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[0,1]" : "+v"(xp[v & 7]) : "v"(cp));
                else if (KIND == 1) { f32x4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(lds0 + voff)); sink += r; }
                else asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sp), "s"(lds0) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i] + xp[i].x + xp[i].y;
    s += sink[0] + sink[1] + sink[2] + sink[3] + lds[threadIdx.x];
    if (s == 123.456f) out[1] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int G, int NV, int KIND>
static void runb(const char* tag, int threads)
{
    const int blocks = 256, iters = 1000;
    float* out; long long* cyc; float* src;
    hipMalloc(&out, 64); hipMemset(out, 0, 64);
    hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    hipMalloc(&cyc, blocks * 8 * sizeof(long long)); hipMemset(cyc, 0, blocks * 8 * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kb<G, NV, KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, src, 20);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kb<G, NV, KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double tot = 0; int n = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < nw; ++w) { tot += h[b * 8 + w]; ++n; }
    const double per = 16.0 * iters;
    printf("%-28s G=%2d fillers/MFMA=%d waves/SIMD=%d  cycles per MFMA per wave %.1f -> per SIMD %.1f   wall %.3f ms\n", tag, G, NV, nw / 4, tot / n / per, tot / n / per / (nw / 4), ms);
    hipFree(out); hipFree(cyc); hipFree(src);
}

template <int NV, int MODE, int FMA>
static void run(const char* tag, int threads)
{
    const int blocks = 256, iters = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, 64); hipMemset(out, 0, 64);
    hipMalloc(&cyc, blocks * 8 * sizeof(long long)); hipMemset(cyc, 0, blocks * 8 * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, MODE, FMA>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, MODE, FMA>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double lo = 0, hi = 0; int nlo = 0, nhi = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < nw; ++w) { if (w < 4) { lo += h[b * 8 + w]; ++nlo; } else { hi += h[b * 8 + w]; ++nhi; } }
    const double per = 16.0 * iters;
    printf("%-44s NV=%d waves/SIMD=%d  wall %.3f ms  cycles/MFMA-slot: waves0-3 %.1f", tag, NV, nw / 4, ms, lo / nlo / per);
    if (nhi) printf("  waves4-7 %.1f", hi / nhi / per);
    printf("   (wall ns per slot %.2f)\n", ms * 1e6 / per);
    hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv)
{
    if (argc > 1 && !strcmp(argv[1], "pk")) {                                   // packed fp32 VALU beside fp32 MFMA
#define PK(G, NV) runb<G, NV, 0>("burst v_add", 512); runb<G, NV, 4>("burst v_pk_add", 512); runb<G, NV, 5>("burst v_pk_fma", 512); runb<G, NV, 6>("burst v_pk_add op_sel/neg", 512);
        PK(1, 1) PK(4, 1) PK(8, 1) PK(8, 2) PK(16, 2)
        runb<8, 1, 4>("burst v_pk_add", 256); runb<8, 2, 4>("burst v_pk_add", 256); runb<8, 1, 0>("burst v_add", 256);
        return 0;
    }
#define ROW(NV) run<NV, 0, 0>("f32 16x16x4 + v_add per MFMA, all waves", 256); run<NV, 0, 0>("f32 16x16x4 + v_add per MFMA, all waves", 512);
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(6)
    run<2, 0, 1>("f32 16x16x4 + v_fma per MFMA, all waves", 256); run<2, 0, 1>("f32 16x16x4 + v_fma per MFMA, all waves", 512);
    run<4, 0, 1>("f32 16x16x4 + v_fma per MFMA, all waves", 256);
#define SPL(NV) run<NV, 1, 0>("f32: waves0-3 MFMA only | waves4-7 v_add only", 512);
    SPL(1) SPL(2) SPL(4) SPL(6) SPL(8)
#define BF(NV) run<NV, 2, 0>("bf16 32x32x16 + v_add per MFMA, all waves", 256); run<NV, 2, 0>("bf16 32x32x16 + v_add per MFMA, all waves", 512);
    BF(0) BF(2) BF(4) BF(6)
#define BURST(G, NV) runb<G, NV, 0>("burst v_add", 256); runb<G, NV, 0>("burst v_add", 512);
    BURST(1, 1) BURST(4, 1) BURST(8, 1) BURST(16, 1) BURST(1, 2) BURST(4, 2) BURST(8, 2) BURST(16, 2)
    runb<4, 2, 3>("burst v_mov", 512); runb<8, 1, 3>("burst v_mov", 512);
    runb<4, 1, 1>("ds_read_b128 1/MFMA", 256); runb<4, 1, 1>("ds_read_b128 1/MFMA", 512); runb<8, 1, 1>("ds_read_b128 1/MFMA", 512);
    runb<16, 0, 2>("no DMA (baseline)", 512);
    // LDS-DMA: NV pieces (1 KiB each, global_load_lds_dwordx4) per G MFMAs
    runb<16, 1, 2>("LDS-DMA 1 piece / 16 MFMA", 256); runb<16, 1, 2>("LDS-DMA 1 piece / 16 MFMA", 512);
    runb<8, 1, 2>("LDS-DMA 1 piece / 8 MFMA", 256); runb<8, 1, 2>("LDS-DMA 1 piece / 8 MFMA", 512);
    runb<4, 1, 2>("LDS-DMA 1 piece / 4 MFMA", 512); runb<8, 2, 2>("LDS-DMA 2 pieces / 8 MFMA", 512);
    return 0;
}
