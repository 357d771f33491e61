// scripts/ubench/tile_reads.hip — what HBM delivers for up6's input access pattern, planar vs channel-last.
//
// up6 reads, per workgroup, a 10 x 66 pixel patch of 32 channels from planar [C][H][W] tensors: 320 row pieces of 264 bytes, 2 KiB
// (next row) and 256 KiB (next channel) apart.  This kernel issues exactly those loads (same lane -> address map as srt_up6_kernel:
// sub-tiles of 32 pixels, a lane loads its pixel of 16 channel pairs) and only sums them, for (a) the planar layout and (b) a
// channel-last layout [H][W][16] x 2 tensors (a pixel's 16 channels = 64 contiguous bytes, lane loads 2 x float4).
//   hipcc --offload-arch=gfx950 -O3 -o tile_reads tile_reads.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define H 128
#define W 512
#define C 32
#define TH 8
#define TW 64
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const float* __restrict__ x, float* out, int ninst)
{
    constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW, NSUB = (NPIX + 31) / 32;
    const int tilesX = W / TW, nsp = tilesX * (H / TH);
    // XCD order as the product kernels: consecutive positions on one XCD
    const int total = nsp * ninst, L = blockIdx.x, xcd = L & 7, j = L >> 3, q = total >> 3, r = total & 7;
    const int pos = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    const int sp = pos % nsp, inst = pos / nsp, tx0 = (sp % tilesX) * TW, ty0 = (sp / tilesX) * TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const size_t hw = (size_t)H * W;
    const float* xi = x + (size_t)inst * C * hw;
    float s = 0.f;
    for (int sub = wave; sub < NSUB; sub += 4) {
        const int pix = sub * 32 + l31, pr = pix / PW, pc = pix % PW;
        const int gy = ty0 + pr - 1, gx = tx0 + pc - 1;
        const bool ok = pix < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t off = ok ? (size_t)gy * W + gx : 0;
        if (MODE == 0) {
#pragma unroll
            for (int cp = 0; cp < C / 2; ++cp) { const float v = xi[(size_t)(2 * cp + half) * hw + off]; s += ok ? v : 0.f; }
        } else if (MODE == 2) {
        } else {
            // two [H][W][16] tensors; lane half h takes channels 8h..8h+7 of each (2 x float4 per tensor)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4* p4 = reinterpret_cast<const float4*>(xi + (size_t)t * 16 * hw + off * 16 + half * 8);
                const float4 a = p4[0], b = p4[1];
                s += ok ? (a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w) : 0.f;
            }
        }
    }
    if (MODE == 2) {
        // planar, but as aligned float4 row segments: element e = (channel, patch row, 4-pixel group), groups tx0-4 .. tx0+TW+3 (18 per row)
        constexpr int G = (TW + 8) / 4, NE = C * PH * G;
        for (int e = threadIdx.x; e < NE; e += 256) {
            const int g = e % G, r = (e / G) % PH, c = e / (G * PH);
            const int gy = ty0 + r - 1, gx = tx0 - 4 + 4 * g;
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
            const float4 v = *reinterpret_cast<const float4*>(xi + (size_t)c * hw + (ok ? (size_t)gy * W + gx : 0));
            s += ok ? v.x + v.y + v.z + v.w : 0.f;
        }
    }
    if (s == 123.456f) out[0] = s;
}
int main()
{
    const int ninst = 256;
    const size_t n = (size_t)ninst * C * H * W;
    float* x; float* out;
    hipMalloc(&x, n * 4); hipMemset(x, 0, n * 4); hipMalloc(&out, 64);
    const int grid = (W / TW) * (H / TH) * ninst;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, x, out, ninst);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, x, out, ninst);
            else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, x, out, ninst);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.3f ms for %.2f GB of tensor (%.2f TB/s algorithmic; patch bytes incl. halo %.2f GB)\n", mode == 2 ? "planar [C][H][W], aligned float4 row segments    " : mode ? "channel-last [H][W][16] x2, 2 x float4 per lane" : "planar [C][H][W], 16 dword loads per lane  ",
                   ms, n * 4 / 1e9, n * 4 / 1e9 / ms, (double)grid * 660 * C * 4 / 1e9);
        }
    }
    return 0;
}
