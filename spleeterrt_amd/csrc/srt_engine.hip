// srt_engine.hip — engine object, HBM layout and the device-resident C ABI (include/spleeterrt_amd.h).
//
// HBM layout (all fp32, instance = (stem, tile), CHW planar, row = time, contiguous = frequency):
//   coeff[stem]           raw spleeterCoeff blob (Executable/spleeter.h:5-31), biases / BN read in place
//   wpack[stem][layer]    GEMM-ready weights [Cin][25][CP]
//   raw[i]  i=0..5        encoder conv+bias outputs   [stem][tile][Cout][H>>i+1][W>>i+1]: the skip tensors AND the next encoder
//                         layer's input (its BN + activation is applied by the consumer while staging; nothing is stored twice)
//   up[i]   i=0..5        decoder outputs (new channels only; the concat is done by pointer)
//   spec / mag / masks / frames / pcm for the DSP stages
#include "srt_internal.h"
#include "../../include/spleeterrt_amd.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <cxxabi.h>
#include <mutex>

static std::recursive_mutex g_setup_mutex;
SrtSetupLock::SrtSetupLock() { g_setup_mutex.lock(); }
SrtSetupLock::~SrtSetupLock() { g_setup_mutex.unlock(); }

static thread_local char g_err[512] = "";
static thread_local hipError_t g_hip_noted = hipSuccess;               // HIP error behind the last failed launch (srt_launch_status)
void srt_note_hip_error(hipError_t e) { g_hip_noted = e; }
int srt_set_error(int code, const char* fmt, const char* detail)      // shared with the drop-in layers (srt_compat.hip, srt_stream.hip)
{
    const int n = snprintf(g_err, sizeof g_err, fmt, detail);
    if (g_hip_noted != hipSuccess && n > 0 && (size_t)n < sizeof g_err - 8)
        snprintf(g_err + n, sizeof g_err - n, " [HIP: %s]", hipGetErrorString(g_hip_noted));
    g_hip_noted = hipSuccess;
    return code;
}
static int fail(int code, const char* fmt, const char* detail = "") { return srt_set_error(code, fmt, detail); }

// first kernel the calling thread launched since the last reset (SRT_LAUNCH, srt_internal.h)
static thread_local const void* g_kfn = nullptr;
static thread_local const char* g_ktext = nullptr;
void srt_kernel_note(const void* fn, const char* text) { if (!g_kfn) { g_kfn = fn; g_ktext = text; } }
void srt_kernel_note_reset() { g_kfn = nullptr; g_ktext = nullptr; }
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(-2, "HIP error: %s", hipGetErrorString(_e)); } while (0)

static const int ENC_CH[6][2] = { {2, 16}, {16, 32}, {32, 64}, {64, 128}, {128, 256}, {256, 512} };
static const int DEC_CH[6][2] = { {512, 256}, {512, 128}, {256, 64}, {128, 32}, {64, 16}, {32, 1} };

struct LayerOff { size_t w, b, bn; int cin, cout, cp; };
#define SRT_W16CS_U5 ((size_t)(64 / 16) * 15 * 2 * 32 * 8)    // halves per stem of up5's class-stacked fp16 weights
struct Layout { LayerOff down[6], up[6]; size_t head_w, head_b, total; };

static Layout make_layout()
{
    Layout lo; size_t o = 0;
    for (int i = 0; i < 6; ++i) {
        LayerOff& L = lo.down[i]; L.cin = ENC_CH[i][0]; L.cout = ENC_CH[i][1]; L.cp = (L.cout + 31) / 32 * 32;
        L.w = o; o += (size_t)25 * L.cin * L.cout; L.b = o; o += L.cout; L.bn = o; if (i < 5) o += 2 * (size_t)L.cout;
    }
    for (int i = 0; i < 6; ++i) {
        LayerOff& L = lo.up[i]; L.cin = DEC_CH[i][0]; L.cout = DEC_CH[i][1]; L.cp = (L.cout + 31) / 32 * 32;
        L.w = o; o += (size_t)25 * L.cin * L.cout; L.b = o; o += L.cout; L.bn = o; o += 2 * (size_t)L.cout;
    }
    lo.head_w = o; o += 32; lo.head_b = o; o += 2; lo.total = o;
    return lo;
}

struct TimingEntry { std::string name; hipEvent_t a, b; const void* kfn; const char* ktext; };
#define SRT_TIMING_MAX 65536        // launches recorded per srtSetTiming(1) window; later launches run untimed

// Persistent staging of srtSeparateHostStream / srtSeparateCliHost: device double buffers, copy streams and events are
// created on first use, grown when a call needs more, and freed with the engine (not allocated per call).
struct HostStaging {
    float* d_in[2]; float* d_out[2]; float* d_carry;
    size_t in_cap, out_cap, carry_cap;                 // floats
    hipStream_t s_in, s_out;
    hipEvent_t ev_in[2], ev_cmp[2], ev_out[2];
    bool ready;
};

// hipGraph replay of the launch sequences a low-latency caller repeats with the same arguments (srtSetGraphMode): the real-time
// plugin runs srtForward on the same two mask buffers for ever, the tile API on one pair of buffers.  A sequence is captured
// the first time its argument tuple is seen and replayed afterwards: one host call instead of ~25 launches on the audio thread.
struct GraphKey { int kind; const void* p0; const void* p1; void* p2; size_t n, frames, rows; int ntiles, s0, ns; };
struct GraphSlot { GraphKey key; hipGraph_t graph; hipGraphExec_t exec; unsigned long used; };
#define SRT_GRAPH_SLOTS 4

// Every entry point that allocates or launches runs on the device the engine was created on, whatever the caller's
// current device is (a host thread that switched devices after srtCreate must not mix device-A streams with device-B memory).
struct DeviceScope {
    int prev; bool sw;
    explicit DeviceScope(int dev) : prev(-1), sw(false) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) sw = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() { if (sw) hipSetDevice(prev); }
};

struct srt_engine {
    srt_config cfg;
    int device;
    hipStream_t stream;
    HostStaging hs;
    Layout lo;
    float* coeff_all;                                  // [n_stems][SRT_COEFF_STRIDE]
    float* wpack_down[6]; float* wpack_up[6];          // per layer: [n_stems][Cin*25*CP]
    size_t wpack_down_stem[6], wpack_up_stem[6];
    uint16_t* wpack16_down[6]; uint16_t* wpack16_up[6];  // fp16-MFMA packs [n_stems][Cin/16][25][2][CP][8] (precision != F32 only)
    size_t wpack16_down_stem[6], wpack16_up_stem[6];
    float* wpack2_d1;                                  // down1 stem-stacked [2][25][CP2], repacked per launch group (tiny)
    float* wpack2_u5;                                  // up5 class-stacked [n_stems][64][15][32]
    float* wino_u[6]; size_t wino_u_stem[6];           // Winograd-transformed decoder weights (srt_nn4.hip), layers in srt_wino_mask() only
    float* wino_e[6]; size_t wino_e_stem[6];           // the same for the encoder layers that can run in Winograd form (down3..down6)
    float* act32[6];                                   // fp32 act(BN(raw_i)) copies, i = 2..4: the inputs of those layers (written by their producers)
    bool   have_coeff[SRT_MAX_STEMS];
    float* raw[6]; float* up[6];
    size_t raw_tile[6], up_tile[6];                    // elements per instance
    float* act16buf[5];                                // fp16-storage mode only: act(bn(raw_i)) as halves, written by the producer
    bool act16;                                        // raw[0..5] and up[0..4] hold IEEE halves (precision F16 on a supported geometry)
    uint16_t* wpack16cs_u5;                            // act16 only: up5's class-stacked fp16 weights [n_stems][4][15][2][32][8] (srt_nn5.hip)
    SrtConvParams up6_params; int up6_s0; unsigned up6_stale;          // the last forward ran up6 + head in one pass (no up6 plane stored): srtCopyTensor("up6") re-launches up6 alone from these
    bool last_c8_l1;                                   // ... and conv1 / act1 too (down1 ran on its streamed kernels)
    bool masks16_req, last_masks16;                    // srtSeparate in the fp16 mode: the engine's OWN mask buffer may hold halves (asked for by separate_issue / what the last forward did)
    bool last_c8;                                      // the last forward stored raw2..6 / act2..5 / up1..4 channel-interleaved by eight (srt_nn5.hip): srtCopyTensor's view
    float* ws; size_t ws_floats;                       // split-K partial sums of small-batch launches (allocated on the first one)
    int graph_mode; unsigned long gclock; GraphSlot gslots[SRT_GRAPH_SLOTS];
    // DSP
    float *preWin, *postWin; float2* twiddle;
    float2* spec; float2* spec2; float* mag; float* masks; float* frames;   // spec2: residual spectrum of the CLI chain (on first use)
    size_t rows_cap, frames_rows;
    int last_ntiles;
    // timing
    bool timing; std::vector<TimingEntry> tlog;
};

const char* srtLastError(void) { return g_err; }
size_t srtCoeffBytes(void) { return (size_t)SRT_COEFF_FLOATS * 4; }
size_t srtStftRows(size_t n) { return (n + SRT_HOP - 1) / SRT_HOP; }
size_t srtStftFrames(size_t n) { return n < SRT_FFT ? 0 : (n - SRT_FFT + SRT_HOP / 4) / SRT_HOP + 1; }   // stftFix.c:378 + tail frame
size_t srtIstftLength(size_t rows) { return rows * SRT_HOP + (SRT_FFT - SRT_HOP); }

// element offset into an activation tensor whose elements are halves (act16) or floats
static inline float* eoff(const srt_engine* e, float* base, size_t elems) { return (float*)((char*)base + elems * (e->act16 ? 2 : 4)); }

struct TimerScope {
    srt_engine* e; size_t idx; bool on;
    TimerScope(srt_engine* e_, const char* name) : e(e_), idx(0), on(e_->timing && e_->tlog.size() < SRT_TIMING_MAX) {
        if (!on) return;
        TimingEntry t; t.name = name; t.kfn = nullptr; t.ktext = nullptr;
        srt_kernel_note_reset();
        if (hipEventCreate(&t.a) != hipSuccess) { on = false; return; }
        if (hipEventCreate(&t.b) != hipSuccess) { hipEventDestroy(t.a); on = false; return; }
        hipEventRecord(t.a, e->stream);
        e->tlog.push_back(t); idx = e->tlog.size() - 1;
    }
    ~TimerScope() { if (on) { hipEventRecord(e->tlog[idx].b, e->stream); e->tlog[idx].kfn = g_kfn; e->tlog[idx].ktext = g_ktext; } }
};

static void free_staging(srt_engine* e)
{
    HostStaging& h = e->hs;
    for (int b = 0; b < 2; ++b) {
        if (h.d_in[b]) hipFree(h.d_in[b]);
        if (h.d_out[b]) hipFree(h.d_out[b]);
        if (h.ev_in[b]) hipEventDestroy(h.ev_in[b]);
        if (h.ev_cmp[b]) hipEventDestroy(h.ev_cmp[b]);
        if (h.ev_out[b]) hipEventDestroy(h.ev_out[b]);
    }
    if (h.d_carry) hipFree(h.d_carry);
    if (h.s_in) hipStreamDestroy(h.s_in);
    if (h.s_out) hipStreamDestroy(h.s_out);
    memset(&h, 0, sizeof h);
}

static void free_graphs(srt_engine* e)
{
    for (GraphSlot& g : e->gslots) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
        memset(&g, 0, sizeof g);
    }
}

static void free_all(srt_engine* e)
{
    free_graphs(e);
    free_staging(e);
    if (e->coeff_all) hipFree(e->coeff_all);
    for (int i = 0; i < 6; ++i) { if (e->wpack16_down[i]) hipFree(e->wpack16_down[i]); if (e->wpack16_up[i]) hipFree(e->wpack16_up[i]); }
    if (e->ws) hipFree(e->ws);
    if (e->wpack2_d1) hipFree(e->wpack2_d1);
    if (e->wpack2_u5) hipFree(e->wpack2_u5);
    if (e->wpack16cs_u5) hipFree(e->wpack16cs_u5);
    for (int i = 0; i < 6; ++i) { if (e->wino_u[i]) hipFree(e->wino_u[i]); if (e->wino_e[i]) hipFree(e->wino_e[i]); if (e->act32[i]) hipFree(e->act32[i]); }
    for (int i = 0; i < 6; ++i) { if (e->wpack_down[i]) hipFree(e->wpack_down[i]); if (e->wpack_up[i]) hipFree(e->wpack_up[i]); }
    for (int i = 0; i < 6; ++i) { if (e->raw[i]) hipFree(e->raw[i]); if (e->up[i]) hipFree(e->up[i]); if (i < 5 && e->act16buf[i]) hipFree(e->act16buf[i]); }
    void* misc[] = { e->preWin, e->postWin, e->twiddle, e->spec, e->spec2, e->mag, e->masks, e->frames };
    for (void* m : misc) if (m) hipFree(m);
    for (auto& t : e->tlog) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
}

int srtCreate(const srt_config* cfg, void* stream, srt_engine** out)
{
    if (!cfg || !out) return fail(-1, "srtCreate: null argument");
    SrtSetupLock setup;
    if (cfg->F < 64 || cfg->F > 2048 || cfg->F % 64 || cfg->T < 64 || cfg->T % 64)
        return fail(-1, "srtCreate: F and T must be multiples of 64 (F <= 2048)");     // spleeter.c:113-119 floor-divides by 64
    if (cfg->n_stems < 1 || cfg->n_stems > SRT_MAX_STEMS || cfg->max_tiles < 1) return fail(-1, "srtCreate: bad n_stems / max_tiles");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(-3, "srtCreate: no HIP device (this library has no CPU path)");
    srt_engine* e = new srt_engine();
    memset(&e->hs, 0, sizeof e->hs);
    e->ws = nullptr; e->ws_floats = 0;
    e->graph_mode = 0; e->gclock = 0; memset(e->gslots, 0, sizeof e->gslots);
    if (hipGetDevice(&e->device) != hipSuccess) { delete e; return fail(-3, "srtCreate: no current HIP device"); }
    memset(e->wpack16_down, 0, sizeof e->wpack16_down); memset(e->wpack16_up, 0, sizeof e->wpack16_up);
    memset(e->wino_u, 0, sizeof e->wino_u); memset(e->wino_u_stem, 0, sizeof e->wino_u_stem);
    memset(e->wino_e, 0, sizeof e->wino_e); memset(e->wino_e_stem, 0, sizeof e->wino_e_stem); memset(e->act32, 0, sizeof e->act32);
    e->wpack16cs_u5 = nullptr; e->last_c8 = false; e->last_c8_l1 = false; e->masks16_req = e->last_masks16 = false; e->up6_stale = 0; e->up6_s0 = 0;
    e->coeff_all = nullptr; e->wpack2_d1 = e->wpack2_u5 = nullptr; memset(e->wpack_down, 0, sizeof e->wpack_down); memset(e->wpack_up, 0, sizeof e->wpack_up);
    memset(e->have_coeff, 0, sizeof e->have_coeff);
    memset(e->raw, 0, sizeof e->raw); memset(e->up, 0, sizeof e->up); memset(e->act16buf, 0, sizeof e->act16buf);
    e->preWin = e->postWin = nullptr; e->twiddle = nullptr; e->spec = nullptr; e->spec2 = nullptr; e->mag = e->masks = e->frames = nullptr;
    e->cfg = *cfg;
    { const char* bi = getenv("SPLEETERRT_BATCH_INVARIANT"); if (bi && bi[0] == '1') e->cfg.batch_invariant = 1; }
    e->stream = (hipStream_t)stream; e->lo = make_layout(); e->timing = false; e->last_ntiles = cfg->max_tiles;
    if (e->lo.total != SRT_COEFF_FLOATS) { delete e; return fail(-4, "internal: weight layout size mismatch"); }
    const size_t S = cfg->n_stems, NT = cfg->max_tiles, HW = (size_t)cfg->T * cfg->F;
#define EALLOC(ptr, nfloats) do { if (hipMalloc((void**)&(ptr), (nfloats) * sizeof(float)) != hipSuccess) { free_all(e); delete e; return fail(-2, "srtCreate: hipMalloc failed"); } } while (0)
    EALLOC(e->coeff_all, S * SRT_COEFF_STRIDE);
    if (cfg->precision != SRT_PREC_F32) {
        for (int i = 0; i < 6; ++i) {
            const LayerOff& D = e->lo.down[i]; const LayerOff& U = e->lo.up[i];
            e->wpack16_down_stem[i] = D.cin % 16 ? 0 : (size_t)(D.cin / 16) * 400 * D.cp;      // halves
            e->wpack16_up_stem[i] = U.cin % 16 ? 0 : (size_t)(U.cin / 16) * 400 * U.cp;
            if (e->wpack16_down_stem[i] && hipMalloc((void**)&e->wpack16_down[i], S * e->wpack16_down_stem[i] * 2) != hipSuccess) { free_all(e); delete e; return fail(-2, "srtCreate: hipMalloc failed"); }
            if (e->wpack16_up_stem[i] && hipMalloc((void**)&e->wpack16_up[i], S * e->wpack16_up_stem[i] * 2) != hipSuccess) { free_all(e); delete e; return fail(-2, "srtCreate: hipMalloc failed"); }
        }
    }
    EALLOC(e->wpack2_d1, (size_t)2 * 25 * 128);
    EALLOC(e->wpack2_u5, S * 64 * 15 * 32);
    // The Winograd-form layers only ever run on launches above 16 instances (forward_range: `few`) or under batch_invariant: an engine that can never
    // hold that many (every drop-in tile-API / plugin instance: 1 tile x 1-4 stems) does not allocate their transformed weights and input copies.
    const bool wino_possible = S * NT > 16 || e->cfg.batch_invariant || srt_wino_force();
    for (int i = 0; i < 6; ++i) {
        e->wpack_down_stem[i] = (size_t)e->lo.down[i].cin * 25 * e->lo.down[i].cp;
        e->wpack_up_stem[i] = (size_t)e->lo.up[i].cin * 25 * e->lo.up[i].cp;
        EALLOC(e->wpack_down[i], S * e->wpack_down_stem[i]);
        EALLOC(e->wpack_up[i], S * e->wpack_up_stem[i]);
        // Winograd form of the decoder layers named by srt_wino_mask() (fp32 MFMA path): [Cin/4][Cout/16][4][16][52] per stem
        const LayerOff& U = e->lo.up[i];
        if (wino_possible && cfg->impl == SRT_IMPL_MFMA && cfg->precision == SRT_PREC_F32 && ((srt_wino_mask() >> i) & 1) && U.cout % 16 == 0 && U.cin % 4 == 0) {
            e->wino_u_stem[i] = (size_t)U.cin * U.cout * 52;
            EALLOC(e->wino_u[i], S * e->wino_u_stem[i]);
        }
        const LayerOff& Dn = e->lo.down[i];
        if (wino_possible && cfg->impl == SRT_IMPL_MFMA && cfg->precision == SRT_PREC_F32 && i >= 1 && srt_enc_wino_covers(Dn.cin, Dn.cout, cfg->T >> i, cfg->F >> i)) {
            e->wino_e_stem[i] = (size_t)Dn.cin * Dn.cout * 52;
            EALLOC(e->wino_e[i], S * e->wino_e_stem[i]);
            EALLOC(e->act32[i - 1], S * NT * ((size_t)ENC_CH[i - 1][1] * (HW >> (2 * i))));        // act(BN(raw_{i-1})): this layer's input
        }
    }
    // fp16 activation storage: every layer between down1 and up6 must run on the fp16-MFMA kernels, which stage aligned
    // 4-pixel row segments at every level (up1's input is F/64 wide): F % 256 == 0.  Other geometries keep fp32 tensors.
    e->act16 = cfg->precision == SRT_PREC_F16 && cfg->impl == SRT_IMPL_MFMA && cfg->F % 256 == 0;
    if (e->act16 && hipMalloc((void**)&e->wpack16cs_u5, S * SRT_W16CS_U5 * 2) != hipSuccess) { free_all(e); delete e; return fail(-2, "srtCreate: hipMalloc failed"); }
    for (int i = 0; i < 6; ++i) {
        e->raw_tile[i] = (size_t)ENC_CH[i][1] * (HW >> (2 * (i + 1)));
        EALLOC(e->raw[i], (S * NT * e->raw_tile[i] + (e->act16 ? 1 : 0)) / (e->act16 ? 2 : 1));
        if (e->act16 && i < 5) EALLOC(e->act16buf[i], (S * NT * e->raw_tile[i] + 1) / 2);
        e->up_tile[i] = (size_t)DEC_CH[i][1] * (HW >> (2 * (5 - i)));
        EALLOC(e->up[i], (S * NT * e->up_tile[i] + (e->act16 && i < 5 ? 1 : 0)) / (e->act16 && i < 5 ? 2 : 1));      // up6's output (the head's input) stays fp32
    }
    e->rows_cap = NT * cfg->T;
    EALLOC(e->preWin, SRT_FFT); EALLOC(e->postWin, SRT_FFT); EALLOC(e->twiddle, 2 * SRT_FFT);
    EALLOC(e->spec, 2 * 2 * e->rows_cap * SRT_SPEC_LD);
    EALLOC(e->mag, NT * 2 * HW);
    EALLOC(e->masks, S * NT * 2 * HW);
    e->frames_rows = 0;                                 // windowed-frame scratch is allocated on first use (ensure_frames)
#undef EALLOC
    // tables: same formulas and float rounding as InitSTFT (stftFix.c:302-313)
    std::vector<float> pre(SRT_FFT), post(SRT_FFT), tw(2 * SRT_FFT), sig(1026);
    const double w0 = 6.283185307179586476925286766559 / SRT_FFT;
    const float postScale = (float)SRT_FFT * ((1.0f / 2.0f) / (3.0f / 8.0f));
    for (int i = 0; i < SRT_FFT; ++i) {
        const float hs = (float)((1.0 / SRT_FFT) * (0.5 * (1.0 - cos(w0 * (i + 0.5)))));
        pre[i] = hs * (2.0f / 4.0f);
        post[i] = hs * postScale * 0.5f;
        tw[2 * i] = (float)cos(w0 * i); tw[2 * i + 1] = (float)(-sin(w0 * i));
    }
    // logistic LUT of the Executable flavour (spleeter.c:29): 1025 samples on [-7,7] + trailing 1.0, regenerated in closed form
    for (int i = 0; i < 1025; ++i) sig[i] = (float)(1.0 / (1.0 + exp(7.0 - 0.013671875 * i)));
    sig[1025] = 1.0f;
    if (hipMemcpy(e->preWin, pre.data(), SRT_FFT * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->postWin, post.data(), SRT_FFT * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->twiddle, tw.data(), 2 * SRT_FFT * 4, hipMemcpyHostToDevice) != hipSuccess ||
        srt_set_sigmoid_table(sig.data()) != 0) { free_all(e); delete e; return fail(-2, "srtCreate: table upload failed"); }
    *out = e;
    return 0;
}

void srtDestroy(srt_engine* e)
{
    if (!e) return;
    DeviceScope ds(e->device);
    hipStreamSynchronize(e->stream);
    free_all(e);
    delete e;
}

// SRT_PREC_F16X2 promises fp32-level parity, which holds only while every weight that goes through the fp16 pack is an fp16 value: refuse a blob that
// is not (the caller picks SRT_PREC_F32, or SRT_PREC_F16 with its 2e-2 tolerance class, knowingly) instead of silently rounding it.
static int check_f16x2_weights(srt_engine* e, int stem)
{
    unsigned* d_count = nullptr; unsigned bad = 0; size_t total = 0;
    HIPCHK(hipMalloc((void**)&d_count, sizeof(unsigned)));
    hipError_t er = hipMemsetAsync(d_count, 0, sizeof(unsigned), e->stream);
    const float* c = e->coeff_all + (size_t)stem * SRT_COEFF_STRIDE;
    for (int i = 0; i < 6 && er == hipSuccess; ++i) {
        const LayerOff& D = e->lo.down[i]; const LayerOff& U = e->lo.up[i];
        if (e->wpack16_down[i]) { total += (size_t)25 * D.cin * D.cout; if (srt_launch_count_not_fp16(c + D.w, (size_t)25 * D.cin * D.cout, d_count, e->stream)) er = hipErrorLaunchFailure; }
        if (e->wpack16_up[i]) { total += (size_t)25 * U.cin * U.cout; if (srt_launch_count_not_fp16(c + U.w, (size_t)25 * U.cin * U.cout, d_count, e->stream)) er = hipErrorLaunchFailure; }
    }
    if (er == hipSuccess) er = hipMemcpyAsync(&bad, d_count, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(e->stream);
    hipFree(d_count);
    HIPCHK(er);
    if (bad) {
        e->have_coeff[stem] = false;
        char what[96];
        snprintf(what, sizeof what, "sub-network %d: %u of %zu are not and would be rounded", stem, bad, total);
        return fail(-5, "srtSetCoeff: SRT_PREC_F16X2 needs fp16-representable conv weights (%s); use SRT_PREC_F32, or SRT_PREC_F16 for the fp16 tolerance class", what);
    }
    return 0;
}

static int pack_stem(srt_engine* e, int stem)
{
    if (e->cfg.precision == SRT_PREC_F16X2 && e->cfg.impl == SRT_IMPL_MFMA) { const int rc = check_f16x2_weights(e, stem); if (rc) return rc; }
    for (int i = 0; i < 6; ++i) {
        const LayerOff& D = e->lo.down[i]; const LayerOff& U = e->lo.up[i];
        const float* c = e->coeff_all + (size_t)stem * SRT_COEFF_STRIDE;
        if (srt_launch_pack_enc(c + D.w, e->wpack_down[i] + stem * e->wpack_down_stem[i], D.cin, D.cout, D.cp, e->stream)) return fail(-2, "pack launch failed");
        if (srt_launch_pack_dec(c + U.w, e->wpack_up[i] + stem * e->wpack_up_stem[i], U.cin, U.cout, U.cp, e->stream)) return fail(-2, "pack launch failed");
        if (e->wpack16_down[i] && srt_launch_pack16(c + D.w, e->wpack16_down[i] + stem * e->wpack16_down_stem[i], D.cin, D.cout, D.cp, 0, e->stream)) return fail(-2, "pack launch failed");
        if (e->wpack16_up[i] && srt_launch_pack16(c + U.w, e->wpack16_up[i] + stem * e->wpack16_up_stem[i], U.cin, U.cout, U.cp, 1, e->stream)) return fail(-2, "pack launch failed");
        if (e->wino_u[i] && srt_launch_pack_wino(c + U.w, e->wino_u[i] + stem * e->wino_u_stem[i], U.cin, U.cout, e->stream)) return fail(-2, "pack launch failed");
        if (e->wino_e[i] && srt_launch_pack_wino_enc(c + D.w, e->wino_e[i] + stem * e->wino_e_stem[i], D.cin, D.cout, e->stream)) return fail(-2, "pack launch failed");
    }
    if (srt_launch_pack_classstack(e->coeff_all + (size_t)stem * SRT_COEFF_STRIDE + e->lo.up[4].w, e->wpack2_u5 + (size_t)stem * 64 * 15 * 32, 64, 16, e->stream))
        return fail(-2, "pack launch failed");
    if (e->wpack16cs_u5 && srt_launch_pack16_classstack(e->coeff_all + (size_t)stem * SRT_COEFF_STRIDE + e->lo.up[4].w, e->wpack16cs_u5 + (size_t)stem * SRT_W16CS_U5, 64, 16, e->stream))
        return fail(-2, "pack launch failed");
    e->have_coeff[stem] = true;
    return 0;
}

int srtSetCoeffHost(srt_engine* e, int stem, const void* h)
{
    if (!e || !h || stem < 0 || stem >= e->cfg.n_stems) return fail(-1, "srtSetCoeffHost: bad argument");
    SrtSetupLock setup;
    DeviceScope ds(e->device);
    HIPCHK(hipMemcpyAsync(e->coeff_all + (size_t)stem * SRT_COEFF_STRIDE, h, srtCoeffBytes(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return pack_stem(e, stem);
}
int srtSetCoeffDevice(srt_engine* e, int stem, const void* d)
{
    if (!e || !d || stem < 0 || stem >= e->cfg.n_stems) return fail(-1, "srtSetCoeffDevice: bad argument");
    SrtSetupLock setup;
    DeviceScope ds(e->device);
    HIPCHK(hipMemcpyAsync(e->coeff_all + (size_t)stem * SRT_COEFF_STRIDE, d, srtCoeffBytes(), hipMemcpyDeviceToDevice, e->stream));
    return pack_stem(e, stem);
}
// read the fp32 blob of a sub-network back (what srtSetCoeff* stored: for the fp16 container, the expanded values)
int srtGetCoeffHost(srt_engine* e, int stem, void* h)
{
    if (!e || !h || stem < 0 || stem >= e->cfg.n_stems) return fail(-1, "srtGetCoeffHost: bad argument");
    SrtSetupLock setup;
    DeviceScope ds(e->device);
    HIPCHK(hipMemcpyAsync(h, e->coeff_all + (size_t)stem * SRT_COEFF_STRIDE, srtCoeffBytes(), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
int srtSetCoeffFp16Host(srt_engine* e, int stem, const uint16_t* h)
{
    if (!e || !h || stem < 0 || stem >= e->cfg.n_stems) return fail(-1, "srtSetCoeffFp16Host: bad argument");
    SrtSetupLock setup;
    DeviceScope ds(e->device);
    uint16_t* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)SRT_COEFF_FLOATS * 2));
    hipError_t er = hipMemcpyAsync(d, h, (size_t)SRT_COEFF_FLOATS * 2, hipMemcpyHostToDevice, e->stream);
    if (er == hipSuccess) { srt_fp16_expand(d, e->coeff_all + (size_t)stem * SRT_COEFF_STRIDE, SRT_COEFF_FLOATS, e->stream); er = hipStreamSynchronize(e->stream); }
    hipFree(d);
    HIPCHK(er);
    return pack_stem(e, stem);
}


// Small batches (the real-time plugin: 1 tile x 4 stems; BASELINE configs[1]: 1 x 2) leave most CUs idle in the deep layers:
// the launchers get a workspace so they can cut those layers' K loops into slices (srt_nn2.hip, split-K).
static void ensure_ws(srt_engine* e, size_t instances)
{
    if (e->ws || instances > 16 || e->cfg.batch_invariant || e->cfg.impl != SRT_IMPL_MFMA || e->cfg.precision != SRT_PREC_F32) return;
    const size_t want = (size_t)16 << 20;                  // 64 MiB: 8 slices of the largest split layer at 8 instances
    if (hipMalloc((void**)&e->ws, want * sizeof(float)) == hipSuccess) e->ws_floats = want;
    else { e->ws = nullptr; (void)hipGetLastError(); }
}

// Run `issue` (a function that only enqueues work on e->stream) through the graph cache when graph mode is on.  `valid` says
// whether the arguments passed the entry point's checks: an invalid call is issued eagerly (it fails with its own error code and
// nothing is captured).  Graph mode is switched off for good only when the capture / instantiate API itself fails - a user error
// such as missing weights leaves it on.  If the caller's stream is already capturing (the caller builds its own graph), the
// launches simply join that capture.
template <class F>
static int run_graphed(srt_engine* e, const GraphKey& key, bool valid, F&& issue)
{
    if (!e->graph_mode || e->timing || !e->stream || !valid) return issue();      // the legacy null stream cannot be captured
    for (GraphSlot& g : e->gslots)
        if (g.exec && !memcmp(&g.key, &key, sizeof key)) {
            g.used = ++e->gclock;
            HIPCHK(hipGraphLaunch(g.exec, e->stream));
            return 0;
        }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(e->stream, &cap) != hipSuccess) { (void)hipGetLastError(); return issue(); }
    if (cap != hipStreamCaptureStatusNone) return issue();                 // inside the caller's capture: do not nest one
    if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); e->graph_mode = 0; return issue(); }
    const int rc = issue();
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipError_t er = hipStreamEndCapture(e->stream, &graph);
    if (rc) {
        // A launch failed INSIDE the capture.  The arguments were validated above, so the likeliest cause is the capture itself having been
        // invalidated from outside (another host thread's legacy-stream operation while this stream was capturing): run the sequence once
        // more, eagerly - if it fails again the error is real and is returned; graph mode stays on either way.
        if (graph) hipGraphDestroy(graph);
        (void)hipGetLastError();
        return issue();
    }
    if (er == hipSuccess && graph) er = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (er != hipSuccess || !exec) {                                       // capture is not available here: plain launches from now on
        if (graph) hipGraphDestroy(graph);
        (void)hipGetLastError();
        e->graph_mode = 0;
        return issue();
    }
    GraphSlot* v = &e->gslots[0];
    for (GraphSlot& g : e->gslots) if (!g.exec || g.used < v->used) { v = &g; if (!g.exec) break; }
    if (v->exec) hipGraphExecDestroy(v->exec);
    if (v->graph) hipGraphDestroy(v->graph);
    v->key = key; v->graph = graph; v->exec = exec; v->used = ++e->gclock;
    HIPCHK(hipGraphLaunch(exec, e->stream));
    return 0;
}

// Sub-networks [s0, s0+ns) on ntiles tiles.  Buffers keep their all-stem layout (stem stride = ntiles instances), so a
// later call for other stems of the same batch lands beside this one's results.
static int forward_range(srt_engine* e, const float* d_mag, int ntiles, float* d_masks, int s0, int ns)
{
    if (!e || !d_mag || !d_masks) return fail(-1, "srtForward: null argument");
    DeviceScope ds(e->device);
    if (ntiles < 1 || ntiles > e->cfg.max_tiles) return fail(-1, "srtForward: ntiles exceeds max_tiles");
    const int S = e->cfg.n_stems, T = e->cfg.T, F = e->cfg.F;
    if (s0 < 0 || ns < 1 || s0 + ns > S) return fail(-1, "srtForward: stem range outside the engine's sub-networks");
    for (int s = s0; s < s0 + ns; ++s) if (!e->have_coeff[s]) return fail(-5, "srtForward: weights not set for every stem");
    const size_t HW = (size_t)T * F;
    e->last_ntiles = ntiles;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (e->stream) (void)hipStreamIsCapturing(e->stream, &cap);
    if (cap == hipStreamCaptureStatusNone) ensure_ws(e, (size_t)ns * ntiles);      // (no allocation inside a capture: callers pre-allocate)
    // Kernel choice by launch size: at most 16 instances keep the direct kernels and (when the workspace exists) cut the deep layers' K loops
    // into slices; larger launches run up1..up5 and down3..down6 in Winograd form.  batch_invariant: never split-K (ws stays null) and the
    // Winograd-form layers whenever the layer geometry fits, whatever the batch - the same bits for a tile in any batch.
    const bool few = (size_t)ns * ntiles <= 16 && !e->cfg.batch_invariant;
    const bool small = few && e->ws;
    // fp16 storage, launches above 16 instances: the tensors between down2 and up5 are kept channel-interleaved by eight and the layers run on the DMA-fed
    // kernels of srt_nn5.hip (SPLEETERRT_C8=0: the planar kernels of srt_nn3.hip everywhere, for A/B runs)
    const char* c8v = getenv("SPLEETERRT_C8");                                     // (read per forward: parity tests compare the two layouts inside one process)
    const bool c8_env = !(c8v && c8v[0] == '0');
    const bool c8 = e->act16 && !few && c8_env;
    // ... and down1's two outputs (raw1: up6's skip input, act1: down2's input) where down1 runs on its streamed kernels (SPLEETERRT_C8L1=0: planar, for A/B runs)
    const char* c8l1v = getenv("SPLEETERRT_C8L1");
    const bool c8_l1 = c8 && !(c8l1v && c8l1v[0] == '0') && e->cfg.impl == SRT_IMPL_MFMA && srt_down1_c8_ok(T, F, ntiles, (size_t)ntiles * e->raw_tile[0]);
    e->last_c8 = c8; e->last_c8_l1 = c8_l1; e->last_masks16 = false;
    {
        // all stems go in one launch per layer; the activation pair is per stem (spleeter.c:130-139) and travels as a bit mask
        unsigned elu_mask = 0;
        for (int s = 0; s < ns; ++s) if (e->cfg.stem_mode[s0 + s]) elu_mask |= 1u << s;
        const int actE = SRT_ACT_LEAKY, actD = SRT_ACT_RELU;
        const float* cbase = e->coeff_all + (size_t)s0 * SRT_COEFF_STRIDE;
        char nm[32];
        bool act_ready = false;                                                 // act32[i-1] holds act(BN(raw_{i-1})) of this batch
        for (int i = 0; i < 6; ++i) {                                           // encoder (spleeter.c:182-238)
            const LayerOff& L = e->lo.down[i];
            SrtConvParams p; memset(&p, 0, sizeof p);
            p.Cin = L.cin; p.Cout = L.cout; p.H = T >> i; p.W = F >> i; p.CA = L.cin; p.ntiles = ntiles; p.nstems = ns;
            if (i == 0) { p.srcA = d_mag; p.srcA_stem = 0; p.srcA_tile = 2 * HW; }                 // every stem reads the same magnitudes
            else {
                // the previous layer's conv + bias; its batch-norm + activation (spleeter.c:188) is applied by this layer while staging
                const LayerOff& P = e->lo.down[i - 1];
                p.srcA_stem = (size_t)ntiles * e->raw_tile[i - 1]; p.srcA_tile = e->raw_tile[i - 1];
                if (e->act16) {            // fp16 storage: the producer wrote act(bn(raw)) as a second fp16 tensor (see srt_enc_f16)
                    p.srcA = eoff(e, e->act16buf[i - 1], (size_t)s0 * ntiles * e->raw_tile[i - 1]);
                    p.in16 = 1;
                } else {
                    p.srcA = e->raw[i - 1] + (size_t)s0 * ntiles * e->raw_tile[i - 1];
                    p.inShift = cbase + P.bn; p.inScale = cbase + P.bn + P.cout;
                }
            }
            p.srcB = p.srcA;
            p.wraw = cbase + L.w; p.bias = cbase + L.b;
            p.coeff_stem = SRT_COEFF_STRIDE;
            p.wpack = e->wpack_down[i] + (size_t)s0 * e->wpack_down_stem[i]; p.wpack_stem = e->wpack_down_stem[i];
            p.CP = L.cp;
            p.outRaw = eoff(e, e->raw[i], (size_t)s0 * ntiles * e->raw_tile[i]);
            p.out16 = e->act16;
            if (e->act16 && i < 5) {
                p.outAct = eoff(e, e->act16buf[i], (size_t)s0 * ntiles * e->raw_tile[i]);
                p.bnShift = cbase + L.bn; p.bnScale = cbase + L.bn + L.cout;
            }
            p.out_stem = (size_t)ntiles * e->raw_tile[i]; p.out_tile = e->raw_tile[i];
            p.act = actE; p.elu_mask = elu_mask; p.variant = e->cfg.variant;
            if (small) { p.ws = e->ws; p.ws_floats = e->ws_floats; }
            if (i == 0) p.c8out = c8_l1;
            // fp16 mode, C8 outputs: down1's products on the fp16 MFMA too, every stem of the call (up to six) in ONE launch (srt_down1_f16_kernel;
            // SPLEETERRT_D1F16=0: the fp32-MFMA streamed kernels below, read per forward for A/B runs and the parity test)
            if (i == 0 && c8_l1 && e->cfg.precision == SRT_PREC_F16 && p.nstems <= 6 && !small) {
                const char* dv = getenv("SPLEETERRT_D1F16");
                if (!(dv && dv[0] == '0')) {
                    p.stack = p.nstems;
                    TimerScope tg(e, "down1");
                    const int rg = srt_launch_down1_f16(p, e->stream);
                    if (rg == 0) continue;
                    if (rg != 1) return fail(-2, "encoder launch failed");
                    p.stack = 0;
                }
            }
            if (i == 0 && e->cfg.impl == SRT_IMPL_MFMA) {                    // stem-stacked M: all stems of the group share the input
                // more than four sub-networks (BASELINE configs[4]: five): the streamed down1 kernel stacks at most 4 x 16 rows, so the first whole groups of
                // four go out here, each as its own stacked launch; the remainder (1..4 stems) follows the common path below
                while (p.nstems > 4) {
                    SrtConvParams q = p;
                    q.nstems = 4; q.stack = 4; q.CP2 = 64; q.wpack2 = e->wpack2_d1; q.wpack2_stem = 0;
                    if (srt_launch_pack_stemstack(q.wraw, SRT_COEFF_STRIDE, 4, e->wpack2_d1, L.cin, L.cout, q.CP2, e->stream)) return fail(-2, "pack launch failed");
                    { TimerScope tg(e, "down1"); const int rg = srt_launch_enc2(q, e->stream);
                      if (rg == 1) return fail(-4, "internal: stem-stacked down1 group not covered by its launcher");
                      if (rg) return fail(-2, "encoder launch failed"); }
                    p.nstems -= 4; p.elu_mask >>= 4;
                    p.wpack += 4 * p.wpack_stem;                                 // (the per-stem packs advance with the group too - the remainder launch's fallbacks read them)
                    p.wraw += 4 * (size_t)SRT_COEFF_STRIDE; p.bias += 4 * (size_t)SRT_COEFF_STRIDE;
                    if (p.bnShift) { p.bnShift += 4 * (size_t)SRT_COEFF_STRIDE; p.bnScale += 4 * (size_t)SRT_COEFF_STRIDE; }
                    p.outRaw = eoff(e, p.outRaw, 4 * p.out_stem);
                    if (p.outAct) p.outAct = eoff(e, p.outAct, 4 * p.out_stem);
                }
                p.stack = p.nstems; p.CP2 = (p.nstems * 16 + 63) / 64 * 64; p.wpack2 = e->wpack2_d1; p.wpack2_stem = 0;
                if (srt_launch_pack_stemstack(p.wraw, SRT_COEFF_STRIDE, p.nstems, e->wpack2_d1, L.cin, L.cout, p.CP2, e->stream)) return fail(-2, "pack launch failed");
            }
            // Winograd form (down3..down6 of launches above 16 instances): reads the act(BN(raw)) copy of its input, writes raw + its own copy when
            // the next layer runs here too.  The first such layer's input copy is written by the direct layer in front (below); only when that layer
            // ran on a kernel without the second output does a batched bn+act pass over its raw tensor make the copy here.
            const bool ewino = !few && e->wino_e[i] && !e->act16;
            if (ewino) {
                if (!act_ready) {
                    const LayerOff& P = e->lo.down[i - 1];
                    TimerScope ta(e, "actcopy");
                    if (srt_launch_bn_act_batch(e->raw[i - 1] + (size_t)s0 * ntiles * e->raw_tile[i - 1], e->act32[i - 1] + (size_t)s0 * ntiles * e->raw_tile[i - 1],
                                                cbase + P.bn + P.cout, cbase + P.bn, SRT_COEFF_STRIDE, ns, ntiles, P.cout, e->raw_tile[i - 1] / P.cout,
                                                actE, elu_mask, e->cfg.variant, e->stream)) return fail(-2, "bn-act launch failed");
                }
                p.srcA = e->act32[i - 1] + (size_t)s0 * ntiles * e->raw_tile[i - 1]; p.srcB = p.srcA;
                p.inScale = p.inShift = nullptr;
                act_ready = i + 1 < 6 && e->wino_e[i + 1] != nullptr;       // this layer writes the next one's input copy
                if (act_ready) {
                    p.outAct = e->act32[i] + (size_t)s0 * ntiles * e->raw_tile[i];
                    p.bnShift = cbase + L.bn; p.bnScale = cbase + L.bn + L.cout;
                }
            } else act_ready = false;
            // a direct layer in front of the first Winograd-form one writes that copy itself (srt_enc_mfma2, second fp32 output): one more store per
            // element instead of a pass that reads the tensor back.  Only the plain MFMA kernel does that (not the split-K one of small batches).
            const bool next_wino = !few && i > 0 && i + 1 < 6 && e->wino_e[i + 1] && e->act32[i] && !e->act16;
            bool producer_copy = false;
            if (!ewino && next_wino && e->cfg.impl == SRT_IMPL_MFMA && !small && srt_enc_producer_copy()) {
                p.outAct = e->act32[i] + (size_t)s0 * ntiles * e->raw_tile[i];
                p.bnShift = cbase + L.bn; p.bnScale = cbase + L.bn + L.cout;
                producer_copy = true;
            }
            snprintf(nm, sizeof nm, "down%d", i + 1);
            TimerScope ts(e, nm);
            int rc2 = 1;
            if (ewino) {
                rc2 = srt_launch_enc_wino(p, e->wino_e[i] + (size_t)s0 * e->wino_e_stem[i], e->wino_e_stem[i], e->stream);
                if (rc2 == 1) return fail(-4, "internal: Winograd encoder layer not covered by its launcher");
            }
            if (rc2 == 1 && e->cfg.impl == SRT_IMPL_MFMA && e->wpack16_down[i]) {
                p.wpack16 = e->wpack16_down[i] + (size_t)s0 * e->wpack16_down_stem[i]; p.wpack16_stem = e->wpack16_down_stem[i];
                p.nsplit = e->cfg.precision == SRT_PREC_F16X2 ? 2 : 1;
                if (c8 && (i >= 2 || c8_l1)) {                                  // C8 in, C8 out (srt_nn5.hip); down2 when down1 wrote its act copy C8
                    rc2 = srt_launch_enc_c8(p, e->stream);
                    if (rc2 == 1) return fail(-4, "internal: C8 activation layout but no C8 kernel for an encoder layer");
                } else {
                    p.c8out = c8 && i == 1;                                     // down2: planar input (down1's act copy), C8 outputs
#ifdef SRT_TUNING
                    // timing experiment (wrong results): down2 on the C8 kernel as if its input were C8 - what moving down1's act copy to C8 would buy
                    static const bool d2c8 = []() { const char* v = getenv("SRT_TUNE_D2C8"); return v && v[0] == '1'; }();
                    if (d2c8 && p.c8out) rc2 = srt_launch_enc_c8(p, e->stream); else
#endif
                    rc2 = srt_launch_enc_f16(p, e->stream);
                }
            }
            if (e->act16 && i > 0 && rc2 == 1) return fail(-4, "internal: fp16 activation storage but no fp16 kernel for an encoder layer");
            if (rc2 == 1 && e->cfg.impl == SRT_IMPL_MFMA) { rc2 = srt_launch_enc2(p, e->stream); if (rc2 == 0 && producer_copy) act_ready = true; }
            if (rc2 < 0) return fail(-2, "encoder launch failed");
            if (rc2 == 1 && srt_launch_enc(p, e->cfg.impl, e->stream)) return fail(-2, "encoder launch failed");
        }
        SrtHeadParams head; memset(&head, 0, sizeof head);                     // head (spleeter.c:295-300)
        head.H = T; head.W = F; head.ntiles = ntiles; head.nstems = ns;
        head.src = e->up[5] + (size_t)s0 * ntiles * e->up_tile[5]; head.src_stem = (size_t)ntiles * e->up_tile[5]; head.src_tile = e->up_tile[5];
        head.w = cbase + e->lo.head_w; head.bias = cbase + e->lo.head_b; head.coeff_stem = SRT_COEFF_STRIDE;
        head.out = d_masks + (size_t)s0 * ntiles * 2 * HW; head.out_stem = (size_t)ntiles * 2 * HW; head.out_tile = 2 * HW;
        head.variant = e->cfg.variant;
        head.out16 = e->masks16_req && d_masks == e->masks && srt_head_out16_ok(head);
        e->last_masks16 = head.out16 != 0;
        bool head_done = false;
        for (int i = 0; i < 6; ++i) {                                           // decoder (spleeter.c:239-294)
            const LayerOff& L = e->lo.up[i];
            SrtConvParams p; memset(&p, 0, sizeof p);
            p.Cin = L.cin; p.Cout = L.cout; p.H = T >> (6 - i); p.W = F >> (6 - i); p.ntiles = ntiles; p.nstems = ns;
            const int sk = 5 - i;                                               // skip tensor = raw[5-i]; up1 consumes conv6 alone
            p.srcA = eoff(e, e->raw[sk], (size_t)s0 * ntiles * e->raw_tile[sk]); p.srcA_stem = (size_t)ntiles * e->raw_tile[sk]; p.srcA_tile = e->raw_tile[sk];
            if (i == 0) { p.CA = L.cin; p.srcB = p.srcA; p.srcB_stem = p.srcA_stem; p.srcB_tile = p.srcA_tile; }
            else {
                p.CA = L.cin / 2;
                p.srcB = eoff(e, e->up[i - 1], (size_t)s0 * ntiles * e->up_tile[i - 1]); p.srcB_stem = (size_t)ntiles * e->up_tile[i - 1]; p.srcB_tile = e->up_tile[i - 1];
            }
            p.in16 = e->act16; p.out16 = e->act16 && i < 5;
            p.c8srcB = c8 && i == 5;                                            // up6 reads up5's C8 output
            p.c8srcA = c8_l1 && i == 5;                                         // ... and down1's raw tensor, its skip input, where that is C8 too
            p.wraw = cbase + L.w; p.bias = cbase + L.b; p.bnShift = cbase + L.bn; p.bnScale = cbase + L.bn + L.cout;
            p.coeff_stem = SRT_COEFF_STRIDE;
            p.wpack = e->wpack_up[i] + (size_t)s0 * e->wpack_up_stem[i]; p.wpack_stem = e->wpack_up_stem[i];
            p.CP = L.cp;
            p.outRaw = nullptr;
            p.outAct = i < 5 ? eoff(e, e->up[i], (size_t)s0 * ntiles * e->up_tile[i]) : e->up[i] + (size_t)s0 * ntiles * e->up_tile[i];
            p.out_stem = (size_t)ntiles * e->up_tile[i]; p.out_tile = e->up_tile[i];
            p.act = actD; p.elu_mask = elu_mask; p.variant = e->cfg.variant;
            if (small) { p.ws = e->ws; p.ws_floats = e->ws_floats; }
            if (i == 4) { p.wpack2 = e->wpack2_u5 + (size_t)s0 * 64 * 15 * 32; p.wpack2_stem = 64 * 15 * 32; p.CP2 = 32; }
            snprintf(nm, sizeof nm, "up%d", i + 1);
            TimerScope ts(e, nm);
            int rc2 = 1;
            if (e->cfg.impl == SRT_IMPL_MFMA && e->wpack16_up[i]) {
                p.wpack16 = e->wpack16_up[i] + (size_t)s0 * e->wpack16_up_stem[i]; p.wpack16_stem = e->wpack16_up_stem[i];
                p.nsplit = e->cfg.precision == SRT_PREC_F16X2 ? 2 : 1;
                if (c8 && i < 5) {                                              // C8 in, C8 out (up5: class-stacked weights)
                    p.wpack16cs = e->wpack16cs_u5 + (size_t)s0 * SRT_W16CS_U5; p.wpack16cs_stem = SRT_W16CS_U5;
                    rc2 = srt_launch_dec_c8(p, e->stream);
                    if (rc2 == 1) return fail(-4, "internal: C8 activation layout but no C8 kernel for a decoder layer");
                } else rc2 = srt_launch_dec_f16(p, e->stream);
            }
            if (i == 5) {                                                       // up6 + head in one pass where covered (srt_up6_head_kernel); the plane is then not stored
                const unsigned bits = ((1u << ns) - 1u) << s0;
                e->up6_stale &= ~bits;
                if (e->cfg.impl == SRT_IMPL_MFMA && !e->graph_mode) {              // (graph replays would not maintain up6_stale; graphs are for launches far below this form's threshold)
                    rc2 = srt_launch_up6_head(p, head, e->stream);
                    if (rc2 == 0) { head_done = true; e->up6_stale |= bits; e->up6_params = p; e->up6_s0 = s0; }
                }
            }
            if (e->act16 && i < 5 && rc2 == 1) return fail(-4, "internal: fp16 activation storage but no fp16 kernel for a decoder layer");
            if (rc2 == 1 && e->wino_u[i] && (!few || srt_wino_force())) rc2 = srt_launch_dec_wino(p, e->wino_u[i] + (size_t)s0 * e->wino_u_stem[i], e->wino_u_stem[i], e->stream);
            if (rc2 == 1 && e->cfg.impl == SRT_IMPL_MFMA) rc2 = srt_launch_dec2(p, e->stream);
            if (rc2 < 0) return fail(-2, "decoder launch failed");
            if (rc2 == 1 && srt_launch_dec(p, e->cfg.impl, e->stream)) return fail(-2, "decoder launch failed");
        }
        if (!head_done) {
            TimerScope ts(e, "up7");
            if (srt_launch_head(head, e->stream)) return fail(-2, "head launch failed");
        }
    }
    return 0;
}

int srtForward(srt_engine* e, const float* d_mag, int ntiles, float* d_masks)
{
    if (!e) return fail(-1, "srtForward: null argument");
    if (!e->graph_mode) return forward_range(e, d_mag, ntiles, d_masks, 0, e->cfg.n_stems);
    DeviceScope ds(e->device);
    bool valid = d_mag && d_masks && ntiles >= 1 && ntiles <= e->cfg.max_tiles;
    for (int s = 0; s < e->cfg.n_stems; ++s) valid = valid && e->have_coeff[s];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (e->stream) (void)hipStreamIsCapturing(e->stream, &cap);
    if (valid && cap == hipStreamCaptureStatusNone) ensure_ws(e, (size_t)e->cfg.n_stems * ntiles);
    GraphKey k; memset(&k, 0, sizeof k);
    k.kind = 1; k.p0 = d_mag; k.p2 = d_masks; k.ntiles = ntiles; k.ns = e->cfg.n_stems;
    const int rc = run_graphed(e, k, valid, [&]() { return forward_range(e, d_mag, ntiles, d_masks, 0, e->cfg.n_stems); });
    if (!rc) e->last_ntiles = ntiles;
    return rc;
}

int srtPrepareForward(srt_engine* e, const float* d_mag, int ntiles, float* d_masks)
{
    if (!e) return fail(-1, "srtPrepareForward: null argument");
    SrtSetupLock setup;
    DeviceScope ds(e->device);
    const int rc = srtForward(e, d_mag, ntiles, d_masks);     // allocates the workspace, captures + instantiates (graph mode), runs once
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

int srtReleaseStaging(srt_engine* e)
{
    if (!e) return fail(-1, "srtReleaseStaging: null argument");
    DeviceScope ds(e->device);
    if (e->hs.ready) { hipStreamSynchronize(e->hs.s_in); hipStreamSynchronize(e->hs.s_out); }
    HIPCHK(hipStreamSynchronize(e->stream));
    free_staging(e);
    return 0;
}

int srtForwardStems(srt_engine* e, const float* d_mag, int ntiles, float* d_masks, int stem0, int nstems)
{
    if (!e) return fail(-1, "srtForward: null argument");
    return forward_range(e, d_mag, ntiles, d_masks, stem0, nstems);
}

int srtRatioMask(srt_engine* e, float* d_masks, int ntiles)
{
    if (!e || !d_masks || ntiles < 1) return fail(-1, "srtRatioMask: bad argument");
    DeviceScope ds(e->device);
    TimerScope ts(e, "ratio");
    if (srt_launch_ratio_mask(d_masks, e->cfg.n_stems, (size_t)ntiles * 2 * e->cfg.T * e->cfg.F, e->stream)) return fail(-2, "ratio-mask launch failed");
    return 0;
}

static SrtDspTables tables_of(const srt_engine* e) { SrtDspTables t; t.preWin = e->preWin; t.postWin = e->postWin; t.twiddle = e->twiddle; return t; }

int srtStftEx(srt_engine* e, const float* d_L, const float* d_R, size_t n, size_t frames, size_t rows, float* d_spec, float* d_mag)
{
    if (!e || !d_L || !d_R || !d_spec) return fail(-1, "srtStft: null argument");
    DeviceScope ds(e->device);
    if (rows < 1 || frames > rows) return fail(-1, "srtStft: need 1 <= frames <= rows");
    const int T = e->cfg.T;
    const size_t ntiles = (rows + T - 1) / T;
    if (d_mag && ntiles > (size_t)e->cfg.max_tiles) return fail(-1, "srtStft: more than max_tiles * T rows");
    SrtStftParams p; memset(&p, 0, sizeof p);
    p.L = d_L; p.R = d_R; p.nsamples = n;
    p.frames_computed = (int)frames;
    p.rows_total = (int)rows;
    p.spec = (float2*)d_spec; p.spec_ch_stride = rows * SRT_SPEC_LD;
    p.mag = nullptr; p.T = T; p.F = e->cfg.F; p.tab = tables_of(e);
    if (d_mag) {
        // magnitude rows exist for whole tiles: rows..ntiles*T are zero (main.c:507-514)
        p.mag = d_mag;
        if (ntiles * T > rows) HIPCHK(hipMemsetAsync(d_mag, 0, ntiles * 2 * (size_t)T * e->cfg.F * sizeof(float), e->stream));
    }
    TimerScope ts(e, "stft");
    if (srt_launch_stft(p, e->stream)) return fail(-2, "stft launch failed");
    return 0;
}

int srtStft(srt_engine* e, const float* d_L, const float* d_R, size_t n, float* d_spec, float* d_mag)
{
    if (n < SRT_FFT) return fail(-1, "srtStft: need at least 4096 samples");          // the reference underflows here (stftFix.c:378)
    return srtStftEx(e, d_L, d_R, n, srtStftFrames(n), srtStftRows(n), d_spec, d_mag);
}

static int istft_issue(srt_engine* e, const float* d_spec, size_t rows, const float* d_masks, float* d_out, bool ratio, bool masks16 = false);
// (the public entry applies the masks as they are given: srtRatioMask is its caller's business)
int srtIstft(srt_engine* e, const float* d_spec, size_t rows, const float* d_masks, float* d_out) { return istft_issue(e, d_spec, rows, d_masks, d_out, false); }
static int istft_issue(srt_engine* e, const float* d_spec, size_t rows, const float* d_masks, float* d_out, bool ratio, bool masks16)
{
    if (!e || !d_spec || !d_out) return fail(-1, "srtIstft: null argument");
    DeviceScope ds(e->device);
    if (rows < 1) return fail(-1, "srtIstft: no rows");
    const int T = e->cfg.T;
    if (d_masks && (rows + T - 1) / T > (size_t)e->cfg.max_tiles) return fail(-1, "srtIstft: rows exceed max_tiles * T");
    SrtIstftParams p; memset(&p, 0, sizeof p);
    p.spec = (const float2*)d_spec; p.spec_ch_stride = rows * SRT_SPEC_LD;
    p.frames = (int)rows; p.masks = d_masks; p.nstems = e->cfg.n_stems; p.ntiles = (int)((rows + T - 1) / T);
    p.T = T; p.F = e->cfg.F;
    for (int s = 0; s < SRT_MAX_STEMS; ++s) p.oob[s] = e->cfg.oob_weight[s];
    p.ratio = ratio ? 1 : 0;
    p.masks16 = masks16 ? 1 : 0;
    p.frames_out = nullptr; p.out = d_out; p.out_len = srtIstftLength(rows); p.tab = tables_of(e);
    TimerScope ts(e, "istft");
    if (srt_launch_istft(p, e->stream)) return fail(-2, "istft launch failed");
    return 0;
}

static int separate_issue(srt_engine* e, const float* d_L, const float* d_R, size_t n, size_t frames, size_t rows, float* d_out)
{
    const int T = e->cfg.T;
    const size_t ntiles = (rows + T - 1) / T;
    int rc = srtStftEx(e, d_L, d_R, n, frames, rows, (float*)e->spec, e->mag);
    if (rc) return rc;
    // fp16 mode: the masks between the head and the inverse transform - the engine's own buffer, never seen by a caller - are halves where both kernels take them
    // (srt_head_rows_kernel<.., true> / srt_istft_ola3_kernel<.., true>: F <= 1024, no ratio mask, head launches of >= 1024 workgroups; SPLEETERRT_M16=0: floats, for A/B runs)
    const char* m16v = getenv("SPLEETERRT_M16");
    e->masks16_req = e->cfg.precision == SRT_PREC_F16 && e->act16 && !e->cfg.ratio_mask && e->cfg.F <= 1024 && !(m16v && m16v[0] == '0');
    rc = forward_range(e, e->mag, (int)ntiles, e->masks, 0, e->cfg.n_stems);
    e->masks16_req = false;
    if (rc) return rc;
    // ratio_mask: normalised across the stems inside the inverse kernel's prologue (srt_ratio_of, srt_dsp.hip) - e->masks keeps the raw sigmoid masks
    return istft_issue(e, (const float*)e->spec, rows, e->masks, d_out, e->cfg.ratio_mask != 0, e->last_masks16);
}

int srtSeparateEx(srt_engine* e, const float* d_L, const float* d_R, size_t n, size_t frames, size_t rows, float* d_out)
{
    if (!e) return fail(-1, "srtSeparate: null engine");
    const int T = e->cfg.T;
    const size_t ntiles = (rows + T - 1) / T;
    if (rows < 1 || ntiles > (size_t)e->cfg.max_tiles) return fail(-1, "srtSeparate: signal longer than max_tiles * T frames");
    if (!e->graph_mode) return separate_issue(e, d_L, d_R, n, frames, rows, d_out);
    DeviceScope ds(e->device);
    bool valid = d_L && d_R && d_out && frames <= rows;
    for (int s = 0; s < e->cfg.n_stems; ++s) valid = valid && e->have_coeff[s];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (e->stream) (void)hipStreamIsCapturing(e->stream, &cap);
    if (valid && cap == hipStreamCaptureStatusNone) ensure_ws(e, (size_t)e->cfg.n_stems * ntiles);
    GraphKey k; memset(&k, 0, sizeof k);
    k.kind = 2; k.p0 = d_L; k.p1 = d_R; k.p2 = d_out; k.n = n; k.frames = frames; k.rows = rows;
    const int rc = run_graphed(e, k, valid, [&]() { return separate_issue(e, d_L, d_R, n, frames, rows, d_out); });
    if (!rc) e->last_ntiles = (int)ntiles;
    return rc;
}

int srtSetGraphMode(srt_engine* e, int enable)
{
    if (!e) return fail(-1, "null engine");
    DeviceScope ds(e->device);
    if (!enable) { hipStreamSynchronize(e->stream); free_graphs(e); }
    e->graph_mode = enable != 0;
    return 0;
}

// iSTFT of one spectrum under ONE stem's mask (or none) into a [2][len] destination
static int istft_one(srt_engine* e, const float2* spec, size_t rows, const float* mask_stem, float oob, float* d_dst, const char* tag)
{
    DeviceScope ds(e->device);
    const int T = e->cfg.T;
    SrtIstftParams p; memset(&p, 0, sizeof p);
    p.spec = spec; p.spec_ch_stride = rows * SRT_SPEC_LD;
    p.frames = (int)rows; p.masks = mask_stem; p.nstems = 1; p.ntiles = (int)((rows + T - 1) / T);
    p.T = T; p.F = e->cfg.F; p.oob[0] = oob;
    p.frames_out = nullptr; p.out = d_dst; p.out_len = srtIstftLength(rows); p.tab = tables_of(e);
    TimerScope ts(e, tag);
    if (srt_launch_istft(p, e->stream)) return fail(-2, "istft launch failed");
    return 0;
}

// The offline CLI's separation flows, device resident (Executable/main.c:776-798 two outputs, :845-928 three outputs).
// Sub-network 0 is the CLI's net[0] (drum, mode 1), sub-network 1 its net[1] (vocal, mode 0)  (main.c:759-760).
//   2: vocal = istft(mask1 . S);  accompaniment = input - vocal                                  (time-domain residual)
//   3: drum = istft(mask0 . S);  R = S - mask0 . S;  vocal = istft(mask1(|R|) . R);  accompaniment = istft(R) - vocal
// d_out: [stems][2][srtIstftLength(rows)] in the CLI's file order: (Vocal, Accompaniment) or (Drum, Vocal, Accompaniment).
// Explicit geometry as srtSeparateEx (a tile range of a longer file).  residual_now = false leaves the last, time-domain subtraction
// to the caller: the chunked pipeline first adds the previous chunk's overlap to every plane (the accompaniment slot then holds
// the UNsubtracted term: nothing for 2 outputs, istft(R) for 3) and subtracts afterwards (cli_time_residual).
// samples [lo, hi) of the planes (hi = 0: all of them)
static int cli_time_residual(srt_engine* e, const float* d_L, const float* d_R, size_t n, int stems, float* d_out, size_t len, size_t lo = 0, size_t hi = 0)
{
    TimerScope ts(e, "residual");
    if (!hi) hi = len;
    const int rc = stems == 2 ? srt_launch_time_residual(d_L, d_R, n, d_out, len, d_out + 2 * len, lo, hi, e->stream)
                              : srt_launch_time_residual(d_out + 4 * len, d_out + 5 * len, len, d_out + 2 * len, len, d_out + 4 * len, lo, hi, e->stream);
    return rc ? fail(-2, "residual launch failed") : 0;
}

static int cli_issue(srt_engine* e, const float* d_L, const float* d_R, size_t n, size_t frames, size_t rows, int stems, float* d_out, bool residual_now)
{
    const int T = e->cfg.T;
    const size_t ntiles = (rows + T - 1) / T, len = srtIstftLength(rows);
    const size_t HW2 = 2 * (size_t)T * e->cfg.F;
    float* mask0 = e->masks;                                  // stem stride of this batch = ntiles instances
    float* mask1 = e->masks + ntiles * HW2;
    int rc = srtStftEx(e, d_L, d_R, n, frames, rows, (float*)e->spec, e->mag);
    if (rc) return rc;
    if (stems == 2) {
        if ((rc = forward_range(e, e->mag, (int)ntiles, e->masks, 1, 1))) return rc;
        if ((rc = istft_one(e, e->spec, rows, mask1, e->cfg.oob_weight[1], d_out, "istft"))) return rc;
        return residual_now ? cli_time_residual(e, d_L, d_R, n, stems, d_out, len) : 0;
    }
    if (!e->spec2) HIPCHK(hipMalloc((void**)&e->spec2, (size_t)2 * e->rows_cap * SRT_SPEC_LD * sizeof(float2)));
    if ((rc = forward_range(e, e->mag, (int)ntiles, e->masks, 0, 1))) return rc;
    if ((rc = istft_one(e, e->spec, rows, mask0, e->cfg.oob_weight[0], d_out, "istft"))) return rc;                 // Drum
    {
        SrtResidualParams r; memset(&r, 0, sizeof r);
        r.spec = e->spec; r.res = e->spec2; r.spec_ch_stride = rows * SRT_SPEC_LD; r.rows = (int)rows; r.T = T; r.F = e->cfg.F;
        r.mask = mask0; r.oob = e->cfg.oob_weight[0]; r.mag = e->mag;
        TimerScope ts(e, "residual");
        if (srt_launch_residual(r, e->stream)) return fail(-2, "residual launch failed");
    }
    if ((rc = istft_one(e, e->spec2, rows, nullptr, 1.0f, d_out + 4 * len, "istft"))) return rc;                      // accompaniment + vocal
    if ((rc = forward_range(e, e->mag, (int)ntiles, e->masks, 1, 1))) return rc;
    if ((rc = istft_one(e, e->spec2, rows, mask1, e->cfg.oob_weight[1], d_out + 2 * len, "istft"))) return rc;       // Vocal
    return residual_now ? cli_time_residual(e, d_L, d_R, n, stems, d_out, len) : 0;
}

static int cli_check(srt_engine* e, int stems)
{
    if (stems != 2 && stems != 3) return fail(-1, "srtSeparateCli: stems must be 2 or 3");
    if (e->cfg.n_stems < 2) return fail(-1, "srtSeparateCli: the engine needs sub-networks 0 (drum) and 1 (vocal)");
    if (e->cfg.ratio_mask) return fail(-1, "srtSeparateCli: ratio_mask does not apply to the CLI flows (the sub-networks see different inputs)");
    return 0;
}

int srtSeparateCli(srt_engine* e, const float* d_L, const float* d_R, size_t n, int stems, float* d_out)
{
    if (!e || !d_L || !d_R || !d_out) return fail(-1, "srtSeparateCli: null argument");
    DeviceScope ds(e->device);
    int rc = cli_check(e, stems);
    if (rc) return rc;
    if (n < SRT_FFT) return fail(-1, "srtSeparateCli: need at least 4096 samples");
    const int T = e->cfg.T;
    const size_t rows = srtStftRows(n), frames = srtStftFrames(n), ntiles = (rows + T - 1) / T;
    if (ntiles > (size_t)e->cfg.max_tiles) return fail(-1, "srtSeparateCli: signal longer than max_tiles * T frames (srtSeparateCliHost walks longer files chunk by chunk)");
    return cli_issue(e, d_L, d_R, n, frames, rows, stems, d_out, true);
}

// Grow-only device staging shared by the host-buffer entry points (kept in the engine, freed by srtDestroy).
static int ensure_staging(srt_engine* e, size_t in_floats, size_t out_floats, size_t carry_floats, int nbuf)
{
    HostStaging& h = e->hs;
    if (!h.ready) {
        HIPCHK(hipStreamCreateWithFlags(&h.s_in, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&h.s_out, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIPCHK(hipEventCreateWithFlags(&h.ev_in[b], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h.ev_cmp[b], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h.ev_out[b], hipEventDisableTiming));
        }
        h.ready = true;
    }
    if (in_floats > h.in_cap || out_floats > h.out_cap || carry_floats > h.carry_cap) {
        // buffers may still be in use by an earlier call's asynchronous work: drain before replacing them
        HIPCHK(hipStreamSynchronize(h.s_in)); HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipStreamSynchronize(h.s_out));
    }
    for (int b = 0; b < 2; ++b) {
        if (in_floats > h.in_cap || (b < nbuf && !h.d_in[b])) {
            if (h.d_in[b]) { hipFree(h.d_in[b]); h.d_in[b] = nullptr; }
            if (b < nbuf) HIPCHK(hipMalloc((void**)&h.d_in[b], (in_floats > h.in_cap ? in_floats : h.in_cap) * sizeof(float)));
        }
        if (out_floats > h.out_cap || (b < nbuf && !h.d_out[b])) {
            if (h.d_out[b]) { hipFree(h.d_out[b]); h.d_out[b] = nullptr; }
            if (b < nbuf) HIPCHK(hipMalloc((void**)&h.d_out[b], (out_floats > h.out_cap ? out_floats : h.out_cap) * sizeof(float)));
        }
    }
    if (in_floats > h.in_cap) h.in_cap = in_floats;
    if (out_floats > h.out_cap) h.out_cap = out_floats;
    if (carry_floats > h.carry_cap) {
        if (h.d_carry) { hipFree(h.d_carry); h.d_carry = nullptr; }
        HIPCHK(hipMalloc((void**)&h.d_carry, carry_floats * sizeof(float)));
        h.carry_cap = carry_floats;
    }
    return 0;
}

// seam: how a tile RANGE of a longer stream joins its neighbours when they run on other devices (srt_multi.hip); {0, nullptr, false} = the whole stream
struct SrtSeam { size_t out_stride; float* h_tail; bool head; };
static int host_stream(srt_engine* e, const float* h_L, const float* h_R, size_t n, size_t frames, size_t rows, float* h_out, unsigned flags, int cli_stems, SrtSeam seam = SrtSeam{0, nullptr, false});

// Host-buffer form of srtSeparateCli for plain-C callers (the CLI harness): synchronous.  A file that fits the engine's capacity
// (max_tiles tiles) is one resident batch: H2D, chain, D2H.  A longer one - any length, as the reference's tile loop over a
// host-resident spectrogram handles (main.c:455-495) - goes through the chunked pipeline of srtSeparateHostStream: max_tiles tiles
// at a time, copies overlapped with compute, the residual chain evaluated per chunk (it is row-local) and the time-domain
// subtraction applied after the 3072-sample chunk overlaps have been added on the device.
int srtSeparateCliHost(srt_engine* e, const float* h_L, const float* h_R, size_t n, int stems, float* h_out)
{
    if (!e || !h_L || !h_R || !h_out) return fail(-1, "srtSeparateCliHost: null argument");
    DeviceScope ds(e->device);
    int rc = cli_check(e, stems);
    if (rc) return rc;
    if (n < SRT_FFT) return fail(-1, "srtSeparateCli: need at least 4096 samples");
    const size_t rows = srtStftRows(n), len = srtIstftLength(rows);
    if ((rows + e->cfg.T - 1) / e->cfg.T > (size_t)e->cfg.max_tiles)
        return host_stream(e, h_L, h_R, n, srtStftFrames(n), rows, h_out, 0, stems);
    rc = ensure_staging(e, 2 * n, (size_t)stems * 2 * len, 0, 1);
    if (rc) return rc;
    float *d_in = e->hs.d_in[0], *d_out = e->hs.d_out[0];
    hipError_t er = hipMemcpyAsync(d_in, h_L, n * sizeof(float), hipMemcpyHostToDevice, e->stream);
    if (er == hipSuccess) er = hipMemcpyAsync(d_in + n, h_R, n * sizeof(float), hipMemcpyHostToDevice, e->stream);
    rc = er == hipSuccess ? srtSeparateCli(e, d_in, d_in + n, n, stems, d_out) : fail(-2, "HIP error: %s", hipGetErrorString(er));
    if (!rc) {
        er = hipMemcpyAsync(h_out, d_out, (size_t)stems * 2 * len * sizeof(float), hipMemcpyDeviceToHost, e->stream);
        if (er == hipSuccess) er = hipStreamSynchronize(e->stream);
        if (er != hipSuccess) rc = fail(-2, "HIP error: %s", hipGetErrorString(er));
    }
    hipStreamSynchronize(e->stream);
    return rc;                                                 // (the staging is grow-only and reused by later calls; srtReleaseStaging frees it)
}

int srtSeparate(srt_engine* e, const float* d_L, const float* d_R, size_t n, float* d_out)
{
    if (n < SRT_FFT) return fail(-1, "srtSeparate: need at least 4096 samples");
    return srtSeparateEx(e, d_L, d_R, n, srtStftFrames(n), srtStftRows(n), d_out);
}

// Long host-resident stream through one GPU: the stream is cut into chunks of max_tiles tiles (tiles are independent,
// main.c:455-495); chunk c+1's PCM goes up and chunk c-1's stems come down on two copy streams while chunk c computes,
// and the 3072-sample overlap between consecutive chunks is added on the device (srt_carry_kernel), so every output
// sample crosses PCIe exactly once.  Geometry as srtSeparateEx (a tile range of a longer stream: rows = whole tiles,
// frames = rows; the whole stream: rows = srtStftRows(n), frames = srtStftFrames(n)).
// cli_stems = 0: the n_stems sub-networks on the same input (srtSeparateEx per chunk); 2 / 3: the CLI's flows (cli_issue per chunk)
// seam (multi-device ranges, srt_multi.hip): h_out points at the range's first sample inside planes of seam.out_stride floats (the whole
// stream's output length); with seam.h_tail the range's last 3072 samples - the overlap-add contribution that belongs to the NEXT range's
// first samples - go to h_tail [planes][3072] instead of h_out; seam.head says a previous range will add its tail to this range's first
// 3072 samples.  The CLI flows' time-domain subtraction is left out on exactly those seam samples (both contributions have to be added
// first: the joiner does that, as the reference's main() does its subtraction on the host, main.c:794-798).
static int host_stream(srt_engine* e, const float* h_L, const float* h_R, size_t n, size_t frames, size_t rows, float* h_out, unsigned flags, int cli_stems, SrtSeam seam)
{
    if (!e || !h_L || !h_R || !h_out) return fail(-1, "srtSeparateHostStream: null argument");
    if (rows < 1 || frames > rows) return fail(-1, "srtSeparateHostStream: need 1 <= frames <= rows");
    DeviceScope ds(e->device);
    const int S = cli_stems ? cli_stems : e->cfg.n_stems, T = e->cfg.T, NP = S * 2;
    const size_t chunk_rows = (size_t)e->cfg.max_tiles * T, tail = SRT_FFT - SRT_HOP;
    const size_t nchunks = (rows + chunk_rows - 1) / chunk_rows, total_len = seam.out_stride ? seam.out_stride : srtIstftLength(rows);
    const size_t in_cap = chunk_rows * SRT_HOP + tail, out_cap = srtIstftLength(chunk_rows);
    int rc = ensure_staging(e, 2 * in_cap, (size_t)NP * out_cap, (size_t)NP * tail, 2);
    if (rc) return rc;
    HostStaging& h = e->hs;
    // Page-locked caller buffers let the copies run asynchronously.  SRT_HOST_PINNED: the caller guarantees they already are
    // (hipHostMalloc / hipHostRegister / torch pin_memory) and nothing is registered here; otherwise the three buffers are
    // registered for the duration of the call (if that fails the copies still work, staged by the runtime).
    bool pinL = false, pinR = false, pinO = false;
    if (!(flags & SRT_HOST_PINNED)) {
        pinL = hipHostRegister((void*)h_L, n * sizeof(float), hipHostRegisterDefault) == hipSuccess;
        pinR = hipHostRegister((void*)h_R, n * sizeof(float), hipHostRegisterDefault) == hipSuccess;
        pinO = !seam.out_stride && hipHostRegister((void*)h_out, (size_t)NP * total_len * sizeof(float), hipHostRegisterDefault) == hipSuccess;
        (void)hipGetLastError();
    }
    hipError_t er = hipSuccess;
#define STEP(x) do { if (er == hipSuccess) er = (x); } while (0)
    for (size_t c = 0; c < nchunks && er == hipSuccess && rc == 0; ++c) {
        const int b = (int)(c & 1);
        const size_t row0 = c * chunk_rows, row1 = row0 + chunk_rows < rows ? row0 + chunk_rows : rows, crow = row1 - row0;
        const size_t s0 = row0 * SRT_HOP, send = row1 * SRT_HOP + tail < n ? row1 * SRT_HOP + tail : n;
        const size_t ns = send > s0 ? send - s0 : 0;
        const size_t cfr = frames > row0 ? (frames - row0 < crow ? frames - row0 : crow) : 0;
        const size_t clen = srtIstftLength(crow);
        // upload: the input buffer is free once the compute that read it two chunks ago has finished
        if (c >= 2) STEP(hipStreamWaitEvent(h.s_in, h.ev_cmp[b], 0));
        if (ns) {
            STEP(hipMemcpyAsync(h.d_in[b], h_L + s0, ns * sizeof(float), hipMemcpyHostToDevice, h.s_in));
            STEP(hipMemcpyAsync(h.d_in[b] + in_cap, h_R + s0, ns * sizeof(float), hipMemcpyHostToDevice, h.s_in));
        }
        STEP(hipEventRecord(h.ev_in[b], h.s_in));
        // compute: needs this chunk's PCM and the output buffer drained by the download of two chunks ago
        STEP(hipStreamWaitEvent(e->stream, h.ev_in[b], 0));
        if (c >= 2) STEP(hipStreamWaitEvent(e->stream, h.ev_out[b], 0));
        if (er != hipSuccess) break;
        rc = cli_stems ? cli_issue(e, h.d_in[b], h.d_in[b] + in_cap, ns, cfr, crow, cli_stems, h.d_out[b], false)
                       : srtSeparateEx(e, h.d_in[b], h.d_in[b] + in_cap, ns, cfr, crow, h.d_out[b]);
        if (rc) break;
        if (srt_launch_carry(h.d_out[b], clen, NP, crow * SRT_HOP, h.d_carry, c == 0, c + 1 == nchunks, e->stream)) { rc = fail(-2, "carry launch failed"); break; }
        // CLI flows: the time-domain subtraction comes after the seam has been added (the planes now hold the stitched signals)
        const bool to_tail = c + 1 == nchunks && seam.h_tail;                      // the range's last 3072 samples belong to the next range's seam
        if (cli_stems) {
            const size_t lo = c == 0 && seam.head ? tail : 0, hi = to_tail ? crow * SRT_HOP : clen;
            if (hi > lo && (rc = cli_time_residual(e, h.d_in[b], h.d_in[b] + in_cap, ns, cli_stems, h.d_out[b], clen, lo, hi))) break;
        }
        STEP(hipEventRecord(h.ev_cmp[b], e->stream));
        // download: every plane's [0, crow*1024) (+ the final 3072 on the last chunk) lands at its place in h_out
        STEP(hipStreamWaitEvent(h.s_out, h.ev_cmp[b], 0));
        const size_t take = c + 1 == nchunks && !to_tail ? clen : crow * SRT_HOP;
        STEP(hipMemcpy2DAsync(h_out + s0, total_len * sizeof(float), h.d_out[b], clen * sizeof(float), take * sizeof(float), NP, hipMemcpyDeviceToHost, h.s_out));
        if (to_tail) STEP(hipMemcpy2DAsync(seam.h_tail, tail * sizeof(float), h.d_out[b] + crow * SRT_HOP, clen * sizeof(float), tail * sizeof(float), NP, hipMemcpyDeviceToHost, h.s_out));
        STEP(hipEventRecord(h.ev_out[b], h.s_out));
    }
#undef STEP
    hipStreamSynchronize(h.s_in);
    hipStreamSynchronize(e->stream);
    hipStreamSynchronize(h.s_out);
    if (rc == 0 && er != hipSuccess) rc = fail(-2, "HIP error: %s", hipGetErrorString(er));
    if (pinL) hipHostUnregister((void*)h_L);
    if (pinR) hipHostUnregister((void*)h_R);
    if (pinO) hipHostUnregister((void*)h_out);
    return rc;
}

int srt_engine_host_range(srt_engine* e, const float* h_L, const float* h_R, size_t n, size_t frames, size_t rows, float* h_out, size_t out_stride,
                          float* h_tail, bool head, unsigned flags, int cli_stems)
{
    if (cli_stems) { DeviceScope ds(e->device); const int rc = cli_check(e, cli_stems); if (rc) return rc; }
    return host_stream(e, h_L, h_R, n, frames, rows, h_out, flags, cli_stems, SrtSeam{out_stride, h_tail, head});
}
const float* srt_engine_coeff_device(const srt_engine* e, int stem) { return e->coeff_all + (size_t)stem * SRT_COEFF_STRIDE; }
int srt_engine_device(const srt_engine* e) { return e->device; }
void* srt_engine_stream(const srt_engine* e) { return (void*)e->stream; }

int srtSeparateHostStreamEx(srt_engine* e, const float* h_L, const float* h_R, size_t n, size_t frames, size_t rows, float* h_out, unsigned flags)
{
    return host_stream(e, h_L, h_R, n, frames, rows, h_out, flags, 0);
}

int srtSeparateHostStream(srt_engine* e, const float* h_L, const float* h_R, size_t n, size_t frames, size_t rows, float* h_out)
{
    return srtSeparateHostStreamEx(e, h_L, h_R, n, frames, rows, h_out, 0);
}

int srtCopyTensor(srt_engine* e, const char* name, int stem, int tile, float* h_dst, size_t max_floats)
{
    if (!e || !name || !h_dst) return fail(-1, "srtCopyTensor: null argument");
    DeviceScope ds(e->device);
    if (!name[0]) return fail(-1, "srtCopyTensor: empty tensor name");
    if (stem < 0 || stem >= e->cfg.n_stems || tile < 0 || tile >= e->last_ntiles) return fail(-1, "srtCopyTensor: stem / tile outside the last forward batch");
    const int idx = name[strlen(name) - 1] - '1';
    const float* base = nullptr; size_t per = 0; bool derived = false;
    if (!strncmp(name, "conv", 4) && idx >= 0 && idx < 6) { base = e->raw[idx]; per = e->raw_tile[idx]; }
    else if (!strncmp(name, "act", 3) && idx >= 0 && idx < 5) { base = e->raw[idx]; per = e->raw_tile[idx]; derived = true; }
    else if (!strncmp(name, "up", 2) && idx >= 0 && idx < 6) { base = e->up[idx]; per = e->up_tile[idx]; }
    else return fail(-1, "srtCopyTensor: unknown tensor %s", name);
    if (per > max_floats) return fail(-1, "srtCopyTensor: destination too small");
    // instance stride = ntiles of the last srtForward call
    if (name[0] == 'u' && idx == 5 && ((e->up6_stale >> stem) & 1u)) {       // the fused launch kept the plane in LDS: materialise the tap with the plain up6 kernel
        if (stem < e->up6_s0 || stem >= e->up6_s0 + e->up6_params.nstems) return fail(-1, "srtCopyTensor: up6 of this stem was not stored by the fused up6 + head launch of an earlier stem range");
        if (srt_launch_dec(e->up6_params, e->cfg.impl, e->stream)) return fail(-2, "up6 launch failed");
    }
    const bool halves = e->act16 && !(name[0] == 'u' && idx == 5);           // up6 (the head's input) is always fp32
    const float* src = halves ? eoff(e, const_cast<float*>(base), ((size_t)stem * e->last_ntiles + tile) * per) : base + ((size_t)stem * e->last_ntiles + tile) * per;
    float* tmp = nullptr;
    // raw2..raw6 (and the act taps derived from them) and up1..up5 of a large fp16-storage batch are channel-interleaved by eight (srt_nn5.hip): the tap
    // is returned planar, like every other
    const bool c8 = halves && e->last_c8 && ((name[0] == 'u') ? idx <= 4 : (idx >= 1 || e->last_c8_l1));
    float* planar = nullptr;
    if (c8) {
        const int C = name[0] == 'u' ? DEC_CH[idx][1] : ENC_CH[idx][1];
        HIPCHK(hipMalloc((void**)&planar, per * sizeof(float)));
        if (srt_launch_c8_to_float(src, planar, C, per / C, e->stream)) { hipFree(planar); return fail(-2, "conversion launch failed"); }
        src = planar;
    }
    const bool halves_in = halves && !c8;
    if (derived) {
        // "actN" is no longer stored: the next encoder layer applies act(bn(convN)) while staging.  Materialise it for the tap.
        const LayerOff& L = e->lo.down[idx];
        const float* c = e->coeff_all + (size_t)stem * SRT_COEFF_STRIDE;
        HIPCHK(hipMalloc((void**)&tmp, per * sizeof(float)));
        if (srt_launch_bn_act(src, halves_in, tmp, c + L.bn + L.cout, c + L.bn, L.cout, per / L.cout, e->cfg.stem_mode[stem] ? SRT_ACT_ELU : SRT_ACT_LEAKY, e->cfg.variant, e->stream)) { hipFree(tmp); if (planar) hipFree(planar); return fail(-2, "bn-act launch failed"); }
        src = tmp;
    } else if (halves_in) {
        HIPCHK(hipMalloc((void**)&tmp, per * sizeof(float)));
        if (srt_launch_half_to_float(src, tmp, per, e->stream)) { hipFree(tmp); return fail(-2, "conversion launch failed"); }
        src = tmp;
    }
    hipError_t er = hipStreamSynchronize(e->stream);
    if (er == hipSuccess) er = hipMemcpy(h_dst, src, per * sizeof(float), hipMemcpyDeviceToHost);
    if (tmp) hipFree(tmp);
    if (planar) hipFree(planar);
    HIPCHK(er);
    return 0;
}

int srtSetTiming(srt_engine* e, int enable)
{
    if (!e) return fail(-1, "null engine");
    DeviceScope ds(e->device);
    for (auto& t : e->tlog) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    e->tlog.clear();
    e->timing = enable != 0;
    return 0;
}

int srtGetTiming(srt_engine* e, char* names, size_t names_bytes, float* ms, int max_entries)
{
    if (!e || !ms || max_entries < 0) return fail(-1, "srtGetTiming: bad argument");
    DeviceScope ds(e->device);
    HIPCHK(hipStreamSynchronize(e->stream));
    int n = 0; size_t used = 0;
    if (names && names_bytes) names[0] = 0;
    for (auto& t : e->tlog) {
        if (n >= max_entries) break;
        float v = 0; hipEventElapsedTime(&v, t.a, t.b);
        ms[n++] = v;
        if (names && used + t.name.size() + 2 < names_bytes) { memcpy(names + used, t.name.c_str(), t.name.size()); used += t.name.size(); names[used++] = ','; names[used] = 0; }
    }
    return n;
}

// The kernel that ran each timed launch (same order and count as srtGetTiming), ';'-separated, as rocprofv3 prints kernel names:
// the demangled symbol without its return type and argument list, e.g. "srt_dec_wino<4, 16, 1, 0>".  For a split-K layer this is
// the layer kernel (the reduction that follows it is not named).  If the runtime cannot resolve a symbol the launch expression's
// source text is reported instead.
int srtGetTimingKernels(srt_engine* e, char* kernels, size_t kernels_bytes)
{
    if (!e || !kernels || !kernels_bytes) return fail(-1, "srtGetTimingKernels: bad argument");
    DeviceScope ds(e->device);
    kernels[0] = 0;
    size_t used = 0; int n = 0;
    for (auto& t : e->tlog) {
        std::string nm;
        const char* sym = t.kfn ? hipKernelNameRefByPtr(t.kfn, e->stream) : nullptr;
        (void)hipGetLastError();
        if (sym && sym[0]) {
            int st = 0;
            char* dm = abi::__cxa_demangle(sym, nullptr, nullptr, &st);
            nm = (st == 0 && dm) ? dm : sym;
            free(dm);
            if (!nm.compare(0, 5, "void ")) nm.erase(0, 5);
            // cut the argument list: the '(' that closes the name is the first one outside template brackets
            int depth = 0;
            for (size_t i = 0; i < nm.size(); ++i) {
                if (nm[i] == '<') ++depth; else if (nm[i] == '>') --depth;
                else if (nm[i] == '(' && depth == 0) { nm.erase(i); break; }
            }
        } else if (t.ktext) {
            nm = t.ktext;
            if (nm.size() >= 2 && nm.front() == '(' && nm.back() == ')') nm = nm.substr(1, nm.size() - 2);
        } else nm = "?";
        if (used + nm.size() + 2 >= kernels_bytes) break;
        memcpy(kernels + used, nm.c_str(), nm.size()); used += nm.size(); kernels[used++] = ';'; kernels[used] = 0;
        ++n;
    }
    return n;
}
