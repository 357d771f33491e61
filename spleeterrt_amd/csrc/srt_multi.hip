// srt_multi.hip — one node, several devices: the reference CLI's tile-range fan-out (Executable/main.c:544-673, `processMT`: spawnNthreads
// workers, each with its own `nn` instance and a contiguous tile range, one shared read-only weight blob) with a device where the reference
// has a CPU thread.  Plain C ABI (include/spleeterrt_amd.h, srtMulti*): host programs stay C.
//
//   * one srt_engine per entry of the device list, created on that device; one host thread per engine while a call runs (std::thread here
//     = pthread_create in processMT);
//   * partition: srtRankSpan() - rank g gets tiles [g * ceil(N/G), ...), PCM samples [tile0*T*1024, tile1*T*1024 + 3072), the same arithmetic as
//     spleeterrt_amd/stream.py:rank_span (tests/test_sharding.py holds the two against each other);
//   * weights: read once by the caller, uploaded to the first device and distributed with ONE ncclBroadcast per blob over the devices' RCCL
//     clique (ncclCommInitAll: single process, one communicator per device; xGMI between the GPUs of a node).  librccl is dlopen'ed on first use,
//     so the plugin / single-device users of this library never load it; SPLEETERRT_NO_RCCL=1 (or a missing librccl) falls back to hipMemcpyPeer;
//   * no data-path collective: each range's overlap-add contribution is complete except for the 3072 samples it shares with its neighbour; every
//     engine writes its samples straight into the caller's output planes and hands its last 3072 samples ("tail") to the joiner, which adds them
//     to the next range's first samples on the host - 3072 x planes x (G - 1) additions - and finishes the CLI flows' time-domain subtraction
//     (main.c:794-798, 924-928) on those seam samples (the devices have done it everywhere else).
#include "srt_internal.h"
#include "spleeterrt_amd.h"
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ---- the slice of the RCCL C API used here (rccl.h declares the same prototypes; resolved at run time)
typedef struct ncclComm* srt_ncclComm_t;
typedef int (*fn_ncclCommInitAll)(srt_ncclComm_t*, int, const int*);
typedef int (*fn_ncclCommDestroy)(srt_ncclComm_t);
typedef int (*fn_ncclBroadcast)(const void*, void*, size_t, int, int, srt_ncclComm_t, hipStream_t);
typedef int (*fn_ncclGroup)(void);
typedef const char* (*fn_ncclGetErrorString)(int);
#define SRT_NCCL_FLOAT 7          // ncclFloat32 (rccl.h: ncclDataType_t)

struct srt_multi {
    srt_config cfg;
    std::vector<int> dev;                  // device of engine g
    std::vector<srt_engine*> eng;
    std::vector<hipStream_t> stream;       // engine g's compute stream (created on its device)
    std::vector<int> udev;                 // distinct devices, in first-seen order; udev[0] is the broadcast root
    std::vector<int> uidx;                 // engine g -> index into udev
    std::vector<float*> blob;              // per distinct device: staging for one spleeterCoeff blob (39 MB)
    std::vector<hipStream_t> ustream;
    void* rccl;                            // dlopen handle (nullptr: peer copies)
    std::vector<srt_ncclComm_t> comm;
    fn_ncclCommDestroy p_destroy; fn_ncclBroadcast p_bcast; fn_ncclGroup p_gstart, p_gend; fn_ncclGetErrorString p_err;
    float* tails; size_t tails_floats;     // page-locked [G-1][planes][3072]
    unsigned long broadcasts;              // RCCL broadcasts issued (srtMultiInfo)
};

static int mfail(int code, const char* fmt, const char* detail = "") { return srt_set_error(code, fmt, detail); }

int srtDeviceCount(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// Executable/main.c:545-575 hands contiguous tile ranges to its threads; SURVEY 8e: rank g gets tiles [g * ceil(N/G), ...)
int srtRankSpan(size_t n, int T, int rank, int world, srt_span* out)
{
    if (!out || T < 1 || world < 1 || rank < 0 || rank >= world) return mfail(-1, "srtRankSpan: bad argument");
    const size_t rows = srtStftRows(n), frames = srtStftFrames(n), ntiles = (rows + T - 1) / T;
    const size_t per = (ntiles + world - 1) / world;
    size_t t0 = (size_t)rank * per, t1 = t0 + per;
    if (t0 > ntiles) t0 = ntiles;
    if (t1 > ntiles) t1 = ntiles;
    const size_t row0 = t0 * T, row1 = t1 * T < rows ? t1 * T : rows;
    const size_t s0 = row0 * SRT_HOP, send = row1 * SRT_HOP + (SRT_FFT - SRT_HOP) < n ? row1 * SRT_HOP + (SRT_FFT - SRT_HOP) : n;
    out->tile0 = t0; out->tile1 = t1; out->sample0 = s0;
    out->nsamples = send > s0 ? send - s0 : 0;
    out->rows = row1 > row0 ? row1 - row0 : 0;
    out->frames = frames > row0 ? (frames - row0 < out->rows ? frames - row0 : out->rows) : 0;
    out->out_offset = s0;
    return 0;
}

static void multi_free(srt_multi* m)
{
    for (size_t g = 0; g < m->eng.size(); ++g) {
        if (m->eng[g]) srtDestroy(m->eng[g]);
        if (g < m->stream.size() && m->stream[g]) { hipSetDevice(m->dev[g]); hipStreamDestroy(m->stream[g]); }
    }
    for (size_t i = 0; i < m->udev.size(); ++i) {
        hipSetDevice(m->udev[i]);
        if (i < m->comm.size() && m->comm[i] && m->p_destroy) m->p_destroy(m->comm[i]);
        if (i < m->blob.size() && m->blob[i]) hipFree(m->blob[i]);
        if (i < m->ustream.size() && m->ustream[i]) hipStreamDestroy(m->ustream[i]);
    }
    if (m->tails) hipHostFree(m->tails);
    if (m->rccl) dlclose(m->rccl);
    delete m;
}

// RCCL clique over the distinct devices (single process: ncclCommInitAll).  Failure is not fatal: the weights then travel by hipMemcpyPeer.
static void multi_init_rccl(srt_multi* m)
{
    const char* off = getenv("SPLEETERRT_NO_RCCL");
    if (off && off[0] == '1') return;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    fn_ncclCommInitAll p_init = (fn_ncclCommInitAll)dlsym(h, "ncclCommInitAll");
    m->p_destroy = (fn_ncclCommDestroy)dlsym(h, "ncclCommDestroy");
    m->p_bcast = (fn_ncclBroadcast)dlsym(h, "ncclBroadcast");
    m->p_gstart = (fn_ncclGroup)dlsym(h, "ncclGroupStart");
    m->p_gend = (fn_ncclGroup)dlsym(h, "ncclGroupEnd");
    m->p_err = (fn_ncclGetErrorString)dlsym(h, "ncclGetErrorString");
    if (!p_init || !m->p_destroy || !m->p_bcast || !m->p_gstart || !m->p_gend) { dlclose(h); return; }
    m->comm.assign(m->udev.size(), nullptr);
    const int rc = p_init(m->comm.data(), (int)m->udev.size(), m->udev.data());
    if (rc != 0) {
        fprintf(stderr, "spleeterrt_amd: ncclCommInitAll over %zu device(s) failed (%s); weights will be copied peer to peer\n", m->udev.size(), m->p_err ? m->p_err(rc) : "?");
        m->comm.clear(); dlclose(h); (void)hipGetLastError();
        return;
    }
    m->rccl = h;
}

int srtMultiCreate(const srt_config* cfg, const int* devices, int ndev, srt_multi** out)
{
    if (!cfg || !out || ndev < 1 || ndev > 64) return mfail(-1, "srtMultiCreate: bad argument");
    const int have = srtDeviceCount();
    if (have == 0) return mfail(-3, "srtMultiCreate: no HIP device (this library has no CPU path)");
    srt_multi* m = new srt_multi();
    m->cfg = *cfg; m->rccl = nullptr; m->tails = nullptr; m->tails_floats = 0; m->broadcasts = 0;
    m->p_destroy = nullptr; m->p_bcast = nullptr; m->p_gstart = m->p_gend = nullptr; m->p_err = nullptr;
    int prev = 0; (void)hipGetDevice(&prev);
    for (int g = 0; g < ndev; ++g) {
        const int d = devices ? devices[g] : g;
        if (d < 0 || d >= have) { multi_free(m); hipSetDevice(prev); return mfail(-1, "srtMultiCreate: device index outside the node's devices"); }
        m->dev.push_back(d);
        size_t u = 0;
        while (u < m->udev.size() && m->udev[u] != d) ++u;
        if (u == m->udev.size()) m->udev.push_back(d);
        m->uidx.push_back((int)u);
    }
    m->eng.assign(ndev, nullptr); m->stream.assign(ndev, nullptr);
    for (int g = 0; g < ndev; ++g) {
        if (hipSetDevice(m->dev[g]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[g], hipStreamNonBlocking) != hipSuccess) { multi_free(m); hipSetDevice(prev); return mfail(-2, "srtMultiCreate: cannot create a stream on a device"); }
        const int rc = srtCreate(cfg, m->stream[g], &m->eng[g]);
        if (rc) { multi_free(m); hipSetDevice(prev); return rc; }
    }
    m->blob.assign(m->udev.size(), nullptr); m->ustream.assign(m->udev.size(), nullptr);
    for (size_t i = 0; i < m->udev.size(); ++i) {
        if (hipSetDevice(m->udev[i]) != hipSuccess || hipMalloc((void**)&m->blob[i], (size_t)SRT_COEFF_STRIDE * 4) != hipSuccess ||
            hipStreamCreateWithFlags(&m->ustream[i], hipStreamNonBlocking) != hipSuccess) { multi_free(m); hipSetDevice(prev); return mfail(-2, "srtMultiCreate: hipMalloc failed"); }
    }
    multi_init_rccl(m);
    hipSetDevice(prev);
    *out = m;
    return 0;
}

void srtMultiDestroy(srt_multi* m)
{
    if (!m) return;
    int prev = 0; (void)hipGetDevice(&prev);
    multi_free(m);
    hipSetDevice(prev);
}

int srtMultiInfo(const srt_multi* m, char* text, size_t bytes)
{
    if (!m || !text || !bytes) return mfail(-1, "srtMultiInfo: bad argument");
    std::string s = "engines=" + std::to_string(m->eng.size()) + " devices=";
    for (size_t g = 0; g < m->dev.size(); ++g) s += (g ? "," : "") + std::to_string(m->dev[g]);
    s += " distinct=" + std::to_string(m->udev.size()) + " weights=" + (m->rccl ? "rccl" : "peer-copy") + " broadcasts=" + std::to_string(m->broadcasts);
    snprintf(text, bytes, "%s", s.c_str());
    return (int)m->eng.size();
}

// blob[0] (root device) -> every distinct device -> every engine
static int multi_distribute(srt_multi* m, int stem, int skip_engine)
{
    const size_t count = SRT_COEFF_FLOATS;
    if (m->rccl) {
        int rc = m->p_gstart();
        for (size_t i = 0; i < m->udev.size() && rc == 0; ++i) {
            hipSetDevice(m->udev[i]);
            rc = m->p_bcast(m->blob[i], m->blob[i], count, SRT_NCCL_FLOAT, 0, m->comm[i], m->ustream[i]);
        }
        const int rc2 = m->p_gend();
        if (rc == 0) rc = rc2;
        if (rc != 0) return mfail(-2, "srtMultiSetCoeff: ncclBroadcast failed: %s", m->p_err ? m->p_err(rc) : "?");
        ++m->broadcasts;
        for (size_t i = 0; i < m->udev.size(); ++i) { hipSetDevice(m->udev[i]); if (hipStreamSynchronize(m->ustream[i]) != hipSuccess) return mfail(-2, "srtMultiSetCoeff: broadcast stream failed"); }
    } else {
        hipSetDevice(m->udev[0]);
        for (size_t i = 1; i < m->udev.size(); ++i)
            if (hipMemcpyPeerAsync(m->blob[i], m->udev[i], m->blob[0], m->udev[0], count * 4, m->ustream[0]) != hipSuccess) return mfail(-2, "srtMultiSetCoeff: hipMemcpyPeer failed");
        if (hipStreamSynchronize(m->ustream[0]) != hipSuccess) return mfail(-2, "srtMultiSetCoeff: peer copy failed");
    }
    for (size_t g = 0; g < m->eng.size(); ++g) {
        if ((int)g == skip_engine) continue;
        hipSetDevice(m->dev[g]);
        const int rc = srtSetCoeffDevice(m->eng[g], stem, m->blob[m->uidx[g]]);
        if (rc) return rc;
        if (hipStreamSynchronize(m->stream[g]) != hipSuccess) return mfail(-2, "srtMultiSetCoeff: stream failed");
    }
    return 0;
}

int srtMultiSetCoeffHost(srt_multi* m, int stem, const void* h_coeff)
{
    if (!m || !h_coeff || stem < 0 || stem >= m->cfg.n_stems) return mfail(-1, "srtMultiSetCoeffHost: bad argument");
    int prev = 0; (void)hipGetDevice(&prev);
    hipSetDevice(m->udev[0]);
    int rc = hipMemcpyAsync(m->blob[0], h_coeff, srtCoeffBytes(), hipMemcpyHostToDevice, m->ustream[0]) == hipSuccess && hipStreamSynchronize(m->ustream[0]) == hipSuccess
                 ? 0 : mfail(-2, "srtMultiSetCoeffHost: upload failed");
    if (!rc) rc = multi_distribute(m, stem, -1);
    hipSetDevice(prev);
    return rc;
}

// fp16 container (spleeterQuantizedSubNet, main.c:423-443): expanded ONCE on the root device by its first engine, then distributed as fp32
int srtMultiSetCoeffFp16Host(srt_multi* m, int stem, const uint16_t* h_halfs)
{
    if (!m || !h_halfs || stem < 0 || stem >= m->cfg.n_stems) return mfail(-1, "srtMultiSetCoeffFp16Host: bad argument");
    int prev = 0; (void)hipGetDevice(&prev);
    hipSetDevice(m->dev[0]);                                    // engine 0 lives on udev[0]
    int rc = srtSetCoeffFp16Host(m->eng[0], stem, h_halfs);
    // (on the engine's own stream and waited for: a device-to-device hipMemcpy on the null stream may return before the copy has run, and the
    //  non-blocking streams that read the staging buffer next do not wait for the null stream)
    if (!rc && (hipMemcpyAsync(m->blob[0], srt_engine_coeff_device(m->eng[0], stem), srtCoeffBytes(), hipMemcpyDeviceToDevice, m->stream[0]) != hipSuccess ||
                hipStreamSynchronize(m->stream[0]) != hipSuccess)) rc = mfail(-2, "srtMultiSetCoeffFp16Host: copy failed");
    if (!rc) rc = multi_distribute(m, stem, 0);
    hipSetDevice(prev);
    return rc;
}

// cli_stems 0: the engine's n_stems sub-networks on the same input (srtSeparateHostStream); 2 | 3: the CLI's flows (srtSeparateCliHost)
static int multi_run(srt_multi* m, const float* h_L, const float* h_R, size_t n, float* h_out, unsigned flags, int cli_stems)
{
    if (!m || !h_L || !h_R || !h_out) return mfail(-1, "srtMultiSeparate: null argument");
    if (n < SRT_FFT) return mfail(-1, "srtMultiSeparate: need at least 4096 samples");
    if (cli_stems && cli_stems != 2 && cli_stems != 3) return mfail(-1, "srtMultiSeparateCliHost: stems must be 2 or 3");
    const int G = (int)m->eng.size(), T = m->cfg.T, NP = 2 * (cli_stems ? cli_stems : m->cfg.n_stems);
    const size_t rows = srtStftRows(n), total_len = srtIstftLength(rows), tail = SRT_FFT - SRT_HOP;
    std::vector<srt_span> sp(G);
    for (int g = 0; g < G; ++g) srtRankSpan(n, T, g, G, &sp[g]);                  // (a range longer than max_tiles tiles is walked chunk by chunk inside its engine)
    int prev = 0; (void)hipGetDevice(&prev);
    const size_t need = (size_t)(G > 1 ? G - 1 : 1) * NP * tail;
    if (need > m->tails_floats) {
        if (m->tails) hipHostFree(m->tails);
        m->tails = nullptr; m->tails_floats = 0;
        if (hipHostMalloc((void**)&m->tails, need * sizeof(float), hipHostMallocPortable) != hipSuccess) { hipSetDevice(prev); return mfail(-2, "srtMultiSeparate: page-locked allocation failed"); }
        m->tails_floats = need;
    }
    // page-lock the caller's buffers once for all devices (each engine would otherwise register its own slice per call)
    bool pinL = false, pinR = false, pinO = false;
    if (!(flags & SRT_HOST_PINNED)) {
        pinL = hipHostRegister((void*)h_L, n * sizeof(float), hipHostRegisterPortable) == hipSuccess;
        pinR = hipHostRegister((void*)h_R, n * sizeof(float), hipHostRegisterPortable) == hipSuccess;
        pinO = hipHostRegister((void*)h_out, (size_t)NP * total_len * sizeof(float), hipHostRegisterPortable) == hipSuccess;
        (void)hipGetLastError();
    }
    std::vector<int> rc(G, 0);
    std::vector<std::string> err(G);
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) {
        if (!sp[g].rows) continue;
        th.emplace_back([&, g]() {                              // one worker per device, as processMT's pthread_create per tile range
            hipSetDevice(m->dev[g]);
            const bool next = g + 1 < G && sp[g + 1].rows > 0;
            rc[g] = srt_engine_host_range(m->eng[g], h_L + sp[g].sample0, h_R + sp[g].sample0, sp[g].nsamples, sp[g].frames, sp[g].rows,
                                          h_out + sp[g].out_offset, total_len, next ? m->tails + (size_t)g * NP * tail : nullptr, g > 0,
                                          SRT_HOST_PINNED, cli_stems);
            if (rc[g]) err[g] = srtLastError();
        });
    }
    for (auto& t : th) t.join();
    if (pinL) hipHostUnregister((void*)h_L);
    if (pinR) hipHostUnregister((void*)h_R);
    if (pinO) hipHostUnregister((void*)h_out);
    hipSetDevice(prev);
    for (int g = 0; g < G; ++g) if (rc[g]) return mfail(rc[g], "%s", err[g].c_str());
    // join: the seams (the reference joins its threads and is done - its tiles do not overlap in the spectrogram; the overlap-add of the last
    // three frames of a range into the next range's first 3072 samples is what a per-range iSTFT adds)
    for (int g = 1; g < G; ++g) {
        if (!sp[g].rows) break;
        const float* tl = m->tails + (size_t)(g - 1) * NP * tail;
        const size_t s0 = sp[g].out_offset;
        for (int pl = 0; pl < NP; ++pl) {
            float* o = h_out + (size_t)pl * total_len + s0;
            for (size_t i = 0; i < tail; ++i) o[i] += tl[(size_t)pl * tail + i];
        }
        if (cli_stems == 2) {                                   // accompaniment = input - vocal (main.c:794-798)
            for (int ch = 0; ch < 2; ++ch) {
                const float* in = ch ? h_R : h_L;
                const float* v = h_out + (size_t)ch * total_len; float* a = h_out + (size_t)(2 + ch) * total_len;
                for (size_t j = s0; j < s0 + tail; ++j) a[j] = (j < n ? in[j] : 0.0f) - v[j];
            }
        } else if (cli_stems == 3) {                            // accompaniment = istft(residual) - vocal (main.c:924-928)
            for (int ch = 0; ch < 2; ++ch) {
                const float* v = h_out + (size_t)(2 + ch) * total_len; float* a = h_out + (size_t)(4 + ch) * total_len;
                for (size_t j = s0; j < s0 + tail; ++j) a[j] = a[j] - v[j];
            }
        }
    }
    return 0;
}

int srtMultiSeparateHost(srt_multi* m, const float* h_L, const float* h_R, size_t n, float* h_out, unsigned flags)
{
    return multi_run(m, h_L, h_R, n, h_out, flags, 0);
}

int srtMultiSeparateCliHost(srt_multi* m, const float* h_L, const float* h_R, size_t n, int stems, float* h_out)
{
    if (stems != 2 && stems != 3) return mfail(-1, "srtMultiSeparateCliHost: stems must be 2 or 3");
    return multi_run(m, h_L, h_R, n, h_out, 0, stems);
}

srt_engine* srtMultiEngine(srt_multi* m, int g)
{
    return (m && g >= 0 && g < (int)m->eng.size()) ? m->eng[g] : nullptr;
}

// ---- resident throughput on every engine at once (bench.py --host native): the C host of main.c:544-673's shape - one process, one worker
// thread + engine per device - timed the way the per-process bench times its ranks: warm-up, a barrier over the workers, `steps` passes of
// the whole path (srtSeparate, PCM and stems resident in each device's HBM), device synchronised, barrier; the clock runs from the first
// worker leaving the first barrier to the last one finishing.
namespace {
struct WorkerBarrier {
    std::mutex mu; std::condition_variable cv; int n, waiting = 0; unsigned long gen = 0;
    explicit WorkerBarrier(int n_) : n(n_) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};
}

int srtMultiBenchResident(srt_multi* m, int tiles, int steps, int warmup, double* seconds, double* seconds_events)
{
    if (!m || !seconds || tiles < 1 || tiles > m->cfg.max_tiles || steps < 1 || warmup < 0) return mfail(-1, "srtMultiBenchResident: bad argument");
    const int G = (int)m->eng.size(), NP = 2 * m->cfg.n_stems;
    const size_t n = (size_t)tiles * m->cfg.T * SRT_HOP, olen = srtIstftLength(srtStftRows(n));
    typedef std::chrono::steady_clock clk;
    std::vector<int> rc(G, 0);
    std::vector<std::string> err(G);
    std::vector<clk::time_point> t0(G), t1(G), e0(G), e1(G);
    WorkerBarrier bar(G);
    int prev = 0; (void)hipGetDevice(&prev);
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) {
        th.emplace_back([&, g]() {
            hipSetDevice(m->dev[g]);
            float *dL = nullptr, *dR = nullptr, *dO = nullptr;
            std::vector<float> h(2 * n);
            uint32_t st = 777u + (uint32_t)g;                       // SURVEY 8d's LCG, white noise of +-0.1
            for (size_t i = 0; i < 2 * n; ++i) { st = st * 1664525u + 1013904223u; h[i] = 0.2f * ((float)(st >> 8) * (1.0f / 16777216.0f) - 0.5f); }
            bool ok = hipMalloc((void**)&dL, n * 4) == hipSuccess && hipMalloc((void**)&dR, n * 4) == hipSuccess && hipMalloc((void**)&dO, (size_t)NP * olen * 4) == hipSuccess &&
                      hipMemcpy(dL, h.data(), n * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dR, h.data() + n, n * 4, hipMemcpyHostToDevice) == hipSuccess;
            if (!ok) { rc[g] = -2; err[g] = "srtMultiBenchResident: device allocation / upload failed"; }
            auto pass = [&](int k) { for (int i = 0; i < k && !rc[g]; ++i) if ((rc[g] = srtSeparate(m->eng[g], dL, dR, n, dO)) != 0) err[g] = srtLastError(); };
            auto drain = [&]() { if (!rc[g] && hipStreamSynchronize(m->stream[g]) != hipSuccess) { rc[g] = -2; err[g] = "srtMultiBenchResident: stream failed"; } };
            pass(warmup); drain();
            bar.wait(); t0[g] = clk::now();
            pass(steps); drain();
            t1[g] = clk::now(); bar.wait();
            if (seconds_events) {                                   // the same K passes with HIP events around every launch of engine 0 (srtGetTiming)
                if (g == 0 && !rc[g]) srtSetTiming(m->eng[0], 1);
                bar.wait(); e0[g] = clk::now();
                pass(steps); drain();
                e1[g] = clk::now(); bar.wait();
            }
            if (dL) hipFree(dL);
            if (dR) hipFree(dR);
            if (dO) hipFree(dO);
        });
    }
    for (auto& t : th) t.join();
    hipSetDevice(prev);
    for (int g = 0; g < G; ++g) if (rc[g]) return mfail(rc[g], "%s", err[g].c_str());
    auto span = [&](std::vector<clk::time_point>& a, std::vector<clk::time_point>& b) {
        clk::time_point lo = a[0], hi = b[0];
        for (int g = 1; g < G; ++g) { if (a[g] < lo) lo = a[g]; if (b[g] > hi) hi = b[g]; }
        return std::chrono::duration<double>(hi - lo).count();
    };
    *seconds = span(t0, t1);
    if (seconds_events) *seconds_events = span(e0, e1);
    return 0;
}
