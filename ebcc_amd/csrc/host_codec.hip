// host_codec.hip - the reference's C API (include/ebcc_codec.h) and HDF5 filter glue on top of the
// MI355X engine: frame-codec orchestration (/root/reference/src/ebcc_codec.c:607-918 encode,
// :1215-1320 decode), the EBCK chunk container (:920-1090, :1322-1449) and the filter plugin
// (/root/reference/src/h5z_ebcc.c).  The host only steers: per-frame scalars come back from the device
// after every probe, every per-sample operation runs in the kernels.  There is no CPU fallback: without
// a HIP device every entry point fails loudly.
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <ctime>
#include <sched.h>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <tuple>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <memory>
#include <thread>
#include <string>
#include <vector>

#include "../../include/ebcc_hip.h"
#include "engine.hpp"
#include "j2k.hpp"
#include "search.hpp"

namespace ebcc {
bool j2k_parse_codestream(const uint8_t *cs, size_t n, const J2kGeom &g, int *table);
bool j2k_peek_dims(const uint8_t *cs, size_t n, int *W, int *H, int *tile_w, int *tile_h);
bool j2k_parse_tiled(const uint8_t *cs, size_t n, const J2kBuffers &jb, int tiles, int *tables, size_t *part_off, size_t *part_len);
}
using namespace ebcc;

// ================================================================================================
// logging (reference src/log/, level from EBCC_LOG_LEVEL, default WARN; src/ebcc_codec.c:431-448)
// ================================================================================================
namespace {
int g_log_level = 3;
const char *kLevelNames[] = {"TRACE", "DEBUG", "INFO", "WARN", "ERROR", "FATAL"};
void log_at(int level, const char *fmt, ...)
{
    if (level < g_log_level) return;
    char tb[16];
    time_t t = time(nullptr);
    struct tm lt;
    localtime_r(&t, &lt);
    strftime(tb, sizeof tb, "%H:%M:%S", &lt);
    fprintf(stderr, "%s %-5s ebcc-mi355x: ", tb, kLevelNames[level]);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
#define log_trace(...) log_at(0, __VA_ARGS__)
#define log_info(...) log_at(2, __VA_ARGS__)
#define log_warn(...) log_at(3, __VA_ARGS__)
#define log_fatal(...) log_at(5, __VA_ARGS__)

// ================================================================================================
// zstd stays on the host (north star); dlopen'd so the library has no link-time dependency
// ================================================================================================
struct Zstd {
    size_t (*bound)(size_t) = nullptr;
    size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
    size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    unsigned (*version)(void) = nullptr;
    bool ok = false;
    Zstd()
    {
        const char *names[] = {"/opt/conda/lib/libzstd.so.1", "libzstd.so.1", "libzstd.so", nullptr};
        void *h = nullptr;
        for (int i = 0; names[i] && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);   // DEEPBIND: never mix with another zstd already in the process
        if (!h) return;
        bound = (size_t(*)(size_t)) dlsym(h, "ZSTD_compressBound");
        compress = (size_t(*)(void *, size_t, const void *, size_t, int)) dlsym(h, "ZSTD_compress");
        decompress = (size_t(*)(void *, size_t, const void *, size_t)) dlsym(h, "ZSTD_decompress");
        is_error = (unsigned (*)(size_t)) dlsym(h, "ZSTD_isError");
        version = (unsigned (*)(void)) dlsym(h, "ZSTD_versionNumber");
        ok = bound && compress && decompress;
    }
};
Zstd &zstd()
{
    static Zstd z;
    return z;
}

// ------------------------------------------------------------------------------------------------
// A lower bound on the size of the zstd frame ZSTD_compress writes for [src, src + n), at any level.
//
// The reference compresses the kept SPIHT prefix at level 22 (:813-817) and then compares its size z with the pure
// base-layer alternative (:838, len2 < z + len1).  Whenever a bound z >= F already gives len2 < F + len1 the comparison is
// decided without z, and the (discarded) compression with it.  The bound, from the format alone (RFC 8878):
//   * a frame is >= 9 bytes of magic, frame header and one block header around its blocks;
//   * a block regenerates its bytes from literals and matches; a match copies >= 3 bytes (Match_Length code 0 = 3) that
//     occurred EARLIER in the regenerated data (offsets are positive; no dictionary), so a byte at position i can only be
//     part of a match if one of the three-byte windows [q, q + 3), q in {i - 2, i - 1, i}, repeats a three-byte string that
//     starts before q.  All other positions ("uncoverable") are literals of their block in every valid frame;
//   * the literals of a block are stored raw (8 bits each), as one repeated byte (only if they are all equal) or under
//     one prefix code per block (Huffman, at most 11 bits - still a prefix code; a first block cannot reuse a table), so
//     they cost at least their empirical entropy m log2 m - sum_s c_s log2 c_s, and that function only grows when further
//     literals join the multiset: the uncoverable positions alone bound it from below;
//   * libzstd before 1.5 cuts the input into blocks of min(128 KB, window) bytes and nothing finer (no block splitter,
//     no target block size unless asked for): the bound is taken block by block at those boundaries (one block up to
//     128 KB) - zstd_floor_usable() checks the library's version, tests/test_zstd_floor.py checks the block structure and
//     the bound itself against the library on the fixtures and on random material.
// Cost: one pass with a 2^24-bit table of the three-byte strings seen, a few microseconds per KB.
// ------------------------------------------------------------------------------------------------
constexpr size_t kZstdBlockBytes = (size_t) 128 << 10;                  // ZSTD_BLOCKSIZE_MAX: libzstd < 1.5 cuts longer inputs into blocks of this size
constexpr size_t kZstdFloorMaxBytes = (size_t) 4 << 20;
bool zstd_floor_usable()
{
    return zstd().ok && zstd().version && zstd().version() < 10500;
}
size_t zstd_size_lower_bound(const uint8_t *src, size_t n)
{
    if (n < 8 || n > kZstdFloorMaxBytes) return 0;
    thread_local std::vector<uint64_t> seen;                             // one bit per three-byte string
    thread_local std::vector<uint8_t> cov;
    if (seen.empty()) seen.assign((size_t) 1 << 18, 0);
    cov.assign(n, 0);
    auto tri = [&](size_t q) { return ((uint32_t) src[q] << 16) | ((uint32_t) src[q + 1] << 8) | (uint32_t) src[q + 2]; };
    // (matches reach back across block boundaries - the window holds the whole input - so the strings seen are kept for
    //  the whole input; a match itself lies inside one block, which only makes fewer positions coverable than counted here)
    for (size_t q = 0; q + 3 <= n; q++) {
        const uint32_t t = tri(q);
        uint64_t &w = seen[t >> 6];
        const uint64_t bit = 1ull << (t & 63);
        if (w & bit) { cov[q] = cov[q + 1] = cov[q + 2] = 1; } else w |= bit;
    }
    for (size_t q = 0; q + 3 <= n; q++) { const uint32_t t = tri(q); seen[t >> 6] = 0; }     // (leave the table clean for the next call)
    // the literals of every block under that block's own prefix code: the entropy of its uncoverable bytes
    double bits_total = 0;
    size_t blocks = 0;
    for (size_t b0 = 0; b0 < n; b0 += kZstdBlockBytes, blocks++) {
        const size_t b1 = std::min(n, b0 + kZstdBlockBytes);
        size_t cnt[256] = {0}, m = 0;
        for (size_t i = b0; i < b1; i++) if (!cov[i]) { cnt[src[i]]++; m++; }
        if (m == 0) continue;
        double bits = (double) m * std::log2((double) m);
        for (size_t c : cnt) if (c) bits -= (double) c * std::log2((double) c);
        if (bits > 0) bits_total += bits;
    }
    // magic, frame header, a 3-byte header per block; a byte less per block than the arithmetic gives (rounded logarithms)
    const double bytes = std::floor(bits_total / 8.0) - (double) blocks;
    return 6 + 3 * blocks + (bytes > 0 ? (size_t) bytes : 0);
}

// ================================================================================================
// stream headers (src/ebcc_codec.c:190-213)
// ================================================================================================
#pragma pack(push, 1)
struct FrameHeader {
    uint8_t magic[4]; uint8_t version; uint8_t flags; uint16_t reserved;
    uint32_t minval_bits, maxval_bits; uint64_t coeffs_size;
    uint32_t rmin_bits, rmax_bits; uint64_t compressed_size; uint64_t tail_size;
};
struct ChunkHeader {
    uint8_t magic[4]; uint32_t version, ndims, reserved;
    uint64_t dims[3], chunk_dims[3], num_chunks, chunk_size;
};
#pragma pack(pop)
static_assert(sizeof(FrameHeader) == 48, "EBCC header must be 48 bytes");
static_assert(sizeof(ChunkHeader) == 80, "EBCK header must be 80 bytes");
uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

bool dims_are_valid(const size_t d[3])
{
    // :286-297
    if (d[0] == 0 || d[1] == 0) return false;
    size_t hh = d[0] * d[1];
    if (hh / d[0] != d[1]) return false;
    return hh >= EBCC_MIN_INTERNAL_IMAGE_DIM && hh <= EBCC_MAX_INTERNAL_IMAGE_DIM && d[2] >= EBCC_MIN_INTERNAL_IMAGE_DIM &&
           d[2] <= EBCC_MAX_INTERNAL_IMAGE_DIM;
}

// ================================================================================================
// rate search of src/ebcc_codec.c:545-596 as a resumable state machine (one probe per step)
// ================================================================================================
struct RateSearch {
    float lo = 0, hi = 0, cr = 0, result = 0;
    double q = 0, q0 = 0, qt = 0;
    int phase = 4;           // 0 halving, 1 doubling, 2 bisect, 3 final probe, 4 done
    float pending = 0;
    void start(float cr0, double q_init, double q_target)
    {
        lo = hi = cr = cr0; q = q0 = q_init; qt = q_target; phase = 0;
    }
    bool done() const { return phase == 4; }
    // returns true and sets `out` if a probe at rate `out` is needed next
    bool next(float &out)
    {
        for (;;) {
            if (phase == 0) {
                if (q < qt && lo >= 1. / 2) { lo /= 2; out = pending = lo; return true; }      // :559-563
                q = q0; phase = 1;
            } else if (phase == 1) {
                if (q >= qt && hi <= 1000) { hi *= 2; out = pending = hi; return true; }       // :565-569
                if (q >= qt) { result = hi; phase = 4; return false; }                         // :571-574
                q = q0; phase = 2;
            } else if (phase == 2) {
                const double eps = 1e-8;
                if ((std::fabs(q - qt) > eps || q == 1.0) && hi - lo > 1.) {                   // :579-588
                    cr = (lo + hi) / 2; out = pending = cr; return true;
                }
                phase = 3; out = pending = lo; return true;                                    // :590
            } else {
                return false;
            }
        }
    }
    void feed(double quantile)
    {
        q = quantile;
        if (phase == 2) { if (q < qt) hi = cr; else lo = cr; }
        else if (phase == 3) { result = lo; phase = 4; }
    }
};

// ================================================================================================
// devices and the context cache: one engine per (device, frame geometry), grown on demand
// ================================================================================================
// The reference-compatible entry points have no device argument.  They run on
//   EBCC_HIP_DEVICE=<n>          if set, else on the calling thread's current HIP device (what torch.cuda.set_device or
//                                hipSetDevice chose; 0 in a process that never chose), and
//   EBCC_HIP_DEVICES=all|a,b,..  lets the chunking entry points spread their chunk list over several devices (default:
//                                all visible devices in a stand-alone process, the one device above when the process is
//                                one rank of a multi-process job - LOCAL_WORLD_SIZE / WORLD_SIZE > 1).
// Every entry point makes its device current for the call and restores the caller's on return.
int resolve_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return 0;         // (create_engine reports the missing device)
    if (const char *e = getenv("EBCC_HIP_DEVICE")) { int d = atoi(e); return d >= 0 && d < n ? d : 0; }
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return d;
}
bool multi_process_job()
{
    for (const char *v : {"LOCAL_WORLD_SIZE", "WORLD_SIZE"})
        if (const char *e = getenv(v)) if (atoi(e) > 1) return true;
    return false;
}
std::vector<int> device_list()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return {0};
    const char *e = getenv("EBCC_HIP_DEVICES");
    std::vector<int> out;
    if (e && strcmp(e, "all") != 0) {
        for (const char *p = e; *p;) {
            char *end;
            long d = strtol(p, &end, 10);
            if (end == p) break;
            // (a device named twice counts once - unless EBCC_HIP_DEVICES_KEEP_REPEATS=1, the tests' way to drive the
            //  several-devices path of run_on_devices on a one-GPU box: the per-device lock then serialises the blocks)
            static const bool keep = getenv("EBCC_HIP_DEVICES_KEEP_REPEATS") != nullptr;
            if (d >= 0 && d < n && (keep || std::find(out.begin(), out.end(), (int) d) == out.end())) out.push_back((int) d);
            p = *end == ',' ? end + 1 : end;
        }
    } else if (e || !multi_process_job()) {
        for (int d = 0; d < n; d++) out.push_back(d);
    }
    if (out.empty()) out.push_back(resolve_device());
    return out;
}
struct DeviceScope {               // the engine's device for the duration of a call, the caller's afterwards
    int prev = -1;
    explicit DeviceScope(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; EBCC_HIP_CHECK(hipSetDevice(dev)); }
    ~DeviceScope() { if (prev >= 0) hipSetDevice(prev); }
};
// One lock per device (HDF5 serialises filter calls anyway; a multi-threaded writer gets one call per device at a time).
std::mutex &device_mutex(int dev)
{
    static std::mutex m[64];
    return m[dev & 63];
}
std::mutex g_map_mutex;
std::map<std::tuple<int, int, int, int>, ebcc_hip_ctx *> g_ctx;

// `period` > 1: the frames are the tiles of images of that many tiles each, every tile position with its own
// JPEG 2000 geometry (j2k.hpp).  Called with the device's lock held and the device current.
ebcc_hip_ctx *get_context(int device, int H, int W, size_t frames, int period = 1)
{
    std::lock_guard<std::mutex> lock(g_map_mutex);
    auto key = std::make_tuple(device, H, W, period);
    auto it = g_ctx.find(key);
    if (it != g_ctx.end() && it->second->max_frames >= frames) return it->second;
    if (it != g_ctx.end()) { ebcc_hip_destroy(it->second); g_ctx.erase(it); }
    ebcc_hip_ctx *c = create_engine(device, frames, (size_t) H, (size_t) W, period);
    if (!c) {                                                   // out of device memory: drop this device's engines of other geometries
        bool dropped = false;
        for (auto i = g_ctx.begin(); i != g_ctx.end();)
            if (std::get<0>(i->first) == device) { ebcc_hip_destroy(i->second); i = g_ctx.erase(i); dropped = true; } else ++i;
        if (dropped) c = create_engine(device, frames, (size_t) H, (size_t) W, period);
    }
    if (c) g_ctx[key] = c;
    return c;
}

// the device image of a host array handed to the reference API: kept in the context between calls
float *io_buffer(ebcc_hip_ctx *ctx, size_t bytes)
{
    if (ctx->io_cap >= bytes) return ctx->d_io;
    if (ctx->d_io) { hipFree(ctx->d_io); ctx->d_io = nullptr; ctx->io_cap = 0; }
    void *p = nullptr;
    hipError_t e = device_malloc(&p, bytes);
    if (e != hipSuccess) { char b[128]; snprintf(b, sizeof b, "device buffer of %zu bytes: %s", bytes, hipGetErrorString(e)); throw HipFailure(b); }
    ctx->d_io = (float *) p; ctx->io_cap = bytes;
    return ctx->d_io;
}

// A pageable host array <-> the device image, through two pinned buffers: the DMA engine fills (or drains) one while a few
// host threads copy the other to (or from) the caller's memory.  hipMemcpy on pageable memory stages through one internal
// buffer on one thread: ~10 GB/s, 100 ms for the 1.06 GB of a 256-frame batch - four times the decode itself.
constexpr size_t kBounceBytes = (size_t) 32 << 20;
static void host_copy_parallel(void *dst, const void *src, size_t bytes)
{
    const size_t nthreads = std::min<size_t>(8, std::max<size_t>(1, bytes >> 20));
    std::vector<std::thread> pool;
    size_t started = 1;
    try {
        for (size_t t = 1; t < nthreads; t++, started++)
            pool.emplace_back([=]() { const size_t lo = bytes / nthreads * t, hi = t + 1 == nthreads ? bytes : bytes / nthreads * (t + 1);
                                      memcpy((char *) dst + lo, (const char *) src + lo, hi - lo); });
    } catch (const std::exception &) {}                                 // (thread limit: this thread copies what is left)
    memcpy(dst, src, bytes / nthreads);
    if (started < nthreads) { const size_t lo = bytes / nthreads * started; memcpy((char *) dst + lo, (const char *) src + lo, bytes - lo); }
    for (auto &t : pool) t.join();
}
static void copy_pageable(ebcc_hip_ctx *ctx, void *host, void *dev, size_t bytes, bool to_host)
{
    if (bytes < 2 * kBounceBytes) {
        EBCC_HIP_CHECK(hipMemcpy(to_host ? host : dev, to_host ? dev : host, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice));
        return;
    }
    if (!ctx->h_bounce) EBCC_HIP_CHECK(hipHostMalloc((void **) &ctx->h_bounce, 2 * kBounceBytes));
    hipStream_t s = ctx->stream;
    const size_t chunks = (bytes + kBounceBytes - 1) / kBounceBytes;
    auto len = [&](size_t i) { return std::min(kBounceBytes, bytes - i * kBounceBytes); };
    if (to_host) {
        EBCC_HIP_CHECK(hipMemcpyAsync(ctx->h_bounce, dev, len(0), hipMemcpyDeviceToHost, s));
        for (size_t i = 0; i < chunks; i++) {
            wait_stream(s);                                       // chunk i has arrived
            if (i + 1 < chunks)
                EBCC_HIP_CHECK(hipMemcpyAsync(ctx->h_bounce + ((i + 1) & 1) * kBounceBytes, (char *) dev + (i + 1) * kBounceBytes, len(i + 1), hipMemcpyDeviceToHost, s));
            host_copy_parallel((char *) host + i * kBounceBytes, ctx->h_bounce + (i & 1) * kBounceBytes, len(i));
        }
    } else {
        for (size_t i = 0; i < chunks; i++) {
            host_copy_parallel(ctx->h_bounce + (i & 1) * kBounceBytes, (const char *) host + i * kBounceBytes, len(i));
            if (i >= 1) wait_stream(s);                           // (chunk i - 1 has left: its buffer is filled next)
            EBCC_HIP_CHECK(hipMemcpyAsync((char *) dev + i * kBounceBytes, ctx->h_bounce + (i & 1) * kBounceBytes, len(i), hipMemcpyHostToDevice, s));
        }
        wait_stream(s);
    }
}

// Frames per device batch of the host-pointer entry points: EBCC_HIP_MAX_BATCH (default 256), reduced for large
// frames so that an engine's workspace (about 160 bytes per pixel and frame with the worst-case slots) stays
// under ~48 GB.
size_t batch_capacity(size_t n_pix)
{
    const char *e = getenv("EBCC_HIP_MAX_BATCH");
    size_t v = e ? strtoul(e, nullptr, 10) : 256;
    if (!v) v = 256;
    const size_t fit = ((size_t) 48 << 30) / (n_pix * 160 + 1);
    return std::max<size_t>(1, std::min(v, fit));
}

struct EncodeEnv {
    double base_error_quantile = 1e-6;
    bool no_fallback = false, no_consistency = false, no_mean_adjust = false;
    int zstd_level = 22;
    EncodeEnv()
    {
        // :634-649
        if (const char *e = getenv("EBCC_INIT_BASE_ERROR_QUANTILE")) base_error_quantile = strtod(e, nullptr);
        no_fallback = getenv("EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK") != nullptr;
        no_consistency = getenv("EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK_CONSISTENCY") != nullptr;
        no_mean_adjust = getenv("EBCC_DISABLE_MEAN_ADJUSTMENT") != nullptr;
        // Level of the residual's zstd stage (:816 uses 22).  Any level gives streams every EBCC decoder reads, but
        // only 22 reproduces the reference's bytes, so this is an opt-in knob (SURVEY section 8(f) n3 study).
        if (const char *e = getenv("EBCC_ZSTD_LEVEL")) { long v = strtol(e, nullptr, 10); if (v >= 1 && v <= 22) zstd_level = (int) v; }
    }
};

// wall-clock phase report on stderr when EBCC_HIP_PHASE_TIMING is set (diagnostics only)
struct PhaseTimer {
    bool on = getenv("EBCC_HIP_PHASE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char *what)
    {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "ebcc-mi355x phase %-28s %9.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// Hand-over point between the slices of a batch (run_slices): slice i + 1 starts when slice i has issued its
// heavy first stage, so that the slices run out of phase and the GPU-bound stage of one overlaps the host- or
// latency-bound stages of the other.
struct SliceGate {
    std::mutex m;
    std::condition_variable cv;
    bool open = false;
    void release() { { std::lock_guard<std::mutex> l(m); open = true; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return open; }); }
};

constexpr int kJ2kMainHeaderBytes = 135;     // SOC, SIZ, COD, QCD, COM of every codestream the codec writes

struct ProbeRec { float cr = -1; unsigned long long nbad = 0; int stream_bytes = 0; double err_sum = 0; bool complete = true; };

struct Job {                     // host-side state of one frame being encoded
    bool const_field = false;
    float minv = 0, maxv = 0, target = -1, cr = -1;
    double mean_err = 0, q = 0, q_first = 0;
    float rmin = 0, rmax = 0;
    bool skip = true, need_pure = false;
    float best_err = -1;
    size_t coeffs_orig = 0, coeffs_size = 0, len1 = 0;
    double t_hi = 0, t_lo = 0, t_best = 0;
    bool trunc_active = false;
    std::vector<uint8_t> tail, zbytes;
    // rate searches: [0] error-bounded (:728), [1] pure base layer (:836).  A probe's outcome depends only on
    // (frame, rate), so both searches share one record of the probes made so far.
    RateSearch rs[2];
    bool want[2] = {false, false};    // search k waits for the probe at want_cr[k]
    float want_cr[2] = {0, 0};
    ProbeRec last[2];                 // the final probe of search k (phase 3)
    std::vector<ProbeRec> probes;
    const ProbeRec *find_probe(float cr) const
    {
        for (const ProbeRec &r : probes) if (r.cr == cr) return &r;
        return nullptr;
    }
};

// The engine's second stream is created on first use: every stream beyond the runtime's few hardware queues
// shares one, and kernels that share a queue run one after the other.
hipStream_t second_stream(ebcc_hip_ctx *c)
{
    if (!c->stream2) EBCC_HIP_CHECK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    return c->stream2;
}

// CPUs this process may really use: the affinity mask, cut down to the container's CPU quota where one is set (cgroup v2
// cpu.max "quota period", cgroup v1 cpu.cfs_quota_us / cpu.cfs_period_us).  The MI355X box of this project is a 16-CPU
// quota on a 256-thread host: the mask says 256, and a pool sized from it bursts into the quota, gets the whole cgroup
// throttled for the rest of the 100 ms period - the threads that steer the GPU included.
static double cgroup_cpu_quota()
{
    auto read_two = [](const char *path, long long &a, long long &b) {
        FILE *f = fopen(path, "r");
        if (!f) return false;
        char tok[64];
        bool ok = fscanf(f, "%63s %lld", tok, &b) == 2;
        fclose(f);
        if (!ok || !strcmp(tok, "max")) return false;
        a = atoll(tok);
        return a > 0 && b > 0;
    };
    auto read_one = [](const char *path, long long &v) {
        FILE *f = fopen(path, "r");
        if (!f) return false;
        bool ok = fscanf(f, "%lld", &v) == 1;
        fclose(f);
        return ok;
    };
    if (const char *e = getenv("EBCC_HOST_CPU_QUOTA")) return std::max(0.0, strtod(e, nullptr));    // (containers that hide their cgroup; tests)
    long long q = 0, per = 0;
    if (read_two("/sys/fs/cgroup/cpu.max", q, per)) return (double) q / (double) per;
    for (const char *dir : {"/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"}) {
        char a[128], b[128];
        snprintf(a, sizeof a, "%s/cpu.cfs_quota_us", dir);
        snprintf(b, sizeof b, "%s/cpu.cfs_period_us", dir);
        if (read_one(a, q) && read_one(b, per) && q > 0 && per > 0) return (double) q / (double) per;
    }
    return 0;                                                   // no quota
}
static unsigned affinity_cpus()
{
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned) CPU_COUNT(&set);
    return std::max(1u, n);
}
unsigned usable_cpus()
{
    static const unsigned cached = [] {
        unsigned n = affinity_cpus();
        const double q = cgroup_cpu_quota();
        if (q > 0) n = std::min(n, (unsigned) std::max(1.0, std::floor(q + 0.5)));
        return std::max(1u, n);
    }();
    return cached;
}
// Width of the pool.  A quota of Q CPUs is Q x 100 ms of CPU time per 100 ms period, not a limit on how many threads run at
// once: work that comes in bursts - the entropy stage, once per slice - may run wider than Q as long as a period's total
// stays below the quota, and finishes sooner for it.  Round 2 ran 64 threads into the 16-CPU quota of the MI355X box with
// 1.3-1.9 core-seconds of zstd per step: throttled in every second period (cpu.stat), the steering threads with it -
// the "two timing modes".  Since the encoder only compresses the prefixes whose size can matter, a step needs 0.6-0.9
// core-seconds, and a burst TWICE the quota wide stays clear of it (tools/gpu/host_sweep.sh: 32 threads 159-167 ms per
// step and no throttled period, 15 threads 172-177, 8 threads 197).  So: min(affinity, 2 x quota) divided by the ranks
// that share the host (LOCAL_WORLD_SIZE), minus the threads that steer the GPU; EBCC_HOST_THREADS overrides.  The pool is
// per process and shared by the slices of every call (HostPool).
unsigned entropy_threads_for(unsigned cpus, unsigned local_world, unsigned slices)
{
    const unsigned share = std::max(1u, cpus / std::max(1u, local_world));
    const unsigned steer = std::min(slices, share > 4 ? 2u : 0u);
    return std::max(1u, std::min(64u, share - steer));
}
static unsigned burst_cpus()
{
    static const unsigned cached = [] {
        unsigned n = affinity_cpus();
        const double q = cgroup_cpu_quota();
        if (q > 0) n = std::min(n, (unsigned) std::max(1.0, std::floor(2.0 * q + 0.5)));
        return std::max(1u, n);
    }();
    return cached;
}
unsigned entropy_threads(unsigned slices = 1)
{
    if (const char *e = getenv("EBCC_HOST_THREADS")) return (unsigned) std::max(1L, strtol(e, nullptr, 10));
    unsigned lws = 1;
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) lws = (unsigned) std::max(1, atoi(e));
    return entropy_threads_for(burst_cpus(), lws, slices);
}

// Host-side accounting of the entropy stage since the last reset (ebcc_hip_host_stats: bench.py prints it per rank so that
// a multi-GPU run that is bound by the host's CPUs can be told from one that is bound by the GPUs).
struct HostStats {
    std::atomic<long long> zstd_core_us{0}, zstd_wait_us{0}, zstd_bytes{0}, batches{0}, skipped_bytes{0};
    void add(long long core_us, long long wait_us, long long bytes) { zstd_core_us += core_us; zstd_wait_us += wait_us; zstd_bytes += bytes; batches++; }
    void reset() { zstd_core_us = 0; zstd_wait_us = 0; zstd_bytes = 0; batches = 0; skipped_bytes = 0; }
};
HostStats &host_stats() { static HostStats h; return h; }

// ------------------------------------------------------------------------------------------------
// HostPool: the process-wide worker threads of the host-side stages (level-22 zstd of the residual prefixes, frame
// parsing and zstd decompression of the decode).  Every slice of every call on every device feeds the same workers, so
// the number of compressing threads is the budget above whatever the slicing - a pool per slice (round 2) doubled it
// with two slices and would multiply it again with several devices in one process.  Workers run at nice 10: below the
// threads that steer the GPU.  A job that throws, or a worker that cannot be started, fails the batch it belongs to -
// nothing on a worker thread can take the process down.
// ------------------------------------------------------------------------------------------------
class HostPool {
  public:
    struct Batch {
        std::function<void(size_t)> fn;
        size_t n = 0;
        std::atomic<size_t> next{0};
        std::atomic<size_t> left{0};
        std::atomic<bool> failed{false};
        std::string error;                                      // first failure's text (under m)
        std::mutex m;
        std::condition_variable cv;
        void fail(const char *what) { std::lock_guard<std::mutex> l(m); if (!failed.exchange(true)) error = what; }
        // the calling thread helps until the indices are handed out, then waits for the stragglers
        bool wait()
        {
            work();
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [&] { return left.load() == 0; });
            return !failed.load();
        }
        void work()
        {
            for (size_t i = next++; i < n; i = next++) {
                try { fn(i); } catch (const std::exception &e) { fail(e.what()); } catch (...) { fail("unknown exception on a host worker"); }
                if (--left == 0) { std::lock_guard<std::mutex> l(m); cv.notify_all(); }
            }
        }
    };
    static HostPool &instance() { static HostPool *p = new HostPool(); return *p; }      // (never destroyed: workers may outlive main)
    // n jobs fn(0 .. n - 1) on up to `width` workers; returns at once.  The caller keeps the batch alive until wait() returned.
    std::shared_ptr<Batch> submit(size_t n, unsigned width, std::function<void(size_t)> fn)
    {
        auto b = std::make_shared<Batch>();
        b->fn = std::move(fn); b->n = n; b->left = n;
        if (n == 0) return b;
        {
            std::lock_guard<std::mutex> l(m_);
            grow(width);
            queue_.push_back(b);
        }
        cv_.notify_all();
        return b;
    }
    unsigned threads() { std::lock_guard<std::mutex> l(m_); return (unsigned) workers_; }

  private:
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Batch>> queue_;
    size_t workers_ = 0;
    void grow(unsigned width)
    {
        while (workers_ < width) {
            try { std::thread([this]() { run(); }).detach(); } catch (const std::exception &) { break; }   // (thread limit: the callers' wait() does the work)
            workers_++;
        }
    }
    void run()
    {
        setpriority(PRIO_PROCESS, (id_t) syscall(SYS_gettid), 10);
        for (;;) {
            std::shared_ptr<Batch> b;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] {
                    while (!queue_.empty() && queue_.front()->next.load() >= queue_.front()->n) queue_.pop_front();
                    return !queue_.empty();
                });
                b = queue_.front();
            }
            b->work();
        }
    }
};

// The base layer of a batch of chunks.  A chunk is one frame, or `tiles` frames stacked along the row axis that
// the reference codes as ONE JPEG 2000 image with one tile per frame (src/ebcc_codec.c:105-180, n_tiles > 1).
// Every tile is a frame of the engine `ctx` (tile t of chunk c at index c * tiles + t); the arrays below are per
// CHUNK - rate, target, error statistics, codestream size - and are expanded to / gathered from the tiles here:
//   * all tiles of a chunk are probed at the same rate; a tile's byte budget subtracts its share of the main
//     header only (opj_j2k_update_rates: 135 / tiles, J2kFrame::hdr_share);
//   * nbad / err_sum add up; the codestream is main header (SIZ rewritten for the stacked image) + the tile-parts
//     (SOT with the tile index) + EOC, byte-identical to OpenJPEG's opj_write_tile sequence.
// The residual layer works on whole chunks in the engine `rc` (== ctx for one-frame chunks).
struct Batch {
    ebcc_hip_ctx *ctx;
    J2kBuffers &jb;
    const float *d_frames;
    size_t n, tiles, nt;           // chunks, tiles per chunk, n * tiles
    std::vector<J2kFrame> jf;      // per chunk
    J2kFrame *tjf;                 // per tile (device image; pinned: ctx->h_jf)
    std::vector<int> active;       // per chunk
    int *tactive, *ractive;        // pinned: ctx->h_act
    std::vector<float> state_cr;   // rate of the decode the engine holds for every chunk (-1: none)
    int *d_active;                 // per tile, on ctx
    hipStream_t s;
    ebcc_hip_ctx *rc;              // residual engine (chunk-sized frames)
    hipStream_t rs;
    Batch(ebcc_hip_ctx *c, const float *d, size_t n_, size_t tiles_ = 1, ebcc_hip_ctx *rc_ = nullptr)
        : ctx(c), jb(*static_cast<J2kBuffers *>(c->j2k)), d_frames(d), n(n_), tiles(tiles_), nt(n_ * tiles_), jf(n_),
          tjf(static_cast<J2kFrame *>(c->h_jf)), active(n_, 0), tactive(c->h_act), ractive(c->h_act + c->max_frames), state_cr(n_, -1.f),
          d_active(c->d_active), s(c->stream), rc(rc_ ? rc_ : c), rs((rc_ ? rc_ : c)->stream)
    {
        memset(tjf, 0, sizeof(J2kFrame) * nt);
    }
    void fetch_jf(hipStream_t on = nullptr)
    {
        if (!on) on = s;
        EBCC_HIP_CHECK(hipMemcpyAsync(tjf, jb.jf, sizeof(J2kFrame) * nt, hipMemcpyDeviceToHost, on));
        wait_stream(on);
        for (size_t c = 0; c < n; c++) {
            J2kFrame &o = jf[c];
            o.nbad = 0; o.err_sum = 0; o.overflow = 0; o.body_bytes = 0;
            for (size_t t = c * tiles; t < (c + 1) * tiles; t++) {
                o.nbad += tjf[t].nbad; o.err_sum += tjf[t].err_sum; o.overflow |= tjf[t].overflow; o.body_bytes += tjf[t].body_bytes;
            }
            o.stream_bytes = kJ2kMainHeaderBytes + (int) tiles * 14 + o.body_bytes + 2;     // main header, SOT + SOD per tile, EOC
        }
    }
    void push_jf()
    {
        for (size_t c = 0; c < n; c++)
            for (size_t t = c * tiles; t < (c + 1) * tiles; t++) {
                tjf[t].cr = jf[c].cr; tjf[t].target = jf[c].target;
                tjf[t].hdr_share = tiles > 1 ? (float) kJ2kMainHeaderBytes / (float) tiles : 0.0f;
            }
        EBCC_HIP_CHECK(hipMemcpyAsync(jb.jf, tjf, sizeof(J2kFrame) * nt, hipMemcpyHostToDevice, s));
    }
    void push_active()
    {
        for (size_t c = 0; c < n; c++)
            for (size_t t = c * tiles; t < (c + 1) * tiles; t++) tactive[t] = active[c];
        EBCC_HIP_CHECK(hipMemcpyAsync(d_active, tactive, sizeof(int) * nt, hipMemcpyHostToDevice, s));
    }
    // the chunk mask for the residual engine's kernels
    void push_ractive()
    {
        for (size_t c = 0; c < n; c++) ractive[c] = active[c];
        EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_active, ractive, sizeof(int) * n, hipMemcpyHostToDevice, rs));
    }
    // one probe of the base layer for the active chunks: rate allocation at jf[c].cr (+ decode and statistics)
    // keep_field = false: only the statistics are wanted, jb.DEC stays what it was
    void launch_probe(bool decode, bool keep_field = true)
    {
        push_jf();
        push_active();
        launch_j2k_rate(jb, (int) nt, d_active, s);
        if (decode) launch_j2k_probe_decode(d_frames, jb, (int) nt, d_active, s, keep_field);
    }
    void probe(bool decode, bool keep_field = true) { launch_probe(decode, keep_field); fetch_jf(); }
    // codestream of the current layer assignment of the active chunks -> jobs[c].tail
    template <class Jobs>
    void collect_tails(Jobs &jobs)
    {
        push_active();
        launch_j2k_write(jb, (int) nt, d_active, s);
        fetch_jf();
        if (tiles == 1) {
            // all codestreams of the batch in one packed download (engine.hip: stage_download)
            std::vector<size_t> len(n, 0), off(n, 0);
            for (size_t f = 0; f < n; f++) if (active[f]) len[f] = (size_t) jf[f].stream_bytes;
            stage_download(ctx, jb.stream, jb.stream_cap, len.data(), off.data(), n, s);
            for (size_t f = 0; f < n; f++)
                if (active[f]) jobs[f].tail.assign(ctx->h_stage + off[f], ctx->h_stage + off[f] + len[f]);
            return;
        }
        // every tile was written as a one-tile codestream into its slot: [main header 135][SOT 12][SOD 2][packets][EOC 2]
        for (size_t c = 0; c < n; c++) {
            if (!active[c]) continue;
            std::vector<uint8_t> &o = jobs[c].tail;
            o.resize((size_t) jf[c].stream_bytes);
            size_t at = kJ2kMainHeaderBytes;
            for (size_t k = 0; k < tiles; k++) {
                const size_t t = c * tiles + k, part = 14 + (size_t) tjf[t].body_bytes;
                const uint8_t *slot = jb.stream + t * jb.stream_cap;
                if (k == 0) EBCC_HIP_CHECK(hipMemcpyAsync(o.data(), slot, kJ2kMainHeaderBytes, hipMemcpyDeviceToHost, s));
                EBCC_HIP_CHECK(hipMemcpyAsync(o.data() + at, slot + kJ2kMainHeaderBytes, part, hipMemcpyDeviceToHost, s));
                at += part;
            }
        }
        wait_stream(s);
        const unsigned H = (unsigned) jb.geom.H;
        for (size_t c = 0; c < n; c++) {
            if (!active[c]) continue;
            std::vector<uint8_t> &o = jobs[c].tail;
            auto put32 = [&](size_t at, unsigned v) { o[at] = (uint8_t) (v >> 24); o[at + 1] = (uint8_t) (v >> 16); o[at + 2] = (uint8_t) (v >> 8); o[at + 3] = (uint8_t) v; };
            put32(12, H * (unsigned) tiles);                             // SIZ: Ysiz (tile size XTsiz/YTsiz stays W x H)
            size_t at = kJ2kMainHeaderBytes;
            for (size_t k = 0; k < tiles; k++) {
                o[at + 4] = (uint8_t) (k >> 8); o[at + 5] = (uint8_t) k;  // SOT: Isot
                at += 14 + (size_t) tjf[c * tiles + k].body_bytes;
            }
            o[at] = 0xFF; o[at + 1] = 0xD9;                               // EOC
        }
    }
};

// Drive rate search k (0: error-bounded :728, 1: pure base layer :836) of every frame to completion; every
// round runs at most one probe per frame.  A search first advances through the probes already on record for its
// frame (the other search, or the first encode, usually made them) and only asks the GPU for rates not seen yet.
// The final probe of search 0 (:590) must leave its decode in the engine - the residual layer is computed from
// it - so it is re-run unless the engine's last decode of the frame was at exactly that rate.
template <class Jobs>
void run_search(Batch &b, int k, Jobs &jobs, size_t n_pix)
{
    const size_t n = b.n;
    auto needs_state = [&](const Job &j, size_t f, float cr) { return k == 0 && j.rs[0].phase == 3 && b.state_cr[f] != cr; };
    auto feed = [&](Job &j, const ProbeRec &rec) {
        RateSearch &rs = j.rs[k];
        if (rs.phase == 3) j.last[k] = rec;
        const double q = 1. - ((double) rec.nbad / (double) n_pix);                            // :512
        if (k == 0) j.q = q;
        rs.feed(q);
    };
    for (;;) {
        bool any = false;
        for (size_t f = 0; f < n; f++) {
            Job &j = jobs[f];
            b.active[f] = 0;
            if (j.const_field) continue;
            while (!j.want[k] && !j.rs[k].done()) {
                float cr;
                if (!j.rs[k].next(cr)) break;
                const ProbeRec *rec = j.find_probe(cr);
                if (rec && !needs_state(j, f, cr)) feed(j, *rec);
                else { j.want[k] = true; j.want_cr[k] = cr; }
            }
            if (j.want[k]) { b.active[f] = 1; b.jf[f].cr = j.want_cr[k]; any = true; }
        }
        if (!any) break;
        b.probe(true, k == 0);                                   // (search 1 uses the statistics only: the field of search 0 stays)
        for (size_t f = 0; f < n; f++) {
            if (!b.active[f]) continue;
            Job &j = jobs[f];
            const J2kFrame &r = b.jf[f];
            if (k == 0) b.state_cr[f] = r.cr;
            if (!j.find_probe(r.cr)) j.probes.push_back(ProbeRec{r.cr, r.nbad, r.stream_bytes, r.err_sum});
            log_trace("frame %zu (search %d): cr %f 1-quantile %.1e jp2_length %d", f, k, r.cr, (double) r.nbad / (double) n_pix,
                      r.stream_bytes);
            feed(j, *j.find_probe(r.cr));
            j.want[k] = false;
        }
    }
    // A search that leaves through the rate > 1000 exit (:571-574) makes no final probe: its result is the last
    // doubling step.  Take that probe's record, and for search 0 make sure its decode is in the engine.
    bool redo = false;
    for (size_t f = 0; f < n; f++) {
        Job &j = jobs[f];
        b.active[f] = 0;
        if (j.const_field) continue;
        if (j.last[k].cr != j.rs[k].result)
            if (const ProbeRec *rec = j.find_probe(j.rs[k].result)) j.last[k] = *rec;
        if (k == 0 && b.state_cr[f] != j.rs[0].result) { b.active[f] = 1; b.jf[f].cr = j.rs[0].result; redo = true; }
    }
    if (redo) {
        b.probe(true);
        for (size_t f = 0; f < n; f++)
            if (b.active[f]) {
                const J2kFrame &r = b.jf[f];
                b.state_cr[f] = r.cr;
                jobs[f].last[0] = ProbeRec{r.cr, r.nbad, r.stream_bytes, r.err_sum};
            }
    }
}

// The same search with its state machine on the device (search.hpp): the rounds are enqueued back to back - advance,
// rate allocation, probe decode - without a host synchronisation in between; the host looks at the states once after
// `rounds` of them (EBCC_HIP_SEARCH_ROUNDS, default 16: more than the usual search needs) and only enqueues more if a
// chunk is still searching.  Same probes, same decisions, same result as run_search (EBCC_HIP_HOST_SEARCH=1 selects that).
int search_rounds()
{
    if (const char *e = getenv("EBCC_HIP_SEARCH_ROUNDS")) return std::max(1, atoi(e));
    return 16;
}
constexpr int kSearchAll = 0, kSearchStart = 1, kSearchFinish = 2;
template <class Jobs>
void device_rate_search(Batch &b, int k, Jobs &jobs, size_t n_pix, unsigned slices, int lane = 0, int part = kSearchAll)
{
    // lane 1: the search runs on the engine's second stream with its own state, counters and active mask, beside whatever
    // the first stream does (search #2 beside the residual layer).  part: enqueue the first batch of rounds only
    // (kSearchStart: no host synchronisation), or take the search up from there (kSearchFinish), or both.
    ebcc_hip_ctx *ctx = b.ctx;
    const size_t n = b.n;
    DevChunk *h = static_cast<DevChunk *>(ctx->h_search) + (size_t) lane * ctx->max_frames, *d = static_cast<DevChunk *>(ctx->d_search) + (size_t) lane * ctx->max_frames;
    int *const d_counter = ctx->d_counter + 4 * lane, *const h_counter = ctx->h_counter + 4 * lane;
    int *const d_active = lane ? ctx->d_active + ctx->max_frames : b.d_active;
    hipStream_t s = lane ? second_stream(ctx) : b.s;
    J2kBuffers &jb = b.jb;
    const int forced = getenv("EBCC_HIP_SPECULATION") ? atoi(getenv("EBCC_HIP_SPECULATION")) != 0 : -1;
    const bool speculate = lane == 0 && (forced >= 0 ? forced != 0 : slices <= 1);   // (lane 1 runs on the stream the candidates would use)
    hipStream_t s2 = nullptr;
    // probes that only steer the search stop counting once they are certainly infeasible (search.hpp); EBCC_HIP_NO_SHORTCUTS=1
    // and TRACE logging (which prints every probe's count) keep every probe exact
    double jobs_qt0 = 0.0;                                               // the error-bounded search's quantile target (0: it never ran)
    for (size_t f = 0; f < n; f++) if (!jobs[f].const_field) { jobs_qt0 = jobs[f].rs[0].qt; break; }
    const double limit_qt = getenv("EBCC_HIP_NO_SHORTCUTS") || g_log_level <= 0 ? 0.0 : std::min(jobs_qt0, 1.0);
    auto advance = [&]() {
        launch_search_advance(d, jb.jf, d_active, (int) n, (int) b.tiles, k, (double) n_pix, d_counter, s,
                              speculate ? jb.cand_cr : nullptr, speculate ? jb.cand_sel : nullptr, limit_qt);
        if (speculate) launch_j2k_rate_publish(jb, (int) b.nt, s);
    };
    auto enqueue_rounds = [&](int rounds) {
        for (int r = 0; r < rounds; r++) {
            launch_j2k_rate(jb, (int) b.nt, d_active, s, speculate ? jb.have_rate : nullptr);
            if (speculate) {
                // the candidates read the record of bisection steps this k_rate may have extended, and the masks / rates of
                // the advance: after both; the next advance reads their results: after them
                EBCC_HIP_CHECK(hipEventRecord(ctx->ev_a, s));
                EBCC_HIP_CHECK(hipStreamWaitEvent(s2, ctx->ev_a, 0));
                launch_j2k_rate_candidates(jb, (int) b.nt, d_active, s2);
                EBCC_HIP_CHECK(hipEventRecord(ctx->ev_b, s2));
            }
            launch_j2k_probe_decode(b.d_frames, jb, (int) b.nt, d_active, s, k == 0 ? 2 : 0);     // (the field is stored where the advance asked for it: search.hip keeps_field)
            if (speculate) EBCC_HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_b, 0));
            advance();
        }
    };
    if (part != kSearchFinish) {
    for (size_t f = 0; f < n; f++) {
        const Job &j = jobs[f];
        DevChunk &c = h[f];
        c.const_field = j.const_field ? 1 : 0;
        c.state_cr = b.state_cr[f];
        c.q = j.q;
        c.n_probes = (int) std::min<size_t>(j.probes.size(), kMaxProbes);
        for (int i = 0; i < c.n_probes; i++) c.probes[i] = DevProbe{j.probes[i].cr, j.probes[i].stream_bytes, j.probes[i].nbad, j.probes[i].err_sum, j.probes[i].complete ? 1 : 0, 0};
        const RateSearch &r = j.rs[k];
        DevRateSearch &o = c.rs[k];
        o.lo = r.lo; o.hi = r.hi; o.cr = r.cr; o.result = r.result; o.pending = r.pending; o.phase = j.const_field ? 6 : r.phase;
        o.q = r.q; o.q0 = r.q0; o.qt = r.qt; o.want = 0; o.want_cr = 0;
        o.last = DevProbe{j.last[k].cr, j.last[k].stream_bytes, j.last[k].nbad, j.last[k].err_sum, 1, 0};
    }
    EBCC_HIP_CHECK(hipMemcpyAsync(d, h, sizeof(DevChunk) * n, hipMemcpyHostToDevice, s));
    EBCC_HIP_CHECK(hipMemsetAsync(d_counter, 0, sizeof(int) * 4, s));
    // a round = the probe the previous advance asked for (rate allocation + decode of the active chunks), then the advance
    // that takes it in and asks for the next one.  Speculative rate allocation: a
    // step of the search can go two ways, so the layers of both rates it may ask for next are worked out on the engine's
    // second stream while the first stream decodes the current probe; the advance then takes the matching one over
    // (k_rate_publish) and the round's own k_rate only runs for the frames whose rate was not among the guesses.
    // It shortens a slice's chain (search #1 of 256 frames in one slice: 33 -> 29 ms) at the price of two more k_rate per
    // round; with several slices in flight the chip has no idle issue slots left to pay with (four slices: encode 7.7 GB/s
    // without, 6.7 with) - so it is on for a batch that runs as one slice, off otherwise; EBCC_HIP_SPECULATION=1 / 0 forces it.
    if (speculate) {
        s2 = second_stream(ctx);
        if (!ctx->ev_a) {
            EBCC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming));
            EBCC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
        }
        EBCC_HIP_CHECK(hipMemsetAsync(jb.cand_cr, 0xFF, sizeof(float) * 2 * b.nt, s));       // (NaN: no candidate matches)
        EBCC_HIP_CHECK(hipMemsetAsync(jb.have_rate, 0, sizeof(int) * b.nt, s));
    }
    advance();
    enqueue_rounds(search_rounds());
    }
    if (part == kSearchStart) return;
    if (speculate && !s2) s2 = second_stream(ctx);
    for (;;) {
        EBCC_HIP_CHECK(hipMemcpyAsync(h, d, sizeof(DevChunk) * n, hipMemcpyDeviceToHost, s));
        EBCC_HIP_CHECK(hipMemcpyAsync(h_counter, d_counter, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
        wait_stream(s);
        bool done = true;
        for (size_t f = 0; f < n; f++) done &= h[f].rs[k].phase == 6;
        if (done) break;
        enqueue_rounds(6);
    }
    log_trace("rate search %d: %d probes of chunks over the rounds", k, h_counter[0]);
    if (getenv("EBCC_HIP_PHASE_TIMING")) {
        int most = 0; long long sum = 0;
        for (size_t f = 0; f < n; f++) { most = std::max(most, h[f].n_probes); sum += h[f].n_probes; }
        fprintf(stderr, "ebcc-mi355x rate search %d: %d probes over the rounds, probes on record per chunk: mean %.1f, most %d\n", k, h_counter[0], (double) sum / (double) n, most);
    }
    if (getenv("EBCC_HIP_T1_STATS") && slices <= 1) j2k_probe_hist_dump(k == 0 ? "search #1" : "search #2");
    for (size_t f = 0; f < n; f++) {
        Job &j = jobs[f];
        if (j.const_field) continue;
        const DevChunk &c = h[f];
        const DevRateSearch &o = c.rs[k];
        RateSearch &r = j.rs[k];
        r.lo = o.lo; r.hi = o.hi; r.cr = o.cr; r.result = o.result; r.pending = o.pending; r.phase = 4; r.q = o.q; r.q0 = o.q0; r.qt = o.qt;
        j.last[k] = ProbeRec{o.last.cr, o.last.nbad, o.last.stream_bytes, o.last.err_sum};
        if (k == 0) j.q = c.q;
        j.probes.clear();
        for (int i = 0; i < c.n_probes; i++) j.probes.push_back(ProbeRec{c.probes[i].cr, c.probes[i].nbad, c.probes[i].stream_bytes, c.probes[i].err_sum, c.probes[i].complete != 0});
        b.state_cr[f] = c.state_cr;
    }
    b.fetch_jf(s);                                                        // (the host mirror of the per-frame scalars follows the device again)
}
template <class Jobs>
void rate_search(Batch &b, int k, Jobs &jobs, size_t n_pix, unsigned slices)
{
    const bool host_loop = getenv("EBCC_HIP_HOST_SEARCH") != nullptr;
    if (host_loop) run_search(b, k, jobs, n_pix); else device_rate_search(b, k, jobs, n_pix, slices);
}

// ------------------------------------------------------------------------------------------------
// ebcc_encode for a batch of device-resident single-frame chunks.  Returns 0, 1 (error) or 2 (NaN/Inf).
// ------------------------------------------------------------------------------------------------
// `n` chunks of `tiles` frames each (tiles == 1: the frame-per-chunk case); `rctx`: residual engine for the stacked
// chunk image when tiles > 1.
// A shard is coded batch after batch (ebcc_hip_encode_shard).  A batch ends with host work - the level-22 zstd of the kept
// prefixes, ~a quarter of its time - during which its engines have nothing to do, so two engine sets alternate: one batch at
// a time is in its GPU phase (GpuPhase), and the next one enters it the moment every slice of the current one has reached
// its entropy stage (PhaseNote).
struct GpuPhase {
    std::mutex m;
    std::condition_variable cv;
    bool busy = false;
    void acquire() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !busy; }); busy = true; }
    void release() { { std::lock_guard<std::mutex> l(m); busy = false; } cv.notify_one(); }
};
struct PhaseNote {
    GpuPhase *phase = nullptr;
    std::atomic<int> total{0}, done{0};
    std::atomic<bool> released{false};
    void expect(int slices) { int e = 0; total.compare_exchange_strong(e, slices); }
    void slice_done() { if (++done == total.load()) release_once(); }
    void release_once() { bool e = false; if (phase && released.compare_exchange_strong(e, true)) phase->release(); }
};

int encode_batch(ebcc_hip_ctx *ctx, const float *d_frames, size_t n, const codec_config_t *cfg, uint8_t **outs, size_t *sizes,
                 SliceGate *next = nullptr, size_t tiles = 1, ebcc_hip_ctx *rctx = nullptr, unsigned slices = 1, PhaseNote *note = nullptr)
{
    struct Release { SliceGate *g; ~Release() { if (g) g->release(); } } release_on_exit{next};   // (error paths too)
    struct NoteOnce { PhaseNote *n; void tell() { if (n) { n->slice_done(); n = nullptr; } } ~NoteOnce() { tell(); } } gpu_phase_over{note};   // (every path reports once)
    const EncodeEnv env;
    const double q_target = 1 - env.base_error_quantile;
    const int mode = (int) cfg->residual_compression_type;
    const bool searching = mode == MAX_ERROR || mode == RELATIVE_ERROR;
    const size_t n_pix = ctx->n_pix * tiles;                           // pixels of a chunk
    const size_t nt = n * tiles;
    Batch b(ctx, d_frames, n, tiles, rctx);
    J2kBuffers &jb = b.jb;
    hipStream_t s = b.s;
    ebcc_hip_ctx *rc = b.rc;                                           // residual engine and its stream (== ctx, s for one-frame chunks)
    hipStream_t rs = b.rs;
    std::vector<Job> jobs(n);
    PhaseTimer pt;

    // ---- statistics, scaling, transform, tier-1: once per frame
    launch_input_stats(d_frames, (int) nt, ctx->n_pix, ctx->rb.fs, s);
    if (tiles > 1) {
        // the reference scales the whole chunk with one (min, max) (:686-689): combine the tiles' statistics before
        // the transform reads them; a tile that happens to be constant inside a varying chunk is coded normally
        fetch_frame_states(ctx, nt);
        for (size_t c = 0; c < n; c++) {
            FrameState *t0 = ctx->h_fs + c * tiles;
            float mn = t0[0].minv, mx = t0[0].maxv;
            int bad = 0;
            for (size_t k = 0; k < tiles; k++) { mn = std::min(mn, t0[k].minv); mx = std::max(mx, t0[k].maxv); bad |= t0[k].has_nonfinite; }
            for (size_t k = 0; k < tiles; k++) { t0[k].minv = mn; t0[k].maxv = mx; t0[k].has_nonfinite = bad; t0[k].const_field = mn == mx; }
        }
        push_frame_states(ctx, nt);
    }
    launch_j2k_analysis(d_frames, jb, (int) nt, s);
    if (next) { next->release(); release_on_exit.g = nullptr; }       // the next slice may start: this one's first stage is queued
    fetch_frame_states(ctx, nt);
    b.fetch_jf();
    if (j2k_tier1_retry(jb, (int) nt, b.tjf, s)) b.fetch_jf();        // (a group's decisions outgrew the segmented encoder's buffer)
    for (size_t f = 0; f < n; f++) {
        const FrameState &t0 = ctx->h_fs[f * tiles];                   // (all tiles of a chunk carry the chunk's statistics)
        if (t0.has_nonfinite) { log_fatal("NaN or Inf found in data of frame %zu", f); return 2; }
        if (b.jf[f].overflow) { log_fatal("code-block byte slot overflow in frame %zu", f); return 1; }
        jobs[f].const_field = t0.const_field != 0;
        jobs[f].minv = t0.minv;
        jobs[f].maxv = t0.maxv;
        b.jf[f].cr = cfg->base_cr;
        b.jf[f].target = 0;
        b.active[f] = jobs[f].const_field ? 0 : 1;
    }
    if (rc != ctx) {                                                   // chunk-level frame states of the residual engine
        for (size_t f = 0; f < n; f++) {
            FrameState &r = rc->h_fs[f];
            r = FrameState{};
            r.minv = jobs[f].minv; r.maxv = jobs[f].maxv; r.const_field = jobs[f].const_field;
        }
        push_frame_states(rc, n);
    }
    pt.mark("analysis (dwt, tier-1, ckpt)");
    const bool need_decode = mode != NONE;
    if (need_decode)
        for (size_t f = 0; f < n; f++) {
            float target = cfg->error;                                                        // :723-726
            if (mode == RELATIVE_ERROR) target *= jobs[f].maxv - jobs[f].minv;
            jobs[f].target = target;
            b.jf[f].target = target;
        }
    // ---- first encode at base_cr (:693) and, unless NONE, its decode (:707-709)
    b.probe(need_decode);
    if (mode == NONE) {
        b.collect_tails(jobs);
    } else {
        for (size_t f = 0; f < n; f++) {
            if (jobs[f].const_field) continue;
            jobs[f].mean_err = b.jf[f].err_sum / (double) n_pix;                              // :709
            jobs[f].q = jobs[f].q_first = 1. - ((double) b.jf[f].nbad / (double) n_pix);
            jobs[f].cr = cfg->base_cr;
            jobs[f].probes.push_back(ProbeRec{b.jf[f].cr, b.jf[f].nbad, b.jf[f].stream_bytes, b.jf[f].err_sum});
            b.state_cr[f] = b.jf[f].cr;
        }
        // residual range of the first decode: only the header fields survive when no search runs (:716)
        launch_residual_minmax(d_frames, jb.DEC, (int) n, n_pix, rc->rb.fs, rs);
        fetch_frame_states(rc, n);
        for (size_t f = 0; f < n; f++) { jobs[f].rmin = rc->h_fs[f].rmin; jobs[f].rmax = rc->h_fs[f].rmax; }
        if (!searching) {                       // stale enum values fall through to a base-only stream (quirk Q2)
            for (size_t f = 0; f < n; f++) b.active[f] = jobs[f].const_field ? 0 : 1;
            b.collect_tails(jobs);
        }
    }

    pt.mark("first probe");
    if (searching) {
        // ---- rate search #1 (:728)
        const bool pure_done = q_target == 1.0;                                               // :738
        const bool want_pure = !pure_done && !env.no_fallback;
        for (size_t f = 0; f < n; f++)
            if (!jobs[f].const_field) jobs[f].rs[0].start(cfg->base_cr, jobs[f].q, q_target);
        rate_search(b, 0, jobs, n_pix, slices);
        for (size_t f = 0; f < n; f++) {
            b.active[f] = jobs[f].const_field ? 0 : 1;
            if (!jobs[f].const_field) { jobs[f].cr = jobs[f].rs[0].result; jobs[f].len1 = (size_t) jobs[f].last[0].stream_bytes; }
        }
        pt.mark("rate search 1");
        // base layer of search #1.  (Sending the codestreams off without waiting for them - written and packed on the second
        // search's stream, fetched at the assembly - was measured: the slice's next stages are queued 2 ms earlier and the step
        // gets 1 - 5 ms LONGER, three alternating runs on two boxes; the wait stays.)
        b.collect_tails(jobs);
        auto start_search2 = [&]() {
            for (size_t f = 0; f < n; f++) {
                if (jobs[f].const_field) continue;
                if (env.no_consistency) jobs[f].rs[1].start(jobs[f].cr, jobs[f].q, 1.0);      // from search #1's state
                else jobs[f].rs[1].start(cfg->base_cr, jobs[f].q_first, 1.0);                 // :829-833 == the first probe
            }
        };
        // ---- the pure base-layer search (:819-836) depends on nothing the residual layer produces: its rounds are queued on the
        //      engine's second stream now (own state, counters and mask: device_rate_search lane 1) and run beside the residual
        //      layer and the truncation search; it is taken up again where the reference runs it (below).  Round 2 measured
        //      this slower - the search was hidden behind the level-22 zstd of every prefix then; with the entropy stage cut
        //      down to the prefixes whose size can matter, the search was what the slice waited for, and its sizes are what
        //      decides which prefixes those are.  (With EBCC_HIP_HOST_SEARCH=1 it runs in the reference's place.)
        const bool overlap2 = want_pure && tiles == 1 && rc == ctx && !getenv("EBCC_HIP_HOST_SEARCH");
        struct DrainSecond {               // an error return between here and the take-up must not leave rounds in flight
            ebcc_hip_ctx *c; bool armed;
            ~DrainSecond() { if (armed && c->stream2) hipStreamSynchronize(c->stream2); }
        } drain2{ctx, false};
        if (overlap2) {
            start_search2();
            device_rate_search(b, 1, jobs, n_pix, slices, 1, kSearchStart);
            drain2.armed = true;
        }
        launch_residual_minmax(d_frames, jb.DEC, (int) n, n_pix, rc->rb.fs, rs);             // :730-733
        fetch_frame_states(rc, n);
        bool any_resid = false;
        for (size_t f = 0; f < n; f++) {
            Job &j = jobs[f];
            b.active[f] = 0;
            if (j.const_field) continue;
            j.rmin = rc->h_fs[f].rmin; j.rmax = rc->h_fs[f].rmax;
            float cur = fmaxf(fabsf(j.rmin), fabsf(j.rmax));                                  // :735
            j.skip = cur <= j.target;                                                         // :737
            if (!j.skip) { b.active[f] = 1; any_resid = true; }
        }
        pt.mark("tails + residual range");

        if (!zstd().ok) { log_fatal("libzstd not available"); return 1; }
        // ---- the entropy stage's state (the stage itself follows the truncation search; the frames whose search ends first
        //      - the early group, below - enter it while the others still search).
        //      jobs on the process-wide pool (HostPool): every slice of a batch feeds the same workers, so the host is never
        //      oversubscribed however many slices run
        std::vector<const uint8_t *> coeff_ptr(n, nullptr);             // the kept prefix of a frame in pinned host memory
        std::atomic<long long> zstd_us{0}, zstd_max_us{0}, zstd_bytes{0}, bound_us{0};    // core time, longest job, bytes
        enum : uint8_t { kZNone = 0, kZQueued = 1, kZRunning = 2, kZSkipped = 3 };
        std::unique_ptr<std::atomic<uint8_t>[]> zstate(new std::atomic<uint8_t>[n]);
        for (size_t f = 0; f < n; f++) zstate[f] = kZNone;
        std::vector<size_t> zfloor(n, 0);                               // lower bound of z (0: none)
        std::vector<char> floor_done(n, 0);
        using PoolBatches = std::vector<std::shared_ptr<HostPool::Batch>>;
        PoolBatches zbatches, fbatches;                                 // level-22 jobs; lower bounds
        struct WaitOnExit { PoolBatches &v; ~WaitOnExit() { for (auto &b : v) if (b) b->wait(); } } wait_on_exit{zbatches}, wait_on_exit_f{fbatches};   // (error paths too: the jobs point into this frame)
        // level-22 zstd of the frames in `list`, in the order given; a frame that was decided in the meantime (kZSkipped)
        // is passed over
        auto submit_zstd = [&](std::vector<size_t> list) {
            for (size_t f : list) zstate[f] = kZQueued;
            auto order = std::make_shared<std::vector<size_t>>(std::move(list));
            zbatches.push_back(HostPool::instance().submit(order->size(), entropy_threads(slices), [&, order](size_t i) {
                const size_t f = (*order)[i];
                uint8_t expect = kZQueued;
                if (!zstate[f].compare_exchange_strong(expect, kZRunning)) return;
                Job &j = jobs[f];
                const auto z0 = std::chrono::steady_clock::now();
                j.zbytes.resize(zstd().bound(j.coeffs_size));
                const size_t z = zstd().compress(j.zbytes.data(), j.zbytes.size(), coeff_ptr[f], j.coeffs_size, env.zstd_level);
                if ((zstd().is_error && zstd().is_error(z)) || z > j.zbytes.size()) throw std::runtime_error("ZSTD_compress failed on a residual prefix");
                j.zbytes.resize(z);
                const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - z0).count();
                zstd_us += us; zstd_bytes += (long long) j.coeffs_size;
                long long m = zstd_max_us.load(); while (us > m && !zstd_max_us.compare_exchange_weak(m, us)) {}
            }));
        };
        // the lower bounds of z for the frames in `list` (zstd_size_lower_bound)
        auto submit_floors = [&](const std::vector<size_t> &frames_) {
            auto list = std::make_shared<std::vector<size_t>>(frames_);
            fbatches.push_back(HostPool::instance().submit(list->size(), entropy_threads(slices), [&, list](size_t i) {
                const size_t f = (*list)[i];
                const auto z0 = std::chrono::steady_clock::now();
                zfloor[f] = zstd_size_lower_bound(coeff_ptr[f], jobs[f].coeffs_size);
                floor_done[f] = 1;
                bound_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - z0).count();
            }));
        };
        // longest first: level 22 takes ~0.2 ms per KB on one core and a batch has frames whose prefix is ten times the
        // average - started last, such a frame alone decides when the slice can go on
        auto longest_first = [&](std::vector<size_t> &list) {
            std::stable_sort(list.begin(), list.end(), [&](size_t a, size_t c) { return jobs[a].coeffs_size > jobs[c].coeffs_size; });
        };
        long long wait_us = 0;
        auto join = [&](PoolBatches &v) -> bool {
            const auto w0 = std::chrono::steady_clock::now();
            bool ok = true;
            std::string why;
            for (auto &b : v) if (b && !b->wait()) { ok = false; if (why.empty()) why = b->error; }
            v.clear();
            wait_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - w0).count();
            if (!ok) { log_fatal("entropy stage failed: %s", why.c_str()); set_error("%s", why.c_str()); }
            return ok;
        };
        auto zjoin = [&]() -> bool { const bool a = join(fbatches), c = join(zbatches); return a && c; };
        const bool use_floor = want_pure && zstd_floor_usable() && !getenv("EBCC_HIP_NO_SHORTCUTS");
        if (any_resid) {
            // ---- residual layer: SPIHT with a budget of the base layer's size (:744-754)
            b.push_ractive();
            launch_pad_and_dc(d_frames, jb.DEC, rc->rb, (int) n, rc->d_active, rs);
            launch_analysis(rc->rb, (int) n, rc->d_active, rs);
            // (budgets, the encoder, the cut "everything" and its probe are queued without a look at the frame states in
            //  between: the budget follows from the base layer's size, the whole stream's length stays on the device)
            for (size_t f = 0; f < n; f++) rc->h_u64a[f] = (unsigned long long) jobs[f].len1 * 8 + 128;   // bits0 = trunc_bits + 128
            EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_u64a, rc->h_u64a, n * sizeof(unsigned long long), hipMemcpyHostToDevice, rs));
            launch_residual_budget(rc->rb, (int) n, rc->d_u64a, rc->d_active, rs);
            launch_spiht_encode(rc->rb, (int) n, rc->d_u64a, rc->d_active, rs);
            launch_whole_stream_cut(rc->rb, (int) n, rc->d_u64b, rc->d_active, rs);
            launch_prefix_synthesis_stats(d_frames, jb.DEC, rc->rb, (int) n, rc->d_u64b, rc->d_active, rs);   // full decode, :749
            fetch_frame_states(rc, n);
            for (size_t f = 0; f < n; f++) {
                if (!b.active[f]) continue;
                jobs[f].coeffs_orig = rc->h_fs[f].stream_bytes;
                jobs[f].coeffs_size = jobs[f].coeffs_orig;
                rc->h_u64b[f] = (unsigned long long) jobs[f].coeffs_orig * 8;
            }
            auto probe_residual = [&]() {
                EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_u64b, rc->h_u64b, n * sizeof(unsigned long long), hipMemcpyHostToDevice, rs));
                b.push_ractive();
                launch_prefix_synthesis_stats(d_frames, jb.DEC, rc->rb, (int) n, rc->d_u64b, rc->d_active, rs);
                fetch_frame_states(rc, n);
            };
            pt.mark("residual: analysis, SPIHT, whole-stream probe");
            for (size_t f = 0; f < n; f++) {
                Job &j = jobs[f];
                if (!b.active[f]) continue;
                float cur = u2f(rc->h_fs[f].maxerr_bits);                                    // :754
                if (cur > j.target) {                                                         // :755-759
                    log_info("frame %zu: could not reach error target %f (%f instead); retry with pure base compression", f, j.target, cur);
                    j.skip = true; j.need_pure = true;
                } else {
                    j.best_err = cur;
                    j.mean_err = rc->h_fs[f].err_sum / (double) n_pix;                       // :762
                    j.t_hi = (double) j.coeffs_size * 8; j.t_lo = 112.0; j.t_best = j.t_hi;   // :766-776
                    j.trunc_active = true;
                }
            }
            // ---- truncation bisection (:777-795): all frames advance one cut per round
            const bool host_loop = getenv("EBCC_HIP_HOST_SEARCH") != nullptr;
            if (!host_loop) {
                // state machine on the device (search.hpp): advance, reconstruct the decoder state at the cut, synthesis +
                // statistics - enqueued back to back, one look at the states after `rounds` of them
                DevChunk *h = static_cast<DevChunk *>(rc->h_search), *d = static_cast<DevChunk *>(rc->d_search);
                for (size_t f = 0; f < n; f++) {
                    const Job &j = jobs[f];
                    DevChunk &c = h[f];
                    c.t_hi = j.t_hi; c.t_lo = j.t_lo; c.t_best = j.t_best; c.mean_err = j.mean_err; c.best_err = j.best_err;
                    c.target = j.target; c.trunc_active = j.trunc_active ? 1 : 0; c.trunc_pending = 0;
                }
                EBCC_HIP_CHECK(hipMemcpyAsync(d, h, sizeof(DevChunk) * n, hipMemcpyHostToDevice, rs));
                EBCC_HIP_CHECK(hipMemsetAsync(rc->d_counter, 0, sizeof(int) * 4, rs));
                // rounds a chunk can still need: a cut halves the interval (rounded up to a byte: + 8 bits at most) until it is
                // 32 bits wide (:777), then one more advance sees that nothing is left
                auto rounds_left = [](const DevChunk &c) {
                    if (!c.trunc_active) return 0;
                    double w = c.t_hi - c.t_lo;
                    int r = 1;
                    while (w > 32 && r < 64) { w = w / 2 + 8; r++; }
                    return r;
                };
                const bool forced_rounds = getenv("EBCC_HIP_SEARCH_ROUNDS") != nullptr;
                int cuts_left = 1;                                       // cuts the longest search still visits, + the advance that ends it
                for (size_t f = 0; f < n; f++) cuts_left = std::max(cuts_left, rounds_left(h[f]));
                auto take_result = [&](const DevChunk &c, Job &j) {
                    j.t_hi = c.t_hi; j.t_lo = c.t_lo; j.t_best = c.t_best; j.mean_err = c.mean_err; j.best_err = c.best_err;
                    j.trunc_active = false;
                };
                // ---- Look-ahead (search.hpp: launch_trunc_advance_multi): a round probes the cut :779 chooses now and the cuts
                //      either outcome leads to - `levels` levels of the bisection tree, 2^levels - 1 cut slots per frame - so the
                //      search takes 1 / levels of the rounds.  The rounds are latency (a chain of six small launches beside the
                //      other slices' work), the probes off the path mostly stop early (a cut shorter than an infeasible one is
                //      infeasible too: its first wave over the target ends it).  EBCC_HIP_TRUNC_LEVELS=1: one cut per round.
                int levels = getenv("EBCC_HIP_TRUNC_LEVELS") ? std::min(3, std::max(1, atoi(getenv("EBCC_HIP_TRUNC_LEVELS")))) : 2;
                while (levels > 1 && !ensure_cut_slots(rc, (int) n * ((1 << levels) - 1))) levels--;
                if (levels > 1) {
                    const CutSlots &cs = rc->cut;
                    const int n_slots = (int) n * ((1 << levels) - 1);
                    launch_trunc_advance_multi(d, rc->rb.fs, cs, (int) n, (double) n_pix, levels, nullptr, rc->d_counter, rs);
                    int rounds = forced_rounds ? search_rounds() : (cuts_left + levels - 1) / levels + 1;
                    for (;;) {
                        for (int r = 0; r < rounds; r++) {
                            launch_prefix_synthesis_slots(d_frames, jb.DEC, rc->rb, cs, n_slots, rs);
                            launch_trunc_advance_multi(d, rc->rb.fs, cs, (int) n, (double) n_pix, levels, nullptr, rc->d_counter, rs);
                        }
                        EBCC_HIP_CHECK(hipMemcpyAsync(h, d, sizeof(DevChunk) * n, hipMemcpyDeviceToHost, rs));
                        wait_stream(rs);
                        bool done = true;
                        for (size_t f = 0; f < n; f++) done &= !h[f].trunc_active;
                        if (done) break;
                        rounds = 3;
                    }
                    for (size_t f = 0; f < n; f++) if (jobs[f].trunc_active) take_result(h[f], jobs[f]);
                } else {
                    launch_trunc_advance(d, rc->rb.fs, rc->d_u64b, rc->d_active, (int) n, (double) n_pix, rc->d_counter, rs);
                    int rounds = forced_rounds ? search_rounds() : cuts_left + 1;
                    for (;;) {
                        for (int r = 0; r < rounds; r++) {
                            launch_prefix_synthesis_stats(d_frames, jb.DEC, rc->rb, (int) n, rc->d_u64b, rc->d_active, rs);
                            launch_trunc_advance(d, rc->rb.fs, rc->d_u64b, rc->d_active, (int) n, (double) n_pix, rc->d_counter, rs);
                        }
                        EBCC_HIP_CHECK(hipMemcpyAsync(h, d, sizeof(DevChunk) * n, hipMemcpyDeviceToHost, rs));
                        wait_stream(rs);
                        bool done = true;
                        for (size_t f = 0; f < n; f++) done &= !h[f].trunc_active;
                        if (done) break;
                        rounds = 6;
                    }
                    for (size_t f = 0; f < n; f++) if (jobs[f].trunc_active) take_result(h[f], jobs[f]);
                }
            } else
            for (;;) {
                const double eps = 1e-8;
                bool any = false;
                for (size_t f = 0; f < n; f++) {
                    Job &j = jobs[f];
                    b.active[f] = 0;
                    if (!j.trunc_active) continue;
                    if (((j.target - j.best_err) / j.target > eps) && (j.t_hi - j.t_lo > 8 * 4)) {
                        size_t tb = ((size_t) ceill((long double) ((j.t_hi + j.t_lo) / 2 / 8))) * 8;
                        rc->h_u64b[f] = tb;
                        b.active[f] = 1;
                        any = true;
                    } else {
                        j.trunc_active = false;
                    }
                }
                if (!any) break;
                probe_residual();
                for (size_t f = 0; f < n; f++) {
                    Job &j = jobs[f];
                    if (!b.active[f]) continue;
                    const double tb = (double) rc->h_u64b[f];
                    float cur = u2f(rc->h_fs[f].maxerr_bits);
                    if (cur > j.target) j.t_lo = tb;
                    else {
                        j.t_hi = tb;
                        if (cur >= j.best_err) { j.best_err = cur; j.t_best = tb; j.mean_err = rc->h_fs[f].err_sum / (double) n_pix; }
                    }
                    log_trace("frame %zu: trunc_lo %.1f trunc_hi %.1f max error %f", f, j.t_lo, j.t_hi, cur);
                }
            }
            for (size_t f = 0; f < n; f++) {
                Job &j = jobs[f];
                if (j.const_field || j.skip) { if (j.need_pure) j.coeffs_size = j.coeffs_orig; else j.coeffs_size = 0; }
                else j.coeffs_size = (size_t) (j.t_best / 8.);                                // :796
            }
        }
        pt.mark("truncation search");
        // ---- entropy stage of the kept SPIHT prefix on host cores (:811-817) and the pure base-layer fallback (:819-854).
        //      Level-22 zstd is by far the longest host step (~160 ns per byte on one core: 1.3 core-seconds per 256 frames
        //      of the bench workload on a box whose container has 16 CPUs), and the reference throws most of it away: the
        //      compressed size z is compared with what the pure base-layer search gives (:838: len2 < z + len1), and for
        //      ~95 % of ERA5-like frames the base layer alone wins.  So z is only worked out where it can matter: a frame
        //      whose z is PROVABLY above len2 - len1 (zstd_size_lower_bound: the literals no match can cover cost at least
        //      their entropy) takes the pure base layer without being compressed - the same decision, bytes unchanged.
        // the kept SPIHT prefixes of the batch (but those of the early group, which left during the truncation search) in one
        // packed download; the workers read them where they land (the staging buffer of the residual engine is not touched
        // again before they are done)
        std::vector<size_t> coeff_len(n, 0), coeff_off(n, 0);
        for (size_t f = 0; f < n; f++) {
            Job &j = jobs[f];
            if (j.coeffs_size <= 16) j.coeffs_size = 0;
            if (!coeff_ptr[f]) coeff_len[f] = j.coeffs_size;
        }
        stage_download(rc, (const uint8_t *) rc->rb.stream, rc->rb.stream_words * sizeof(uint32_t), coeff_len.data(), coeff_off.data(), n, rs);
        for (size_t f = 0; f < n; f++) if (coeff_len[f]) coeff_ptr[f] = rc->h_stage + coeff_off[f];
        std::vector<size_t> with_prefix;
        for (size_t f = 0; f < n; f++) if (jobs[f].coeffs_size > 0) with_prefix.push_back(f);
        std::vector<size_t> cand;
        auto not_started = [&](std::vector<size_t> v) {                 // (the early group's jobs are on their way)
            v.erase(std::remove_if(v.begin(), v.end(), [&](size_t f) { return zstate[f] != kZNone; }), v.end());
            return v;
        };
        if (!want_pure) {
            std::vector<size_t> rest = not_started(with_prefix);
            longest_first(rest);
            submit_zstd(rest);                                          // no fallback: every prefix is part of its stream
        } else {
            // frames whose residual layer could not reach the target are coded by the base layer whatever z is (:838 need_pure)
            std::vector<size_t> now;
            for (size_t f : with_prefix) {
                if (jobs[f].need_pure) { zstate[f] = kZSkipped; continue; }
                if (use_floor && jobs[f].coeffs_size <= kZstdFloorMaxBytes) cand.push_back(f); else now.push_back(f);
            }
            now = not_started(now);
            longest_first(now);
            if (!now.empty()) submit_zstd(now);
            std::vector<size_t> want_floor;
            for (size_t f : cand) if (!floor_done[f]) want_floor.push_back(f);
            if (!want_floor.empty()) submit_floors(want_floor);
        }
        pt.mark("zstd: queued");
        if (want_pure) {
            // The pure-base-layer search restarts from base_cr with the quantile of a re-encode at base_cr
            // (:829-833), i.e. of the first probe above (unless that consistency step is disabled), and re-uses
            // every probe search #1 made; it runs here, while host cores work on the prefixes: first the floors (a few
            // microseconds per KB), then - the search still running on the GPU, len2 not known yet - zstd of the
            // candidates in the order in which they are likely to need it (lowest floor per byte first); the moment the
            // search's sizes are in, the candidates they decide are struck from the queue.
            const bool host_loop = getenv("EBCC_HIP_HOST_SEARCH") != nullptr;
            if (overlap2) {
                // the search has been running beside the residual layer: its sizes are (nearly) there, the floors take a
                // millisecond - the prefixes that are still open after that are compressed, longest first
                device_rate_search(b, 1, jobs, n_pix, slices, 1, kSearchFinish);              // :836
                drain2.armed = false;
                pt.mark("rate search 2");
                gpu_phase_over.tell();                                                        // (what follows is host work and one small launch)
                if (!join(fbatches)) return 1;                                                // (the floors)
            } else {
                start_search2();
                if (!host_loop) device_rate_search(b, 1, jobs, n_pix, slices, 0, kSearchStart);
                if (!cand.empty()) {
                    if (!join(fbatches)) return 1;                                            // (the floors)
                    std::vector<size_t> spec = not_started(cand);
                    std::stable_sort(spec.begin(), spec.end(), [&](size_t a, size_t c) {
                        return (double) zfloor[a] * (double) jobs[c].coeffs_size < (double) zfloor[c] * (double) jobs[a].coeffs_size; });
                    submit_zstd(spec);
                }
                if (host_loop) run_search(b, 1, jobs, n_pix); else device_rate_search(b, 1, jobs, n_pix, slices, 0, kSearchFinish);   // :836
                pt.mark("rate search 2");
                gpu_phase_over.tell();
            }
            long long skipped_bytes = 0, skipped = 0;
            for (size_t f : cand) {
                const Job &j = jobs[f];
                const size_t len2 = (size_t) j.last[1].stream_bytes;
                // z >= zfloor: len2 < zfloor + len1 implies len2 < z + len1 - the base layer alone wins (:838)
                if (!(zfloor[f] > 0 && len2 < zfloor[f] + j.len1)) continue;
                // (decided before a worker took it up: not queued yet, or queued - by the early group or the speculative list)
                uint8_t expect = kZNone;
                bool struck = zstate[f].compare_exchange_strong(expect, kZSkipped);
                if (!struck) { expect = kZQueued; struck = zstate[f].compare_exchange_strong(expect, kZSkipped); }
                if (struck) { skipped++; skipped_bytes += (long long) j.coeffs_size; }
            }
            if (overlap2) {
                std::vector<size_t> open;
                for (size_t f : cand) if (zstate[f] == kZNone) open.push_back(f);
                longest_first(open);
                if (!open.empty()) submit_zstd(open);
            }
            if (!zjoin()) return 1;
            pt.mark("zstd: wait for the workers");
            for (size_t f : with_prefix) if (jobs[f].need_pure) { skipped++; skipped_bytes += (long long) jobs[f].coeffs_size; }
            if (pt.on) fprintf(stderr, "ebcc-mi355x zstd: %.1f ms of core time for %lld bytes, longest job %.1f ms; floors %.1f ms; %lld of %zu prefixes (%lld bytes) not compressed\n",
                               zstd_us.load() / 1e3, zstd_bytes.load(), zstd_max_us.load() / 1e3, bound_us.load() / 1e3, skipped, with_prefix.size(), skipped_bytes);
            host_stats().skipped_bytes += skipped_bytes;
            if (pt.on && getenv("EBCC_HIP_ZSTD_TRACE"))
                for (size_t f : with_prefix) {
                    const Job &j = jobs[f];
                    const long long x = (long long) j.last[1].stream_bytes - (long long) j.len1;
                    fprintf(stderr, "zstd-trace c %zu floor %zu X %lld z %zu state %d need_pure %d nbad1 %llu len1 %zu orig %zu\n", j.coeffs_size, zfloor[f], x, j.zbytes.size(), (int) zstate[f].load(), (int) j.need_pure,
                            (unsigned long long) j.last[0].nbad, j.len1, j.coeffs_orig);
                }
            bool any_pure = false;
            for (size_t f = 0; f < n; f++) {
                Job &j = jobs[f];
                b.active[f] = 0;
                if (j.const_field) continue;
                const size_t len2 = (size_t) j.last[1].stream_bytes;
                const bool decided = zfloor[f] > 0 && len2 < zfloor[f] + j.len1;              // (z may not be known - only that it loses)
                if (decided || len2 < j.zbytes.size() + j.len1 || j.need_pure) {              // :838
                    if (decided) log_info("frame %zu: pure base compression (%zu) beats base (%zu) + residual (at least %zu)", f, len2, j.len1, zfloor[f]);
                    else if (len2 < j.zbytes.size() + j.len1)
                        log_info("frame %zu: pure base compression (%zu) beats base (%zu) + residual (%zu)", f, len2, j.len1, j.zbytes.size());
                    j.mean_err = j.last[1].err_sum / (double) n_pix;                          // :843
                    j.zbytes.clear(); j.coeffs_size = 0;
                    b.active[f] = 1; b.jf[f].cr = j.rs[1].result; any_pure = true;
                }
            }
            if (any_pure) {
                // the layer assignment of that rate again (allocation only, no decode), then its codestream
                b.push_jf();
                b.push_active();
                launch_j2k_rate(jb, (int) nt, b.d_active, s);
                b.collect_tails(jobs);
            }
        }
        if (!zjoin()) return 1;
        host_stats().add(zstd_us.load() + bound_us.load(), wait_us, zstd_bytes.load());
    }

    pt.mark("fallback search + tails");
    // ---- assemble (:863-907)
    for (size_t f = 0; f < n; f++) {
        Job &j = jobs[f];
        float minv = j.minv, maxv = j.maxv;
        log_info("frame %zu: mean of compression error %e", f, j.mean_err);
        if (!env.no_mean_adjust && std::fabs(j.mean_err) > 1e-18) {
            minv += j.mean_err;
            maxv += j.mean_err;
        }
        const size_t codec_size = j.const_field ? sizeof(uint64_t) : j.tail.size();
        const size_t total = sizeof(FrameHeader) + j.zbytes.size() + codec_size;
        uint8_t *o = (uint8_t *) malloc(total), *p = o;
        if (!o) { log_fatal("out of memory"); return 1; }
        FrameHeader hd;
        memset(&hd, 0, sizeof hd);
        memcpy(hd.magic, EBCC_HEADER_MAGIC, 4);
        hd.version = EBCC_HEADER_VERSION;
        if (j.const_field) hd.flags |= EBCC_HEADER_FLAG_CONST_FIELD;
        hd.minval_bits = f2u(minv); hd.maxval_bits = f2u(maxv);
        hd.coeffs_size = j.coeffs_size;
        hd.rmin_bits = f2u(j.const_field ? 0.0f : j.rmin); hd.rmax_bits = f2u(j.const_field ? 0.0f : j.rmax);
        hd.compressed_size = j.zbytes.size(); hd.tail_size = codec_size;
        memcpy(p, &hd, sizeof hd); p += sizeof hd;
        if (!j.zbytes.empty()) { memcpy(p, j.zbytes.data(), j.zbytes.size()); p += j.zbytes.size(); }
        if (j.const_field) { uint64_t cnt = n_pix; memcpy(p, &cnt, 8); }
        else memcpy(p, j.tail.data(), j.tail.size());
        log_info("frame %zu: coeffs_size %zu compressed_size %zu jp2_length %zu ratio %f", f, j.coeffs_size, j.zbytes.size(),
                 codec_size, (double) (n_pix * 4) / (double) total);
        outs[f] = o;
        sizes[f] = total;
    }
    pt.mark("assemble");
    return 0;
}

// One frame stream, either format: the 48-byte "EBCC" header (:190-202, :1234-1260) or the legacy header-less
// prefix `f32 min, f32 max, u64 coeffs_size, f32 rmin, f32 rmax, u64 compressed_size` (ebcc_decode_legacy,
// :1147-1213), where a constant field is signalled by min == max.
struct ParsedFrame {
    float minv = 0, maxv = 0, rmin = 0, rmax = 0;
    bool const_field = false;
    size_t coeffs_size = 0, compressed_size = 0, tail_size = 0;
    const uint8_t *z = nullptr, *tail = nullptr;
};

bool parse_frame(const uint8_t *d, size_t len, ParsedFrame &pf)
{
    if (len >= sizeof(FrameHeader) && memcmp(d, EBCC_HEADER_MAGIC, 4) == 0) {
        FrameHeader hd;
        memcpy(&hd, d, sizeof hd);
        if (hd.version != EBCC_HEADER_VERSION) { log_fatal("Unsupported EBCC header version: %u", hd.version); return false; }
        size_t used = sizeof hd;
        if (hd.compressed_size > len - used) { log_fatal("Invalid encoded data: truncated payload"); return false; }   // :1249
        used += hd.compressed_size;
        if (hd.tail_size > len - used) { log_fatal("Invalid encoded data: truncated payload"); return false; }         // :1254
        used += hd.tail_size;
        if (used != len) { log_fatal("Invalid encoded data: payload size mismatch"); return false; }                  // :1314
        pf.minv = u2f(hd.minval_bits); pf.maxv = u2f(hd.maxval_bits);
        pf.rmin = u2f(hd.rmin_bits); pf.rmax = u2f(hd.rmax_bits);
        pf.const_field = (hd.flags & EBCC_HEADER_FLAG_CONST_FIELD) != 0;
        pf.coeffs_size = hd.coeffs_size; pf.compressed_size = hd.compressed_size; pf.tail_size = hd.tail_size;
        pf.z = d + sizeof hd; pf.tail = pf.z + hd.compressed_size;
        if (pf.const_field && hd.tail_size != sizeof(uint64_t)) {
            log_fatal("Invalid encoded data: const-field payload must contain uint64_t length");
            return false;
        }
    } else {
        const size_t prefix = 4 + 4 + 8 + 4 + 4 + 8;
        if (len < prefix) { log_fatal("Invalid legacy encoded data: truncated header"); return false; }
        uint64_t cs, zs;
        memcpy(&pf.minv, d, 4); memcpy(&pf.maxv, d + 4, 4); memcpy(&cs, d + 8, 8);
        memcpy(&pf.rmin, d + 16, 4); memcpy(&pf.rmax, d + 20, 4); memcpy(&zs, d + 24, 8);
        if (zs > len - prefix) { log_fatal("Invalid legacy encoded data: truncated residual payload"); return false; }
        pf.coeffs_size = cs; pf.compressed_size = zs;
        pf.z = d + prefix; pf.tail = pf.z + zs; pf.tail_size = len - prefix - zs;
        pf.const_field = pf.minv == pf.maxv;
        if (pf.const_field && pf.tail_size < sizeof(uint64_t)) { log_fatal("Invalid legacy encoded data: missing const-field length"); return false; }
    }
    if (pf.const_field && pf.compressed_size > 0 && pf.coeffs_size > 0) {
        log_fatal("Invalid encoded data: residual data cannot be applied to const field");
        return false;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// ebcc_decode for a batch of single-frame EBCC streams -> device buffer d_out [n][H*W]
// ------------------------------------------------------------------------------------------------
int decode_batch(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n, float *d_out,
                 SliceGate *next = nullptr)
{
    struct Release { SliceGate *g; ~Release() { if (g) g->release(); } } release_on_exit{next};
    J2kBuffers &jb = *static_cast<J2kBuffers *>(ctx->j2k);
    hipStream_t s = ctx->stream;
    const size_t n_pix = ctx->n_pix;
    const J2kGeom &g = jb.geom;
    int *const table = ctx->h_table;                                  // (pinned)
    const size_t table_ints = n * (size_t) g.stride * 4;
    memset(table, 0, table_ints * sizeof(int));
    // pieces to upload: codestream k = f, SPIHT bytes k = n + f - staged in pinned memory and sent as one copy
    std::vector<size_t> piece(2 * n, 0), piece_off(2 * n, 0);
    std::vector<ParsedFrame> heads(n);
    PhaseTimer pt;
    fetch_frame_states(ctx, n);
    // the host side of a batch - frame headers, packet headers, zstd of the residual streams - is per frame and runs on a
    // few host threads (decode is one slice: nothing else hides it); a frame's failure fails the batch
    std::atomic<bool> failed{false}, resid{false};
    // (the reason a frame was rejected is written on the worker's thread - set_error's text is per thread; the first one is
    //  carried over to the calling thread, where ebcc_hip_last_error is read)
    auto for_frames = [&](auto body) {
        const unsigned width = (unsigned) std::min<size_t>({(size_t) 16, (size_t) entropy_threads(1), (n + 7) / 8});
        auto batch = HostPool::instance().submit(n, width, [&](size_t f) {
            if (failed.load(std::memory_order_relaxed)) return;
            clear_error();
            if (!body(f)) {
                failed = true;
                const char *why = ebcc_hip_last_error();
                throw std::runtime_error(why && *why ? why : "invalid encoded data");
            }
        });
        if (!batch->wait()) { failed = true; set_error("%s", batch->error.c_str()); }
    };
    for_frames([&](size_t f) -> bool {
        const uint8_t *d = streams[f];
        const size_t len = sizes[f];
        ctx->h_active[f] = 0;
        ParsedFrame &hd = heads[f];
        if (!parse_frame(d, len, hd)) return false;
        FrameState &fs = ctx->h_fs[f];
        fs.minv = hd.minv; fs.maxv = hd.maxv;
        fs.rmin = hd.rmin; fs.rmax = hd.rmax;
        fs.const_field = hd.const_field ? 1 : 0;
        if (fs.const_field) {
            uint64_t cnt = 0;
            memcpy(&cnt, hd.tail, 8);
            if (cnt != n_pix) { log_fatal("const-field length %llu does not match the frame", (unsigned long long) cnt); return false; }
        } else {
            if (hd.tail_size > jb.stream_cap) { log_fatal("codestream larger than the device slot"); return false; }
            if (!j2k_parse_codestream(hd.tail, hd.tail_size, g, table + f * g.stride * 4)) return false;
            piece[f] = hd.tail_size;
            if (hd.compressed_size > 0 && hd.coeffs_size > 0) {                                                    // :1294-1304
                if (!zstd().ok) { log_fatal("libzstd not available"); return false; }
                if (hd.coeffs_size > ctx->rb.stream_words * 4 - 64) { log_fatal("residual stream larger than the device slot"); return false; }
                piece[n + f] = hd.coeffs_size;
                ctx->h_active[f] = 1;
                resid = true;
            }
        }
        return true;
    });
    if (failed) return 1;
    const bool any_resid = resid;
    stage_reserve(ctx, piece.data(), piece_off.data(), 2 * n);
    for_frames([&](size_t f) -> bool {
        const ParsedFrame &hd = heads[f];
        if (piece[f]) memcpy(ctx->h_stage + piece_off[f], hd.tail, hd.tail_size);
        if (piece[n + f]) {
            // the residual stream: exactly coeffs_size bytes (the staging buffer holds whatever an earlier call left),
            // a SPIHT header for this grid and a bit budget the decoder can work with (:1294-1304)
            const size_t got = zstd().decompress(ctx->h_stage + piece_off[n + f], hd.coeffs_size, hd.z, hd.compressed_size);
            if ((zstd().is_error && zstd().is_error(got)) || got != hd.coeffs_size) { log_fatal("Invalid encoded data: residual payload does not decompress to %zu bytes", hd.coeffs_size); return false; }
            if (check_ims_header(ctx, ctx->h_stage + piece_off[n + f], hd.coeffs_size, hd.coeffs_size * 8)) { log_fatal("Invalid encoded data: %s", ebcc_hip_last_error()); return false; }
        }
        return true;
    });
    if (failed) return 1;
    stage_send(ctx, 2 * n, s);
    stage_scatter(ctx, jb.stream, jb.stream_cap, 0, n, s);
    pt.mark("decode: parse, zstd, uploads");
    push_frame_states(ctx, n);
    // The residual layer (SPIHT decode + synthesis: one wave per frame, latency-bound) does not depend on the
    // base layer until the final addition, so it runs on the engine's second stream beside the tier-1 decode.
    hipStream_t s2 = s;
    if (any_resid) {
        s2 = second_stream(ctx);
        if (!ctx->ev_a) {
            EBCC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming));
            EBCC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
        }
        EBCC_HIP_CHECK(hipEventRecord(ctx->ev_a, s));                       // frame states and the staged pieces are on the device
        EBCC_HIP_CHECK(hipStreamWaitEvent(s2, ctx->ev_a, 0));
    }
    // the residual stream is fed first: its one-wave-per-frame kernel has to find a wave slot on every CU, and once the
    // tier-1 decoder's ~10^4 workgroups (longest code-blocks first) hold the slots none frees up for milliseconds
    if (any_resid) {
        const size_t slot = ctx->rb.stream_words * 4;
        for (size_t f = 0; f < n; f++) {
            ctx->h_u64a[f] = piece[n + f];
            ctx->h_u64b[f] = piece[n + f] * 8;
        }
        stage_scatter(ctx, (uint8_t *) ctx->rb.stream, slot, n, n, s2);
        EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_u64a, ctx->h_u64a, n * sizeof(unsigned long long), hipMemcpyHostToDevice, s2));
        EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_u64b, ctx->h_u64b, n * sizeof(unsigned long long), hipMemcpyHostToDevice, s2));
        EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_active, ctx->h_active, n * sizeof(int), hipMemcpyHostToDevice, s2));
        launch_spiht_decode((const uint8_t *) ctx->rb.stream, slot, ctx->d_u64a, ctx->d_u64b, ctx->rb, (int) n, ctx->d_active, s2);
        launch_synthesis_head(ctx->rb, (int) n, ctx->d_active, s2);
        if (s2 != s) EBCC_HIP_CHECK(hipEventRecord(ctx->ev_b, s2));
    }
    EBCC_HIP_CHECK(hipMemcpyAsync(jb.dec_table, table, table_ints * sizeof(int), hipMemcpyHostToDevice, s));
    launch_j2k_decode(jb, (int) n, s);
    if (next) { next->release(); release_on_exit.g = nullptr; }     // host parsing done, kernels queued
    if (any_resid) {
        if (s2 != s) EBCC_HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_b, 0));
        launch_synthesis_tail_add(jb.DEC, ctx->rb, (int) n, ctx->d_active, s);       // last row pass: DEC += residual
    }
    EBCC_HIP_CHECK(hipMemcpyAsync(d_out, jb.DEC, n * n_pix * sizeof(float), hipMemcpyDeviceToDevice, s));
    // constant fields: fill on the host side of the copy (rare path)
    for (size_t f = 0; f < n; f++)
        if (ctx->h_fs[f].const_field) {
            std::vector<float> v(n_pix, ctx->h_fs[f].minv);
            EBCC_HIP_CHECK(hipMemcpyAsync(d_out + f * n_pix, v.data(), n_pix * sizeof(float), hipMemcpyHostToDevice, s));
            wait_stream(s);
        }
    wait_stream(s);
    pt.mark("decode: kernels");
    return 0;
}

// Chunks of several frames: the tail is one codestream with a tile per frame (reference :121-125); the tiles are
// decoded as frames of `ctx`, the residual of the whole chunk image in `rc`.
// Frame heights a chunk of several frames can have: OpenJPEG cannot set up 6 resolutions on smaller tiles (the
// reference crashes on them), and the chunk image itself is bounded by the reference's 2047-row limit.
bool tile_height_supported(size_t h) { return h >= 32 && h <= 1023; }
// Heights for which every tile has the geometry of a tile at the origin (sub-band extents, parity and code-block
// partition repeat): the context then needs a single geometry for all tile positions.
bool tile_geometry_uniform(size_t h) { return h >= 32 && h <= 1024 && (h & (h - 1)) == 0; }

int decode_tiled(ebcc_hip_ctx *ctx, ebcc_hip_ctx *rc, const uint8_t *const *streams, const size_t *sizes, size_t n, size_t tiles,
                 float *d_out)
{
    J2kBuffers &jb = *static_cast<J2kBuffers *>(ctx->j2k);
    hipStream_t s = ctx->stream, rs = rc->stream;
    const J2kGeom &g = jb.geom;
    const size_t tile_pix = ctx->n_pix, n_pix = tile_pix * tiles, nt = n * tiles;
    std::vector<int> table(nt * (size_t) g.stride * 4, 0);
    std::vector<std::vector<uint8_t>> coeffs(n);
    std::vector<size_t> off(tiles), len(tiles);
    bool any_resid = false;
    for (size_t c = 0; c < n; c++) {
        ParsedFrame hd;
        if (!parse_frame(streams[c], sizes[c], hd)) return 1;
        rc->h_active[c] = 0;
        FrameState &r = rc->h_fs[c];
        r = FrameState{};
        r.minv = hd.minv; r.maxv = hd.maxv; r.rmin = hd.rmin; r.rmax = hd.rmax; r.const_field = hd.const_field ? 1 : 0;
        for (size_t k = 0; k < tiles; k++) {
            FrameState &t = ctx->h_fs[c * tiles + k];
            t = FrameState{};
            t.minv = hd.minv; t.maxv = hd.maxv; t.const_field = r.const_field;
        }
        if (hd.const_field) {
            uint64_t cnt = 0;
            memcpy(&cnt, hd.tail, 8);
            if (cnt != n_pix) { log_fatal("const-field length %llu does not match the chunk", (unsigned long long) cnt); return 1; }
            continue;
        }
        if (!j2k_parse_tiled(hd.tail, hd.tail_size, jb, (int) tiles, table.data() + c * tiles * g.stride * 4, off.data(), len.data())) {
            log_fatal("Invalid encoded data: %s", ebcc_hip_last_error());
            return 1;
        }
        for (size_t k = 0; k < tiles; k++) {
            if (len[k] > jb.stream_cap) { log_fatal("tile-part larger than the device slot"); return 1; }
            EBCC_HIP_CHECK(hipMemcpyAsync(jb.stream + (c * tiles + k) * jb.stream_cap, hd.tail + off[k], len[k], hipMemcpyHostToDevice, s));
        }
        if (hd.compressed_size > 0 && hd.coeffs_size > 0) {
            if (hd.coeffs_size > rc->rb.stream_words * 4 - 64) { log_fatal("residual stream larger than the device slot"); return 1; }
            coeffs[c].assign(hd.coeffs_size, 0);
            const size_t got = zstd().decompress(coeffs[c].data(), hd.coeffs_size, hd.z, hd.compressed_size);
            if ((zstd().is_error && zstd().is_error(got)) || got != hd.coeffs_size) { log_fatal("Invalid encoded data: residual payload does not decompress to %zu bytes", hd.coeffs_size); return 1; }
            if (check_ims_header(rc, coeffs[c].data(), hd.coeffs_size, hd.coeffs_size * 8)) { log_fatal("Invalid encoded data: %s", ebcc_hip_last_error()); return 1; }
            rc->h_active[c] = 1;
            any_resid = true;
        }
    }
    push_frame_states(ctx, nt);
    EBCC_HIP_CHECK(hipMemcpyAsync(jb.dec_table, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice, s));
    launch_j2k_decode(jb, (int) nt, s);
    wait_stream(s);
    if (any_resid) {
        push_frame_states(rc, n);
        const size_t slot = rc->rb.stream_words * 4;
        for (size_t c = 0; c < n; c++) {
            rc->h_u64a[c] = coeffs[c].size();
            rc->h_u64b[c] = coeffs[c].size() * 8;
            if (rc->h_active[c])
                EBCC_HIP_CHECK(hipMemcpyAsync((uint8_t *) rc->rb.stream + c * slot, coeffs[c].data(), coeffs[c].size(), hipMemcpyHostToDevice, rs));
        }
        EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_u64a, rc->h_u64a, n * sizeof(unsigned long long), hipMemcpyHostToDevice, rs));
        EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_u64b, rc->h_u64b, n * sizeof(unsigned long long), hipMemcpyHostToDevice, rs));
        EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_active, rc->h_active, n * sizeof(int), hipMemcpyHostToDevice, rs));
        launch_spiht_decode((const uint8_t *) rc->rb.stream, slot, rc->d_u64a, rc->d_u64b, rc->rb, (int) n, rc->d_active, rs);
        launch_synthesis_head(rc->rb, (int) n, rc->d_active, rs);
        launch_synthesis_tail_add(jb.DEC, rc->rb, (int) n, rc->d_active, rs);
        wait_stream(rs);
    }
    EBCC_HIP_CHECK(hipMemcpyAsync(d_out, jb.DEC, n * n_pix * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (size_t c = 0; c < n; c++)
        if (rc->h_fs[c].const_field) {
            std::vector<float> v(n_pix, rc->h_fs[c].minv);
            EBCC_HIP_CHECK(hipMemcpyAsync(d_out + c * n_pix, v.data(), n_pix * sizeof(float), hipMemcpyHostToDevice, s));
            wait_stream(s);
        }
    wait_stream(s);
    return 0;
}

// the engine of the tiles and, for chunks of several frames, the engine of the stacked chunk image
bool chunk_engines(int device, int H, int W, size_t chunks, size_t tiles, ebcc_hip_ctx **ctx, ebcc_hip_ctx **rc)
{
    *rc = nullptr;
    if (tiles > 1) {
        *rc = get_context(device, (int) (tiles * (size_t) H), W, chunks);
        if (!*rc) return false;
    }
    const int period = tiles > 1 && !tile_geometry_uniform((size_t) H) ? (int) tiles : 1;
    *ctx = get_context(device, H, W, chunks * tiles, period);
    if (!*ctx) return false;
    if (tiles > 1) {                                             // (creating the second engine may have evicted the first)
        *rc = get_context(device, (int) (tiles * (size_t) H), W, chunks);
        if (!*rc) return false;
        std::lock_guard<std::mutex> lock(g_map_mutex);
        if (g_ctx.find(std::make_tuple(device, H, W, period)) == g_ctx.end()) return false;
    }
    return true;
}

// Chunk <-> array copies of the chunking entry points (reference :311-370) as row copies: a chunk is a box, its rows
// are contiguous in the array; rows / frames / columns past the array's edge repeat the last one (index clamping).
struct ChunkBox {
    size_t dims[3], cd[3], cnt[3];
    size_t csize() const { return cd[0] * cd[1] * cd[2]; }
    void origin(size_t cl, size_t org[3]) const { for (int d = 3; d-- > 0;) { org[d] = (cl % cnt[d]) * cd[d]; cl /= cnt[d]; } }
    bool inside(size_t cl) const { size_t o[3]; origin(cl, o); return o[0] + cd[0] <= dims[0] && o[1] + cd[1] <= dims[1] && o[2] + cd[2] <= dims[2]; }
    // chunks that are whole frames of the array: chunk cl is the contiguous range [cl * csize, (cl + 1) * csize)
    bool slabs() const { return cd[1] == dims[1] && cd[2] == dims[2]; }
    void gather(const float *data, size_t cl, float *dst) const
    {
        size_t org[3];
        origin(cl, org);
        const size_t w = std::min(cd[2], dims[2] - org[2]);
        for (size_t z = 0; z < cd[0]; z++) {
            const size_t zi = std::min(org[0] + z, dims[0] - 1);
            for (size_t y = 0; y < cd[1]; y++) {
                const size_t yi = std::min(org[1] + y, dims[1] - 1);
                const float *src = data + (zi * dims[1] + yi) * dims[2] + org[2];
                float *row = dst + (z * cd[1] + y) * cd[2];
                memcpy(row, src, w * sizeof(float));
                for (size_t x = w; x < cd[2]; x++) row[x] = src[w - 1];
            }
        }
    }
    void scatter(const float *src, size_t cl, float *out) const
    {
        size_t org[3];
        origin(cl, org);
        const size_t w = std::min(cd[2], dims[2] - org[2]);
        for (size_t z = 0; z < cd[0] && org[0] + z < dims[0]; z++)
            for (size_t y = 0; y < cd[1] && org[1] + y < dims[1]; y++)
                memcpy(out + ((org[0] + z) * dims[1] + org[1] + y) * dims[2] + org[2], src + (z * cd[1] + y) * cd[2], w * sizeof(float));
    }
};

int run_encode_slices(ebcc_hip_ctx *ctx, const float *d_frames, size_t n_frames, const codec_config_t *cfg, uint8_t **outs, size_t *sizes,
                      GpuPhase *phase = nullptr);
int run_decode_slices(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n_frames, float *d_out);

// n_frames one-frame chunks in batches of the context's capacity, alternately on the context's engines and on a second set
// (ebcc_hip_ctx::twin, made on first use; without memory for it the batches run one after the other on the first):
// one batch at a time is in its GPU phase, the next enters it when every slice of the current one has reached its
// entropy stage (GpuPhase / PhaseNote).  stage(set, first frame, count) -> where the batch's frames are on the device
// (a host array is uploaded there: that copy runs beside the other batch's kernels too).
template <class Stage>
int encode_batches_alternating(ebcc_hip_ctx *ctx, size_t n_frames, const codec_config_t *cfg, uint8_t **outs, size_t *sizes, Stage stage)
{
    const size_t cap = ctx->max_frames, batches = (n_frames + cap - 1) / cap;
    if (batches == 1) return run_encode_slices(ctx, stage(ctx, (size_t) 0, n_frames), n_frames, cfg, outs, sizes);
    // (a second set that could not be made is not tried again at every call - tens of GB allocated and freed each time -
    //  until ebcc_hip_release_engines / a new context gives the memory a chance to have changed)
    if (!ctx->twin && !ctx->twin_failed) {
        ctx->twin = ebcc_hip_create(ctx->device, cap, (size_t) ctx->height, (size_t) ctx->width);
        if (!ctx->twin) ctx->twin_failed = true;
    }
    ebcc_hip_ctx *const set[2] = {ctx, ctx->twin};
    GpuPhase phase;
    std::atomic<int> worst{0};
    std::atomic<size_t> next{0};
    std::mutex redo_m;
    std::vector<size_t> redo;                                        // batches the second set could not stage
    std::string err[2];
    auto work = [&](int t) {
        try {
            EBCC_HIP_CHECK(hipSetDevice(ctx->device));
            for (;;) {
                size_t b = next++;
                if (b >= batches) {
                    if (t != 0) break;
                    std::lock_guard<std::mutex> l(redo_m);
                    if (redo.empty()) break;
                    b = redo.back(); redo.pop_back();
                }
                if (worst.load()) break;
                const size_t lo = b * cap, cnt = std::min(cap, n_frames - lo);
                const float *where = nullptr;
                try { where = stage(set[t], lo, cnt); }
                catch (const std::exception &e) {
                    // the second set has no room for its image of the frames: the first set does its batches after its own
                    if (t == 0) throw;
                    log_warn("second engine set: %s - its batches run on the first", e.what());
                    clear_error();
                    std::lock_guard<std::mutex> l(redo_m);
                    redo.push_back(b);
                    for (size_t r = next++; r < batches; r = next++) redo.push_back(r);
                    return;
                }
                const int r = run_encode_slices(set[t], where, cnt, cfg, outs + lo, sizes + lo, &phase);
                if (r) { err[t] = ebcc_hip_last_error(); int e = 0; worst.compare_exchange_strong(e, r); }
            }
        } catch (const std::exception &e) { err[t] = e.what(); int z = 0; worst.compare_exchange_strong(z, 1); }
    };
    if (set[1]) {
        std::thread second(work, 1);
        work(0);
        second.join();
        work(0);                                                      // (what the second set handed back after the first had finished)
    } else {
        work(0);
    }
    if (worst.load()) set_error("%s", (err[0].empty() ? err[1] : err[0]).c_str());
    return worst.load();
}

// The decode counterpart: batches of `cap` frames alternately on the two engine sets, both free-running - one batch's
// download (or, for long residual streams, its one-wave-per-frame SPIHT chains, which leave most of the chip idle) beside
// the other's kernels.  each(set, first frame, count) decodes one batch and puts its output where it belongs.
template <class Each>
int decode_batches_alternating(ebcc_hip_ctx *ctx, size_t n_frames, size_t cap, Each each)
{
    const size_t batches = (n_frames + cap - 1) / cap;
    if (batches == 1) return each(ctx, (size_t) 0, n_frames);
    if (!ctx->twin && !ctx->twin_failed) {
        ctx->twin = ebcc_hip_create(ctx->device, ctx->max_frames, (size_t) ctx->height, (size_t) ctx->width);
        if (!ctx->twin) ctx->twin_failed = true;
    }
    ebcc_hip_ctx *const set[2] = {ctx, ctx->twin};
    std::atomic<int> worst{0};
    std::string err[2];
    auto work = [&](int t) {
        try {
            EBCC_HIP_CHECK(hipSetDevice(ctx->device));
            for (size_t b = (size_t) t; b < batches && !worst.load(); b += set[1] ? 2 : 1) {
                const size_t lo = b * cap;
                const int r = each(set[t], lo, std::min(cap, n_frames - lo));
                if (r) { err[t] = ebcc_hip_last_error(); int e = 0; worst.compare_exchange_strong(e, r); }
            }
        } catch (const std::exception &e) { err[t] = e.what(); int z = 0; worst.compare_exchange_strong(z, 1); }
    };
    if (set[1]) {
        std::thread second(work, 1);
        work(0);
        second.join();
    } else {
        work(0);
    }
    if (worst.load()) set_error("%s", (err[0].empty() ? err[1] : err[0]).c_str());
    return worst.load();
}

// host-pointer convenience used by the reference-compatible entry points
// n chunks of `tiles` frames of H x W each, contiguous in host memory, on device `device`.  Returns 0 ok, 1 error (logged),
// 2 NaN / Inf in the data (the caller exits as the reference does, /root/reference/src/ebcc_codec.c:598-605).
int encode_host_frames(int device, const float *data, size_t n, int H, int W, const codec_config_t *cfg, uint8_t **outs, size_t *sizes,
                       size_t tiles = 1)
{
    try {
        std::lock_guard<std::mutex> lock(device_mutex(device));
        DeviceScope scope(device);
        const size_t n_pix = (size_t) H * W * tiles;
        const size_t cap = std::min(n, batch_capacity(n_pix));
        ebcc_hip_ctx *ctx = nullptr, *rc = nullptr;
        if (!chunk_engines(device, H, W, cap, tiles, &ctx, &rc)) { log_fatal("no MI355X engine available: %s", ebcc_hip_last_error()); return 1; }
        // (a batch is uploaded in one go: uploads issued from inside the slices slow every slice down - measured in round 1 with
        //  pageable copies, 5.6 against 3.7 GB/s, and again in round 2 through the bounce buffers, 7.1 against 6.5)
        if (tiles == 1 && ctx->max_frames == cap)                   // one-frame chunks: concurrent slices, batches on alternating engine sets
            return encode_batches_alternating(ctx, n, cfg, outs, sizes, [&](ebcc_hip_ctx *set, size_t lo, size_t cnt) {
                float *d = io_buffer(set, cap * n_pix * sizeof(float));
                copy_pageable(set, const_cast<float *>(data + lo * n_pix), d, cnt * n_pix * sizeof(float), false);
                return (const float *) d;
            });
        float *d = io_buffer(ctx, cap * n_pix * sizeof(float));
        size_t done = 0;
        while (done < n) {
            size_t k = std::min(cap, n - done);
            copy_pageable(ctx, const_cast<float *>(data + done * n_pix), d, k * n_pix * sizeof(float), false);
            const int rcode = tiles == 1 ? run_encode_slices(ctx, d, k, cfg, outs + done, sizes + done)
                                         : encode_batch(ctx, d, k, cfg, outs + done, sizes + done, nullptr, tiles, rc);
            if (rcode) return rcode;
            done += k;
        }
        return 0;
    } catch (const std::exception &e) {
        log_fatal("MI355X engine failure: %s", e.what());
        set_error("%s", e.what());
        return 1;
    }
}

// A list of independent chunks spread over the devices of device_list(): contiguous blocks, one host thread per device
// (/root/reference/src/ebcc_codec.c:1007-1046 is a serial loop over the chunks; the order of the results is that of the
// chunks).  fn(device, first, count) -> status; returns the worst status.
template <class Fn>
int run_on_devices(size_t n_chunks, Fn fn)
{
    std::vector<int> devs = device_list();
    if (devs.size() > n_chunks) devs.resize(std::max<size_t>(1, n_chunks));
    if (devs.size() == 1) return fn(devs[0], (size_t) 0, n_chunks);
    const size_t per = (n_chunks + devs.size() - 1) / devs.size();
    std::vector<int> rc(devs.size(), 0);
    std::vector<std::thread> th;
    for (size_t i = 0; i < devs.size(); i++) {
        const size_t lo = i * per, hi = std::min(n_chunks, lo + per);
        if (lo >= hi) break;
        th.emplace_back([&, i, lo, hi]() { rc[i] = fn(devs[i], lo, hi - lo); });
    }
    for (auto &t : th) t.join();
    int worst = 0;
    for (int r : rc) worst = std::max(worst, r);
    return worst;
}

}  // namespace

// Slices of a batch: EBCC_HIP_SLICES (encode, 1 = off) / EBCC_HIP_DECODE_SLICES engines of max_frames / slices
// frames each, created on first use.  Small batches stay on the context's own engine.  More than two slices only
// pay when the HIP runtime has a hardware queue for each stream (GPU_MAX_HW_QUEUES, default 4, shared with the
// application's streams; it is read when the runtime starts, so the application sets it): streams that share a
// queue run one after the other.  Default for encode: THREE slices (default_encode_slices below).
// Decode runs as ONE slice since round 2: its launch is as long as the longest SPIHT stream of the batch whatever the
// batch size, and the tier-1 decoder is bound by vector issue slots - two half batches side by side only shared them
// (A/B on one box, tools/gpu/ab_dec.sh: 33 GB/s with one slice, 27 with two).
constexpr size_t kDefaultDecodeSlices = 1;
static size_t default_encode_slices()
{
    // Four with eight hardware queues in round 1; two at the end of round 2 (the search loops had moved to the device and every
    // further slice repeated the latency-bound launches for fewer frames: 8.4-8.5 GB/s with two, 6.8-7.5 with four); three
    // in round 3 - the probes are cheaper (early exits, leaner fused levels) and the host no longer compresses every prefix,
    // so a third slice finds idle GPU and idle cores again (alternating runs on one box, tools/gpu/host_sweep.sh: 154-160 ms
    // per step with two, 146-153 with three, 160-165 with four).
    return 3;
}
static size_t slice_engines(ebcc_hip_ctx *ctx, size_t n_frames, const char *env_name, size_t k)
{
    if (const char *e = getenv(env_name)) k = (size_t) std::max(1L, strtol(e, nullptr, 10));
    k = std::min<size_t>(k, 8);
    if (k < 2 || n_frames < 4 * k) return 1;
    const size_t per = (ctx->max_frames + k - 1) / k;
    for (size_t i = 0; i + 1 < k; i++) {                    // slice 0 runs on the context's own engine
        if (i < ctx->lanes.size() && ctx->lanes[i]->max_frames >= per) continue;
        // (a lane made for a finer slicing - encode and decode choose their own - is too small for this one)
        if (i < ctx->lanes.size()) { ebcc_hip_destroy(ctx->lanes[i]); ctx->lanes[i] = nullptr; }
        ebcc_hip_ctx *c = ebcc_hip_create(ctx->device, per, (size_t) ctx->height, (size_t) ctx->width);
        if (i < ctx->lanes.size()) ctx->lanes[i] = c; else if (c) ctx->lanes.push_back(c);
        if (!c) {                                           // out of memory: fall back to the single engine
            ctx->lanes.erase(std::remove(ctx->lanes.begin(), ctx->lanes.end(), (ebcc_hip_ctx *) nullptr), ctx->lanes.end());
            return 1;
        }
    }
    return k;
}

template <class Fn>
static int run_slices(ebcc_hip_ctx *ctx, size_t n_frames, Fn fn, const char *env_name, size_t default_slices)
{
    const size_t k = slice_engines(ctx, n_frames, env_name, default_slices);
    if (k == 1) return fn(ctx, (size_t) 0, n_frames, (SliceGate *) nullptr, 1u);
    const size_t per = (n_frames + k - 1) / k;
    const unsigned started = (unsigned) ((n_frames + per - 1) / per);   // (the last slices of a fine slicing can be empty: 8 slices of 33 frames)
    std::vector<int> rc(k, 0);
    std::vector<std::string> err(k);
    std::vector<SliceGate> gates(k);
    std::vector<std::thread> th;
    for (size_t i = 0; i < k; i++) {
        const size_t lo = i * per, hi = std::min(n_frames, lo + per);
        if (lo >= hi) break;
        th.emplace_back([&, i, lo, hi]() {
            // (a thread has its own current device and its own last-error text: the slice reports through rc / err)
            try {
                EBCC_HIP_CHECK(hipSetDevice(ctx->device));
                if (i > 0) gates[i - 1].wait();
                rc[i] = fn(i == 0 ? ctx : ctx->lanes[i - 1], lo, hi - lo, &gates[i], started);
                if (rc[i]) err[i] = ebcc_hip_last_error();
            } catch (const std::exception &e) {
                rc[i] = 1; err[i] = e.what();
                gates[i].release();                                    // (never leave the next slice waiting)
            }
        });
    }
    for (auto &t : th) t.join();
    int worst = 0;
    for (size_t i = 0; i < k; i++) {
        if (rc[i] && !err[i].empty()) set_error("%s", err[i].c_str());
        worst = std::max(worst, rc[i]);
    }
    return worst;
}

namespace {

// n_frames one-frame chunks as concurrent slices
int run_encode_slices(ebcc_hip_ctx *ctx, const float *d_frames, size_t n_frames, const codec_config_t *cfg, uint8_t **outs, size_t *sizes,
                      GpuPhase *phase)
{
    const size_t n_pix = ctx->n_pix;
    PhaseNote note;
    note.phase = phase;
    struct Over { PhaseNote &n; ~Over() { n.release_once(); } } over{note};                 // (whatever happened to the slices)
    if (phase) phase->acquire();
    return run_slices(ctx, n_frames, [&](ebcc_hip_ctx *c, size_t lo, size_t cnt, SliceGate *next, unsigned slices) {
        note.expect((int) slices);
        return encode_batch(c, d_frames + lo * n_pix, cnt, cfg, outs + lo, sizes + lo, next, 1, nullptr, slices, phase ? &note : nullptr);
    }, "EBCC_HIP_SLICES", default_encode_slices());
}

// the decode counterpart (decode overlaps its two layers on the engine's two streams, decode_batch; a second slice hides
// the host side - parsing, zstd, uploads - of one half behind the kernels of the other when there are hardware queues
// for four streams)
int run_decode_slices(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n_frames, float *d_out)
{
    const size_t n_pix = ctx->n_pix;
    return run_slices(ctx, n_frames, [&](ebcc_hip_ctx *c, size_t lo, size_t cnt, SliceGate *next, unsigned) {
        return decode_batch(c, streams + lo, sizes + lo, cnt, d_out + lo * n_pix, next);
    }, "EBCC_HIP_DECODE_SLICES", kDefaultDecodeSlices);
}

}  // namespace

extern "C" {

void free_buffer(void *p) { if (p) free(p); }

void log_set_level_from_env(void)
{
    g_log_level = 3;
    if (const char *e = getenv("EBCC_LOG_LEVEL")) {
        char *end;
        long v = strtol(e, &end, 10);
        if (*end != '\0') log_warn("Ignore log level: %s, should be in [0, 5]", e);
        else g_log_level = (int) v;
    }
}

void print_config(codec_config_t *c)
{
    static const char *names[] = {"NONE", "MAX_ERROR", "RELATIVE_ERROR"};
    log_info("dimensions:\t(%lu, %lu, %lu)", c->dims[0], c->dims[1], c->dims[2]);
    log_info("chunk dimensions:\t(%lu, %lu, %lu)", c->chunk_dims[0], c->chunk_dims[1], c->chunk_dims[2]);
    log_info("base_cr:\t%f", c->base_cr);
    unsigned t = (unsigned) c->residual_compression_type;
    log_info("residual type:\t%s", t < 3 ? names[t] : "?");
    if (t == MAX_ERROR) log_info("max error:\t%f", c->error);
    if (t == RELATIVE_ERROR) log_info("relative error:\t%f", c->error);
}

int ebcc_hip_host_threads(int slices) { return (int) entropy_threads((unsigned) std::max(1, slices)); }
int ebcc_hip_default_encode_slices(void) { return (int) default_encode_slices(); }

// out[0..6] = usable CPUs (affinity mask cut to the cgroup quota), CPU quota (0: none), zstd core-seconds, seconds the
// slices waited for the zstd workers, bytes compressed, entropy batches, prefix bytes whose compression was proved
// unnecessary - since the last call with reset != 0
// arithmetic identities the kernels rely on, checked on the host (0 = all hold): the division-free s / 65535.0f of the fused
// inverse level for every s in [0, 65535]; v / 255.0f and x / kXi of the residual synthesis (all significands)
int ebcc_hip_selfcheck(void) { return j2k_selfcheck_div65535() + residual_selfcheck_divisions(); }

// the lower bound of zstd_size_lower_bound (0: not applicable - longer than 4 MB, or a libzstd that may split blocks)
size_t ebcc_hip_zstd_floor(const uint8_t *src, size_t n) { return zstd_floor_usable() ? zstd_size_lower_bound(src, n) : 0; }

void ebcc_hip_host_stats(double *out, int reset)
{
    HostStats &h = host_stats();
    if (out) {
        out[0] = (double) usable_cpus(); out[1] = cgroup_cpu_quota();
        out[2] = h.zstd_core_us.load() / 1e6; out[3] = h.zstd_wait_us.load() / 1e6;
        out[4] = (double) h.zstd_bytes.load(); out[5] = (double) h.batches.load(); out[6] = (double) h.skipped_bytes.load();
    }
    if (reset) h.reset();
}

// A fresh allocation of hundreds of MB is unmapped pages: a download into it would fault them in one by one on the copying
// threads.  A few host threads ask for huge pages and touch them meanwhile (while the GPU decodes).
struct Prefault {
    std::vector<std::thread> pool;
    Prefault(void *p, size_t bytes)
    {
        const size_t nthreads = bytes >= ((size_t) 64 << 20) ? std::min<size_t>(16, std::max(1u, (unsigned) entropy_threads(1))) : 0;
        if (nthreads) {                                             // huge pages where the system grants them: 512 x fewer faults
            const uintptr_t a = ((uintptr_t) p + ((size_t) 2 << 20) - 1) & ~(((uintptr_t) 2 << 20) - 1), e = ((uintptr_t) p + bytes) & ~(((uintptr_t) 2 << 20) - 1);
            if (e > a) madvise((void *) a, e - a, MADV_HUGEPAGE);
        }
        try {
            for (size_t t = 0; t < nthreads; t++)
                pool.emplace_back([=]() {
                    volatile char *c = (volatile char *) p;
                    const size_t lo = bytes / nthreads * t, hi = t + 1 == nthreads ? bytes : bytes / nthreads * (t + 1);
                    for (size_t i = lo; i < hi; i += 4096) c[i] = 0;
                });
        } catch (const std::exception &) {}                         // (no thread to be had: the download faults the pages in itself)
    }
    std::mutex m;                                                   // (one device thread per device may come here)
    void join() { std::lock_guard<std::mutex> g(m); for (auto &t : pool) if (t.joinable()) t.join(); }
    ~Prefault() { join(); }
};

// A pageable host array <-> device memory at PCIe speed: through the engine's two pinned bounce buffers with several host
// threads copying (copy_pageable) instead of hipMemcpy's single staging thread (~10 GB/s, and a fresh destination's page
// faults on top) - what ebcc_decode_chunking does for its own output, for callers of the frames API that keep their
// frames in host memory (ebcc_amd/h5_batch.py).  Return 0 = ok.
int ebcc_hip_upload(ebcc_hip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    EBCC_API_TRY
    if (!ctx || !d_dst || !h_src) { set_error("ebcc_hip_upload: null argument"); return 1; }
    std::lock_guard<std::mutex> lock(device_mutex(ctx->device));
    DeviceScope scope(ctx->device);
    copy_pageable(ctx, const_cast<void *>(h_src), d_dst, bytes, false);
    return 0;
    EBCC_API_CATCH(1)
}
int ebcc_hip_download(ebcc_hip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    EBCC_API_TRY
    if (!ctx || !h_dst || !d_src) { set_error("ebcc_hip_download: null argument"); return 1; }
    std::lock_guard<std::mutex> lock(device_mutex(ctx->device));
    DeviceScope scope(ctx->device);
    if (bytes >= ((size_t) 64 << 20)) {                                   // huge pages where the system grants them: 512 x fewer faults on a fresh array
        const uintptr_t a = ((uintptr_t) h_dst + ((size_t) 2 << 20) - 1) & ~(((uintptr_t) 2 << 20) - 1), e = ((uintptr_t) h_dst + bytes) & ~(((uintptr_t) 2 << 20) - 1);
        if (e > a) madvise((void *) a, e - a, MADV_HUGEPAGE);
    }
    copy_pageable(ctx, h_dst, const_cast<void *>(d_src), bytes, true);
    return 0;
    EBCC_API_CATCH(1)
}

// The pages of a host array that is about to receive a download, mapped by several threads (huge pages where granted);
// returns when they are.  Meant to run on a caller's thread beside ebcc_hip_decode_frames.  Return 0 = ok.
int ebcc_hip_prefault(void *h_dst, size_t bytes)
{
    EBCC_API_TRY
    if (!h_dst) { set_error("ebcc_hip_prefault: null argument"); return 1; }
    Prefault pf(h_dst, bytes);
    pf.join();
    return 0;
    EBCC_API_CATCH(1)
}

// Frames in pageable host memory <-> streams through a caller's context, for callers that keep the chunks themselves
// (ebcc_amd/h5_batch.py: HDF5 direct chunk writes / reads): what ebcc_encode_chunking / ebcc_decode_chunking do between the
// array and the EBCK container - uploads / downloads through the pinned bounce buffers, batches of the context's capacity
// on the two alternating engine sets, the output's pages mapped while the GPU decodes.  Any number of frames.  0 = ok;
// on error every stream made so far has been freed.
int ebcc_hip_encode_host_frames(ebcc_hip_ctx *ctx, const float *h_frames, size_t n_frames, const codec_config_t *config,
                                uint8_t **out_streams, size_t *out_sizes)
{
    EBCC_API_TRY
    if (!ctx || !h_frames || !config || !out_streams || !out_sizes || n_frames < 1) { set_error("ebcc_hip_encode_host_frames: bad arguments"); return 1; }
    if (config->dims[0] != 1 || (int) config->dims[1] != ctx->height || (int) config->dims[2] != ctx->width) {
        set_error("ebcc_hip_encode_host_frames: config dims must be (1, %d, %d)", ctx->height, ctx->width);
        return 1;
    }
    std::lock_guard<std::mutex> lock(device_mutex(ctx->device));
    DeviceScope scope(ctx->device);
    log_set_level_from_env();
    if (!zstd().ok) { log_fatal("libzstd not available"); return 1; }
    for (size_t f = 0; f < n_frames; f++) { out_streams[f] = nullptr; out_sizes[f] = 0; }
    const size_t n_pix = ctx->n_pix, cap = ctx->max_frames;
    const int rc = encode_batches_alternating(ctx, n_frames, config, out_streams, out_sizes, [&](ebcc_hip_ctx *set, size_t lo, size_t cnt) {
        float *d = io_buffer(set, cap * n_pix * sizeof(float));
        copy_pageable(set, const_cast<float *>(h_frames + lo * n_pix), d, cnt * n_pix * sizeof(float), false);
        return (const float *) d;
    });
    if (rc) {
        if (rc == 2) set_error("ebcc_hip_encode_host_frames: NaN or Inf in the data");
        for (size_t f = 0; f < n_frames; f++) { free(out_streams[f]); out_streams[f] = nullptr; out_sizes[f] = 0; }
    }
    return rc;
    EBCC_API_CATCH(1)
}
int ebcc_hip_decode_host_frames(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n_frames, float *h_frames_out)
{
    EBCC_API_TRY
    if (!ctx || !streams || !sizes || !h_frames_out || n_frames < 1) { set_error("ebcc_hip_decode_host_frames: bad arguments"); return 1; }
    std::lock_guard<std::mutex> lock(device_mutex(ctx->device));
    DeviceScope scope(ctx->device);
    if (!zstd().ok) { log_fatal("libzstd not available"); return 1; }
    const size_t n_pix = ctx->n_pix, cap = ctx->max_frames;
    Prefault prefault(h_frames_out, n_frames * n_pix * sizeof(float));
    return decode_batches_alternating(ctx, n_frames, cap, [&](ebcc_hip_ctx *set, size_t lo, size_t k) {
        float *d = io_buffer(set, cap * n_pix * sizeof(float));
        const int r = run_decode_slices(set, streams + lo, sizes + lo, k, d);
        if (r) return r;
        prefault.join();
        copy_pageable(set, h_frames_out + lo * n_pix, d, k * n_pix * sizeof(float), true);
        return 0;
    });
    EBCC_API_CATCH(1)
}

// The engines the reference-compatible entry points keep between calls (one per device and frame geometry, with their slice
// engines and second set: tens of GB of device memory for 256 frames of 721 x 1440) are destroyed; the next call makes them
// again.  Contexts made with ebcc_hip_create are the caller's and are not touched.
void ebcc_hip_release_engines(void)
{
    EBCC_API_TRY
    std::vector<std::pair<int, ebcc_hip_ctx *>> victims;
    {
        std::lock_guard<std::mutex> lock(g_map_mutex);
        for (auto &kv : g_ctx) victims.emplace_back(std::get<0>(kv.first), kv.second);
        g_ctx.clear();
    }
    for (auto &v : victims) {
        std::lock_guard<std::mutex> lock(device_mutex(v.first));       // (a call that is using the engine finishes first)
        DeviceScope scope(v.first);
        ebcc_hip_destroy(v.second);
    }
    EBCC_API_CATCH_VOID
}

int ebcc_hip_prepare(ebcc_hip_ctx *ctx, size_t n_frames)
{
    EBCC_API_TRY
    if (!ctx || n_frames < 1 || n_frames > ctx->max_frames) { set_error("ebcc_hip_prepare: bad batch"); return 1; }
    std::lock_guard<std::mutex> lock(device_mutex(ctx->device));
    DeviceScope scope(ctx->device);
    slice_engines(ctx, n_frames, "EBCC_HIP_DECODE_SLICES", kDefaultDecodeSlices);                     // (the coarser slicing first:
    slice_engines(ctx, n_frames, "EBCC_HIP_SLICES", default_encode_slices());                        //  its lanes serve both)
    second_stream(ctx);
    for (ebcc_hip_ctx *c : ctx->lanes) second_stream(c);
    return 0;
    EBCC_API_CATCH(1)
}

int ebcc_hip_encode_frames(ebcc_hip_ctx *ctx, const float *d_frames, size_t n_frames, const codec_config_t *config,
                           uint8_t **out_streams, size_t *out_sizes)
{
    EBCC_API_TRY
    if (!ctx || n_frames < 1 || n_frames > ctx->max_frames) { set_error("ebcc_hip_encode_frames: bad batch"); return 1; }
    if (config->dims[0] != 1 || (int) config->dims[1] != ctx->height || (int) config->dims[2] != ctx->width) {
        set_error("ebcc_hip_encode_frames: config dims must be (1, %d, %d)", ctx->height, ctx->width);
        return 1;
    }
    std::lock_guard<std::mutex> lock(device_mutex(ctx->device));
    DeviceScope scope(ctx->device);
    log_set_level_from_env();
    if (!zstd().ok) { log_fatal("libzstd not available"); return 1; }
    for (size_t f = 0; f < n_frames; f++) { out_streams[f] = nullptr; out_sizes[f] = 0; }   // on error: free the non-null ones
    return run_encode_slices(ctx, d_frames, n_frames, config, out_streams, out_sizes);
    EBCC_API_CATCH(1)
}

// Any number of frames resident on the device, coded in batches of the context's capacity on two alternating engine sets
// (GpuPhase above): the entropy stage of batch k runs beside the kernels of batch k + 1.  Same streams as
// ebcc_hip_encode_frames batch by batch.  On error every stream made so far is freed and the call returns 1.
int ebcc_hip_encode_shard(ebcc_hip_ctx *ctx, const float *d_frames, size_t n_frames, const codec_config_t *config,
                          uint8_t **out_streams, size_t *out_sizes)
{
    EBCC_API_TRY
    if (!ctx || !d_frames || !config || !out_streams || !out_sizes || n_frames < 1) { set_error("ebcc_hip_encode_shard: bad arguments"); return 1; }
    if (config->dims[0] != 1 || (int) config->dims[1] != ctx->height || (int) config->dims[2] != ctx->width) {
        set_error("ebcc_hip_encode_shard: config dims must be (1, %d, %d)", ctx->height, ctx->width);
        return 1;
    }
    std::lock_guard<std::mutex> lock(device_mutex(ctx->device));
    DeviceScope scope(ctx->device);
    log_set_level_from_env();
    if (!zstd().ok) { log_fatal("libzstd not available"); return 1; }
    for (size_t f = 0; f < n_frames; f++) { out_streams[f] = nullptr; out_sizes[f] = 0; }
    const size_t n_pix = ctx->n_pix;
    const int rc = encode_batches_alternating(ctx, n_frames, config, out_streams, out_sizes,
                                              [&](ebcc_hip_ctx *, size_t lo, size_t) { return d_frames + lo * n_pix; });
    if (rc)
        for (size_t f = 0; f < n_frames; f++) { free(out_streams[f]); out_streams[f] = nullptr; out_sizes[f] = 0; }
    return rc;
    EBCC_API_CATCH(1)
}

int ebcc_hip_decode_frames(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n_frames,
                           float *d_frames_out)
{
    EBCC_API_TRY
    if (!ctx || n_frames < 1 || n_frames > ctx->max_frames) { set_error("ebcc_hip_decode_frames: bad batch"); return 1; }
    std::lock_guard<std::mutex> lock(device_mutex(ctx->device));
    DeviceScope scope(ctx->device);
    if (!zstd().ok) { log_fatal("libzstd not available"); return 1; }
    return run_decode_slices(ctx, streams, sizes, n_frames, d_frames_out);
    EBCC_API_CATCH(1)
}

// Any number of streams decoded to consecutive frames on the device, in batches of the context's capacity on the two
// engine sets side by side (decode_batches_alternating).  Same frames as ebcc_hip_decode_frames batch by batch.
int ebcc_hip_decode_shard(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n_frames,
                          float *d_frames_out)
{
    EBCC_API_TRY
    if (!ctx || !streams || !sizes || !d_frames_out || n_frames < 1) { set_error("ebcc_hip_decode_shard: bad arguments"); return 1; }
    std::lock_guard<std::mutex> lock(device_mutex(ctx->device));
    DeviceScope scope(ctx->device);
    if (!zstd().ok) { log_fatal("libzstd not available"); return 1; }
    const size_t n_pix = ctx->n_pix;
    return decode_batches_alternating(ctx, n_frames, ctx->max_frames, [&](ebcc_hip_ctx *set, size_t lo, size_t k) {
        return run_decode_slices(set, streams + lo, sizes + lo, k, d_frames_out + lo * n_pix);
    });
    EBCC_API_CATCH(1)
}

size_t ebcc_encode(float *data, codec_config_t *config, uint8_t **out_buffer)
{
    log_set_level_from_env();
    if (!dims_are_valid(config->dims)) {
        log_fatal("Invalid EBCC dimensions: product(dims[0..1]) and dims[2] must be between %d and %d",
                  EBCC_MIN_INTERNAL_IMAGE_DIM, EBCC_MAX_INTERNAL_IMAGE_DIM);
        return 0;
    }
    print_config(config);
    if (config->dims[0] != 1 && !tile_height_supported(config->dims[1])) {
        // the reference codes such a chunk as one tiled JPEG 2000 image (src/ebcc_codec.c:121-125,167-171) and crashes
        // inside OpenJPEG when the tiles are too small for 6 resolutions
        log_fatal("chunks holding %lu frames of %lu rows are not supported (a chunk of several frames is a tiled JPEG 2000 "
                  "image: tiles need at least 32 rows)", config->dims[0], config->dims[1]);
        return 0;
    }
    size_t size = 0;
    uint8_t *o = nullptr;
    const int rcode = encode_host_frames(resolve_device(), data, 1, (int) config->dims[1], (int) config->dims[2], config, &o, &size, config->dims[0]);
    if (rcode == 2) exit(1);                                                                   // check_nan_inf, :598-605
    if (rcode) { free(o); return 0; }
    *out_buffer = o;
    return size;
}

size_t ebcc_decode(uint8_t *data, size_t data_size, float **out_buffer)
{
    ParsedFrame hd;                                                                             // both stream formats
    if (!parse_frame(data, data_size, hd)) return 0;
    const uint8_t *tail = hd.tail;
    if (hd.const_field) {                                                                      // :1265-1281 / :1175-1183, no device work
        uint64_t cnt;
        memcpy(&cnt, tail, 8);
        float *o = (float *) malloc(cnt * sizeof(float));
        if (!o) { log_fatal("out of memory"); return 0; }
        for (uint64_t i = 0; i < cnt; i++) o[i] = hd.minv;
        *out_buffer = o;
        return (size_t) cnt;
    }
    int H = 0, W = 0, tw = 0, th = 0;
    if (!j2k_peek_dims(tail, hd.tail_size, &W, &H, &tw, &th) || H < 1 || W < 1 || H > 2047 || W > 2047 || th < 1 || tw != W || H % th != 0) {
        log_fatal("Invalid encoded data: no usable JPEG 2000 codestream in the tail");
        return 0;
    }
    const size_t tiles = (size_t) (H / th);
    if (tiles > 1 && !tile_height_supported((size_t) th)) {
        log_fatal("streams with %zu tiles of %d rows are not supported", tiles, th);
        return 0;
    }
    const int device = resolve_device();
    try {
        std::lock_guard<std::mutex> lock(device_mutex(device));
        DeviceScope scope(device);
        ebcc_hip_ctx *ctx = nullptr, *rc = nullptr;
        if (!chunk_engines(device, th, W, 1, tiles, &ctx, &rc)) { log_fatal("no MI355X engine available: %s", ebcc_hip_last_error()); return 0; }
        const size_t n_pix = (size_t) H * W;
        float *d = io_buffer(ctx, n_pix * sizeof(float));
        const uint8_t *sp = data;
        int rcode = tiles > 1 ? decode_tiled(ctx, rc, &sp, &data_size, 1, tiles, d) : decode_batch(ctx, &sp, &data_size, 1, d);
        if (rcode) return 0;
        // :1126-1128: honour a caller-provided buffer
        float *o = *out_buffer ? *out_buffer : (float *) malloc(n_pix * sizeof(float));
        if (!o) { log_fatal("out of memory"); return 0; }
        EBCC_HIP_CHECK(hipMemcpy(o, d, n_pix * sizeof(float), hipMemcpyDeviceToHost));
        *out_buffer = o;
        return n_pix;
    } catch (const std::exception &e) {
        log_fatal("MI355X engine failure: %s", e.what());
        set_error("%s", e.what());
        return 0;
    }
}

// ---- EBCK chunk container (:920-1052) -------------------------------------------------------------
static size_t cdiv(size_t a, size_t b) { return a / b + (a % b != 0); }

size_t ebcc_encode_chunking(float *data, codec_config_t *config, uint8_t **out_buffer)
{
    log_set_level_from_env();
    size_t cd[3];
    bool all_zero = true;
    for (int i = 0; i < 3; i++) { cd[i] = config->chunk_dims[i]; if (cd[i]) all_zero = false; }
    if (all_zero) for (int i = 0; i < 3; i++) cd[i] = config->dims[i];
    if (!dims_are_valid(cd)) {
        log_fatal("Invalid chunking dimensions: product(chunk_dims[0..1]) and chunk_dims[2] must be between %d and %d",
                  EBCC_MIN_INTERNAL_IMAGE_DIM, EBCC_MAX_INTERNAL_IMAGE_DIM);
        return 0;
    }
    size_t cnt[3];
    for (int i = 0; i < 3; i++) {
        if (config->dims[i] == 0 || cd[i] == 0) { log_fatal("Invalid chunking dimensions: dims and chunk_dims must be non-zero"); return 0; }
        cnt[i] = cdiv(config->dims[i], cd[i]);
    }
    if (cd[0] != 1 && !tile_height_supported(cd[1])) {
        log_fatal("chunks holding %lu frames of %lu rows are not supported (a chunk of several frames is a tiled JPEG 2000 "
                  "image: tiles need at least 32 rows); use chunk_dims[0] = 1", cd[0], cd[1]);
        return 0;
    }
    const size_t csize = cd[0] * cd[1] * cd[2], nchunks = cnt[0] * cnt[1] * cnt[2];
    const size_t total = config->dims[0] * config->dims[1] * config->dims[2];
    const size_t padded = csize * nchunks;
    if (padded > total && padded - total > total / 10)
        log_warn("Chunk padding adds %lu values over %lu real values (%.2f%%)", padded - total, total,
                 ((double) (padded - total) / (double) total) * 100.0);
    // the chunks in C order of chunk index (:311-318), edge chunks padded by index clamping (:339-351).  Chunks that
    // are whole frames of the array need no copy at all.
    ChunkBox box;
    for (int i = 0; i < 3; i++) { box.dims[i] = config->dims[i]; box.cd[i] = cd[i]; box.cnt[i] = cnt[i]; }
    bool in_place = box.slabs();
    for (size_t cl = 0; cl < nchunks && in_place; cl++) in_place = box.inside(cl);
    std::vector<float> gathered;
    if (!in_place) {
        gathered.resize(nchunks * csize);
        for (size_t cl = 0; cl < nchunks; cl++) box.gather(data, cl, gathered.data() + cl * csize);
    }
    const float *chunk_data = in_place ? data : gathered.data();
    codec_config_t cc = *config;
    for (int i = 0; i < 3; i++) { cc.dims[i] = cd[i]; cc.chunk_dims[i] = 0; }
    std::vector<uint8_t *> outs(nchunks, nullptr);
    std::vector<size_t> sizes(nchunks, 0);
    const int rcode = run_on_devices(nchunks, [&](int device, size_t first, size_t count) {
        return encode_host_frames(device, chunk_data + first * csize, count, (int) cd[1], (int) cd[2], &cc, outs.data() + first, sizes.data() + first, cd[0]);
    });
    if (rcode == 2) exit(1);                                                                   // check_nan_inf, :598-605
    if (rcode) {
        for (auto p : outs) free(p);
        return 0;
    }
    size_t len = sizeof(ChunkHeader);
    for (size_t c = 0; c < nchunks; c++) len += 8 + sizes[c];
    uint8_t *o = (uint8_t *) malloc(len), *p = o;
    if (!o) { log_fatal("out of memory"); for (auto q : outs) free(q); return 0; }
    ChunkHeader hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, EBCC_CHUNKING_HEADER_MAGIC, 4);
    hd.version = EBCC_CHUNKING_HEADER_VERSION; hd.ndims = NDIMS;
    for (int i = 0; i < 3; i++) { hd.dims[i] = config->dims[i]; hd.chunk_dims[i] = cd[i]; }
    hd.num_chunks = nchunks; hd.chunk_size = csize;
    memcpy(p, &hd, sizeof hd); p += sizeof hd;
    for (size_t c = 0; c < nchunks; c++) {
        uint64_t nb = sizes[c];
        memcpy(p, &nb, 8); p += 8;
        memcpy(p, outs[c], sizes[c]); p += sizes[c];
        free(outs[c]);
    }
    *out_buffer = o;
    return len;
}

size_t ebcc_encode_chunking_compat(float *data, codec_config_t *config, uint8_t **out_buffer)
{
    // :1054-1090
    log_set_level_from_env();
    codec_config_t c = *config;
    if (!c.chunk_dims[0] && !c.chunk_dims[1] && !c.chunk_dims[2]) {
        c.chunk_dims[0] = 1;
        c.chunk_dims[1] = c.dims[1] > EBCC_MAX_INTERNAL_IMAGE_DIM ? 1024 : c.dims[1];
        c.chunk_dims[2] = c.dims[2] > EBCC_MAX_INTERNAL_IMAGE_DIM ? 1024 : c.dims[2];
        log_info("ebcc_encode_chunking_compat chunk dimensions: (%lu, %lu, %lu)", c.chunk_dims[0], c.chunk_dims[1], c.chunk_dims[2]);
    }
    if (c.residual_compression_type == RELATIVE_ERROR) {
        size_t total = c.dims[0] * c.dims[1] * c.dims[2];
        if (total == 0) { log_fatal("Invalid EBCC dimensions: size overflow or zero-sized data"); return 0; }
        float mn = data[0], mx = data[0];
        for (size_t i = 0; i < total; i++) {
            if (std::isnan(data[i]) || std::isinf(data[i])) { log_fatal("NaN or Inf found in data at index %lu", i); exit(1); }
            if (data[i] > mx) mx = data[i];
            if (data[i] < mn) mn = data[i];
        }
        c.error *= mx - mn;                                                                    // global range, :1085
        c.residual_compression_type = MAX_ERROR;
    }
    return ebcc_encode_chunking(data, &c, out_buffer);
}

size_t ebcc_decode_chunking(uint8_t *data, size_t data_size, float **out_buffer)
{
    // :1322-1449
    log_set_level_from_env();
    if (data_size < sizeof(ChunkHeader) || memcmp(data, EBCC_CHUNKING_HEADER_MAGIC, 4) != 0) return ebcc_decode(data, data_size, out_buffer);
    ChunkHeader hd;
    memcpy(&hd, data, sizeof hd);
    if (hd.version != EBCC_CHUNKING_HEADER_VERSION) { log_fatal("Unsupported EBCC chunking header version: %u", hd.version); return 0; }
    if (hd.ndims != NDIMS) { log_fatal("Unsupported EBCC chunking dimensionality: %u", hd.ndims); return 0; }
    size_t dims[3], cd[3], cnt[3];
    for (int i = 0; i < 3; i++) { dims[i] = hd.dims[i]; cd[i] = hd.chunk_dims[i]; }
    if (!dims_are_valid(cd)) { log_fatal("Invalid chunked EBCC data: bad chunk dimensions"); return 0; }
    for (int i = 0; i < 3; i++) {
        if (!dims[i] || !cd[i]) { log_fatal("Invalid chunked EBCC data: dims and chunk_dims must be non-zero"); return 0; }
        cnt[i] = cdiv(dims[i], cd[i]);
    }
    const size_t csize = cd[0] * cd[1] * cd[2], nchunks = cnt[0] * cnt[1] * cnt[2], total = dims[0] * dims[1] * dims[2];
    if (hd.chunk_size != csize || hd.num_chunks != nchunks) { log_fatal("Invalid chunked EBCC data: inconsistent chunk metadata"); return 0; }
    if (cd[0] != 1 && !tile_height_supported(cd[1])) {
        log_fatal("chunks holding %lu frames of %lu rows are not supported", cd[0], cd[1]);
        return 0;
    }
    std::vector<const uint8_t *> ptrs(nchunks);
    std::vector<size_t> lens(nchunks);
    const uint8_t *p = data + sizeof hd, *end = data + data_size;
    for (size_t c = 0; c < nchunks; c++) {
        uint64_t nb;
        if ((size_t) (end - p) < 8) { log_fatal("Invalid chunked EBCC data: missing chunk size"); return 0; }
        memcpy(&nb, p, 8); p += 8;
        if (nb > (size_t) (end - p)) { log_fatal("Invalid chunked EBCC data: truncated chunk payload"); return 0; }
        ptrs[c] = p; lens[c] = nb; p += nb;
    }
    if (p != end) { log_fatal("Invalid chunked EBCC data: trailing payload bytes"); return 0; }
    const int H = (int) cd[1], W = (int) cd[2];
    ChunkBox box;
    for (int i = 0; i < 3; i++) { box.dims[i] = dims[i]; box.cd[i] = cd[i]; box.cnt[i] = cnt[i]; }
    bool in_place = box.slabs();                               // chunks = whole frames of the array: decode straight into it
    for (size_t cl = 0; cl < nchunks && in_place; cl++) in_place = box.inside(cl);
    float *o = (float *) malloc(total * sizeof(float));
    if (!o) { log_fatal("Failed to allocate chunked EBCC decode output"); return 0; }
    std::vector<float> chunks;
    if (!in_place) chunks.resize(nchunks * csize);
    float *h_chunks = in_place ? o : chunks.data();
    // a fresh allocation of this size is unmapped pages: the download would fault them in one by one on the copying thread.
    // A few host threads touch them while the GPU decodes (the reference-compatible output must be a malloc'd buffer).
    Prefault prefault(in_place ? (void *) o : nullptr, in_place ? total * sizeof(float) : 0);
    const size_t tiles = cd[0];
    const int rcode = run_on_devices(nchunks, [&](int device, size_t first, size_t count) {
        try {
            std::lock_guard<std::mutex> lock(device_mutex(device));
            DeviceScope scope(device);
            const size_t cap = std::min(count, batch_capacity(csize));
            ebcc_hip_ctx *ctx = nullptr, *rc = nullptr;
            if (!chunk_engines(device, H, W, cap, tiles, &ctx, &rc)) { log_fatal("no MI355X engine available: %s", ebcc_hip_last_error()); return 1; }
            if (!zstd().ok) { log_fatal("libzstd not available"); return 1; }
            // (one download per batch: copies issued from inside the slices slowed them down)
            auto one_batch = [&](ebcc_hip_ctx *set, size_t lo, size_t k) {
                PhaseTimer pt;
                float *d = io_buffer(set, cap * csize * sizeof(float));
                pt.mark("decode_chunking: engine, device image");
                const size_t done = first + lo;
                const int r = tiles > 1 ? decode_tiled(set, rc, ptrs.data() + done, lens.data() + done, k, tiles, d)
                                        : run_decode_slices(set, ptrs.data() + done, lens.data() + done, k, d);
                if (r) return r;
                pt.mark("decode_chunking: decode");
                prefault.join();
                pt.mark("decode_chunking: output pages");
                copy_pageable(set, h_chunks + done * csize, d, k * csize * sizeof(float), true);
                pt.mark("decode_chunking: download");
                return 0;
            };
            // one-frame chunks in several batches: on two alternating engine sets, a batch's download beside the next one's kernels
            if (tiles == 1 && ctx->max_frames == cap) return decode_batches_alternating(ctx, count, cap, one_batch);
            for (size_t lo = 0; lo < count; lo += cap) {
                const int r = one_batch(ctx, lo, std::min(cap, count - lo));
                if (r) return r;
            }
            return 0;
        } catch (const std::exception &e) {
            log_fatal("MI355X engine failure: %s", e.what());
            set_error("%s", e.what());
            return 1;
        }
    });
    if (rcode) { prefault.join(); free(o); return 0; }         // (the page-touching threads write into `o` until they are joined)
    if (!in_place)
        for (size_t cl = 0; cl < nchunks; cl++) box.scatter(chunks.data() + cl * csize, cl, o);          // :353-370
    *out_buffer = o;
    return total;
}

// ---- HDF5 filter plugin, /root/reference/src/h5z_ebcc.c ---------------------------------------------
#define H5Z_FLAG_REVERSE 0x0100
typedef size_t (*H5Z_func_t)(unsigned int, size_t, const unsigned int[], size_t, size_t *, void **);
struct H5Z_class2_t {
    int version; int id; unsigned encoder_present; unsigned decoder_present; const char *name;
    void *can_apply; void *set_local; H5Z_func_t filter;
};

void populate_config(codec_config_t *config, size_t cd_nelmts, const unsigned int cd_values[], size_t buf_size)
{
    // h5z_ebcc.c:38-93, including exit(1) on invalid parameters
    if (cd_nelmts < 4) { log_fatal("EBCC filter requires at least 4 configuration values, got %lu", cd_nelmts); exit(1); }
    for (int i = 0; i < NDIMS; i++) config->chunk_dims[i] = 0;
    size_t th = cd_values[0], tw = cd_values[1];
    if (th < EBCC_MIN_INTERNAL_IMAGE_DIM || tw < EBCC_MIN_INTERNAL_IMAGE_DIM || th > EBCC_MAX_INTERNAL_IMAGE_DIM ||
        tw > EBCC_MAX_INTERNAL_IMAGE_DIM) {
        log_fatal("Tile size %lu x %lu is invalid, each dimension must be between %d and %d", th, tw,
                  EBCC_MIN_INTERNAL_IMAGE_DIM, EBCC_MAX_INTERNAL_IMAGE_DIM);
        exit(1);
    }
    size_t tile = th * tw;
    config->dims[0] = buf_size / sizeof(float);
    if (config->dims[0] < tile) { log_fatal("Buffer size %lu is smaller than the tile size %lu x %lu = %lu", config->dims[0], th, tw, tile); exit(1); }
    if (config->dims[0] % tile != 0) { log_fatal("Buffer size %lu is not divisible by the tile size %lu x %lu = %lu", config->dims[0], th, tw, tile); exit(1); }
    for (size_t i = 0; i < 2; i++) {
        size_t cur = cd_values[i];
        config->dims[0] /= cur;
        config->dims[i + 1] = cur;
    }
    if (config->dims[1] != 0 && config->dims[0] > EBCC_MAX_INTERNAL_IMAGE_DIM / config->dims[1]) {
        log_fatal("Flattened EBCC image height %lu x %lu exceeds the limit of %d", config->dims[0], config->dims[1],
                  EBCC_MAX_INTERNAL_IMAGE_DIM);
        exit(1);
    }
    config->base_cr = u2f(cd_values[2]);
    config->residual_compression_type = (residual_t) cd_values[3];
    if (config->residual_compression_type == MAX_ERROR || config->residual_compression_type == RELATIVE_ERROR) {
        if (cd_nelmts != 5) { log_fatal("EBCC filter: modes 1 and 2 need 5 configuration values"); exit(1); }
        config->error = u2f(cd_values[4]);
    }
}

static size_t H5Z_filter_ebcc(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes,
                              size_t *buf_size, void **buf)
{
    if (flags & H5Z_FLAG_REVERSE) {
        float *out = nullptr;
        *buf_size = ebcc_decode((uint8_t *) *buf, nbytes, &out);                               // element count (quirk Q1)
        free_buffer(*buf);
        *buf = out;
        return *buf_size;
    }
    codec_config_t config;
    memset(&config, 0, sizeof config);
    populate_config(&config, cd_nelmts, cd_values, *buf_size);
    uint8_t *out = nullptr;
    *buf_size = ebcc_encode((float *) *buf, &config, &out);
    free_buffer(*buf);
    *buf = out;
    return *buf_size;
}

static const H5Z_class2_t H5Z_EBCC[1] = {{1, 308, 1, 1, "HDF5 EBCC filter L&L", nullptr, nullptr, H5Z_filter_ebcc}};

int H5PLget_plugin_type(void) { return 0; }            // H5PL_TYPE_FILTER
const void *H5PLget_plugin_info(void) { return H5Z_EBCC; }

}  // extern "C"
