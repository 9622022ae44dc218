// host.hpp - what the host-side translation units of the library share:
//   host_pool.hip    logging, the dlopen'd libzstd and the provable lower bound of its output, the CPU budget, the
//                    process-wide worker pool of the entropy stage (HostPool) and its accounting
//   batch_codec.hip  one device batch: the frame codec of /root/reference/src/ebcc_codec.c:607-918 (encode_batch) and
//                    :1215-1320 (decode_batch, decode_tiled) on an engine, with its three search loops
//   host_codec.hip   engines per device and geometry, slices and alternating engine sets, host <-> device copies, the EBCK
//                    container, and the C API of include/ebcc_codec.h / include/ebcc_hip.h
//   h5z_filter.hip   the HDF5 filter plugin (id 308) of /root/reference/src/h5z_ebcc.c
#pragma once

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

// ---- logging (reference src/log/, level from EBCC_LOG_LEVEL, default WARN; src/ebcc_codec.c:431-448): host_pool.hip
extern int g_log_level;
void log_at(int level, const char *fmt, ...);
#define log_trace(...) ::ebcc::log_at(0, __VA_ARGS__)
#define log_info(...) ::ebcc::log_at(2, __VA_ARGS__)
#define log_warn(...) ::ebcc::log_at(3, __VA_ARGS__)
#define log_fatal(...) ::ebcc::log_at(5, __VA_ARGS__)

// ---- zstd stays on the host (north star); dlopen'd so the library has no link-time dependency: host_pool.hip
struct Zstd {
    size_t (*bound)(size_t) = nullptr;
    size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
    size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    unsigned (*version)(void) = nullptr;
    bool ok = false;
    Zstd();
};
Zstd &zstd();
constexpr size_t kZstdFloorMaxBytes = (size_t) 4 << 20;
bool zstd_floor_usable();
size_t zstd_size_lower_bound(const uint8_t *src, size_t n);

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
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

inline bool dims_are_valid(const size_t d[3])
{
    // :286-297
    if (d[0] == 0 || d[1] == 0) return false;
    size_t hh = d[0] * d[1];
    if (hh / d[0] != d[1]) return false;
    return hh >= EBCC_MIN_INTERNAL_IMAGE_DIM && hh <= EBCC_MAX_INTERNAL_IMAGE_DIM && d[2] >= EBCC_MIN_INTERNAL_IMAGE_DIM &&
           d[2] <= EBCC_MAX_INTERNAL_IMAGE_DIM;
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

// The engine's second stream is created on first use: every stream beyond the runtime's few hardware queues
// shares one, and kernels that share a queue run one after the other.
inline hipStream_t second_stream(ebcc_hip_ctx *c)
{
    if (!c->stream2) EBCC_HIP_CHECK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    return c->stream2;
}

// ---- the CPU budget of this process (host_pool.hip)
double cgroup_cpu_quota();
unsigned usable_cpus();
unsigned entropy_threads_for(unsigned cpus, unsigned local_world, unsigned slices);
unsigned entropy_threads(unsigned slices = 1);

// Host-side accounting of the entropy stage since the last reset (ebcc_hip_host_stats: bench.py prints it per rank so that
// a multi-GPU run that is bound by the host's CPUs can be told from one that is bound by the GPUs).
struct HostStats {
    std::atomic<long long> zstd_core_us{0}, zstd_wait_us{0}, zstd_bytes{0}, batches{0}, skipped_bytes{0};
    void add(long long core_us, long long wait_us, long long bytes) { zstd_core_us += core_us; zstd_wait_us += wait_us; zstd_bytes += bytes; batches++; }
    void reset() { zstd_core_us = 0; zstd_wait_us = 0; zstd_bytes = 0; batches = 0; skipped_bytes = 0; }
};
HostStats &host_stats();

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

// ---- one device batch (batch_codec.hip)
// ebcc_encode for a batch of device-resident chunks: `n` chunks of `tiles` frames each (tiles == 1: the frame-per-chunk
// case); `rctx`: residual engine for the stacked chunk image when tiles > 1.  Returns 0, 1 (error) or 2 (NaN/Inf).
int encode_batch(ebcc_hip_ctx *ctx, const float *d_frames, size_t n, const codec_config_t *cfg, uint8_t **outs, size_t *sizes,
                 SliceGate *next = nullptr, size_t tiles = 1, ebcc_hip_ctx *rctx = nullptr, unsigned slices = 1, PhaseNote *note = nullptr);
// One frame stream, either format: the 48-byte "EBCC" header (:190-202, :1234-1260) or the legacy header-less prefix
struct ParsedFrame {
    float minv = 0, maxv = 0, rmin = 0, rmax = 0;
    bool const_field = false;
    size_t coeffs_size = 0, compressed_size = 0, tail_size = 0;
    const uint8_t *z = nullptr, *tail = nullptr;
};
bool parse_frame(const uint8_t *d, size_t len, ParsedFrame &pf);
// ebcc_decode for a batch of single-frame EBCC streams -> device buffer d_out [n][H*W]
int decode_batch(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n, float *d_out, SliceGate *next = nullptr);
// chunks of several frames (one tiled codestream per chunk): frame heights such a chunk can have, heights for which every
// tile has the geometry of a tile at the origin, and the decode
bool tile_height_supported(size_t h);
bool tile_geometry_uniform(size_t h);
int decode_tiled(ebcc_hip_ctx *ctx, ebcc_hip_ctx *rc, const uint8_t *const *streams, const size_t *sizes, size_t n, size_t tiles, float *d_out);

// ---- frames in host memory <-> frame streams on the engines the reference-compatible entry points keep (host_codec.hip):
//      any number of one-frame chunks of H x W, batches of EBCC_HIP_MAX_BATCH on alternating engine sets.  0 ok, 1 error, 2 NaN/Inf
int cached_encode_host_frames(const float *h_frames, size_t n, int H, int W, const codec_config_t *cfg, uint8_t **outs, size_t *sizes);
int cached_decode_host_frames(const uint8_t *const *streams, const size_t *sizes, size_t n, int H, int W, float *h_out);

}  // namespace ebcc
