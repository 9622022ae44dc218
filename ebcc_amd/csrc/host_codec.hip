// host_codec.hip - the reference's C API (include/ebcc_codec.h) on top of the MI355X engine: engines per device and
// frame geometry, concurrent slices and alternating engine sets of a batch, host <-> device copies, the EBCK chunk
// container (/root/reference/src/ebcc_codec.c:920-1090, :1322-1449), and the batch entry points of include/ebcc_hip.h.
// The frame codec itself is batch_codec.hip, the HDF5 plugin h5z_filter.hip, the host services host_pool.hip (host.hpp).
// There is no CPU fallback: without a HIP device every entry point fails loudly.
#include "host.hpp"

using namespace ebcc;

namespace {

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
        PhaseTimer pt;
        if (!chunk_engines(device, H, W, cap, tiles, &ctx, &rc)) { log_fatal("no MI355X engine available: %s", ebcc_hip_last_error()); return 1; }
        pt.mark("host frames: engine");
        // (a batch is uploaded in one go: uploads issued from inside the slices slow every slice down - measured in round 1 with
        //  pageable copies, 5.6 against 3.7 GB/s, and again in round 2 through the bounce buffers, 7.1 against 6.5)
        if (tiles == 1 && ctx->max_frames == cap)                   // one-frame chunks: concurrent slices, batches on alternating engine sets
            return encode_batches_alternating(ctx, n, cfg, outs, sizes, [&](ebcc_hip_ctx *set, size_t lo, size_t cnt) {
                float *d = io_buffer(set, cap * n_pix * sizeof(float));
                copy_pageable(set, const_cast<float *>(data + lo * n_pix), d, cnt * n_pix * sizeof(float), false);
                pt.mark("host frames: upload");
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
constexpr size_t kSliceFromFrames = 96;                             // smaller batches run as one slice unless the environment says otherwise
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
    // (a batch below about a hundred frames is a chain whatever it is cut into - 43 frames of 721 x 1440: 53.9 ms per step as
    //  one slice, 56.8 as two, 60.3 as three; 85 frames: 75.8 / 77.4 / 76.4; 128 frames: 85.9 / 84.6 / 83.0, tools/gpu/slices_frames.sh)
    if (const char *e = getenv(env_name)) k = (size_t) std::max(1L, strtol(e, nullptr, 10));
    else if (n_frames < kSliceFromFrames) k = 1;
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

namespace ebcc {

int cached_encode_host_frames(const float *h_frames, size_t n, int H, int W, const codec_config_t *cfg, uint8_t **outs, size_t *sizes)
{
    return encode_host_frames(resolve_device(), h_frames, n, H, W, cfg, outs, sizes);
}

int cached_decode_host_frames(const uint8_t *const *streams, const size_t *sizes, size_t n, int H, int W, float *h_out)
{
    try {
        const int device = resolve_device();
        std::lock_guard<std::mutex> lock(device_mutex(device));
        DeviceScope scope(device);
        if (!zstd().ok) { log_fatal("libzstd not available"); return 1; }
        const size_t n_pix = (size_t) H * W, cap = std::min(n, batch_capacity(n_pix));
        ebcc_hip_ctx *ctx = get_context(device, H, W, cap);
        if (!ctx) { log_fatal("no MI355X engine available: %s", ebcc_hip_last_error()); return 1; }
        return decode_batches_alternating(ctx, n, ctx->max_frames, [&](ebcc_hip_ctx *set, size_t lo, size_t k) {
            float *d = io_buffer(set, ctx->max_frames * n_pix * sizeof(float));
            const int r = run_decode_slices(set, streams + lo, sizes + lo, k, d);
            if (r) return r;
            copy_pageable(set, h_out + lo * n_pix, d, k * n_pix * sizeof(float), true);
            return 0;
        });
    } catch (const std::exception &e) {
        log_fatal("MI355X engine failure: %s", e.what());
        set_error("%s", e.what());
        return 1;
    }
}

}  // namespace ebcc

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
int ebcc_hip_encode_slices_for(size_t n_frames)
{
    size_t k = default_encode_slices();
    if (const char *e = getenv("EBCC_HIP_SLICES")) k = (size_t) std::max(1L, strtol(e, nullptr, 10));
    else if (n_frames < kSliceFromFrames) k = 1;
    k = std::min<size_t>(k, 8);
    return k < 2 || n_frames < 4 * k ? 1 : (int) k;
}

// out[0..6] = usable CPUs (affinity mask cut to the cgroup quota), CPU quota (0: none), zstd core-seconds, seconds the
// slices waited for the zstd workers, bytes compressed, entropy batches, prefix bytes whose compression was proved
// unnecessary - since the last call with reset != 0
// arithmetic identities the kernels rely on, checked on the host (0 = all hold): the division-free s / 65535.0f of the fused
// inverse level for every s in [0, 65535]; v / 255.0f and x / kXi of the residual synthesis (all significands)
int ebcc_hip_selfcheck(void) { return j2k_selfcheck_div65535() + residual_selfcheck_divisions(); }
void ebcc_hip_plan_decode_lanes(const int *table, int n_code_blocks, int out[4]) { plan_decode_lanes(table, std::max(0, n_code_blocks), out); }

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

// The second engine set of a caller's context (made by the first shard / host-frames call of more than one batch: as much
// device memory as the context itself) is destroyed; the next such call makes it again.
void ebcc_hip_release_second_set(ebcc_hip_ctx *ctx)
{
    EBCC_API_TRY
    if (!ctx) return;
    std::lock_guard<std::mutex> lock(device_mutex(ctx->device));
    DeviceScope scope(ctx->device);
    if (ctx->twin) { ebcc_hip_destroy(ctx->twin); ctx->twin = nullptr; }
    ctx->twin_failed = false;
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
        PhaseTimer pt;
        float *d = io_buffer(ctx, n_pix * sizeof(float));
        const uint8_t *sp = data;
        int rcode = tiles > 1 ? decode_tiled(ctx, rc, &sp, &data_size, 1, tiles, d) : decode_batch(ctx, &sp, &data_size, 1, d);
        if (rcode) return 0;
        pt.mark("ebcc_decode: decode_batch");
        // :1126-1128: honour a caller-provided buffer
        float *o = *out_buffer ? *out_buffer : (float *) malloc(n_pix * sizeof(float));
        if (!o) { log_fatal("out of memory"); return 0; }
        EBCC_HIP_CHECK(hipMemcpy(o, d, n_pix * sizeof(float), hipMemcpyDeviceToHost));
        pt.mark("ebcc_decode: download");
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

}  // extern "C"
