// engine.hip - context management and the residual-layer C-ABI (include/ebcc_hip.h).
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <mutex>
#include <string>
#include <cstring>
#include <vector>

#include "../../include/ebcc_hip.h"
#include "search.hpp"

namespace ebcc {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    std::fprintf(stderr, "ebcc-hip: %s\n", buf);
}

void wait_stream(hipStream_t s)
{
    // events come from a per-device free list (an event belongs to the device that was current when it was made; slice
    // threads are short-lived, so nothing is kept per thread)
    static std::mutex m;
    static std::vector<hipEvent_t> spare[64];
    int dev = 0;
    EBCC_HIP_CHECK(hipGetDevice(&dev));
    hipEvent_t e = nullptr;
    {
        std::lock_guard<std::mutex> lock(m);
        auto &v = spare[dev & 63];
        if (!v.empty()) { e = v.back(); v.pop_back(); }
    }
    if (!e) EBCC_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventBlockingSync | hipEventDisableTiming));
    struct Return { hipEvent_t e; int dev; ~Return() { std::lock_guard<std::mutex> lock(m); spare[dev & 63].push_back(e); } } back{e, dev};
    EBCC_HIP_CHECK(hipEventRecord(e, s));
    EBCC_HIP_CHECK(hipEventSynchronize(e));
}

void clear_error() { g_last_error.clear(); }

hipError_t device_malloc(void **p, size_t bytes)
{
    static const long fail_at = getenv("EBCC_HIP_FAIL_ALLOC") ? strtol(getenv("EBCC_HIP_FAIL_ALLOC"), nullptr, 10) : 0;
    static std::atomic<long> count{0};
    if (fail_at > 0 && ++count == fail_at) { *p = nullptr; return hipErrorOutOfMemory; }
    const hipError_t e = hipMalloc(p, bytes);
    // (a failed allocation is reported by its return value; the runtime also remembers it as the "last error", which the
    //  next launch check - hipGetLastError - would take for its own: an engine that goes on without the memory would see
    //  its next healthy launch fail with "out of memory")
    if (e != hipSuccess) { *p = nullptr; (void) hipGetLastError(); }
    return e;
}

template <typename T>
T *ctx_alloc(ebcc_hip_ctx *ctx, size_t count)
{
    void *p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t e = device_malloc(&p, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    ctx->allocs.push_back(p);
    ctx->bytes += bytes;
    return static_cast<T *>(p);
}
template float *ctx_alloc<float>(ebcc_hip_ctx *, size_t);
template int32_t *ctx_alloc<int32_t>(ebcc_hip_ctx *, size_t);
template uint32_t *ctx_alloc<uint32_t>(ebcc_hip_ctx *, size_t);
template uint16_t *ctx_alloc<uint16_t>(ebcc_hip_ctx *, size_t);
template uint8_t *ctx_alloc<uint8_t>(ebcc_hip_ctx *, size_t);
template double *ctx_alloc<double>(ebcc_hip_ctx *, size_t);
template unsigned long long *ctx_alloc<unsigned long long>(ebcc_hip_ctx *, size_t);
template FrameState *ctx_alloc<FrameState>(ebcc_hip_ctx *, size_t);

// ---- staging of per-frame pieces --------------------------------------------------------------------
namespace {

// pack = true:  staged[off[f] .. + len[f]) = slot f; false: the other way.  Slots and offsets are 4-byte aligned;
// lengths are rounded up to whole words (inside the slot / the 16-byte aligned piece).
__global__ __launch_bounds__(256) void k_stage(uint8_t *slots, size_t stride, const unsigned long long *pack, uint8_t *staged, int to_stage)
{
    const int f = blockIdx.x;
    const unsigned long long off = pack[2 * f], len = pack[2 * f + 1];
    if (!len) return;
    uint32_t *a = (uint32_t *) (slots + (size_t) f * stride), *b = (uint32_t *) (staged + off);
    const size_t words = (size_t) ((len + 3) / 4);
    for (size_t i = (size_t) blockIdx.y * blockDim.x + threadIdx.x; i < words; i += (size_t) gridDim.y * blockDim.x) {
        if (to_stage) b[i] = a[i]; else a[i] = b[i];
    }
}

size_t stage_layout(ebcc_hip_ctx *ctx, const size_t *len, size_t *off, size_t n)
{
    size_t total = 0;
    for (size_t f = 0; f < n; f++) {
        off[f] = total;
        ctx->h_pack[2 * f] = total; ctx->h_pack[2 * f + 1] = len[f];
        total += (len[f] + 15) & ~(size_t) 15;
    }
    if (total > ctx->stage_cap) {                                    // grow (rare: sized by the largest batch seen)
        // the context stays cached after a failed call: it must never keep a freed pointer or a capacity it does not have
        uint8_t *h_old = ctx->h_stage, *d_old = ctx->d_stage;
        ctx->h_stage = nullptr; ctx->d_stage = nullptr; ctx->stage_cap = 0;
        if (h_old) hipHostFree(h_old);
        if (d_old) hipFree(d_old);
        const size_t cap = total + total / 2 + 4096;
        uint8_t *h_new = nullptr, *d_new = nullptr;
        EBCC_HIP_CHECK(hipHostMalloc((void **) &h_new, cap));
        const hipError_t e = device_malloc((void **) &d_new, cap);
        if (e != hipSuccess) { hipHostFree(h_new); EBCC_HIP_CHECK(e); }
        ctx->h_stage = h_new; ctx->d_stage = d_new; ctx->stage_cap = cap;
    }
    return total;
}

}  // namespace

void stage_download(ebcc_hip_ctx *ctx, const uint8_t *src, size_t stride, const size_t *len, size_t *off, size_t n, hipStream_t s)
{
    const size_t total = stage_layout(ctx, len, off, n);
    if (!total) return;
    EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_pack, ctx->h_pack, 2 * n * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_stage, dim3((unsigned) n, 4), dim3(256), 0, s, const_cast<uint8_t *>(src), stride, ctx->d_pack, ctx->d_stage, 1);
    EBCC_HIP_CHECK(hipMemcpyAsync(ctx->h_stage, ctx->d_stage, total, hipMemcpyDeviceToHost, s));
    wait_stream(s);
}

void stage_reserve(ebcc_hip_ctx *ctx, const size_t *len, size_t *off, size_t m) { stage_layout(ctx, len, off, m); }

void stage_send(ebcc_hip_ctx *ctx, size_t m, hipStream_t s)
{
    size_t total = 0;
    for (size_t k = 0; k < m; k++) total = std::max<size_t>(total, ctx->h_pack[2 * k] + ((ctx->h_pack[2 * k + 1] + 15) & ~15ull));
    EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_pack, ctx->h_pack, 2 * m * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    if (total) EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_stage, ctx->h_stage, total, hipMemcpyHostToDevice, s));
}

void stage_scatter(ebcc_hip_ctx *ctx, uint8_t *dst, size_t stride, size_t first, size_t count, hipStream_t s)
{
    if (!count) return;
    hipLaunchKernelGGL(k_stage, dim3((unsigned) count, 4), dim3(256), 0, s, dst, stride, ctx->d_pack + 2 * first, ctx->d_stage, 0);
}

bool ensure_cut_slots(ebcc_hip_ctx *ctx, int capacity)
{
    if (ctx->cut.capacity >= capacity) return true;
    if (ctx->cut_failed) return false;
    // sized for what is asked (a slice engine's batches are all of one size), grown if a larger batch comes: the old set goes first
    const int cap = capacity;
    if (!prefix_slots_supported(ctx->rb, cap)) { ctx->cut_failed = true; return false; }
    if (ctx->cut.capacity > 0) {
        EBCC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        void *old[] = {ctx->cut.A, ctx->cut.T, ctx->cut.D, ctx->cut.fs, ctx->cut.partial, ctx->cut.bits, ctx->cut.active, ctx->cut.frame_of};
        for (void *p : old) {
            auto it = std::find(ctx->allocs.begin(), ctx->allocs.end(), p);
            if (it != ctx->allocs.end()) { hipFree(p); ctx->allocs.erase(it); }
        }
        ctx->cut = CutSlots{};
    }
    CutSlots c;
    c.stride = (size_t) (ctx->rb.g.ny >> 1) * (size_t) ctx->rb.g.nx;
    bool ok = true;
    ok &= (c.A = ctx_alloc<float>(ctx, c.stride * cap)) != nullptr;
    ok = ok && (c.T = ctx_alloc<float>(ctx, c.stride * cap)) != nullptr;
    ok = ok && (c.D = ctx_alloc<float>(ctx, c.stride * cap)) != nullptr;
    ok = ok && (c.fs = ctx_alloc<FrameState>(ctx, cap)) != nullptr;
    ok = ok && (c.partial = ctx_alloc<double>(ctx, (size_t) kPartials * cap)) != nullptr;
    ok = ok && (c.bits = ctx_alloc<unsigned long long>(ctx, cap)) != nullptr;
    ok = ok && (c.active = (int *) ctx_alloc<uint32_t>(ctx, cap)) != nullptr;
    ok = ok && (c.frame_of = (int *) ctx_alloc<uint32_t>(ctx, cap)) != nullptr;
    if (!ok) { ctx->cut_failed = true; clear_error(); return false; }    // (what was allocated of the set stays with the context until it goes)
    EBCC_HIP_CHECK(hipMemsetAsync(c.active, 0, sizeof(int) * cap, ctx->stream));
    EBCC_HIP_CHECK(hipMemsetAsync(c.frame_of, 0, sizeof(int) * cap, ctx->stream));
    c.capacity = cap;
    ctx->cut = c;
    return true;
}

void fetch_frame_states(ebcc_hip_ctx *ctx, size_t n)
{
    EBCC_HIP_CHECK(hipMemcpyAsync(ctx->h_fs, ctx->rb.fs, n * sizeof(FrameState), hipMemcpyDeviceToHost, ctx->stream));
    wait_stream(ctx->stream);
}
void push_frame_states(ebcc_hip_ctx *ctx, size_t n)
{
    EBCC_HIP_CHECK(hipMemcpyAsync(ctx->rb.fs, ctx->h_fs, n * sizeof(FrameState), hipMemcpyHostToDevice, ctx->stream));
}

// ---- optional kernel timing -------------------------------------------------------------------------
struct TimedSpan { std::string name; hipEvent_t a, b; hipStream_t s; bool closed; };
static bool g_timing = false;
static std::vector<TimedSpan> g_spans;
static std::mutex g_spans_mutex;             // sub-batches are driven by their own host threads (and streams)

void timing_begin(const char *name, hipStream_t s)
{
    if (!g_timing) return;
    TimedSpan t{name, nullptr, nullptr, s, false};
    EBCC_HIP_CHECK(hipEventCreate(&t.a));
    EBCC_HIP_CHECK(hipEventCreate(&t.b));
    EBCC_HIP_CHECK(hipEventRecord(t.a, s));
    std::lock_guard<std::mutex> lock(g_spans_mutex);
    g_spans.push_back(t);
}
void timing_end(const char *name, hipStream_t s)
{
    if (!g_timing) return;
    std::lock_guard<std::mutex> lock(g_spans_mutex);
    for (size_t i = g_spans.size(); i-- > 0;)
        if (!g_spans[i].closed && g_spans[i].s == s && g_spans[i].name == name) {
            EBCC_HIP_CHECK(hipEventRecord(g_spans[i].b, s));
            g_spans[i].closed = true;
            return;
        }
}

}  // namespace ebcc

using namespace ebcc;

namespace ebcc {
// j2k.hip
bool j2k_create(ebcc_hip_ctx *ctx);
void j2k_destroy(ebcc_hip_ctx *ctx);
}  // namespace ebcc

extern "C" {

const char *ebcc_hip_last_error(void) { return g_last_error.c_str(); }

void ebcc_hip_timing_enable(ebcc_hip_ctx *ctx, int on)
{
    (void) ctx;
    if (on) {
        for (auto &t : g_spans) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
        g_spans.clear();
    }
    g_timing = on != 0;
}

int ebcc_hip_timing_read(ebcc_hip_ctx *ctx, const char *name, double *total_ms, long *launches)
{
    EBCC_API_TRY
    if (ctx) EBCC_HIP_CHECK(hipDeviceSynchronize());
    double tot = 0;
    long n = 0;
    for (auto &t : g_spans)
        if (t.closed && t.name == name) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { tot += ms; n++; }
        }
    *total_ms = tot;
    *launches = n;
    return 0;
    EBCC_API_CATCH(1)
}

int ebcc_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

ebcc_hip_ctx *ebcc_hip_create(int device, size_t max_frames, size_t height, size_t width)
{
    return ebcc::create_engine(device, max_frames, height, width, 1);
}

}  // extern "C"

ebcc_hip_ctx *ebcc::create_engine(int device, size_t max_frames, size_t height, size_t width, int tile_period)
{
    EBCC_API_TRY
    if (height < 1 || width < 1 || height > 2047 || width > 2047 || max_frames < 1 || tile_period < 1 ||
        (tile_period > 1 && (size_t) tile_period * height > 2047)) {
        set_error("ebcc_hip_create: unsupported geometry %zu x %zu x %zu", max_frames, height, width);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error("ebcc_hip_create: no HIP device %d (found %d) - the MI355X engine has no CPU fallback", device, ndev);
        return nullptr;
    }
    EBCC_HIP_CHECK(hipSetDevice(device));
    ebcc_hip_ctx *ctx = new ebcc_hip_ctx();
    try {                                                   // (a throw below must not leak the half-built engine)
    ctx->device = device;
    ctx->max_frames = max_frames;
    ctx->height = (int) height;
    ctx->tile_period = tile_period;
    ctx->width = (int) width;
    ctx->n_pix = height * width;
    EBCC_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));

    ResidualBuffers &rb = ctx->rb;
    rb.g = make_grid((int) height, (int) width, kResidualStages);
    rb.max_frames = (int) max_frames;
    rb.np = (size_t) rb.g.npix();
    const size_t tot = rb.np * max_frames;
    // stream capacity: the reference allocates height*width*4 bytes when trunc_bits == 0 (spiht_re.c:433)
    rb.stream_words = (ctx->n_pix * 4 + 64) / 4 + 64;
    bool ok = true;
    ok &= (rb.A = ctx_alloc<float>(ctx, tot)) != nullptr;
    ok &= (rb.T = ctx_alloc<float>(ctx, tot)) != nullptr;
    ok &= (rb.C = ctx_alloc<int32_t>(ctx, tot)) != nullptr;
    ok &= (rb.D = ctx_alloc<int32_t>(ctx, tot)) != nullptr;
    ok &= (rb.G = ctx_alloc<int32_t>(ctx, tot)) != nullptr;
    ok &= (rb.lip = ctx_alloc<uint32_t>(ctx, tot)) != nullptr;
    ok &= (rb.lsp = ctx_alloc<uint32_t>(ctx, tot)) != nullptr;
    ok &= (rb.lis0 = ctx_alloc<uint32_t>(ctx, tot)) != nullptr;
    ok &= (rb.lis1 = ctx_alloc<uint32_t>(ctx, tot)) != nullptr;
    ok &= (rb.sigord = ctx_alloc<uint32_t>(ctx, tot)) != nullptr;
    ok &= (rb.lspidx = ctx_alloc<uint32_t>(ctx, tot)) != nullptr;
    ok &= (rb.stream = ctx_alloc<uint32_t>(ctx, rb.stream_words * max_frames)) != nullptr;
    ok &= (rb.partial = ctx_alloc<double>(ctx, (size_t) kPartials * max_frames)) != nullptr;
    ok &= (rb.fs = ctx_alloc<FrameState>(ctx, max_frames)) != nullptr;
    ok &= (ctx->d_u64a = ctx_alloc<unsigned long long>(ctx, max_frames)) != nullptr;
    ok &= (ctx->d_u64b = ctx_alloc<unsigned long long>(ctx, max_frames)) != nullptr;
    ok &= (ctx->d_u64c = ctx_alloc<unsigned long long>(ctx, max_frames)) != nullptr;
    ok &= (ctx->d_active = (int *) ctx_alloc<uint32_t>(ctx, 2 * max_frames)) != nullptr;      // (second half: the overlapped search's mask)
    ok &= (ctx->d_pack = ctx_alloc<unsigned long long>(ctx, 4 * max_frames)) != nullptr;
    ok &= (ctx->d_search = ctx_alloc<uint8_t>(ctx, 2 * sizeof(DevChunk) * max_frames)) != nullptr;   // (second half: the overlapped search)
    ok &= (ctx->d_counter = (int *) ctx_alloc<uint32_t>(ctx, 8)) != nullptr;
    if (ok) {
        EBCC_HIP_CHECK(hipHostMalloc((void **) &ctx->h_u64a, max_frames * sizeof(unsigned long long)));
        EBCC_HIP_CHECK(hipHostMalloc((void **) &ctx->h_u64b, max_frames * sizeof(unsigned long long)));
        EBCC_HIP_CHECK(hipHostMalloc((void **) &ctx->h_u64c, max_frames * sizeof(unsigned long long)));
        EBCC_HIP_CHECK(hipHostMalloc((void **) &ctx->h_active, max_frames * sizeof(int)));
        EBCC_HIP_CHECK(hipHostMalloc((void **) &ctx->h_fs, max_frames * sizeof(FrameState)));
        EBCC_HIP_CHECK(hipHostMalloc((void **) &ctx->h_pack, 4 * max_frames * sizeof(unsigned long long)));
        EBCC_HIP_CHECK(hipHostMalloc((void **) &ctx->h_act, 2 * max_frames * sizeof(int)));
        EBCC_HIP_CHECK(hipHostMalloc(&ctx->h_search, 2 * sizeof(DevChunk) * max_frames));
        EBCC_HIP_CHECK(hipHostMalloc((void **) &ctx->h_counter, 8 * sizeof(int)));
        EBCC_HIP_CHECK(hipMemsetAsync(rb.fs, 0, max_frames * sizeof(FrameState), ctx->stream));
        ok = j2k_create(ctx);
    }
    if (!ok) {
        ebcc_hip_destroy(ctx);
        return nullptr;
    }
    wait_stream(ctx->stream);
    return ctx;
    } catch (...) {
        ebcc_hip_destroy(ctx);
        throw;
    }
    EBCC_API_CATCH(nullptr)
}

extern "C" {

void ebcc_hip_destroy(ebcc_hip_ctx *ctx)
{
    EBCC_API_TRY
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (ebcc_hip_ctx *c : ctx->lanes) ebcc_hip_destroy(c);
    ctx->lanes.clear();
    if (ctx->twin) { ebcc_hip_destroy(ctx->twin); ctx->twin = nullptr; }
    j2k_destroy(ctx);
    for (void *p : ctx->allocs) hipFree(p);
    if (ctx->h_u64a) hipHostFree(ctx->h_u64a);
    if (ctx->h_u64b) hipHostFree(ctx->h_u64b);
    if (ctx->h_u64c) hipHostFree(ctx->h_u64c);
    if (ctx->h_active) hipHostFree(ctx->h_active);
    if (ctx->h_fs) hipHostFree(ctx->h_fs);
    if (ctx->h_pack) hipHostFree(ctx->h_pack);
    if (ctx->h_act) hipHostFree(ctx->h_act);
    if (ctx->h_search) hipHostFree(ctx->h_search);
    if (ctx->h_counter) hipHostFree(ctx->h_counter);
    if (ctx->h_jf) hipHostFree(ctx->h_jf);
    if (ctx->h_table) hipHostFree(ctx->h_table);
    if (ctx->h_stage) hipHostFree(ctx->h_stage);
    if (ctx->d_stage) hipFree(ctx->d_stage);
    if (ctx->d_io) hipFree(ctx->d_io);
    if (ctx->h_bounce) hipHostFree(ctx->h_bounce);
    if (ctx->ev_a) hipEventDestroy(ctx->ev_a);
    if (ctx->ev_b) hipEventDestroy(ctx->ev_b);
    if (ctx->stream2) hipStreamDestroy(ctx->stream2);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    EBCC_API_CATCH_VOID
}

void *ebcc_hip_stream(ebcc_hip_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }
size_t ebcc_hip_workspace_bytes(const ebcc_hip_ctx *ctx) { return ctx ? ctx->bytes : 0; }
size_t ebcc_hip_padded_pixels(const ebcc_hip_ctx *ctx) { return ctx ? ctx->rb.np : 0; }

void *ebcc_hip_malloc(size_t bytes)
{
    void *p = nullptr;
    if (device_malloc(&p, bytes ? bytes : 1) != hipSuccess) {
        set_error("ebcc_hip_malloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}
void ebcc_hip_free(void *p) { if (p) hipFree(p); }
int ebcc_hip_memcpy_h2d(void *d, const void *s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyHostToDevice) != hipSuccess; }
int ebcc_hip_memcpy_d2h(void *d, const void *s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyDeviceToHost) != hipSuccess; }

// ------------------------------------------------------------------------------------------------
// residual layer
// ------------------------------------------------------------------------------------------------
static int check_batch(ebcc_hip_ctx *ctx, size_t n_frames, const char *who)
{
    if (!ctx) { set_error("%s: null context", who); return 1; }
    if (n_frames < 1 || n_frames > ctx->max_frames) {
        set_error("%s: %zu frames exceeds the context capacity %zu", who, n_frames, ctx->max_frames);
        return 1;
    }
    EBCC_HIP_CHECK(hipSetDevice(ctx->device));
    return 0;
}

// shared by ebcc_hip_spiht_encode and the frame codec: analysis must already have run
static void run_spiht_encode(ebcc_hip_ctx *ctx, size_t n, const size_t *trunc_bits)
{
    hipStream_t s = ctx->stream;
    fetch_frame_states(ctx, n);
    for (size_t f = 0; f < n; f++) {
        // spiht_re.c:458: bits0 = trunc_bits ? trunc_bits + 128 : 2^28
        unsigned long long bits0 = trunc_bits[f] == 0 ? (1ull << 28) : (unsigned long long) trunc_bits[f] + 128;
        ctx->h_u64a[f] = bits0;
        ctx->h_fs[f].budget = bits0 - 128;
    }
    push_frame_states(ctx, n);
    EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_u64a, ctx->h_u64a, n * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    launch_spiht_encode(ctx->rb, (int) n, ctx->d_u64a, nullptr, s);
}

static int collect_streams(ebcc_hip_ctx *ctx, size_t n, uint8_t **out_streams, size_t *out_sizes)
{
    fetch_frame_states(ctx, n);
    for (size_t f = 0; f < n; f++) {
        size_t nb = ctx->h_fs[f].stream_bytes;
        out_sizes[f] = nb;
        out_streams[f] = (uint8_t *) malloc(nb ? nb : 1);
        if (!out_streams[f]) { set_error("out of host memory"); return 1; }
        EBCC_HIP_CHECK(hipMemcpyAsync(out_streams[f], ctx->rb.stream + f * ctx->rb.stream_words, nb,
                                      hipMemcpyDeviceToHost, ctx->stream));
    }
    wait_stream(ctx->stream);
    return 0;
}

int ebcc_hip_spiht_encode(ebcc_hip_ctx *ctx, const float *d_images, size_t n_frames, const size_t *trunc_bits,
                          uint8_t **out_streams, size_t *out_sizes)
{
    EBCC_API_TRY
    if (check_batch(ctx, n_frames, "ebcc_hip_spiht_encode")) return 1;
    for (size_t f = 0; f < n_frames; f++) {
        // reference buffer is trunc_bits + 1 bytes (spiht_re.c:433); our device slot holds height*width*4
        if (trunc_bits[f] != 0 && (trunc_bits[f] + 121 + 7) / 8 > ctx->rb.stream_words * 4 - 64) {
            set_error("ebcc_hip_spiht_encode: trunc_bits %zu exceeds the stream slot", trunc_bits[f]);
            return 1;
        }
    }
    hipStream_t s = ctx->stream;
    launch_pad_and_dc_from_image(d_images, ctx->rb, (int) n_frames, s);
    launch_analysis(ctx->rb, (int) n_frames, nullptr, s);
    run_spiht_encode(ctx, n_frames, trunc_bits);
    return collect_streams(ctx, n_frames, out_streams, out_sizes);
    EBCC_API_CATCH(1)
}

int ebcc_hip_spiht_coeffs(ebcc_hip_ctx *ctx, const float *d_images, size_t n_frames, int32_t *coeffs, int *dc)
{
    EBCC_API_TRY
    if (check_batch(ctx, n_frames, "ebcc_hip_spiht_coeffs")) return 1;
    hipStream_t s = ctx->stream;
    launch_pad_and_dc_from_image(d_images, ctx->rb, (int) n_frames, s);
    launch_analysis(ctx->rb, (int) n_frames, nullptr, s);
    EBCC_HIP_CHECK(hipMemcpyAsync(coeffs, ctx->rb.C, n_frames * ctx->rb.np * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    fetch_frame_states(ctx, n_frames);
    for (size_t f = 0; f < n_frames; f++) dc[f] = (int) ctx->h_fs[f].dc;
    return 0;
    EBCC_API_CATCH(1)
}

int ebcc_hip_spiht_decode_prefix(ebcc_hip_ctx *ctx, size_t n_frames, const size_t *trunc_bits, float *d_images_out)
{
    EBCC_API_TRY
    if (check_batch(ctx, n_frames, "ebcc_hip_spiht_decode_prefix")) return 1;
    hipStream_t s = ctx->stream;
    fetch_frame_states(ctx, n_frames);
    for (size_t f = 0; f < n_frames; f++) {
        if (trunc_bits[f] <= 128) { set_error("decode_prefix: trunc_bits must exceed 128"); return 1; }
        ctx->h_u64b[f] = trunc_bits[f];
        ctx->h_fs[f].dec_dc = (int) ctx->h_fs[f].dc;
    }
    push_frame_states(ctx, n_frames);
    EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_u64b, ctx->h_u64b, n_frames * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    launch_reconstruct(ctx->rb, (int) n_frames, ctx->d_u64b, nullptr, s);
    launch_synthesis(ctx->rb, (int) n_frames, nullptr, s);
    launch_emit_image(d_images_out, ctx->rb, (int) n_frames, s);
    wait_stream(s);
    return 0;
    EBCC_API_CATCH(1)
}

// parse + validate the 15-byte IMS header on the host (spiht_re.c:480-503)
}  // extern "C"
// header of a SPIHT stream against the context's grid, and a bit budget the decoder can work with (non-zero: reject)
int ebcc::check_ims_header(ebcc_hip_ctx *ctx, const uint8_t *b, size_t n, size_t num_bits)
{
    if (n < 15 || b[0] != 'I' || b[1] != 'M' || b[2] != 'S') { set_error("SPIHT stream: bad magic"); return 1; }
    unsigned long long hi = 0;
    for (int i = 0; i < 8; i++) hi = (hi << 8) | b[3 + i];
    unsigned lo = ((unsigned) b[11] << 8) | b[12];
    unsigned stages = (unsigned) (hi >> 58), sx = (unsigned) (hi >> 46) & 0xFFF, sy = (unsigned) (hi >> 34) & 0xFFF;
    unsigned ex = (unsigned) (hi >> 24) & 0x3FF, ey = (unsigned) (hi >> 14) & 0x3FF;
    unsigned long long bits0 = ((hi & 0x1FFF) << 16) | lo;
    const Grid &g = ctx->rb.g;
    if ((int) stages != g.stages || (int) sx != g.size_x || (int) sy != g.size_y || (int) ex != g.extra_x ||
        (int) ey != g.extra_y) {
        set_error("SPIHT stream geometry (%u stages, %ux%u +%u,%u) does not match the context (%dx%d)", stages, sx, sy,
                  ex, ey, g.size_x, g.size_y);
        return 1;
    }
    unsigned long long nb = num_bits > bits0 ? bits0 : num_bits;
    if (nb <= 128) { set_error("SPIHT stream: num_bits must exceed 128"); return 1; }
    return 0;
}
extern "C" {

int ebcc_hip_spiht_decode(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes,
                          const size_t *num_bits, size_t n_frames, float *d_images_out)
{
    EBCC_API_TRY
    if (check_batch(ctx, n_frames, "ebcc_hip_spiht_decode")) return 1;
    hipStream_t s = ctx->stream;
    const size_t slot = ctx->rb.stream_words * 4;
    for (size_t f = 0; f < n_frames; f++) {
        if (sizes[f] > slot - 64) { set_error("SPIHT stream of %zu bytes exceeds the slot", sizes[f]); return 1; }
        if (check_ims_header(ctx, streams[f], sizes[f], num_bits[f])) return 1;
        ctx->h_u64a[f] = sizes[f];
        ctx->h_u64b[f] = num_bits[f];
        EBCC_HIP_CHECK(hipMemcpyAsync((uint8_t *) ctx->rb.stream + f * slot, streams[f], sizes[f], hipMemcpyHostToDevice, s));
    }
    EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_u64a, ctx->h_u64a, n_frames * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_u64b, ctx->h_u64b, n_frames * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    launch_spiht_decode((const uint8_t *) ctx->rb.stream, slot, ctx->d_u64a, ctx->d_u64b, ctx->rb, (int) n_frames, nullptr, s);
    launch_synthesis(ctx->rb, (int) n_frames, nullptr, s);
    launch_emit_image(d_images_out, ctx->rb, (int) n_frames, s);
    wait_stream(s);
    return 0;
    EBCC_API_CATCH(1)
}

}  // extern "C"
