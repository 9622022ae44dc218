// host_pool.hip - host-side services of the library (host.hpp): logging, libzstd (dlopen'd) and the provable lower
// bound of a zstd frame's size, the CPU budget of the process, the accounting of the entropy stage.  The worker pool itself
// (HostPool) is a class of host.hpp.
#include "host.hpp"

namespace ebcc {

// ================================================================================================
// logging (reference src/log/, level from EBCC_LOG_LEVEL, default WARN; src/ebcc_codec.c:431-448)
// ================================================================================================
int g_log_level = 3;
namespace { const char *kLevelNames[] = {"TRACE", "DEBUG", "INFO", "WARN", "ERROR", "FATAL"}; }
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

// ================================================================================================
// zstd stays on the host (north star); dlopen'd so the library has no link-time dependency
// ================================================================================================
Zstd::Zstd()
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

// CPUs this process may really use: the affinity mask, cut down to the container's CPU quota where one is set (cgroup v2
// cpu.max "quota period", cgroup v1 cpu.cfs_quota_us / cpu.cfs_period_us).  The MI355X box of this project is a 16-CPU
// quota on a 256-thread host: the mask says 256, and a pool sized from it bursts into the quota, gets the whole cgroup
// throttled for the rest of the 100 ms period - the threads that steer the GPU included.
double cgroup_cpu_quota()
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
unsigned entropy_threads(unsigned slices)
{
    if (const char *e = getenv("EBCC_HOST_THREADS")) return (unsigned) std::max(1L, strtol(e, nullptr, 10));
    unsigned lws = 1;
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) lws = (unsigned) std::max(1, atoi(e));
    return entropy_threads_for(burst_cpus(), lws, slices);
}

HostStats &host_stats() { static HostStats h; return h; }

}  // namespace ebcc
