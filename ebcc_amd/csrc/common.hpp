// common.hpp - shared host/device definitions for the MI355X EBCC engine (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace ebcc {

// A failed HIP call on the host side throws; every extern "C" entry point catches at the boundary, reports through
// set_error / the reference's log_fatal and returns its error value (0 bytes / 0 floats / non-zero status) - the
// library never takes the host application down for a failed allocation or copy (reference convention:
// /root/reference/src/ebcc_codec.c:613-617,1230-1257 log and return 0).
struct HipFailure : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] inline void hip_fail(const char *what, const char *file, int line, hipError_t e)
{
    char buf[384];
    std::snprintf(buf, sizeof buf, "%s failed at %s:%d: %s", what, file, line, hipGetErrorString(e));
    throw HipFailure(buf);
}
#define EBCC_HIP_CHECK(expr)                                                                    \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) ::ebcc::hip_fail(#expr, __FILE__, __LINE__, e_);                  \
    } while (0)
// after a group of kernel launches: a launch that the runtime refused (bad configuration, no resources) shows here,
// not at some later synchronisation
#define EBCC_HIP_LAUNCH_CHECK() EBCC_HIP_CHECK(hipGetLastError())
// Every extern "C" function body sits between these two: `ret` is the function's error value.
#define EBCC_API_TRY try {
#define EBCC_API_CATCH(ret)                                                                     \
    } catch (const std::exception &e_) { ::ebcc::set_error("%s", e_.what()); return ret; }
#define EBCC_API_CATCH_VOID                                                                     \
    } catch (const std::exception &e_) { ::ebcc::set_error("%s", e_.what()); }

void set_error(const char *fmt, ...);
void clear_error();                 // this thread's last-error text
// hipMalloc through one door: EBCC_HIP_FAIL_ALLOC=<n> (tests) makes the n-th allocation of the process fail
hipError_t device_malloc(void **p, size_t bytes);

// Host wait for a stream.  By default through an event created with hipEventBlockingSync: the waiting thread sleeps
// instead of spinning (hipStreamSynchronize spins on a core for as long as the GPU works - two slice threads waiting are
// two of the container's CPUs, taken from the zstd workers).
void wait_stream(hipStream_t s);

constexpr int kWave = 64;             // CDNA wavefront
constexpr int kResidualStages = 3;    // WAVELET_LEVELS, reference src/ebcc_codec.c:28

// Padded transform grid of the residual coder (reference src/spiht/dwt.h:41-59).
struct Grid {
    int size_x, size_y;     // frame width / height
    int extra_x, extra_y;   // padding to a multiple of 2^(stages+1)
    int nx, ny;             // padded extents; row stride == nx
    int stages;
    int lx, ly;             // LL band extents (nx >> stages, ny >> stages)
    __host__ __device__ int npix() const { return nx * ny; }
};

inline Grid make_grid(int height, int width, int stages)
{
    Grid g;
    int unit = 1 << (stages + 1);
    g.size_x = width;
    g.size_y = height;
    g.extra_x = (unit - width % unit) % unit;
    g.extra_y = (unit - height % unit) % unit;
    g.nx = width + g.extra_x;
    g.ny = height + g.extra_y;
    g.stages = stages;
    g.lx = g.nx >> stages;
    g.ly = g.ny >> stages;
    return g;
}

template <typename T>
__host__ __device__ inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

// XCD-aware workgroup -> tile mapping for kernels in which the `strips` workgroups of a tile read neighbouring column
// ranges of the same rows (the fused inverse wavelet levels: 60 sample pairs per wave, ranges that are not multiples of a
// 128-byte line, two pairs of halo either side).  MI355X deals workgroups round-robin over its 8 XCDs, each with its own L2
// (observed placement; used for speed only): with the plain (strip, frame, piece) grid the neighbours of a strip run on
// seven OTHER XCDs and every line two strips share - and every halo - is fetched from HBM once per XCD: the top inverse
// level of the base layer moved 2.02x its bytes (profiles/r02_hbm_kernels.json).  Here workgroup b of a 1-D launch of
// strips * tiles workgroups takes strip (b / 8) % strips of tile 8 * (b / 8 / strips) + b % 8: the strips of a tile are
// consecutive workgroups of ONE XCD.  Tiles are numbered piece-major (tile = piece * frames + frame), so that all frames'
// first pieces are dispatched first (the early exit of the search probes relies on that order for speed, not correctness).
struct TileOfBlock { int strip, frame, piece; };
__device__ inline TileOfBlock xcd_tile_of_block(unsigned b, unsigned strips, unsigned frames, unsigned pieces)
{
    const unsigned tiles = frames * pieces, groups = tiles / 8, s = b >> 3;
    unsigned strip, tile;
    if (s < groups * strips) { strip = s % strips; tile = (s / strips) * 8 + (b & 7); }
    else { const unsigned idx = b - groups * 8 * strips; strip = idx % strips; tile = groups * 8 + idx / strips; }   // (the last tiles, fewer than 8)
    return TileOfBlock{(int) strip, (int) (tile % frames), (int) (tile / frames)};
}

// value of the lane below / above (lane 0 / lane 63 keep their own): __shfl_up / __shfl_down by one as a single DPP move
// (v_mov_b32_dpp wave_shr:1 / wave_shl:1, GFX9 family) instead of a trip through the LDS crossbar (ds_bpermute)
__device__ inline float lane_below(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}
__device__ inline float lane_above(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x130 /* wave_shl:1 */, 0xF, 0xF, false));
}

// Inclusive prefix sum over the 64 lanes of a wave in eight DPP additions (row_shr 1/2/3, row_shr:4 / :8 on the upper banks,
// row_bcast:15 / :31 into the rows above - GFX9 family) - no trip through the LDS crossbar, which a wave pays with ~100
// cycles per shuffle and more when other waves keep the LDS busy.  Every lane must be active.
__device__ inline uint32_t wave_prefix_sum(uint32_t x)
{
    const int xi = (int) x;
    int v = xi + __builtin_amdgcn_update_dpp(0, xi, 0x111, 0xF, 0xF, false) + __builtin_amdgcn_update_dpp(0, xi, 0x112, 0xF, 0xF, false) +
            __builtin_amdgcn_update_dpp(0, xi, 0x113, 0xF, 0xF, false);                    // lanes i-3 .. i of the row of 16
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xE, false);                         // + the four before (banks 1-3)
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xC, false);                         // + the eight before (banks 2, 3)
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);                         // rows 1, 3: + the row below's total
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);                         // rows 2, 3: + the total of rows 0, 1
    return (uint32_t) v;
}

// Maximum over the 64 lanes of a wave (non-negative values) by DPP, the result in every lane's copy of lane 63's value: the
// same ladder as the prefix sum - no trip through the LDS crossbar.  Every lane must be active.
__device__ inline int wave_max_nonneg(int x)
{
    int v = x;
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false));                  // row_shr:1 (0 where no lane is)
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false));                  // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false));                  // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false));                  // row_shr:8: lane 15 of a row holds the row's maximum
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));                  // rows 1, 3: the row below's lane 15
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false));                  // rows 2, 3: lane 31
    return __builtin_amdgcn_readlane(v, 63);
}

// Monotone key for float ordering with -0 == +0 (reference compares with < and >, so the two
// zeros tie; ties are then broken by index to reproduce "first occurrence wins").
__host__ __device__ inline uint32_t float_order_key(float f)
{
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    __builtin_memcpy(&u, &f, 4);
#endif
    if (u == 0x80000000u) u = 0;                       // -0 -> +0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Optional per-kernel timing with HIP events recorded on the launching stream (bench.py's roofline leg).
void timing_begin(const char *name, hipStream_t s);
void timing_end(const char *name, hipStream_t s);
struct ScopedTiming {
    const char *name; hipStream_t s;
    ScopedTiming(const char *n, hipStream_t st) : name(n), s(st) { timing_begin(name, s); }
    ~ScopedTiming() { timing_end(name, s); }
};

}  // namespace ebcc
