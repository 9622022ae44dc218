// common.hpp - shared host/device definitions for the MI355X EBCC engine (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace ebcc {

#define EBCC_HIP_CHECK(expr)                                                                    \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "ebcc-hip: %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__, \
                         hipGetErrorString(e_));                                                \
            std::abort();                                                                       \
        }                                                                                       \
    } while (0)

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
