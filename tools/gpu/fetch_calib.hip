// fetch_calib.hip - what FETCH_SIZE / WRITE_SIZE report on gfx950 for the access shapes this codec uses.  The guide's x2
// correction of FETCH_SIZE is calibrated for 16-byte-per-lane streaming reads only ("other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern"): every kernel below moves exactly
// kBytes (1 GiB, four times the Infinity Cache) once.   tools/gpu/fetch_calib.sh runs it under rocprofv3 --pmc.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr size_t kBytes = (size_t) 1 << 30;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// 16 bytes per lane, grid-stride (the guide's calibrated case)
__global__ void calib_read16(const float4 *p, size_t n, float *sink)
{
    float a = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 123.456f) *sink = a;
}
// 4 bytes per lane, consecutive lanes consecutive floats: one wave instruction = 256 contiguous bytes (band rows of the fused inverse levels)
__global__ void calib_read4(const float *p, size_t n, float *sink)
{
    float a = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) a += p[i];
    if (a == 123.456f) *sink = a;
}
// 8 bytes per lane
__global__ void calib_read8(const float2 *p, size_t n, float *sink)
{
    float a = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { float2 v = p[i]; a += v.x + v.y; }
    if (a == 123.456f) *sink = a;
}
// two 4-byte loads per lane at an 8-byte lane stride (columns 2i and 2i + 1 of a frame row: the statistics' reads)
__global__ void calib_read4_pair(const float *p, size_t n, float *sink)
{
    float a = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; 2 * i + 1 < n; i += (size_t) gridDim.x * blockDim.x) { a += p[2 * i]; a += p[2 * i + 1]; }
    if (a == 123.456f) *sink = a;
}
// one wave walks down a column strip: 64 lanes x 4 bytes of one row, then the next row (pitch 5760 bytes) - a 64-thread
// workgroup of the fused level; rows come one after the other, one row requested ahead
__global__ void calib_read4_rows(const float *p, int pitch_f, int rows, int strips, float *sink)
{
    const int strip = blockIdx.x % strips, img = blockIdx.x / strips;
    const float *b = p + (size_t) img * pitch_f * rows + (size_t) strip * 64 + threadIdx.x;
    float a = 0;
    for (int r = 0; r < rows; r++) a += b[(size_t) r * pitch_f];
    if (a == 123.456f) *sink = a;
}
__global__ void calib_write16(float4 *p, size_t n)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = make_float4(1, 2, 3, 4);
}
__global__ void calib_write4(float *p, size_t n)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = 1.0f;
}
// 16-byte pieces at a 1 KB stride per lane (the tier-1 row streams as k_t1_emit writes them)
__global__ void calib_write16_strided(float4 *p, size_t n)
{
    // lane l of wave w writes piece (w * 64 * 64) + r * 64 + l ... for r = 0..63: every store instruction covers 64 x 16 B = 1 KB contiguous
    const size_t wave = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
    for (size_t base = wave * 4096; base + 4096 <= n; base += (size_t) gridDim.x * (blockDim.x / 64) * 4096)
        for (int r = 0; r < 64; r++) p[base + (size_t) lane * 64 + r] = make_float4(1, 2, 3, 4);   // lane-major: 16 B at 1 KB stride
}
int main()
{
    void *buf; float *sink;
    CHECK(hipMalloc(&buf, kBytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, kBytes));
    const int blocks = 256 * 16;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(calib_read16, dim3(blocks), dim3(256), 0, 0, (const float4 *) buf, kBytes / 16, sink);
        hipLaunchKernelGGL(calib_read8, dim3(blocks), dim3(256), 0, 0, (const float2 *) buf, kBytes / 8, sink);
        hipLaunchKernelGGL(calib_read4, dim3(blocks), dim3(256), 0, 0, (const float *) buf, kBytes / 4, sink);
        hipLaunchKernelGGL(calib_read4_pair, dim3(blocks), dim3(256), 0, 0, (const float *) buf, kBytes / 4, sink);
        // 721-row images of 1440 floats: 22 strips of 64 columns (1408 of 1440 columns: 97.8 % of the bytes)
        const int rows = 721, pitch = 1440, strips = 22, imgs = (int) (kBytes / ((size_t) rows * pitch * 4));
        hipLaunchKernelGGL(calib_read4_rows, dim3(strips * imgs), dim3(64), 0, 0, (const float *) buf, pitch, rows, strips, sink);
        hipLaunchKernelGGL(calib_write16, dim3(blocks), dim3(256), 0, 0, (float4 *) buf, kBytes / 16);
        hipLaunchKernelGGL(calib_write4, dim3(blocks), dim3(256), 0, 0, (float *) buf, kBytes / 4);
        hipLaunchKernelGGL(calib_write16_strided, dim3(blocks), dim3(256), 0, 0, (float4 *) buf, kBytes / 16);
        CHECK(hipDeviceSynchronize());
    }
    printf("fetch_calib: every kernel moves %zu bytes (read4_rows: %.1f %% of them)\n", kBytes, 100.0 * 22 * 64 / 1440 * (double) ((kBytes / (721 * 1440 * 4)) * (size_t) 721 * 1440 * 4) / (double) kBytes);
    return 0;
}
