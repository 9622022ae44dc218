// GPU box: does a wave64 vector instruction cost less when few of its lanes are enabled?  (the tier-1 decoder runs 1-4 lanes
// of 64 and is bound by vector issue)   hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_skip tools/gpu/exec_skip.hip && /tmp/exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIter = 8192;
__global__ __launch_bounds__(256) void k(unsigned int *out, unsigned int s, int active)
{
    unsigned int b0 = threadIdx.x + 1, b1 = b0 * 3, b2 = b0 * 5, b3 = b0 * 7;
    if ((int) (threadIdx.x & 63) < active) {
        for (int i = 0; i < kIter; i++) {
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(b0) : "v"(s)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(b1) : "v"(s));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(b2) : "v"(s)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(b3) : "v"(s));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = b0 + b1 + b2 + b3;
}
int main()
{
    unsigned int *d;
    const int blocks = 256 * 8;                                         // 8 waves per SIMD
    CHECK(hipMalloc(&d, (size_t) blocks * 256 * 4));
    for (int active : {64, 32, 16, 4, 1}) {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1u, active);
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1u, active);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        const double instr_per_simd = 8.0 * kIter * 4;                  // waves per SIMD x instructions per wave
        printf("%2d lanes enabled: %.3f ms, %.2f ns per wave instruction and SIMD (2.4 GHz: %.2f cycles)\n", active, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
    return 0;
}
