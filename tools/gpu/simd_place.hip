// GPU box: on which SIMDs do the two waves of a 128-thread workgroup run?  (the MQ pass is two dependent chains per
// workgroup: on one SIMD they would share its issue slots)   hipcc --offload-arch=gfx950 -O2 -o /tmp/simd_place tools/gpu/simd_place.hip && /tmp/simd_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out, int spin)
{
    __shared__ unsigned pad[7800];                 // ~31 KB like k_t1_mqrows
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    pad[threadIdx.x] = id;
    unsigned x = id;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;      // keep the waves resident while the grid fills
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id | (x & 0u) | (pad[threadIdx.x] & 0u);
}
int main()
{
    for (int threads : {128, 256}) {
        const int blocks = 490, waves = blocks * threads / 64;
        unsigned *d; hipMalloc(&d, waves * 4);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 200000);
        std::vector<unsigned> h(waves);
        hipMemcpy(h.data(), d, waves * 4, hipMemcpyDeviceToHost);
        int same_simd = 0, same_cu = 0, hist[4] = {0, 0, 0, 0};
        for (int b = 0; b < blocks; b++) {
            const unsigned a = h[b * (threads / 64)], c = h[b * (threads / 64) + 1];
            const unsigned simd_a = (a >> 4) & 3, simd_c = (c >> 4) & 3;
            same_simd += simd_a == simd_c; same_cu += ((a >> 8) & 0xFF) == ((c >> 8) & 0xFF);
            hist[simd_a]++;
        }
        printf("%d threads per workgroup, %d workgroups: waves 0 and 1 on the same SIMD in %d, same CU/SE bits in %d; wave 0 by SIMD: %d %d %d %d\n", threads, blocks, same_simd, same_cu, hist[0], hist[1], hist[2], hist[3]);
        for (int b = 0; b < 6; b++) { printf("  wg %d:", b); for (int w = 0; w < threads / 64; w++) printf(" %08x(simd %u)", h[b * (threads / 64) + w], (h[b * (threads / 64) + w] >> 4) & 3); printf("\n"); }
        hipFree(d);
    }
    return 0;
}
