// valu_rate.hip - issue rate of a few vector instructions on gfx950 (development tool): every wave runs long chains of the
// instruction, four independent chains per lane so that latency does not limit; the time per instruction and wave shows which
// ones take more than one four-cycle issue slot.   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/gpu/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIter = 4096;
template <int OP>
__global__ __launch_bounds__(256) void k_rate_probe(unsigned long long *out, unsigned int s)
{
    unsigned long long a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    unsigned int b0 = threadIdx.x + 1, b1 = b0 * 3, b2 = b0 * 5, b3 = b0 * 7;
    for (int i = 0; i < kIter; i++) {
        if (OP == 0) { asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(b0) : "v"(s)); asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(b1) : "v"(s));
                       asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(b2) : "v"(s)); asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(b3) : "v"(s)); }
        if (OP == 1) { asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a0) : "v"(s)); asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a1) : "v"(s));
                       asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a2) : "v"(s)); asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a3) : "v"(s)); }
        if (OP == 2) { asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a0) : "v"(s)); asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a1) : "v"(s));
                       asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a2) : "v"(s)); asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a3) : "v"(s)); }
        if (OP == 3) { asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(b0) : "v"(s)); asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(b1) : "v"(s));
                       asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(b2) : "v"(s)); asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(b3) : "v"(s)); }
        if (OP == 4) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b0) : "v"(s)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b1) : "v"(s));
                       asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b2) : "v"(s)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b3) : "v"(s)); }
        if (OP == 5) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(b0) : "v"(s)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(b1) : "v"(s));
                       asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(b2) : "v"(s)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(b3) : "v"(s)); }
        if (OP == 6) { asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(a0)); asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(a1));
                       asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(a2)); asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(a3)); }
        if (OP == 7) { asm volatile("v_mov_b32 %0, %1" : "+v"(b0) : "v"(b1)); asm volatile("v_mov_b32 %0, %1" : "+v"(b1) : "v"(b2));
                       asm volatile("v_mov_b32 %0, %1" : "+v"(b2) : "v"(b3)); asm volatile("v_mov_b32 %0, %1" : "+v"(b3) : "v"(b0)); }
        if (OP == 8) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(b0) : "v"(s) : "s10", "s11"); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(b1) : "v"(s) : "s10", "s11");
                       asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(b2) : "v"(s) : "s10", "s11"); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(b3) : "v"(s) : "s10", "s11"); }
        if (OP == 9) { asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(b0) : "v"(s)); asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(b1) : "v"(s));
                       asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(b2) : "v"(s)); asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(b3) : "v"(s)); }
        if (OP == 10) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(b0) : "v"(s)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(b1) : "v"(s));
                        asm volatile("v_add_u32 %0, %0, %1" : "+v"(b2) : "v"(s)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(b3) : "v"(s)); }
        if (OP == 11) { asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(b0) : "v"(s) : "vcc"); asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(b1) : "v"(s) : "vcc");
                        asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(b2) : "v"(s) : "vcc"); asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(b3) : "v"(s) : "vcc"); }
        if (OP == 12) { asm volatile("v_cmp_eq_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b0) : "v"(s) : "vcc"); asm volatile("v_cmp_eq_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b1) : "v"(s) : "vcc");
                        asm volatile("v_cmp_eq_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b2) : "v"(s) : "vcc"); asm volatile("v_cmp_eq_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b3) : "v"(s) : "vcc"); }
        if (OP == 13) { asm volatile("v_bfe_u32 %0, %0, %1, 9" : "+v"(b0) : "v"(s)); asm volatile("v_bfe_u32 %0, %0, %1, 9" : "+v"(b1) : "v"(s));
                        asm volatile("v_bfe_u32 %0, %0, %1, 9" : "+v"(b2) : "v"(s)); asm volatile("v_bfe_u32 %0, %0, %1, 9" : "+v"(b3) : "v"(s)); }
        if (OP == 14) { asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(b0) : "v"(s)); asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(b1) : "v"(s));
                        asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(b2) : "v"(s)); asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(b3) : "v"(s)); }
        if (OP == 15) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(b0) : "v"(s)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(b1) : "v"(s));
                        asm volatile("v_and_b32 %0, %0, %1" : "+v"(b2) : "v"(s)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(b3) : "v"(s)); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
}
template <int OP>
static void run(const char *name, unsigned long long *d)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int blocks = 256 * 8;                                         // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k_rate_probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_rate_probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double insts_per_simd = (double) blocks * 4 / 1024.0 * kIter * 4;        // wave instructions one SIMD issues
    printf("%-16s %8.3f ms  %.2f ns per wave instruction and SIMD (4 cycles at 2.4 GHz = 1.67 ns)\n", name, ms, ms * 1e6 / insts_per_simd);
}
int main()
{
    unsigned long long *d; CHECK(hipMalloc(&d, (size_t) 256 * 8 * 256 * 8));
    run<0>("v_lshrrev_b32", d); run<1>("v_lshrrev_b64", d); run<2>("v_lshlrev_b64", d); run<3>("v_alignbit_b32", d);
    run<4>("v_cndmask_b32", d); run<5>("v_mul_u32_u24", d); run<6>("v_lshl_add_u64", d); run<7>("v_mov_b32", d);
    run<8>("v_cndmask e64 sgpr", d); run<9>("v_bfi_b32", d); run<10>("v_add_u32", d); run<11>("v_cmp + v_add", d); run<12>("cmp+nop+cndmask", d);
    run<13>("v_bfe_u32", d); run<14>("v_and_or_b32", d); run<15>("v_and_b32", d);
    return 0;
}
