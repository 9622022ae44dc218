// t1_device.hpp - LDS-side helpers shared by the tier-1 kernels (MQ state table and context states as state codes).
#pragma once

#include "common.hpp"
#include "t1_core.hpp"

namespace ebcc {

struct LdsTable2 {
    // t1::mq_entry2 for every state code, 8 bytes per entry; the code IS the entry's byte offset
    uint32_t base;                 // LDS byte address of the table
    __device__ void operator()(uint32_t code, uint32_t &qe, uint32_t &next) const
    {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 e = *(const __attribute__((address_space(3))) u32x2 *) (uintptr_t) (base + code);
        qe = e.x; next = e.y;
    }
};
struct CtxLds2 {
    // the state codes of every lane, 16 bits each: context c of lane l at (c * 64 + l) * 2 - a handle (the byte address)
    // is one shift-add away from the decision byte
    uint32_t base;                 // LDS byte address of this lane's context 0
    static constexpr int kBytes = 32 * 64 * 2;
    __device__ uint32_t handle(uint32_t c) const { return base + (c << 7); }
    __device__ uint32_t ld(uint32_t h) const { return *(const __attribute__((address_space(3))) unsigned short *) (uintptr_t) h; }
    __device__ void st(uint32_t h, uint32_t v) { *(__attribute__((address_space(3))) unsigned short *) (uintptr_t) h = (unsigned short) v; }
    __device__ void words(uint32_t x[5]) const
    {
        for (int j = 0; j < 5; j++) {
            uint32_t v = 0;
            for (int k = 0; k < 4; k++) if (4 * j + k < t1::NCTX) v |= t1::mq_code_state(ld(handle((uint32_t) (4 * j + k)))) << (8 * k);
            x[j] = v;
        }
    }
};
// fills a 128-entry table of t1::mq_entry2 (8 bytes per state code) from all threads of the workgroup; the caller
// synchronises before the first use
__device__ inline void fill_mq_table2(uint2 *tab_store)
{
    for (int i = (int) threadIdx.x; i < 128; i += (int) blockDim.x) { uint32_t qe, nx; t1::mq_entry2(i, qe, nx); tab_store[i] = make_uint2(qe, nx); }
}

}  // namespace ebcc
