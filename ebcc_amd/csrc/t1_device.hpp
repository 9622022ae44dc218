// t1_device.hpp - LDS-side helpers of the tier-1 MQ pass (state table with successor Qe, context slots).
#pragma once

#include "common.hpp"
#include "t1_core.hpp"

namespace ebcc {

struct LdsTableNext {
    // t1::mq_entry_next for every state code, 8 bytes per entry: the code IS the entry's byte offset
    uint32_t base;                 // LDS byte address of the table
    __device__ void operator()(uint32_t code, uint32_t &nxt, uint32_t &nqes) const
    {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 e = *(const __attribute__((address_space(3))) u32x2 *) (uintptr_t) (base + code);
        nxt = e.x; nqes = e.y;
    }
};
struct CtxSlotsLds {
    // the context slots of every lane (t1_core.hpp: mq_rows_interval), one word each: context c of lane l at (c * 64 + l) * 4
    uint32_t base;                 // LDS byte address of this lane's context 0
    static constexpr int kContexts = 20;          // 0 .. 18 and the null context
    static constexpr int kBytes = kContexts * 64 * 4;
    __device__ uint32_t handle(uint32_t c) const { return base + (c << 8); }
    __device__ uint32_t ld(uint32_t h) const { return *(const __attribute__((address_space(3))) uint32_t *) (uintptr_t) h; }
    __device__ void st(uint32_t h, uint32_t v) { *(__attribute__((address_space(3))) uint32_t *) (uintptr_t) h = v; }
    __device__ void words(uint32_t x[5]) const
    {
        for (int j = 0; j < 5; j++) {
            uint32_t v = 0;
            for (int k = 0; k < 4; k++) if (4 * j + k < t1::NCTX) v |= t1::mq_code_state(ld(handle((uint32_t) (4 * j + k))) >> 16) << (8 * k);
            x[j] = v;
        }
    }
};
// fills the 128-entry table of t1::mq_entry_next from all threads of the workgroup; the caller synchronises before the first use
__device__ inline void fill_mq_table_next(uint2 *tab_store)
{
    for (int i = (int) threadIdx.x; i < 128; i += (int) blockDim.x) { uint32_t nxt, nqes; t1::mq_entry_next(t1::mq_code((uint32_t) i), nxt, nqes); tab_store[i] = make_uint2(nxt, nqes); }
}

}  // namespace ebcc
