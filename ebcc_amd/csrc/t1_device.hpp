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

// Context tables of the serial tier-1 passes in LDS (t1_core.hpp: DirectCtx): 512 bytes per orientation class (orientation 0
// and 2 share one) indexed by the nine window bits, 256 bytes of sign contexts (context | xor bit << 7) indexed by sc_index.
struct LdsCtx {
    static constexpr int kBytes = 3 * 512 + 256;
    static constexpr bool kFieldPositions = true;  // the tables hold 7 x (context's index in its word of t1::Contexts): no multiply per decision
    uint32_t base = 0, zc_tab = 0;                 // LDS byte addresses
    __device__ void bind(int orient) { zc_tab = base + 512u * (orient == 1 ? 1u : (orient == 3 ? 2u : 0u)); }
    __device__ int zc(uint32_t idx) const { return (int) *(const __attribute__((address_space(3))) uint8_t *) (uintptr_t) (zc_tab + idx); }
    __device__ int sc(uint32_t idx, int &xb) const
    {
        const uint32_t v = *(const __attribute__((address_space(3))) uint8_t *) (uintptr_t) (base + 1536u + idx);
        xb = (int) (v >> 7);
        return (int) (v & 0x7Fu);
    }
};
// the tables, by evaluating DirectCtx (one definition of the contexts): worked out on the host once per device and kept in
// device memory; a workgroup copies them to LDS with a few 16-byte loads
inline void make_ctx_tables(uint8_t *store)
{
    for (int i = 0; i < 1536; i++) {
        t1::DirectCtx d;
        d.bind(i < 512 ? 0 : (i < 1024 ? 1 : 3));
        store[i] = (uint8_t) (7 * (d.zc((uint32_t) i & 511u) - t1::CTX_ZC0));
    }
    for (int i = 0; i < 256; i++) {
        t1::DirectCtx d;
        int xb = 0;
        const int c = d.sc((uint32_t) i, xb);
        store[1536 + i] = (uint8_t) ((7 * (c - 9)) | (xb << 7));
    }
}
__device__ inline void copy_ctx_tables(uint8_t *lds, const uint8_t *global, int tid, int nthreads)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (int i = tid; i < LdsCtx::kBytes / 16; i += nthreads) ((u32x4 *) lds)[i] = ((const u32x4 *) global)[i];
}

}  // namespace ebcc
