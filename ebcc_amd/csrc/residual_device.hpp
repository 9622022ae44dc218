// residual_device.hpp - device helpers shared by the residual-layer kernels.
#pragma once

#include "residual.hpp"

namespace ebcc {

// Coefficient i of the decoder's grid after the first B SPIHT bits of the stream (spiht_re.c:319-430 semantics), from
// the encoder's bookkeeping: the ordinal of the coefficient's significance bit, its LSP slot and the per-step refinement
// offsets (rbase / rreach: FrameState::refine_base / step_reached, 32 entries each).
// (o, c, slot): the coefficient's significance-bit ordinal, value and LSP slot
__device__ inline float prefix_value_of(uint32_t o, int c, uint32_t slot, unsigned long long B, const unsigned int *rbase, const unsigned int *rreach)
{
    float out = 0.0f;
    if (o != 0xFFFFFFFFu && (unsigned long long) o <= B) {
        const unsigned int a = (unsigned int) (c < 0 ? -c : c);
        const int ss = 31 - __clz(a);
        unsigned int mag = 1u << ss;                                 // spiht_re.c:338,370
        for (int s = ss - 1; s >= 0; --s) {
            if (!rreach[s]) break;
            const unsigned long long ord = (unsigned long long) rbase[s] + slot + 1;   // 1-based ordinal of the refinement bit
            if (ord > B + 1) break;                                  // the bit crossing the budget is still applied (:418-426)
            mag |= a & (1u << s);
        }
        out = c < 0 ? -(float) mag : (float) mag;
    }
    return out;
}
__device__ inline float prefix_value(const int32_t *__restrict__ C, const uint32_t *__restrict__ so, const uint32_t *__restrict__ li, size_t i,
                                     unsigned long long B, const unsigned int *rbase, const unsigned int *rreach)
{
    const uint32_t o = so[i];
    if (o == 0xFFFFFFFFu || (unsigned long long) o > B) return 0.0f;
    return prefix_value_of(o, C[i], li[i], B, rbase, rreach);
}

}  // namespace ebcc
