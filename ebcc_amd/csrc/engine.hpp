// engine.hpp - per-device context of the MI355X EBCC engine.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "residual.hpp"

struct ebcc_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    size_t max_frames = 0;
    int height = 0, width = 0;
    int tile_period = 1;                   // > 1: frame f is tile f % tile_period of an image of that many tiles stacked along y (j2k.hpp)
    size_t n_pix = 0;                      // height * width
    size_t bytes = 0;                      // device bytes owned
    ebcc::ResidualBuffers rb{};
    // per-frame parameter arrays (device) and pinned host mirrors
    unsigned long long *d_u64a = nullptr, *d_u64b = nullptr, *d_u64c = nullptr;
    int *d_active = nullptr;
    unsigned long long *h_u64a = nullptr, *h_u64b = nullptr, *h_u64c = nullptr;
    int *h_active = nullptr;
    ebcc::FrameState *h_fs = nullptr;      // pinned
    // pinned mirrors of the small per-round transfers of the rate searches and of the decode tables: copies from / to
    // pageable memory are synchronous inside the runtime and hold up the launches of the other slices
    void *h_jf = nullptr;                  // [max_frames] J2kFrame (j2k.hpp)
    int *h_act = nullptr;                  // [2 * max_frames]: probe mask, residual mask
    int *h_table = nullptr;                // [max_frames * code-block slots * 4] decode tables (lazily sized by j2k_create)
    void *d_search = nullptr, *h_search = nullptr;   // [2][max_frames] DevChunk: state of the device-driven searches (search.hpp; second half: the overlapped search), pinned mirror
    int *d_counter = nullptr, *h_counter = nullptr;  // [4] small device counters and their pinned mirror
    std::vector<void *> allocs;
    void *j2k = nullptr;                   // base-layer state (j2k.hpp)
    hipStream_t stream2 = nullptr;          // second stream of the engine (decode: residual layer beside the base layer)
    hipEvent_t ev_a = nullptr, ev_b = nullptr;   // ordering between the two streams (created on first use)
    // Sub-batch engines of the frames API: a batch is cut into a few slices that run concurrently, each on its
    // own stream and host thread (the kernels of one slice are latency-bound and leave most of the chip idle).
    std::vector<ebcc_hip_ctx *> lanes;
    ebcc_hip_ctx *twin = nullptr;           // second engine set of ebcc_hip_encode_shard (batch k + 1 computes while batch k is in its entropy stage)
    bool twin_failed = false;               // (it could not be made: not tried again until ebcc_hip_release_engines)
    // Staging for the many small per-frame transfers (codestreams, SPIHT bytes): the pieces are packed by a kernel
    // into one device buffer and cross PCIe as one copy into / out of one pinned buffer (engine.hip: stage_*).
    uint8_t *h_stage = nullptr, *d_stage = nullptr;
    size_t stage_cap = 0;
    // device image of the host arrays the reference-compatible entry points are handed (kept between calls)
    float *d_io = nullptr;
    size_t io_cap = 0;                      // bytes
    uint8_t *h_bounce = nullptr;            // 2 x kBounceBytes pinned: pageable host arrays cross PCIe through it (host_codec.hip)
    unsigned long long *h_pack = nullptr, *d_pack = nullptr;   // [2 pieces per frame][offset, length]
    ebcc::CutSlots cut{};                   // look-ahead storage of the truncation search (made on first use: ensure_cut_slots)
    bool cut_failed = false;                // (no memory for it: the search probes one cut per round)
};

namespace ebcc {

void set_error(const char *fmt, ...);
// ebcc_hip_create for frames that are the tiles of images of `tile_period` tiles each (1: plain frames)
ebcc_hip_ctx *create_engine(int device, size_t max_frames, size_t height, size_t width, int tile_period);
template <typename T>
T *ctx_alloc(ebcc_hip_ctx *ctx, size_t count);

// Device -> host: the first len[f] bytes of the slot src + f * stride of every frame with len[f] > 0, as ONE copy.
// Fills off[f] (offsets into ctx->h_stage, 16-byte aligned) and returns after the data has arrived.
void stage_download(ebcc_hip_ctx *ctx, const uint8_t *src, size_t stride, const size_t *len, size_t *off, size_t n, hipStream_t s);
// Host -> device: stage_reserve makes room for m <= 2 * max_frames pieces of len[k] bytes in ctx->h_stage (off[k] filled
// in); after the caller has written them stage_send ships them as one copy (asynchronous on s), and stage_scatter
// moves pieces first .. first + count - 1 to the slots dst, dst + stride, ... (on any stream ordered after the send).
void stage_reserve(ebcc_hip_ctx *ctx, const size_t *len, size_t *off, size_t m);
void stage_send(ebcc_hip_ctx *ctx, size_t m, hipStream_t s);
void stage_scatter(ebcc_hip_ctx *ctx, uint8_t *dst, size_t stride, size_t first, size_t count, hipStream_t s);

// cut slots for `capacity` simultaneous probes (residual.hpp); false: not available (unsupported grid or no memory)
bool ensure_cut_slots(ebcc_hip_ctx *ctx, int capacity);

// header of a SPIHT stream against the context's grid and a usable bit budget (non-zero: reject, message set)
int check_ims_header(ebcc_hip_ctx *ctx, const uint8_t *b, size_t n, size_t num_bits);

// synchronous copy of the frame states to ctx->h_fs
void fetch_frame_states(ebcc_hip_ctx *ctx, size_t n_frames);
void push_frame_states(ebcc_hip_ctx *ctx, size_t n_frames);

}  // namespace ebcc
