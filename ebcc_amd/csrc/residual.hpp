// residual.hpp - device-side residual layer of the EBCC codec (pad + DC + CDF 9/7 + SPIHT).
// Replaces reference src/spiht/{dwt.h,spiht_re.c,ml.h,bitio.h}; every launcher cites what it covers.
#pragma once

#include "common.hpp"

namespace ebcc {

// Per-frame scalar state living in device memory (one entry per frame of the batch).
struct FrameState {
    // ---- input statistics (reference src/ebcc_codec.c:598-605, :515-533)
    unsigned long long min_key, max_key;   // (order_key << 32) | tie-break index
    float minv, maxv;
    int has_nonfinite;
    int const_field;
    // ---- residual statistics (:712-719, :730-746)
    unsigned long long rmin_key, rmax_key;
    float rmin, rmax;
    // ---- DC removal (src/spiht/dwt.h:319-334)
    double dc_sum;
    float dc;
    int dc_uncertain;
    // ---- SPIHT encode (src/spiht/spiht_re.c:432-475)
    int cmax;                               // max |coefficient|
    int top_step;
    unsigned long long budget;              // bits0 - 128
    unsigned long long emitted;             // SPIHT bits written (<= budget + 1)
    unsigned int stream_bytes;
    unsigned int refine_base[32];           // 0-based index of the first refinement bit of each step
    unsigned int refine_count[32];          // LSP entries refined at that step
    unsigned int step_reached[32];          // 1 if the refinement pass of the step was entered
    // ---- probe statistics (:477-513)
    unsigned int maxerr_bits;               // float bits of max |x - (d + r)| (non-negative => monotone)
    double err_sum;
    int err_sum_inexact;
    unsigned long long nbad;                // count(|x - d| > target)
    float exit_above;                       // > 0: a probe of the truncation search may stop once its running maximum exceeds
                                            // this (the search then only asks "max error > target?", search.hip); 0: exact
    // ---- decode
    int dec_stages, dec_dc, dec_top_step;
    unsigned long long dec_budget;
};

struct ResidualBuffers {
    Grid g;
    int max_frames;
    size_t np;                  // padded pixels per frame
    float *A, *T;               // [frames][np] transform ping-pong
    int32_t *C;                 // [frames][np] integer coefficients
    int32_t *D, *G;             // [frames][np] descendant / grand-descendant max magnitude
    uint32_t *lip, *lsp, *lis0, *lis1;   // [frames][np]
    uint32_t *sigord, *lspidx;  // [frames][np] 1-based ordinal of the significance bit / LSP slot
    uint32_t *stream;           // [frames][stream_words] SPIHT byte stream (device copy)
    size_t stream_words;
    double *partial;            // [frames][kPartials] deterministic reduction scratch
    FrameState *fs;             // [frames]
};
constexpr int kPartials = 256;

// Cut slots: several candidate cuts of a frame's SPIHT stream probed in ONE round of the truncation search
// (/root/reference/src/ebcc_codec.c:777-795 visits one cut per iteration; which cut comes next depends on the outcome, so a
// round probes the current cut AND the cuts either outcome leads to - the bisection tree a few levels deep - and the
// search then walks the tree with the real outcomes: same cuts visited, same result, a fraction of the rounds).
// A probe of slot v reads the bookkeeping, the frame and the base layer of frame `frame_of[v]` and keeps everything it
// writes - the coarse reconstruction, the coarser levels' results, the statistics - in the slot's own storage.
struct CutSlots {
    int capacity = 0;                   // slots
    size_t stride = 0;                  // floats per slot of A / T / D: the rows above the finest level's detail bands (np / 2)
    float *A = nullptr, *T = nullptr, *D = nullptr;   // [capacity][stride] coarse reconstruction, level results (alternating)
    FrameState *fs = nullptr;           // [capacity] the slot's copy of its frame's state + its own probe statistics
    double *partial = nullptr;          // [capacity][kPartials]
    unsigned long long *bits = nullptr; // [capacity] the cut
    int *active = nullptr;              // [capacity]
    int *frame_of = nullptr;            // [capacity]
};
// true: launch_prefix_synthesis_slots can be used for this grid (the fused kernels take it; otherwise: one cut per round)
bool prefix_slots_supported(const ResidualBuffers &rb, int n_slots);
// slot v's state = its frame's (everything a probe reads of it: budget, refinement bookkeeping, dc, residual range)
void launch_slots_setup(const ResidualBuffers &rb, const CutSlots &cs, int n_slots, hipStream_t s);
// launch_prefix_synthesis_stats for the active slots: cut cs.bits[v] of frame cs.frame_of[v] -> cs.fs[v].maxerr_bits / err_sum
void launch_prefix_synthesis_slots(const float *data, const float *decoded, const ResidualBuffers &rb, const CutSlots &cs, int n_slots, hipStream_t s);

// ---------------------------------------------------------------- launchers (all asynchronous on `s`)

// min/max of r = data - decoded  -> fs.rmin/rmax   (src/ebcc_codec.c:712-716,730-733)
void launch_residual_minmax(const float *data, const float *decoded, int n_frames, size_t n_pix,
                            FrameState *fs, hipStream_t s);

// load_image + sub_dc prologue: A = mirror-pad(((r - rmin) / (rmax - rmin)) * 255), fs.dc  (dwt.h:41-78,319-334)
void launch_pad_and_dc(const float *data, const float *decoded, const ResidualBuffers &rb, int n_frames,
                       const int *d_active, hipStream_t s);

// same but from an already normalised [0,1] image (unit tests / spiht_encode entry point)
void launch_pad_and_dc_from_image(const float *image, const ResidualBuffers &rb, int n_frames, hipStream_t s);

// dwt2full + normalize: A (minus dc) -> C, fs.cmax, D, G   (dwt.h:293-303,355-368; spiht_re.c:54-60,160-206)
void launch_analysis(const ResidualBuffers &rb, int n_frames, const int *d_active, hipStream_t s);

// spiht_encode_process + IMS header -> rb.stream, fs.emitted/stream_bytes, sigord/lspidx  (spiht_re.c:208-317,448-464)
// fs[f].budget must be set (bits0 - 128) and bits0 per frame is passed for the header.
void launch_spiht_encode(const ResidualBuffers &rb, int n_frames, const unsigned long long *d_bits0, const int *d_active,
                         hipStream_t s);

// fs[f].budget = d_bits0[f] - 128 (and exact probes) for the active frames: what the host used to set between the analysis
// and the encoder; d_trunc_bits[f] = 8 * fs[f].stream_bytes: the cut "everything", for the probe of src/ebcc_codec.c:749-754
void launch_residual_budget(const ResidualBuffers &rb, int n_frames, const unsigned long long *d_bits0, const int *d_active, hipStream_t s);
void launch_whole_stream_cut(const ResidualBuffers &rb, int n_frames, unsigned long long *d_trunc_bits, const int *d_active, hipStream_t s);

// Decoder state after the first `trunc_bits[f]` stream bits, rebuilt from the encoder's bookkeeping:
// A = coefficient grid as spiht_decode_process would leave it  (spiht_re.c:319-430 semantics)
void launch_reconstruct(const ResidualBuffers &rb, int n_frames, const unsigned long long *d_trunc_bits,
                        const int *d_active, hipStream_t s);

// the same for the LL quadrant of the finest level only (what the coarser synthesis levels read)
void launch_reconstruct_coarse(const ResidualBuffers &rb, int n_frames, const unsigned long long *d_trunc_bits,
                               const int *d_active, hipStream_t s);
// launch_reconstruct + launch_synthesis_stats in one: the probe of the truncation search (the finest level's detail bands
// go from the bookkeeping straight into the streaming column pass)
void launch_prefix_synthesis_stats(const float *data, const float *decoded, const ResidualBuffers &rb, int n_frames,
                                   const unsigned long long *d_trunc_bits, const int *d_active, hipStream_t s);

// idwt2full on A (dwt.h:305-317); result in A
void launch_synthesis(const ResidualBuffers &rb, int n_frames, const int *d_active, hipStream_t s);
// The same with the last row pass consuming the rows instead of storing the grid:
//   _stats:    add_dc + crop + /255 (dwt.h:336-353, spiht_re.c:512-516), r*(rmax-rmin)+rmin and the error statistics
//              of src/ebcc_codec.c:477-501 against data / decoded -> fs.maxerr_bits, fs.err_sum;
//   _tail_add: out += r*(rmax-rmin)+rmin (src/ebcc_codec.c:1306-1308), dc from fs.dec_dc; after launch_synthesis_head
//              (every pass but the last, possibly on another stream).
void launch_synthesis_stats(const float *data, const float *decoded, const ResidualBuffers &rb, int n_frames, const int *d_active,
                            hipStream_t s);
void launch_synthesis_head(const ResidualBuffers &rb, int n_frames, const int *d_active, hipStream_t s);
void launch_synthesis_tail_add(float *out, const ResidualBuffers &rb, int n_frames, const int *d_active, hipStream_t s);

// plain spiht_decode output image in [0,1] (spiht_re.c:508-516) for the unit entry point
void launch_emit_image(float *image_out, const ResidualBuffers &rb, int n_frames, hipStream_t s);

// spiht_decode header parse + spiht_decode_process: stream bytes -> A (float coefficient grid), fs.dec_*
// d_streams: [frames][stream_words*4] bytes; d_sizes / d_num_bits per frame.
void launch_spiht_decode(const uint8_t *d_streams, size_t stream_stride, const unsigned long long *d_sizes,
                         const unsigned long long *d_num_bits, const ResidualBuffers &rb, int n_frames,
                         const int *d_active, hipStream_t s);

// mismatches of the division-free forms of the residual synthesis against the divisions they replace (0 expected)
int residual_selfcheck_divisions();

}  // namespace ebcc
