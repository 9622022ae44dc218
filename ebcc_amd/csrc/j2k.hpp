// j2k.hpp - JPEG 2000 base layer of the EBCC codec on gfx950.
// Replaces the OpenJPEG calls of the reference (/root/reference/src/ebcc_codec.c:105-180 encode,
// :1092-1136 decode): single tile, 1 component, 16-bit unsigned, irreversible 9/7, 5 decompositions,
// 64x64 code-blocks, LRCP, 1 quality layer with rate base_cr/2 - bit-exact with OpenJPEG 2.4.0.
#pragma once

#include <vector>

#include "common.hpp"
#include "residual.hpp"

namespace ebcc {

constexpr int kJ2kRes = 6;
constexpr int kJ2kBands = 3 * kJ2kRes - 2;
constexpr int kJ2kMaxPlanes = 26;       // bit-plane masks kept per code-block (Mb <= 16 + ... + guard)
constexpr int kJ2kMaxPasses = 3 * kJ2kMaxPlanes;
constexpr int kJ2kCblkBytes = 16 * 1024; // byte slot per code-block
constexpr int kT1StateWords = 66 + 4 * 64;   // S[-1..64], NEG, VIS, REF, SPS row masks

struct J2kBand {
    int res, orient, level;
    int x0, y0, x1, y1;          // band-domain bounds
    int offx, offy;              // placement inside the tile buffer
    int ncw, nch;                // code-block grid
    int first_block;             // index of its first code-block
    int expn, mant, numbps;      // QCD entry, Mb
    float step_enc, step_dec;    // encoder step (with sub-band gain) / decoder step
    double norm;                 // synthesis norm used in the distortion weights
    int tree_off;                // offset of its tag-tree nodes in the per-frame node arrays
    int tree_levels;
    int lvl_w[12], lvl_h[12], lvl_off[12];
};

struct J2kBlock {
    int band;
    int x, y;                    // top-left in the tile buffer
    int w, h;
    int cx, cy;                  // position in the band's code-block grid
};

// Geometry of a tile of H x W samples whose first row is row ty0 of the image (device + host copies).  A context
// whose frames are the frames themselves has one geometry (ty0 = 0).  A context whose frames are the tiles of
// chunks of `period` frames stacked along y (one JPEG 2000 image per chunk, a tile per frame) has `period`
// geometries, frame f using number f % period: away from the image origin a tile has its own sub-band extents,
// low/high-pass parity and code-block partition (T.800 B.5-B.7) unless its height is a multiple of 64 * 2^5.
// Per-code-block arrays are laid out with the common `stride` >= nblocks of every geometry of the context.
struct J2kGeom {
    int W, H, nblocks, nbands, tree_nodes;
    int ty0;                      // first row of the tile in the image
    int period, stride;           // geometries in this context; code-block slots per frame
    int rw[kJ2kRes], rh[kJ2kRes];
    int ry0[kJ2kRes];             // first row of every resolution in its own coordinates (its parity = first sample is a high-pass one)
    int res_first[kJ2kRes + 1];   // first code-block of every resolution (packet order)
    J2kBand bands[kJ2kBands];
};

// the geometry, code-blocks and code-block map of a frame of the context
__host__ __device__ inline int j2k_geom_index(const J2kGeom *g, int frame) { return g->period > 1 ? frame % g->period : 0; }
__host__ __device__ inline const J2kGeom &j2k_frame_geom(const J2kGeom *g, int frame) { return g[j2k_geom_index(g, frame)]; }
__host__ __device__ inline const J2kBlock *j2k_frame_blocks(const J2kGeom *g, const J2kBlock *blocks, int frame)
{
    return blocks + (size_t) j2k_geom_index(g, frame) * (size_t) g->stride;
}
__host__ __device__ inline const std::uint16_t *j2k_frame_blkmap(const J2kGeom *g, const std::uint16_t *map, int frame)
{
    return map + (size_t) j2k_geom_index(g, frame) * ((size_t) g->W * (size_t) g->H);
}

constexpr int kJ2kSymCap = 4096 * (kJ2kMaxPlanes + 1) + 2048;  // decision bytes budgeted per code-block (64 of them share a group's rows)
constexpr int kJ2kSegCount = kJ2kMaxPlanes * 3 * 16;           // segments of a code-block: (bit-plane, pass type, stripe), t1_core.hpp
constexpr int kJ2kSymRows = (4096 * (kJ2kMaxPlanes + 2)) / 16 + kJ2kSegCount;   // 16-decision rows of a code-block's stream: decisions (a zero-coding / refinement decision per sample and plane, a sign, at most 3 run-length decisions per column and stripe) + one partial row per segment
constexpr int kJ2kCkptPerBlock = kJ2kMaxPasses * 16;   // checkpoint slots of one code-block (pass-major, then stripe)

// Checkpoint storage: one 32-byte record per slot, the slots of a code-block contiguous: [code-block][slot = pass * 16 +
// stripe][8 words] = { a | ct << 16, pos, the 19 context states one byte each in five words, c }.  A record is one
// 32-byte sector: the MQ pass writes a, the shift count (word 1) and the contexts from its interval chain and the low
// half of the ENCODER's c (word 7) from its code chain; the finalising sweep turns words 0, 1 and 7 into the decoder's
// registers; the restart reads the record whole.
constexpr int kJ2kCkptWords = 8;
__host__ __device__ inline size_t j2k_ckpt_block_bytes() { return (size_t) kJ2kCkptWords * kJ2kCkptPerBlock * 4; }
struct J2kCkptView {
    std::uint32_t *rec;            // first record of this code-block
    __host__ __device__ std::uint32_t *slot(std::uint32_t i) const { return rec + i * kJ2kCkptWords; }
    __host__ __device__ static J2kCkptView of(void *all, size_t gid)
    {
        return J2kCkptView{(std::uint32_t *) ((unsigned char *) all + gid * j2k_ckpt_block_bytes())};
    }
};

struct J2kFrame {                 // per-frame scalars (device)
    float cr;                     // rate of the current probe
    float target;                 // error target of the current search
    int maxlen;
    int body_bytes;               // packet bytes of the current layer assignment
    int stream_bytes;             // whole codestream
    unsigned long long nbad;      // count(|x - d| > target) of the last decode
    double err_sum;               // sum(x - d)
    int overflow;                 // bit 0: a code-block outgrew its byte slot; bit 1: a group's decisions outgrew its rows of SYM (retry: launch_j2k_tier1)
    int keep;                     // probes launched with keep_field == 2 store their decoded field only where this is set (search.hip)
    float hdr_share;              // this tile's share of the main header in the byte budget (opj_j2k_update_rates:
                                  // main header bytes / number of tiles); 0 = a single tile = all 135 bytes
    // A probe whose only use is the search's "feasible or not" (search.hip: every probe but the final one of a search and
    // the one that restores its decode) may stop counting once bad_seen >= bad_limit: from that count on the quantile is
    // below the target by more than the search's tolerance whatever the rest of the frame holds.  0: count everything.
    unsigned int bad_limit;
    unsigned int bad_seen;        // running count of the probe in flight (reset by k_finish_reduce)
};

struct J2kBuffers {
    J2kGeom geom;                 // host copy of the first geometry
    std::vector<J2kGeom> geoms;   // host copies of all of them (J2kGeom::period entries)
    J2kGeom *d_geom;              // [period]
    J2kBlock *d_blocks;           // [period][stride]
    std::uint16_t *d_blkmap;      // [period][H*W] code-block id of every tile-buffer position
    int max_frames;
    float *B;                     // [frames][H*W] tile buffer (coefficients / samples)
    float *B2;                    // [frames][H*W] second tile buffer: the fused inverse levels alternate between the two
    int32_t *Q6;                  // [frames][H*W] quantised coefficients with 6 fractional bits
    float *DEC;                   // [frames][H*W] last decoded field (fp32, de-normalised)
    unsigned long long *BP;       // [groups][planes][64][64] bit-plane row masks, lane-interleaved
    unsigned long long *SGN;      // [groups][64][64] sign row masks
    unsigned long long *SUF;      // [groups][planes+2][64][64] suffix-OR of BP over planes >= p (significance above a plane)
    unsigned long long *SPS;      // [groups][64][64] "became significant in a propagation pass" row masks (encoder)
    unsigned long long *VISP;     // [groups][planes][64][64] visited masks at the end of each plane's propagation pass
    void *ckpt;                   // [frames*nblocks][passes * 16 stripes][8 words] MQ-decoder checkpoints at every stripe start of every coding pass (J2kCkptView)
    uint8_t *SYM;                 // [groups][sym_rows / 2][64 lanes][2 rows][16] decision rows (rows in pairs: j2k_analysis.hip sym_row_offset) of the segmented two-phase encoder (t1_core.hpp: row format); null: single-kernel encoder
    int sym_rows;                 // 1-KB rows of SYM per group (kJ2kSymRows; EBCC_HIP_SYM_ROWS overrides, for tests of the retry)
    std::uint16_t *seglen;        // [groups][kJ2kSegCount][64] decisions of every segment of every code-block, then its first row in the block's stream
    std::uint32_t *lanerows;      // [frames*nblocks] rows in the block's stream
    int *qplane;                  // [frames*nblocks] where the current probe's decode restarts: pass | stripe << 8 (-1: nothing, -2: V is up to date)
    int *lastnp;                  // [frames*nblocks] passes the code-block had in the frame's previous probe decode (-1: none yet; reset by the analysis)
    unsigned long long *T1S;      // [groups][kT1StateWords][64] tier-1 state
    int *blkmax;                  // [frames*nblocks] max |q6|
    int *numbps;                  // [frames*nblocks]
    int *totalpasses;             // [frames*nblocks]
    int *cblk_len;                // [frames*nblocks]
    int *rates;                   // [frames*nblocks][kJ2kMaxPasses]
    double *disto;                // [frames*nblocks][kJ2kMaxPasses]
    int *npass;                   // [frames*nblocks] passes in the current layer
    // speculative rate allocation of the device-driven search (search.hpp): the layers of the two rates the search may ask
    // for next, worked out beside the current probe's decode
    float *cand_cr;               // [frames][2] candidate rates (<= 0: none)
    int *cand_out;                // [frames][2][3] body bytes, stream bytes, byte budget of a candidate's layer
    int *cand_npass;              // [2][frames*nblocks] its passes per code-block
    int *cand_sel;                // [frames] candidate whose layer becomes the current one (-1: none)
    int *have_rate;               // [frames] 1: the current layer was taken over from a candidate (k_rate skips the frame)
    int *rate_path;               // [frames][1536][3] packet bytes at the steps of the rate bisection k_rate has visited, as a trie
    int *rate_path_n;             // [frames] nodes in it; reset by the analysis
    // what every k_rate call for a frame needs and the first one works out (reset by the analysis with the trie): the zero-bit-plane
    // tree minima, the table offsets, the slope range and the pass tables packed the way they are staged in LDS
    struct RateCache {
        int *ok;                  // [frames] 1: filled
        double *mnmx;             // [frames][2]
        short *mval;              // [frames][nodes_cap]
        int *off;                 // [frames][stride + 1]
        unsigned short *crate;    // [frames][cap] rates, code-block b's passes at off[b] ..
        double *cdisto;           // [frames][cap]
        int nodes_cap, cap;
    } rate_cache;
    uint8_t *cblk_bytes;          // [frames*nblocks][kJ2kCblkBytes]
    uint8_t *stream;              // [frames][stream_cap] codestream
    size_t stream_cap;
    int32_t *V;                   // [frames][H*W] tier-1 decoder output (half units), decode path
    int *dec_table;               // [frames*nblocks][4]: offset, len, numbps, npasses (decode path)
    int *dec_order;               // [frames*nblocks] code-blocks by falling segment length, then [128] counters (decode path)
    int *mq_order;                // [groups*64] code-blocks by falling number of decision rows, then [256] counters (the MQ pass takes them in this order)
    J2kFrame *jf;                 // [frames]
    FrameState *fs;               // [frames] (shared with the residual layer)
    double *partial;              // [frames][kPartials]
    unsigned long long *partial_u;// [frames][kPartials]
};

// Code-blocks per wavefront in the tier-1 kernels.  The coders are serial and branchy: a wave executes the
// union of its lanes' paths, and with 76 288 code-blocks a full 64-lane mapping leaves ~1 wave per SIMD, so
// the SIMDs sit idle between dependent instructions.  Fewer code-blocks per wave = more waves in flight and
// smaller unions.  Defaults measured on MI355X (profiles/): decision pass 64 and MQ pass 64 (their lanes share one
// instruction stream), probe restart 4, decode 4 (the per-sample decoder is branchy: fewer lanes, smaller unions); EBCC_T1_LPW="<n>" or "<a>,<b>,<c>,<d>" overrides them.
enum T1Kernel { T1_ENCODE = 0, T1_MQ = 1, T1_RESUME = 2, T1_DECODE = 3 };
int t1_lanes_per_wave(int kernel, int total_blocks = 0);   // (total_blocks: code-blocks in the launch, for the encoder kernels)

J2kGeom make_j2k_geom(int H, int W, std::vector<J2kBlock> &blocks, int ty0 = 0);
int j2k_selfcheck_div65535();   // mismatches of the division-free s / 65535.0f of the fused inverse level (0 expected)

// ---- launchers (asynchronous on s) ---------------------------------------------------------------
// check_nan_inf + findMinMaxf (ebcc_codec.c:598-605,515-533) -> fs.minv/maxv/const_field/has_nonfinite
void launch_input_stats(const float *data, int n_frames, size_t n_pix, FrameState *fs, hipStream_t s);
// scale to u16 (:686-689), DC level shift, forward 9/7, quantisation, tier-1 of every code-block,
// distortion tables: everything of opj_encode that does not depend on the rate
void launch_j2k_analysis(const float *data, const J2kBuffers &jb, int n_frames, hipStream_t s);
// the tier-1 stage of the analysis again with the single-kernel encoder (after J2kFrame::overflow bit 1: the decisions of
// a group of code-blocks outgrew their rows of the segmented encoder's buffer); also called by the analysis itself
void launch_j2k_tier1(const J2kBuffers &jb, int n_frames, hipStream_t s, bool single_kernel);
// to be called with the per-frame scalars fetched after the analysis: runs that retry if any frame asks for it (the
// caller then fetches them again)
bool j2k_tier1_retry(const J2kBuffers &jb, int n_frames, const J2kFrame *host_jf, hipStream_t s);
// rate allocation for jf[f].cr (opj_tcd_rateallocate) -> npass, jf.body_bytes/stream_bytes
void launch_j2k_rate(const J2kBuffers &jb, int n_frames, const int *d_active, hipStream_t s, const int *have_rate = nullptr);
void j2k_probe_hist_dump(const char *what);                          // EBCC_HIP_T1_STATS=1 diagnostics (j2k_rate.hip)
// the same for the candidate rates jb.cand_cr (two per frame) into the candidate slots; launch_j2k_rate_publish makes
// candidate jb.cand_sel[f] the frame's current layer and sets jb.have_rate
void launch_j2k_rate_candidates(const J2kBuffers &jb, int n_frames, const int *d_active, hipStream_t s);
void launch_j2k_rate_publish(const J2kBuffers &jb, int n_frames, hipStream_t s);
// write the codestream of the current layer assignment into jb.stream
void launch_j2k_write(const J2kBuffers &jb, int n_frames, const int *d_active, hipStream_t s);
// what opj_decode returns for the current layer assignment (decoded in place from the code-block
// slots), mapped to fp32 as :1130 -> jb.DEC (keep_field; a probe that is only asked for the statistics leaves jb.DEC
// as it is; keep_field == 2: per frame, where jf[f].keep is set), and the error statistics against `data`
void launch_j2k_probe_decode(const float *data, const J2kBuffers &jb, int n_frames, const int *d_active, hipStream_t s,
                             int keep_field = 1);
// true decode of codestreams whose packet headers were parsed on the host into jb.dec_table
// (fs[f].minv/maxv must hold the header's values); result in jb.DEC.  host_table: the host's copy of jb.dec_table, from
// which the launch sizes its waves (null: the fixed tiers tuned for 256 frames)
void launch_j2k_decode(const J2kBuffers &jb, int n_frames, hipStream_t s, const int *host_table = nullptr);
void plan_decode_lanes(const int *host_table, int total, int out[4]);   // (what launch_j2k_decode chooses; ebcc_hip_plan_decode_lanes)

}  // namespace ebcc
