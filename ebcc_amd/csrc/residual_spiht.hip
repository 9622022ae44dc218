// residual_spiht.hip - SPIHT bit-plane coder on CDNA4 wavefronts (gfx950 only).
//
// The reference coder (src/spiht/spiht_re.c:208-430) walks three lists sequentially; the bit order is
// defined by list order.  Here one workgroup owns one frame and sweeps each list in chunks:
//   - every entry's bit string depends only on coefficient data (significance of a pixel, of a
//     descendant set via the precomputed maxima D/G), so a chunk is a data-parallel map;
//   - bit offsets, LSP/LIP/LIS append slots and the stable compaction of survivors come from one packed
//     exclusive scan (wave shuffles + LDS across waves);
//   - entries appended to the LIS during a pass land behind the sweep pointer and are reached by the same
//     sweep, which reproduces the reference's "list grows while being walked" order (spiht_re.c:237);
//   - bits are OR-ed into an LDS window and flushed as whole big-endian words.
// While encoding, every pixel records the ordinal of its significance bit and its LSP slot.  With the
// per-step refinement offsets this determines the decoder state after ANY prefix of the stream, which
// turns the reference's truncation search (17 full SPIHT decodes, ebcc_codec.c:777-795) into an
// element-wise reconstruction per probe.
#include "residual.hpp"
#include "residual_device.hpp"

namespace ebcc {

namespace {

// list entries swept per step (one per thread).  A step costs its scans and barriers, not its work: 512 entries per
// step take 5.4 ms per launch where 256 took 8.5, 1024 take 3.9 (tools/gpu/spiht_threads.sh, tools/gpu/spiht1024.sh).
#ifndef EBCC_SPIHT_ENC_THREADS
#define EBCC_SPIHT_ENC_THREADS 1024
#endif
constexpr int kEncThreads = EBCC_SPIHT_ENC_THREADS;
constexpr int kEncWaves = kEncThreads / kWave;
constexpr int kHeaderBits = 120;      // IMS header incl. the 8-bit step (spiht_re.c:448-464,63)
constexpr int kMaxEntryBits = 9;      // set bit + 4 x (significance + sign)
constexpr int kWindowWords = (kEncThreads * kMaxEntryBits) / 32 + 4;
// The LIS sweep scans five per-entry counters as fields of ONE 64-bit word; a field must hold the sum over a whole
// sweep of kEncThreads entries: bits <= 9 per entry, new LSP / LIP entries and appended LIS entries <= 4 each, survivors
// <= 1.  (Round 1 used 12-bit fields: 512 entries x 9 bits = 4608 carried into the next field - every child of every
// entry of a sweep significant on one plane - and 1024 entries overflowed on ordinary data.)
constexpr int bits_for(unsigned long long v) { int n = 0; while (v) { n++; v >>= 1; } return n; }
constexpr int kFldBits = bits_for((unsigned long long) kEncThreads * kMaxEntryBits);
constexpr int kFldCnt = bits_for((unsigned long long) kEncThreads * 4);
constexpr int kFldSurv = bits_for((unsigned long long) kEncThreads);
constexpr int kShLsp = kFldBits, kShLip = kShLsp + kFldCnt, kShApp = kShLip + kFldCnt, kShSurv = kShApp + kFldCnt;
static_assert(kShSurv + kFldSurv <= 64, "the packed scan word of the LIS sweep does not fit 64 bits: fewer entries per sweep");
static_assert(kEncThreads % kWave == 0 && kEncThreads <= 1024, "entries per sweep = threads of the workgroup");
constexpr unsigned long long fld_mask(int bits) { return (1ull << bits) - 1; }

__device__ inline int first_child(const Grid &g, int x, int y)
{
    int cx, cy;
    if (x < g.lx && y < g.ly) {                                     // spiht_re.c:133-147
        cx = (x & 1) ? x + g.lx - 1 : x;
        cy = (y & 1) ? y + g.ly - 1 : y;
        if (cx == x && cy == y) return -1;
    } else {                                                        // spiht_re.c:148-154
        cx = 2 * x; cy = 2 * y;
        if (cx >= g.nx || cy >= g.ny) return -1;
    }
    return cx + cy * g.nx;
}
// (row, column) of grid position p: p < 2^24, so the float quotient is off by at most one - far fewer instructions than
// the integer division by a run-time nx
__device__ inline int first_child_of(const Grid &g, int p)
{
    int y = (int) ((float) p * (1.0f / (float) g.nx)), x = p - y * g.nx;
    if (x < 0) { y -= 1; x += g.nx; }
    if (x >= g.nx) { y += 1; x -= g.nx; }
    return first_child(g, x, y);
}

// The decoder keeps grid positions in its lists as packed coordinates, x | y << 12 (the extents are 12-bit header fields):
// the first child of a position and the four children's positions are a few integer operations, where the linear index
// p = x + y * nx costs a division by the run-time nx for every significant set.
constexpr int kPkShift = 12;
constexpr uint32_t kPkMask = (1u << kPkShift) - 1u;
__device__ inline uint32_t pk_of(int x, int y) { return (uint32_t) x | ((uint32_t) y << kPkShift); }
__device__ inline uint32_t pk_index(uint32_t pk, int nx) { return (pk >> kPkShift) * (uint32_t) nx + (pk & kPkMask); }
__device__ inline uint32_t first_child_pk(const Grid &g, uint32_t pk)      // ~0u: no children (first_child)
{
    const int x = (int) (pk & kPkMask), y = (int) (pk >> kPkShift);
    int cx, cy;
    if (x < g.lx && y < g.ly) {
        cx = (x & 1) ? x + g.lx - 1 : x;
        cy = (y & 1) ? y + g.ly - 1 : y;
        if (cx == x && cy == y) return ~0u;
    } else {
        cx = 2 * x; cy = 2 * y;
        if (cx >= g.nx || cy >= g.ny) return ~0u;
    }
    return pk_of(cx, cy);
}

__device__ inline unsigned long long shfl_up_u64(unsigned long long v, int d)
{
    unsigned int lo = __shfl_up((unsigned int) v, d), hi = __shfl_up((unsigned int) (v >> 32), d);
    return ((unsigned long long) hi << 32) | lo;
}

// exclusive scan of a packed counter word across the workgroup; returns the block total in `total`.  `wave_tot` holds two
// sets of wave totals used alternately (`turn`): the readers of one scan never meet the writers of the next, so a scan costs
// ONE barrier - and that barrier also stands between the window flush of the step before and this step's puts.
template <int NWAVES>
__device__ inline unsigned long long block_scan(unsigned long long v, unsigned long long *wave_tot_both, unsigned int &turn,
                                                unsigned long long &total)
{
    unsigned long long *wave_tot = wave_tot_both + (turn & 1u) * NWAVES;
    turn++;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // the wave's inclusive sums from three 32-bit DPP scans (the two halves of the low word cannot overflow on their own; no
    // trip through the LDS crossbar: the twelve shuffles of a 64-bit scan were ~0.3 us of every sweep step)
    const uint32_t lo = (uint32_t) v;
    const uint32_t s0 = wave_prefix_sum(lo & 0xFFFFu), s1 = wave_prefix_sum(lo >> 16), s2 = wave_prefix_sum((uint32_t) (v >> 32));
    unsigned long long x = (unsigned long long) s0 + ((unsigned long long) s1 << 16) + ((unsigned long long) s2 << 32);
    if (NWAVES == 1) {
        unsigned int lo = __shfl((unsigned int) x, 63), hi = __shfl((unsigned int) (x >> 32), 63);
        total = ((unsigned long long) hi << 32) | lo;
        return x - v;
    }
    if (lane == 63) wave_tot[w] = x;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NWAVES; i++) {
        unsigned long long t = wave_tot[i];
        if (i < w) base += t;
        tot += t;
    }
    total = tot;
    return base + x - v;
}

// ---- bit window -------------------------------------------------------------------------------
// `off` = 0-based index of the entry's first SPIHT bit, `limit` = number of SPIHT bits that may be
// written in total (budget + 1, spiht_re.c:225 writes the bit that crosses the budget).
struct BitWindow {
    unsigned int *w;            // LDS words
    unsigned long long base;    // absolute stream bit index of w[0] bit 31
};

__device__ inline void window_put(const BitWindow &bw, unsigned int val, int nb, unsigned long long off,
                                  unsigned long long limit)
{
    if (nb == 0 || off >= limit) return;
    if (off + nb > limit) {
        int keep = (int) (limit - off);
        val >>= (nb - keep);
        nb = keep;
    }
    unsigned long long rel = (kHeaderBits + off) - bw.base;
    int j = (int) (rel >> 5), sh = (int) (rel & 31);
    unsigned long long x = (unsigned long long) val << (64 - nb - sh);
    unsigned int hi = (unsigned int) (x >> 32), lo = (unsigned int) x;
    if (hi) atomicOr(&bw.w[j], hi);
    if (lo) atomicOr(&bw.w[j + 1], lo);
}

// the window of a chunk whose first bit has SPIHT index `first` (the words are zero: the flush before left them so)
__device__ inline void window_open(BitWindow &bw, unsigned long long first)
{
    bw.base = ((kHeaderBits + first) >> 5) << 5;
}

// OR the window into the big-endian byte stream and leave it zeroed.  A barrier stands before the reads (every put of the
// step is in); the one behind them is the caller's next block_scan - a sweep without a scan (the refinement pass) asks for it
// here (`fence_after`).
__device__ inline void window_flush(const BitWindow &bw, unsigned int *stream, size_t stream_words, bool fence_after = false)
{
    __syncthreads();
    size_t w0 = (size_t) (bw.base >> 5);
    for (int i = threadIdx.x; i < kWindowWords; i += blockDim.x) {
        unsigned int v = bw.w[i];
        if (v) {
            bw.w[i] = 0;
            if (w0 + i < stream_words) atomicOr(&stream[w0 + i], __builtin_bswap32(v));
        }
    }
    if (fence_after) __syncthreads();
}

// ================================================================================================
// encoder: one workgroup per frame
// ================================================================================================
__global__ __launch_bounds__(kEncThreads) void k_spiht_encode(
    const int32_t *__restrict__ Cb, const int32_t *__restrict__ Db, const int32_t *__restrict__ Gb, uint32_t *lipb,
    uint32_t *lspb, uint32_t *lis0b, uint32_t *lis1b, uint32_t *sigordb, uint32_t *lspidxb, uint32_t *streamb,
    size_t stream_words, Grid g, size_t np, FrameState *fsb, const unsigned long long *bits0, const int *active)
{
    __shared__ unsigned long long wave_tot[2 * kEncWaves];
    __shared__ unsigned int window[kWindowWords];
    unsigned int turn = 0;
    for (int i = threadIdx.x; i < kWindowWords; i += blockDim.x) window[i] = 0;   // (the first scan's barrier comes before the first put)

    const int frame = blockIdx.x;
    if (active && !active[frame]) return;
    const int tid = threadIdx.x;
    const int32_t *C = Cb + (size_t) frame * np;
    const int32_t *D = Db + (size_t) frame * np;
    const int32_t *G = Gb + (size_t) frame * np;
    uint32_t *lip = lipb + (size_t) frame * np;
    uint32_t *lsp = lspb + (size_t) frame * np;
    uint32_t *cur = lis0b + (size_t) frame * np;
    uint32_t *nxt = lis1b + (size_t) frame * np;
    uint32_t *sigord = sigordb + (size_t) frame * np;
    uint32_t *lspidx = lspidxb + (size_t) frame * np;
    uint32_t *stream = streamb + (size_t) frame * stream_words;
    FrameState &fs = fsb[frame];

    const unsigned long long budget = fs.budget;
    const unsigned long long limit = budget + 1;
    const int top = fs.top_step;
    BitWindow bw{window, 0};

    // ---- IMS header, spiht_re.c:448-464 + step byte :63 (15 bytes, written by one lane)
    if (tid == 0) {
        unsigned long long b0 = bits0[frame];
        unsigned char h[16];
        h[0] = 'I'; h[1] = 'M'; h[2] = 'S';
        // 6b stages | 12b size_x | 12b size_y | 10b extra_x | 10b extra_y | 1b color | 29b bits0 = 80 bits
        // first 64 of the 80 bits: everything up to the top 13 bits of bits0; the low 16 bits of bits0 follow
        unsigned long long hi = 0;
        hi = (hi << 6) | (unsigned) g.stages;
        hi = (hi << 12) | (unsigned) g.size_x;
        hi = (hi << 12) | (unsigned) g.size_y;
        hi = (hi << 10) | (unsigned) g.extra_x;
        hi = (hi << 10) | (unsigned) g.extra_y;
        hi = (hi << 1) | 0u;
        hi = (hi << 13) | ((b0 >> 16) & 0x1FFFull);
        unsigned int lo = (unsigned int) (b0 & 0xFFFF);
        for (int i = 0; i < 8; i++) h[3 + i] = (unsigned char) (hi >> (56 - 8 * i));
        h[11] = (unsigned char) (lo >> 8);
        h[12] = (unsigned char) lo;
        h[13] = (unsigned char) (int) fs.dc;
        h[14] = (unsigned char) top;
        h[15] = 0;
        for (int i = 0; i < 4; i++) {
            unsigned int wv = h[4 * i] | (h[4 * i + 1] << 8) | (h[4 * i + 2] << 16) | ((unsigned) h[4 * i + 3] << 24);
            atomicOr(&stream[i], wv);
        }
        for (int i = 0; i < 32; i++) { fs.refine_base[i] = 0; fs.refine_count[i] = 0; fs.step_reached[i] = 0; }
    }

    // ---- seed lists, spiht_re.c:65-77: LIP = LL raster order; LIS = type-A entries of LL pixels with an odd coordinate
    unsigned int nlip = 0, nlis = 0, nlsp = 0;
    {
        const int nll = g.lx * g.ly;
        for (int base = 0; base < nll; base += kEncThreads) {
            int i = base + tid;
            bool valid = i < nll;
            int y = valid ? i / g.lx : 0, x = valid ? i - y * g.lx : 0;
            unsigned int p = (unsigned int) (x + y * g.nx);
            bool isset = valid && ((x & 1) || (y & 1));
            unsigned long long tot;
            unsigned long long ex = block_scan<kEncWaves>(isset ? 1ull : 0ull, wave_tot, turn, tot);
            if (valid) lip[i] = p;
            if (isset) cur[nlis + (unsigned int) ex] = p << 1;           // bit0 = 0: type A
            nlis += (unsigned int) tot;
        }
        nlip = (unsigned int) nll;
    }
    __syncthreads();

    unsigned long long nbits = 0;          // SPIHT bits produced so far (uniform across the workgroup)
    bool stop = false;

    for (int s = top; s >= 0 && !stop; --s) {
        const unsigned int n_old = nlsp;

        // ---------------- LIP pass, spiht_re.c:220-234
        {
            unsigned int wr = 0;
            for (unsigned int base = 0; base < nlip && !stop; base += kEncThreads) {
                unsigned int i = base + tid;
                bool valid = i < nlip;
                unsigned int p = valid ? lip[i] : 0;
                int c = valid ? C[p] : 0;
                unsigned int a = (unsigned int) (c < 0 ? -c : c);
                bool sig = valid && (a >> s) != 0;
                int nb = valid ? (sig ? 2 : 1) : 0;
                unsigned int val = sig ? (2u | (c > 0 ? 0u : 1u)) : 0u;          // sign: 0 = positive, :229
                static_assert(2 * kEncThreads < 65536, "16-bit fields of the LIP sweep");
                unsigned long long pack = (unsigned long long) nb | ((unsigned long long) (sig ? 1 : 0) << 16) |
                                          ((unsigned long long) ((valid && !sig) ? 1 : 0) << 32);
                unsigned long long tot;
                unsigned long long ex = block_scan<kEncWaves>(pack, wave_tot, turn, tot);
                unsigned long long off = nbits + (ex & 0xFFFF);
                window_open(bw, nbits);
                window_put(bw, val, nb, off, limit);
                if (sig) {
                    unsigned int slot = nlsp + (unsigned int) ((ex >> 16) & 0xFFFF);
                    lsp[slot] = p;
                    sigord[p] = (uint32_t) (off + 1);
                    lspidx[p] = slot;
                } else if (valid) {
                    lip[wr + (unsigned int) (ex >> 32)] = p;
                }
                window_flush(bw, stream, stream_words);
                nbits += tot & 0xFFFF;
                nlsp += (unsigned int) ((tot >> 16) & 0xFFFF);
                wr += (unsigned int) (tot >> 32);
                if (nbits > budget) stop = true;
            }
            nlip = wr;
        }
        if (stop) break;

        // ---------------- LIS pass, spiht_re.c:237-305 (entries appended during the sweep are swept too)
        {
            unsigned int ncur = nlis, nnext = 0;
            unsigned int base = 0;
            while (base < ncur && !stop) {
                unsigned int chunk_end = min(base + (unsigned int) kEncThreads, ncur);
                unsigned int i = base + tid;
                bool valid = i < chunk_end;
                unsigned int e = valid ? cur[i] : 0;
                unsigned int p = e >> 1;
                bool isB = e & 1;
                int nb = 0, n_lsp = 0, n_lip = 0, n_app = 0, surv = 0;
                unsigned int val = 0;
                int ch = -1;
                int cc[4] = {0, 0, 0, 0};
                if (valid) {
                    if (!isB) {
                        bool sig = (D[p] >> s) != 0;                             // is_significant_set_A, :160-182
                        if (sig) {
                            ch = first_child_of(g, (int) p);
                            val = 1; nb = 1;
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                int q = ch + (k & 1) + (k >> 1) * g.nx;          // :255-257 (dy outer, dx inner)
                                int c = C[q];
                                cc[k] = c;
                                unsigned int a = (unsigned int) (c < 0 ? -c : c);
                                if ((a >> s) != 0) {
                                    val = (val << 2) | 2u | (c > 0 ? 0u : 1u);
                                    nb += 2; n_lsp++;
                                } else {
                                    val <<= 1;
                                    nb += 1; n_lip++;
                                }
                            }
                            if (first_child_of(g, ch) >= 0) n_app = 1;          // :273-278
                        } else {
                            val = 0; nb = 1; surv = 1;
                        }
                    } else {
                        bool sig = (G[p] >> s) != 0;                             // is_significant_set_B, :184-206
                        val = sig ? 1u : 0u; nb = 1;
                        if (sig) { ch = first_child_of(g, (int) p); n_app = 4; } // :292-300
                        else surv = 1;
                    }
                }
                unsigned long long pack = (unsigned long long) nb | ((unsigned long long) n_lsp << kShLsp) |
                                          ((unsigned long long) n_lip << kShLip) | ((unsigned long long) n_app << kShApp) |
                                          ((unsigned long long) surv << kShSurv);
                unsigned long long tot;
                unsigned long long ex = block_scan<kEncWaves>(pack, wave_tot, turn, tot);
                unsigned long long off = nbits + (ex & fld_mask(kFldBits));
                window_open(bw, nbits);
                window_put(bw, val, nb, off, limit);
                if (valid) {
                    unsigned int r_lsp = nlsp + (unsigned int) ((ex >> kShLsp) & fld_mask(kFldCnt));
                    unsigned int r_lip = nlip + (unsigned int) ((ex >> kShLip) & fld_mask(kFldCnt));
                    unsigned int r_app = ncur + (unsigned int) ((ex >> kShApp) & fld_mask(kFldCnt));
                    unsigned int r_surv = nnext + (unsigned int) ((ex >> kShSurv) & fld_mask(kFldSurv));
                    if (!isB && ch >= 0) {
                        unsigned long long pos = off + 1;                        // next bit after the set bit
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            unsigned int q = (unsigned int) (ch + (k & 1) + (k >> 1) * g.nx);
                            int c = cc[k];
                            unsigned int a = (unsigned int) (c < 0 ? -c : c);
                            if ((a >> s) != 0) {
                                lsp[r_lsp] = q;
                                sigord[q] = (uint32_t) (pos + 1);
                                lspidx[q] = r_lsp;
                                r_lsp++;
                                pos += 2;
                            } else {
                                lip[r_lip++] = q;
                                pos += 1;
                            }
                        }
                        if (n_app) cur[r_app] = (p << 1) | 1u;                   // -(p+1): type B
                    } else if (isB && ch >= 0) {
                        cur[r_app + 0] = (unsigned int) ch << 1;
                        cur[r_app + 1] = (unsigned int) (ch + 1) << 1;
                        cur[r_app + 2] = (unsigned int) (ch + g.nx) << 1;
                        cur[r_app + 3] = (unsigned int) (ch + g.nx + 1) << 1;
                    }
                    if (surv) nxt[r_surv] = e;
                }
                window_flush(bw, stream, stream_words);
                nbits += tot & fld_mask(kFldBits);
                nlsp += (unsigned int) ((tot >> kShLsp) & fld_mask(kFldCnt));
                nlip += (unsigned int) ((tot >> kShLip) & fld_mask(kFldCnt));
                ncur += (unsigned int) ((tot >> kShApp) & fld_mask(kFldCnt));
                nnext += (unsigned int) ((tot >> kShSurv) & fld_mask(kFldSurv));
                base = chunk_end;
                if (nbits > budget) stop = true;
            }
            uint32_t *t = cur; cur = nxt; nxt = t;
            nlis = nnext;
        }
        if (stop) break;

        // ---------------- refinement pass, spiht_re.c:308-314: one bit for every LSP entry older than this step
        if (tid == 0) {
            fs.refine_base[s] = (unsigned int) nbits;
            fs.refine_count[s] = n_old;
            fs.step_reached[s] = 1;
        }
        __syncthreads();                                                 // (the LIS pass's last flush is still zeroing the window; this sweep has no scan)
        for (unsigned int base = 0; base < n_old && !stop; base += kEncThreads) {
            unsigned int i = base + tid;
            bool valid = i < n_old;
            unsigned int bit = 0;
            if (valid) {
                int c = C[lsp[i]];
                unsigned int a = (unsigned int) (c < 0 ? -c : c);
                bit = (a >> s) & 1u;
            }
            window_open(bw, nbits);
            window_put(bw, bit, valid ? 1 : 0, nbits + tid, limit);
            window_flush(bw, stream, stream_words, true);
            nbits += min((unsigned int) kEncThreads, n_old - base);
            if (nbits > budget) stop = true;
        }
    }

    if (tid == 0) {
        unsigned long long emitted = nbits < limit ? nbits : limit;
        fs.emitted = emitted;
        fs.stream_bytes = (unsigned int) ((kHeaderBits + emitted + 7) >> 3);     // bitio_flush, bitio.h:78-88
    }
}

// ================================================================================================
// reconstruct the decoder's coefficient grid after a prefix of the stream (see file header)
// ================================================================================================
__global__ __launch_bounds__(256) void k_reconstruct(const int32_t *__restrict__ Cb, const uint32_t *__restrict__ sigordb,
                                                      const uint32_t *__restrict__ lspidxb, float *__restrict__ Ab,
                                                      size_t np, const FrameState *fsb,
                                                      const unsigned long long *trunc_bits, const int *active, Grid g, int coarse_only,
                                                      const int *frame_of, size_t a_stride)
{
    // frame_of (cut slots, residual.hpp): blockIdx.y is a slot - state, cut, mask and output are the slot's, the bookkeeping
    // is its frame's
    const int slot = blockIdx.y;
    if (active && !active[slot]) return;
    const int frame = frame_of ? frame_of[slot] : slot;
    const FrameState &fs = fsb[slot];
    // spiht_decode: num_bits = min(num_bits, bits0) - 128   (spiht_re.c:495-500)
    unsigned long long nb = trunc_bits[slot], bits0 = fs.budget + 128;
    if (nb > bits0) nb = bits0;
    const unsigned long long B = nb - 128;
    __shared__ unsigned int rbase[32], rreach[32];
    if (threadIdx.x < 32) {
        rbase[threadIdx.x] = fs.refine_base[threadIdx.x];
        rreach[threadIdx.x] = fs.step_reached[threadIdx.x];
    }
    __syncthreads();
    const int32_t *C = Cb + (size_t) frame * np;
    const uint32_t *so = sigordb + (size_t) frame * np;
    const uint32_t *li = lspidxb + (size_t) frame * np;
    float *A = Ab + (size_t) slot * a_stride;
    // (coarse_only: the LL quadrant of the finest level - what the coarser synthesis levels need; the finest level takes
    //  its three detail bands straight from the bookkeeping, k_cols_inv_stream in residual_dwt.hip)
    const int cx = coarse_only ? g.nx >> 1 : g.nx, cy = coarse_only ? g.ny >> 1 : g.ny;
    const size_t cnt = (size_t) cx * cy;
    for (size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x; t < cnt; t += (size_t) gridDim.x * blockDim.x) {
        const size_t i = coarse_only ? (t / cx) * g.nx + (t % cx) : t;
        A[i] = prefix_value(C, so, li, i, B, rbase, rreach);
    }
}

// ================================================================================================
// decoder: one wavefront per frame; the variable-length parse of a 64-entry chunk is a uniform scalar
// loop over a register-resident bit window, the list/coefficient updates are lane-parallel
// ================================================================================================
__device__ inline unsigned int load_be_word(const uint8_t *s, unsigned long long size, unsigned long long wi)
{
    unsigned long long b = wi * 4;
    if (b + 4 <= size) {
        unsigned int v = *reinterpret_cast<const unsigned int *>(s + b);
        return __builtin_bswap32(v);
    }
    unsigned int v = 0;                                               // bitio.h:60-63: past the end reads as 0
    for (int k = 0; k < 4; k++) v = (v << 8) | (b + k < size ? s[b + k] : 0u);
    return v;
}

// uniform bit fetch from the wave-resident window (word l lives in lane l)
__device__ inline unsigned int win_bit(unsigned int wreg, int o)
{
    unsigned int w = __builtin_amdgcn_readlane(wreg, __builtin_amdgcn_readfirstlane(o >> 5));
    return (w >> (31 - (o & 31))) & 1u;
}

// per-lane fetch of up to 32 bits starting at window offset o
__device__ inline unsigned int win_bits_lane(unsigned int wreg, int o, int n)
{
    unsigned int w0 = __shfl(wreg, o >> 5), w1 = __shfl(wreg, (o >> 5) + 1);
    unsigned long long x = ((unsigned long long) w0 << 32) | w1;
    if (n == 0) return 0;
    return (unsigned int) ((x << (o & 31)) >> (64 - n));
}

__device__ inline unsigned int read_bits_uniform(const uint8_t *s, unsigned long long size, unsigned long long pos, int n)
{
    unsigned long long wi = pos >> 5;
    unsigned long long x = ((unsigned long long) load_be_word(s, size, wi) << 32) | load_be_word(s, size, wi + 1);
    return (unsigned int) ((x << (pos & 31)) >> (64 - n));
}

// EBCC_HIP_SPIHT_PROF=1: cycles and list entries of the three passes, summed over the frames of a launch
__device__ unsigned long long g_spiht_prof[12];

__global__ __launch_bounds__(kWave) void k_spiht_decode(const uint8_t *__restrict__ streams, size_t stream_stride,
                                                        const unsigned long long *sizes,
                                                        const unsigned long long *num_bits_in, int32_t *Cb,
                                                        uint32_t *lipb, uint32_t *lspb, uint32_t *lis0b,
                                                        uint32_t *lis1b, Grid g, size_t np, FrameState *fsb,
                                                        const int *active, int prof)
{
    const int frame = blockIdx.x;
    if (active && !active[frame]) return;
    const unsigned long long prof_c0 = prof ? __builtin_readcyclecounter() : 0ull, prof_r0 = prof ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // one wave, one long dependent chain: it runs beside the tier-1 decoder's thousands of waves (other stream), which
    // keep every SIMD's issue slots busy - ask the arbiter to serve this wave first (10.6 ms alone, 25.7 ms without this)
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x;
    const uint8_t *S = streams + (size_t) frame * stream_stride;
    const unsigned long long size = sizes[frame];
    int32_t *C = Cb + (size_t) frame * np;
    uint32_t *lip = lipb + (size_t) frame * np;
    uint32_t *lsp = lspb + (size_t) frame * np;
    uint32_t *cur = lis0b + (size_t) frame * np;
    uint32_t *nxt = lis1b + (size_t) frame * np;
    FrameState &fs = fsb[frame];

    // header, spiht_re.c:480-503 (geometry was validated on the host against `g`)
    unsigned long long bits0 = read_bits_uniform(S, size, 24 + 6 + 12 + 12 + 10 + 10 + 1, 29);
    unsigned long long nbits_arg = num_bits_in[frame];
    if (nbits_arg > bits0) nbits_arg = bits0;
    const unsigned long long B = nbits_arg - 128;
    const int dc = (int) read_bits_uniform(S, size, 104, 8);
    const int top = (int) read_bits_uniform(S, size, 112, 8);
    if (lane == 0) { fs.dec_dc = dc; fs.dec_top_step = top; fs.dec_budget = B; }

    // seed lists, spiht_re.c:106-115
    unsigned int nlip = 0, nlis = 0, nlsp = 0;
    {
        const int nll = g.lx * g.ly;
        for (int base = 0; base < nll; base += kWave) {
            int i = base + lane;
            bool valid = i < nll;
            int y = valid ? i / g.lx : 0, x = valid ? i - y * g.lx : 0;
            unsigned int p = pk_of(x, y);
            bool isset = valid && ((x & 1) || (y & 1));
            unsigned long long m = __ballot(isset);
            if (valid) lip[i] = p;
            if (isset) cur[nlis + __popcll(m & ((1ull << lane) - 1))] = p << 1;
            nlis += (unsigned int) __popcll(m);
        }
        nlip = (unsigned int) nll;
    }

    unsigned long long cnt = 0;            // SPIHT bits consumed (bit_cnt of the reference)
    bool stop = false;
    const unsigned long long lanemask_lt = (1ull << lane) - 1;
    // The stream window: word wbase + l of the stream in lane l (2048 bits), kept across chunks and re-read only when the
    // next chunk might run off its end - the kernel is one dependent chain per frame, and every load it has to wait for
    // is ~1 us of that chain.
    unsigned long long wbase = ~0ull;
    unsigned int wreg = 0;
    auto window = [&](unsigned long long pos, int need_words) -> int {   // window offset of stream bit `pos`
        const unsigned long long w0 = pos >> 5;
        if (wbase == ~0ull || w0 < wbase || w0 + (unsigned long long) need_words > wbase + kWave) {
            wbase = w0;
            wreg = load_be_word(S, size, wbase + lane);
        }
        return (int) (pos - (wbase << 5));
    };
    // the 64 stream bits from `pos` on, first bit in bit 63 (uniform: scalar registers)
    auto stream64 = [&](unsigned long long pos) -> unsigned long long {
        const int o = window(pos, 4), wi = __builtin_amdgcn_readfirstlane(o >> 5), sh = __builtin_amdgcn_readfirstlane(o & 31);
        // (the builtin returns int: through unsigned, or the sign extends into the upper word)
        const unsigned int w0 = (unsigned int) __builtin_amdgcn_readlane(wreg, wi), w1 = (unsigned int) __builtin_amdgcn_readlane(wreg, wi + 1),
                           w2 = (unsigned int) __builtin_amdgcn_readlane(wreg, wi + 2);
        const unsigned long long hi = ((unsigned long long) w0 << 32) | w1, lo = (unsigned long long) w2 << 32;
        return sh ? (hi << sh) | (lo >> (64 - sh)) : hi;
    };
    // A list is read through three registers per lane holding 192 consecutive entries (entry wb + l in w0 of lane l, entry
    // wb + 64 + l in w1, wb + 128 + l in w2); a chunk reads from the first two, the third is requested when the window moves on
    // and used after the NEXT move, about two chunks later (with two registers the load asked for at the end of one chunk was
    // wanted at the start of the next).  `limit`: entries below it were visible when the registers were loaded.
    struct ListWindow {
        const uint32_t *list;
        unsigned int wb, limit;
        uint32_t w0, w1, w2;
        __device__ void reset(const uint32_t *l, unsigned int base, unsigned int n, int lane)
        {
            list = l; wb = base; limit = n;
            w0 = wb + lane < n ? list[wb + lane] : 0u;
            w1 = wb + kWave + lane < n ? list[wb + kWave + lane] : 0u;
            w2 = wb + 2 * kWave + lane < n ? list[wb + 2 * kWave + lane] : 0u;
        }
        // entry base + i for the lane that asks for i (0 <= i < 64, base - wb <= 64); every lane takes part
        __device__ uint32_t get(unsigned int base, int i) const
        {
            const int idx = (int) (base - wb) + i;
            const uint32_t a = __shfl(w0, idx & 63), b = __shfl(w1, idx & 63);
            return idx < kWave ? a : b;
        }
        // bit 0 of the entries base .. base + 63 (the LIS entries' types; zero past the list's end), in scalar registers
        __device__ unsigned long long types(unsigned int base) const
        {
            const unsigned long long t0 = __ballot(w0 & 1u), t1 = __ballot(w1 & 1u);
            const int off = __builtin_amdgcn_readfirstlane((int) (base - wb));
            return off == 0 ? t0 : (off >= kWave ? t1 : (t0 >> off) | (t1 << (kWave - off)));
        }
        __device__ void advance(unsigned int base, int lane)
        {
            if (base - wb >= (unsigned int) kWave) {
                w0 = w1; w1 = w2; wb += kWave;
                w2 = wb + 2 * kWave + lane < limit ? list[wb + 2 * kWave + lane] : 0u;
            }
        }
    };

    for (int s = top; s >= 0 && !stop; --s) {
        const unsigned int n_old = nlsp;
        const int one = 1 << s;

        // ---------------- LIP pass, spiht_re.c:331-343.  An entry is one bit, or two when the first is set (significant +
        // sign): where the entries start in the next 64 stream bits has a closed form - a set bit at an entry start
        // "escapes" the bit after it, the odd/even run-of-ones argument of simdjson's backslash scanner - so a chunk is
        // all the entries that start in those 64 bits, lane x looking at bit x, and nothing in the pass is a serial loop.
        unsigned long long t0 = prof ? __builtin_readcyclecounter() : 0ull;
        if (prof && lane == 0) atomicAdd(&g_spiht_prof[1], (unsigned long long) nlip);
        {
            __threadfence_block();                                     // (entries the previous LIS pass appended)
            unsigned int wr = 0, base = 0;
            ListWindow lw;
            lw.reset(lip, 0, nlip, lane);
            while (base < nlip && !stop) {
                const unsigned long long X = __builtin_bitreverse64(stream64(kHeaderBits + cnt));   // bit x = stream bit cnt + x
                const unsigned long long even = 0x5555555555555555ull, follows = X << 1;
                const unsigned long long odd_starts = X & ~even & ~follows;
                const unsigned long long signs = (even ^ ((odd_starts + X) << 1)) & follows;        // bits that are sign bits
                const unsigned long long starts = ~signs;
                // a significant entry starting at bit 63 has its sign in the next chunk: it waits for that chunk
                int limit = ((starts & X) >> 63) ? 63 : 64;
                const unsigned int n_rem = nlip - base;
                const int idx = __popcll(starts & lanemask_lt);                                    // entry of the chunk that starts at bit `lane`
                const bool isstart = ((starts >> lane) & 1ull) && lane < limit;
                const unsigned long long mover = __ballot(isstart && (unsigned int) idx >= n_rem);  // starts beyond the end of the list
                if (mover) limit = __builtin_ctzll(mover);
                const bool valid = isstart && lane < limit;
                const unsigned int p = lw.get(base, valid ? idx : 0);
                const bool sig = valid && ((X >> lane) & 1ull);
                const bool neg = (X >> ((lane + 1) & 63)) & 1ull;
                const unsigned long long k1 = cnt + (unsigned long long) lane + 1;                 // ordinal of the significance bit
                const bool act = valid && k1 <= B;                                                 // :334
                const bool overrun = valid && (k1 > B || (sig && k1 + 1 > B));                     // :334,:339
                const unsigned long long msig = __ballot(act && sig), mkeep = __ballot(valid && !(act && sig));
                if (act && sig) {
                    lsp[nlsp + __popcll(msig & lanemask_lt)] = p;
                    C[pk_index(p, g.nx)] = neg ? -one : one;                                       // :338
                } else if (valid) {
                    lip[wr + __popcll(mkeep & lanemask_lt)] = p;                                   // (at or below the entry's own slot)
                }
                nlsp += (unsigned int) __popcll(msig);
                wr += (unsigned int) __popcll(mkeep);
                cnt += (unsigned long long) limit;
                base += (unsigned int) __popcll(__ballot(valid));
                lw.advance(base, lane);
                if (__ballot(overrun)) stop = true;
            }
            nlip = wr;
        }
        if (stop) break;

        if (prof && lane == 0) { const unsigned long long t1 = __builtin_readcyclecounter(); atomicAdd(&g_spiht_prof[0], t1 - t0); t0 = t1; }
        // ---------------- LIS pass, spiht_re.c:346-410.  An entry takes 1 bit, or 5 to 9 when a type-A entry's set is
        // significant, so where an entry starts depends on every entry before it: a scalar loop walks the next 64 stream
        // bits (registers only, ~10 scalar instructions per entry) and marks the starts; a chunk is the entries that start
        // in the first 55 of those bits, lane x again looking at bit x.
        {
            unsigned int ncur = nlis, nnext = 0, base = 0;
            // entries appended to the list in this pass are read back further down the same pass: a fence before the
            // list registers are loaded with entries written since the last one
            __threadfence_block();
            ListWindow lw;
            lw.reset(cur, 0, ncur, lane);
            while (base < ncur && !stop) {
                const int m = (int) min((unsigned int) kWave, ncur - base);
                if (base + (unsigned int) m > lw.limit) {
                    __threadfence_block(); lw.reset(cur, base, ncur, lane);
                    if (prof && lane == 0) atomicAdd(&g_spiht_prof[11], 1ull);
                }
                if (prof && lane == 0) atomicAdd(&g_spiht_prof[10], 1ull);
                const unsigned long long typemask = lw.types(base);                      // types of the entries base .. base + 63
                const unsigned long long X = stream64(kHeaderBits + cnt);                // first bit in bit 63
                // what a type-A entry starting at bit `lane` would take (every lane works its own position out: 1 bit, or the
                // set bit and four children of 1 or 2 bits); the scalar walk then only adds lengths up.  The same steps give the
                // children's significance and sign flags, used when the position turns out to be such an entry.
                const uint32_t Wl = (uint32_t) ((X << lane) >> 32);                // the 32 stream bits from this lane's position on, first in bit 31
                int len_a = 1;
                unsigned int wsig = 0, wneg = 0;
                if (lane < 55 && (Wl >> 31)) {
                    int q = 1;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const unsigned int sb = (Wl >> (31 - q)) & 1u;
                        wsig |= sb << k;
                        wneg |= (sb & (Wl >> (30 - q))) << k;
                        q += 1 + (int) sb;
                    }
                    len_a = q;
                }
                // The walk steps from set bit to set bit: an entry whose first bit is 0 takes that one bit whatever its type, so
                // a run of z zero bits is z entries at once, and only an entry that starts on a set bit needs its type (the number
                // of starts up to it is its place in the list) and the length its lane worked out: ~6 steps of ~20 scalar
                // instructions per chunk where a step per entry took ~35 steps.  A set bit stands guard at position 55, where
                // lanes hold length 1, so the loop needs no test for "no set bit left"; the list's end is cut afterwards.
                constexpr unsigned long long kLow55 = (1ull << 55) - 1ull;
                const unsigned long long Xr = (__builtin_bitreverse64(X) & kLow55) | (1ull << 55);   // bit x = stream bit cnt + x
                unsigned long long starts = 0;
                int rel = 0;
                do {
                    const int z = __builtin_ctzll(Xr >> rel);
                    starts |= ((2ull << z) - 1ull) << rel;                          // z zero bits and the set bit after them
                    const int pos = rel + z;
                    const int la = __builtin_amdgcn_readlane(len_a, pos);
                    rel = pos + (((typemask >> (__popcll(starts) - 1)) & 1ull) ? 1 : la);
                } while (rel < 55);
                rel -= (int) ((starts >> 55) & 1ull);                               // the guard is not an entry: 55 bits taken
                starts &= kLow55;
                int n_ent = __popcll(starts);
                if (n_ent > m) {                                                    // the list ends inside the chunk
                    const unsigned long long past = __ballot(((starts >> lane) & 1ull) && __popcll(starts & lanemask_lt) == m);
                    rel = __builtin_ctzll(past);
                    starts &= (1ull << rel) - 1ull;
                    n_ent = m;
                }
                const bool valid = (starts >> lane) & 1ull;
                const int idx = __popcll(starts & lanemask_lt);
                const unsigned int e = lw.get(base, valid ? idx : 0);
                const unsigned int p = e >> 1;                                         // packed coordinates
                const int isB = (int) (e & 1u);
                const unsigned long long k = cnt + (unsigned long long) lane;          // bits consumed before this entry
                int n_lsp = 0, n_lip = 0, n_app = 0, surv = 0;
                uint32_t ch = ~0u;
                unsigned int csig = 0, cneg = 0, cact = 0;                             // per-child flags
                bool act, overrun;
                if (cnt + 64ull <= B) {
                    // every bit of the chunk lies inside the budget (all chunks of a stream but its last one or two): none of the
                    // budget tests of :358-:394 can fire, and what an entry on a set bit says is what its lane read ahead
                    act = valid; overrun = false;
                    if (valid) {
                        if (Wl >> 31) {
                            ch = first_child_pk(g, p);
                            if (!isB) {
                                csig = wsig; cneg = wneg; cact = 0xFu;
                                n_lsp = __popc(csig); n_lip = 4 - n_lsp;
                                // (a child of a set has children of its own iff its doubled coordinates are inside the grid: :273-278)
                                n_app = (2 * (int) (ch & kPkMask) < g.nx && 2 * (int) (ch >> kPkShift) < g.ny) ? 1 : 0;
                            } else {
                                n_app = 4;
                            }
                        } else {
                            surv = 1;
                        }
                    }
                } else {
                    const unsigned long long later = lane < 63 ? starts >> (lane + 1) : 0ull;
                    const int mylen = valid ? (later ? __builtin_ctzll(later) + 1 : rel - lane) : 0;
                    const unsigned int mybits = mylen ? Wl >> (32 - mylen) : 0u;       // MSB-first, mylen <= 9
                    const bool setbit = valid && mylen > 0 && ((mybits >> (mylen - 1)) & 1u);
                    act = valid && (k + 1 <= B);                                       // :358 / :394
                    overrun = valid && !act;
                    if (act) {
                        if (!isB) {
                            if (setbit) {
                                ch = first_child_pk(g, p);
                                int rem = mylen - 1;
                                unsigned long long kk = k + 1;                          // bits consumed so far
                                bool alive = true;
                                for (int c = 0; c < 4; c++) {
                                    unsigned int sb = (mybits >> (rem - 1)) & 1u;
                                    kk += 1;
                                    if (alive && kk > B) { alive = false; overrun = true; }   // :367
                                    if (alive) {
                                        cact |= 1u << c;
                                        if (sb) {
                                            csig |= 1u << c;
                                            if ((mybits >> (rem - 2)) & 1u) cneg |= 1u << c;
                                            n_lsp++;
                                        } else {
                                            n_lip++;
                                        }
                                    }
                                    rem -= 1;
                                    if (sb) {
                                        kk += 1;
                                        rem -= 1;
                                        if (alive && kk > B) { alive = false; overrun = true; } // :371 (value already assigned)
                                    }
                                }
                                if (alive && ch != ~0u && first_child_pk(g, ch) != ~0u) n_app = 1;
                            } else {
                                surv = 1;
                            }
                        } else {
                            if (setbit) { ch = first_child_pk(g, p); n_app = 4; }
                            else surv = 1;
                        }
                    }
                }
                // wave exclusive scans of the four counters: three of them (at most 4 per lane, 256 per wave) packed into one word
                // and summed by DPP, the survivors (0 / 1) by a ballot
                const uint32_t pack = (uint32_t) n_lsp | ((uint32_t) n_lip << 10) | ((uint32_t) n_app << 20);
                const uint32_t inc = wave_prefix_sum(pack), ex = inc - pack;
                const uint32_t tot = (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
                const unsigned long long msurv = __ballot(surv != 0);
                unsigned int r_lsp = nlsp + (ex & 0x3FFu);
                unsigned int r_lip = nlip + ((ex >> 10) & 0x3FFu);
                unsigned int r_app = ncur + ((ex >> 20) & 0x3FFu);
                unsigned int r_surv = nnext + (unsigned int) __popcll(msurv & lanemask_lt);
                if (act) {
                    if (!isB && ch != ~0u) {
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            if ((cact >> c) & 1u) {                                    // (the flags of the children a budget cut off are clear)
                                const unsigned int q = ch + (unsigned int) (c & 1) + ((unsigned int) (c >> 1) << kPkShift);
                                const unsigned int sg = (csig >> c) & 1u;
                                uint32_t *dst = sg ? lsp + r_lsp : lip + r_lip;        // (one store, not a branch per child)
                                *dst = q;
                                r_lsp += sg; r_lip += 1u - sg;
                                if (sg) C[pk_index(q, g.nx)] = ((cneg >> c) & 1u) ? -one : one;   // :370
                            }
                        }
                        if (n_app) cur[r_app] = (p << 1) | 1u;
                    } else if (isB && ch != ~0u) {
                        cur[r_app + 0] = ch << 1;
                        cur[r_app + 1] = (ch + 1u) << 1;
                        cur[r_app + 2] = (ch + (1u << kPkShift)) << 1;
                        cur[r_app + 3] = (ch + (1u << kPkShift) + 1u) << 1;
                    }
                    if (surv) nxt[r_surv] = e;
                }
                nlsp += tot & 0x3FFu;
                nlip += (tot >> 10) & 0x3FFu;
                ncur += (tot >> 20) & 0x3FFu;
                nnext += (unsigned int) __popcll(msurv);
                cnt += (unsigned long long) rel;
                base += (unsigned int) n_ent;
                lw.advance(base, lane);
                if (__ballot(overrun)) stop = true;
            }
            if (prof && lane == 0) atomicAdd(&g_spiht_prof[3], (unsigned long long) ncur);
            uint32_t *t = cur; cur = nxt; nxt = t;
            nlis = nnext;
        }
        if (prof && lane == 0) { const unsigned long long t1 = __builtin_readcyclecounter(); atomicAdd(&g_spiht_prof[2], t1 - t0); t0 = t1; }
        if (stop) break;


        // ---------------- refinement pass, spiht_re.c:413-428: four bits per lane and round (the update of a coefficient is
        // a dependent load / store pair: four of them in flight instead of one)
        __threadfence_block();
        constexpr unsigned int kRef = 4 * kWave;
        for (unsigned int base = 0; base < n_old && !stop; base += kRef) {
            const unsigned int m = min(kRef, n_old - base);
            const int o0 = window(kHeaderBits + cnt, 11);
            unsigned int pj[4];
            int cj[4];
            bool on[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned int i = (unsigned int) (j * kWave + lane);
                const unsigned long long k = cnt + i + 1;                              // ordinal of this bit
                const unsigned int bit = win_bits_lane(wreg, o0 + (int) i, 1);        // (a cross-lane read: every lane takes part)
                on[j] = i < m && k <= B + 1 && bit;                                    // applied, then checked (:418-426)
                pj[j] = on[j] ? pk_index(lsp[base + i], g.nx) : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) cj[j] = on[j] ? C[pj[j]] : 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (on[j]) C[pj[j]] = cj[j] >= 0 ? (cj[j] | one) : -((-cj[j]) | one);
            cnt += (unsigned long long) m;
            if (cnt > B) stop = true;
        }
        if (prof && lane == 0) { atomicAdd(&g_spiht_prof[4], __builtin_readcyclecounter() - t0); atomicAdd(&g_spiht_prof[5], (unsigned long long) n_old); }
    }
    if (prof && lane == 0) {
        atomicAdd(&g_spiht_prof[6], cnt); atomicMax(&g_spiht_prof[7], cnt);
        atomicMax(&g_spiht_prof[8], __builtin_readcyclecounter() - prof_c0); atomicMax(&g_spiht_prof[9], __builtin_amdgcn_s_memrealtime() - prof_r0);
    }
}

__global__ __launch_bounds__(256) void k_zero_active(int32_t *__restrict__ C, size_t np, const int *active)
{
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 *c = reinterpret_cast<i32x4 *>(C + (size_t) frame * np);
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < np / 4; i += (size_t) gridDim.x * blockDim.x) c[i] = i32x4{0, 0, 0, 0};
}

__global__ __launch_bounds__(256) void k_int_to_float(const int32_t *__restrict__ C, float *__restrict__ A, size_t np,
                                                       const int *active)
{
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    // four coefficients per lane and step (np is a multiple of 4: the grid is padded to multiples of 8 both ways)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const i32x4 *c = reinterpret_cast<const i32x4 *>(C + (size_t) frame * np);
    f32x4 *a = reinterpret_cast<f32x4 *>(A + (size_t) frame * np);
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < np / 4; i += (size_t) gridDim.x * blockDim.x) {
        const i32x4 v = c[i];
        a[i] = f32x4{(float) v.x, (float) v.y, (float) v.z, (float) v.w};
    }
}

}  // namespace

// ================================================================================================
void launch_spiht_encode(const ResidualBuffers &rb, int n_frames, const unsigned long long *d_bits0, const int *d_active,
                         hipStream_t s)
{
    EBCC_HIP_CHECK(hipMemsetAsync(rb.sigord, 0xFF, (size_t) n_frames * rb.np * sizeof(uint32_t), s));
    EBCC_HIP_CHECK(hipMemsetAsync(rb.stream, 0, (size_t) n_frames * rb.stream_words * sizeof(uint32_t), s));
    ScopedTiming t("spiht_encode", s);
    hipLaunchKernelGGL(k_spiht_encode, dim3(n_frames), dim3(kEncThreads), 0, s, rb.C, rb.D, rb.G, rb.lip, rb.lsp,
                       rb.lis0, rb.lis1, rb.sigord, rb.lspidx, rb.stream, rb.stream_words, rb.g, rb.np, rb.fs, d_bits0, d_active);
    EBCC_HIP_LAUNCH_CHECK();
}

void launch_reconstruct(const ResidualBuffers &rb, int n_frames, const unsigned long long *d_trunc_bits,
                        const int *d_active, hipStream_t s)
{
    hipLaunchKernelGGL(k_reconstruct, dim3(128, n_frames), dim3(256), 0, s, rb.C, rb.sigord, rb.lspidx, rb.A, rb.np,
                       rb.fs, d_trunc_bits, d_active, rb.g, 0, (const int *) nullptr, rb.np);
    EBCC_HIP_LAUNCH_CHECK();
}

void launch_reconstruct_coarse(const ResidualBuffers &rb, int n_frames, const unsigned long long *d_trunc_bits,
                               const int *d_active, hipStream_t s)
{
    hipLaunchKernelGGL(k_reconstruct, dim3(64, n_frames), dim3(256), 0, s, rb.C, rb.sigord, rb.lspidx, rb.A, rb.np,
                       rb.fs, d_trunc_bits, d_active, rb.g, 1, (const int *) nullptr, rb.np);
    EBCC_HIP_LAUNCH_CHECK();
}

void launch_reconstruct_coarse_slots(const ResidualBuffers &rb, const CutSlots &cs, int n_slots, hipStream_t s)
{
    hipLaunchKernelGGL(k_reconstruct, dim3(64, n_slots), dim3(256), 0, s, rb.C, rb.sigord, rb.lspidx, cs.A, rb.np,
                       cs.fs, cs.bits, cs.active, rb.g, 1, cs.frame_of, cs.stride);
    EBCC_HIP_LAUNCH_CHECK();
}

void launch_spiht_decode(const uint8_t *d_streams, size_t stream_stride, const unsigned long long *d_sizes,
                         const unsigned long long *d_num_bits, const ResidualBuffers &rb, int n_frames,
                         const int *d_active, hipStream_t s)
{
    // (the decoder's lists hold packed 12-bit coordinates: the extents the stream header can carry, spiht_re.c:448-464)
    if (rb.g.nx > (1 << kPkShift) || rb.g.ny > (1 << kPkShift)) throw HipFailure("ebcc-mi355x: residual grid wider than the SPIHT header's 12-bit extents");
    // spiht_decode_init clears the coefficient grid, spiht_re.c:101 (of the frames that are decoded)
    hipLaunchKernelGGL(k_zero_active, dim3(128, n_frames), dim3(256), 0, s, rb.C, rb.np, d_active);
    static const bool prof = getenv("EBCC_HIP_SPIHT_PROF") != nullptr;
    if (prof) { unsigned long long z[12] = {}; EBCC_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_spiht_prof), z, sizeof z)); }
    timing_begin("spiht_decode", s);
    hipLaunchKernelGGL(k_spiht_decode, dim3(n_frames), dim3(kWave), 0, s, d_streams, stream_stride, d_sizes, d_num_bits,
                       rb.C, rb.lip, rb.lsp, rb.lis0, rb.lis1, rb.g, rb.np, rb.fs, d_active, prof ? 1 : 0);
    timing_end("spiht_decode", s);
    if (prof) {
        unsigned long long v[12];
        wait_stream(s);
        EBCC_HIP_CHECK(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_spiht_prof), sizeof v));
        fprintf(stderr, "spiht_decode profile (%d frames): LIP %.1f Mcycles %llu entries | LIS %.1f Mcycles %llu entries | refinement %.1f Mcycles %llu bits | %llu stream bits, longest %llu | slowest frame: %llu cycle-counter ticks in %.1f us (100 MHz clock)\n",
                n_frames, v[0] / 1e6, v[1], v[2] / 1e6, v[3], v[4] / 1e6, v[5], v[6], v[7], v[8], v[9] / 100.0);
        fprintf(stderr, "spiht_decode profile: %llu chunks in the LIS passes, the list registers loaded afresh %llu times\n", v[10], v[11]);
    }
    hipLaunchKernelGGL(k_int_to_float, dim3(128, n_frames), dim3(256), 0, s, rb.C, rb.A, rb.np, d_active);
    EBCC_HIP_LAUNCH_CHECK();
}

}  // namespace ebcc
