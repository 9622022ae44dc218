// j2k_rate.hip - rate-dependent half of the JPEG 2000 base layer on gfx950:
//   PCRD rate allocation (OpenJPEG 2.4.0 opj_tcd_rateallocate / opj_tcd_makelayer), tier-2 packet
//   headers (tag trees, T.800 B.10), codestream assembly, and the decoder (tier-1 MQ decoding with one
//   code-block per lane, dequantisation, inverse 9/7, level shift).
// Each of the reference's ~22 opj_encode+opj_decode probes per frame
// (/root/reference/src/ebcc_codec.c:535-596) becomes one rate allocation over the tier-1 results that
// were computed once, plus a decode of only the passes that allocation keeps, read in place from the
// encoder's code-block slots (no codestream is assembled or parsed for a probe).
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "j2k.hpp"
#include "t1_core.hpp"
#include "t1_device.hpp"

namespace ebcc {

int j2k_inverse_dwt(float *B, const int32_t *V, const float *data, const J2kBuffers &jb, int n_frames, const FrameState *fs,
                    const int *active, hipStream_t s, int keep_field);

namespace {

constexpr int kMainHeaderBytes = 135;
constexpr int kRateTrieNodes = 1536;  // recorded bisection steps of a frame (k_rate): ~80 per call less what calls share
#ifndef EBCC_RATE_THREADS
#define EBCC_RATE_THREADS 512
#endif
constexpr int kRateThreads = EBCC_RATE_THREADS;     // k_rate: one workgroup per frame, about one code-block per thread
constexpr int kWriteThreads = 256;    // k_write: one workgroup per frame (headers by one lane per resolution, bodies by all)

__device__ inline int floorlog2d(int a) { return a > 1 ? 31 - __clz(a) : 0; }

// ------------------------------------------------------------------------------------------------
// packet-header bit writer (B.10.1): counts bytes; writes them when `out` is non-null
// ------------------------------------------------------------------------------------------------
struct Bio {
    unsigned int buf;
    int ct;
    int n;
    uint8_t *out;
    __device__ void init(uint8_t *o) { buf = 0; ct = 8; n = 0; out = o; }
    __device__ void byteout()
    {
        buf = (buf << 8) & 0xFFFFu;
        ct = buf == 0xFF00u ? 7 : 8;
        if (out) out[n] = (uint8_t) (buf >> 8);
        n++;
    }
    __device__ void bit(int v)
    {
        if (ct == 0) byteout();
        ct--;
        buf |= (unsigned int) (v & 1) << ct;
    }
    __device__ void bits(unsigned int v, int nb) { for (int i = nb - 1; i >= 0; i--) bit((int) ((v >> i) & 1u)); }
    __device__ void flush() { byteout(); if (ct == 7) byteout(); }
};

// ------------------------------------------------------------------------------------------------
// tag trees in LDS: node arrays of one band start at tree_off; level l node (i, j) at lvl_off[l] + j*lvl_w[l] + i
// ------------------------------------------------------------------------------------------------
struct Trees {
    short *ival, *ilow, *mval, *mlow;       // inclusion / zero-bit-plane trees: value, low
    unsigned char *iknown, *mknown;
};

__device__ inline void tgt_encode(Bio &bio, const J2kBand &bd, short *val, short *low, unsigned char *known, int cx, int cy,
                                  int threshold)
{
    // walk root -> leaf (B.10.2); node of level l on the path is (cx >> l, cy >> l)
    int lowv = 0;
    for (int l = bd.tree_levels - 1; l >= 0; l--) {
        int idx = bd.tree_off + bd.lvl_off[l] + (cy >> l) * bd.lvl_w[l] + (cx >> l);
        if (lowv > low[idx]) low[idx] = (short) lowv; else lowv = low[idx];
        const int v = val[idx];
        while (lowv < threshold) {
            if (lowv >= v) {
                if (!known[idx]) { bio.bit(1); known[idx] = 1; }
                break;
            }
            bio.bit(0);
            ++lowv;
        }
        low[idx] = (short) lowv;
    }
}

__device__ inline void put_numpasses(Bio &b, int n)
{
    if (n == 1) b.bits(0, 1);
    else if (n == 2) b.bits(2, 2);
    else if (n <= 5) b.bits(0xCu | (unsigned) (n - 3), 4);
    else if (n <= 36) b.bits(0x1E0u | (unsigned) (n - 6), 9);
    else b.bits(0xFF80u | (unsigned) (n - 37), 16);
}

// header of the packet of resolution r for the layer assignment npass[]; returns header bytes and adds the
// body bytes to *body.  Trees must have been reset (and inclusion values set) by the caller.
__device__ int packet_header(int r, const J2kGeom &g, const Trees &t, const short *npass, const int *rates, int gid0,
                             uint8_t *out, int *body)
{
    Bio bio;
    bio.init(out);
    bio.bit(1);                                                      // OpenJPEG 2.4.0 never writes an empty-packet bit
    int bytes = 0;
    for (int bi = 0; bi < g.nbands; bi++) {
        const J2kBand &bd = g.bands[bi];
        if (bd.res != r || bd.ncw * bd.nch == 0) continue;
        for (int cy = 0; cy < bd.nch; cy++)
            for (int cx = 0; cx < bd.ncw; cx++) {
                const int blk = bd.first_block + cy * bd.ncw + cx;
                tgt_encode(bio, bd, t.ival, t.ilow, t.iknown, cx, cy, 1);
                const int n = npass[blk];
                if (!n) continue;
                tgt_encode(bio, bd, t.mval, t.mlow, t.mknown, cx, cy, 999);
                put_numpasses(bio, n);
                const int seglen = rates[(size_t) (gid0 + blk) * kJ2kMaxPasses + n - 1];
                int inc = floorlog2d(seglen) + 1 - (3 + floorlog2d(n));
                if (inc < 0) inc = 0;
                for (int k = 0; k < inc; k++) bio.bit(1);
                bio.bit(0);
                bio.bits((unsigned) seglen, 3 + inc + floorlog2d(n));
                bytes += seglen;
            }
    }
    bio.flush();
    *body = bytes;
    return bio.n;
}

// k_rate's own arrays behind the carve-up: the per-block table offsets, then three sets of path masks and a flag
__host__ __device__ inline size_t rate_off_bytes(int nblocks) { return ((size_t) (nblocks + 1) * 4 + 15) & ~(size_t) 15; }
__host__ __device__ inline size_t rate_path_bytes(int nblocks) { return (size_t) nblocks * 2 * 8 * 3 + (((size_t) nblocks + 7) & ~(size_t) 7); }

__host__ __device__ inline size_t rate_lds_bytes(int nblocks, int nodes)
{
    return 2 * ((((size_t) nblocks * 2 + 7) / 8) * 8) + (size_t) nodes * 16 + (size_t) nblocks * 20 + 128;
}

// LDS carve-up shared by the rate and write kernels
struct RateLds {
    short *npass;
    Trees t;
    short *mval0;        // static zero-bit-plane node minima
    // parallel size evaluation
    short *prev;         // previous layer assignment (unchanged assignment => unchanged size)
    int *firstinc;       // per tree node: raster index of the first included leaf below it (INT_MAX = none)
    unsigned int *raw;   // unstuffed header bits, 128 bits per code-block, per-resolution regions
    int *leafbits;       // per code-block header bit count (scratch of the scan)
    __device__ static size_t bytes(int nblocks, int nodes)
    {
        return (size_t) nblocks * 2 + (size_t) nodes * (2 * 5 + 2) + 64;
    }
    __device__ void carve(unsigned char *base, int nblocks, int nodes)
    {
        npass = (short *) base; base += (((size_t) nblocks * 2 + 7) / 8) * 8;
        t.ival = (short *) base; base += (size_t) nodes * 2;
        t.ilow = (short *) base; base += (size_t) nodes * 2;
        t.mval = (short *) base; base += (size_t) nodes * 2;
        t.mlow = (short *) base; base += (size_t) nodes * 2;
        mval0 = (short *) base; base += (size_t) nodes * 2;
        t.iknown = base; base += nodes;
        t.mknown = base; base += nodes;
        base = (unsigned char *) ((((size_t) base) + 7) & ~(size_t) 7);
        prev = (short *) base; base += (((size_t) nblocks * 2 + 7) / 8) * 8;
        firstinc = (int *) base; base += (size_t) nodes * 4;
        raw = (unsigned int *) base; base += (size_t) nblocks * 16;
        leafbits = (int *) base;
    }
};

__device__ inline void tree_setmin(const J2kBand &bd, short *val, int cx, int cy, int v)
{
    for (int l = 0; l < bd.tree_levels; l++) {
        int idx = bd.tree_off + bd.lvl_off[l] + (cy >> l) * bd.lvl_w[l] + (cx >> l);
        if (val[idx] > v) val[idx] = (short) v; else break;
    }
}

// static part: node minima of Mb - numbps over ALL code-blocks (opj_tgt_setvalue for every cblk)
template <int NT>
__device__ void trees_static(const J2kGeom &g, RateLds &L, const int *numbps, int gid0, int lane)
{
    // leaves, then every level from its (up to four) children: a node's value is the minimum below it
    for (int b = lane; b < g.nblocks; b += NT) {
        int bi = 0;
        while (bi + 1 < g.nbands && g.bands[bi + 1].first_block <= b) bi++;
        const J2kBand &bd = g.bands[bi];
        const int k = b - bd.first_block, cy = k / bd.ncw, cx = k - cy * bd.ncw;
        L.mval0[bd.tree_off + bd.lvl_off[0] + cy * bd.lvl_w[0] + cx] = (short) (bd.numbps - numbps[gid0 + b]);
    }
    __syncthreads();
    for (int l = 1; l < 12; l++) {
        bool any = false;
        for (int bi = 0; bi < g.nbands; bi++) {
            const J2kBand &bd = g.bands[bi];
            if (l >= bd.tree_levels) continue;
            any = true;
            const int w = bd.lvl_w[l], h = bd.lvl_h[l], cw = bd.lvl_w[l - 1], ch = bd.lvl_h[l - 1];
            for (int n = lane; n < w * h; n += NT) {
                const int j = n / w, i = n - j * w;
                const short *c = L.mval0 + bd.tree_off + bd.lvl_off[l - 1];
                int m = c[(2 * j) * cw + 2 * i];
                if (2 * i + 1 < cw) m = min(m, (int) c[(2 * j) * cw + 2 * i + 1]);
                if (2 * j + 1 < ch) {
                    m = min(m, (int) c[(2 * j + 1) * cw + 2 * i]);
                    if (2 * i + 1 < cw) m = min(m, (int) c[(2 * j + 1) * cw + 2 * i + 1]);
                }
                L.mval0[bd.tree_off + bd.lvl_off[l] + n] = (short) m;
            }
        }
        if (!any) break;
        __syncthreads();
    }
    __syncthreads();
}

template <int NT>
__device__ void trees_reset(const J2kGeom &g, RateLds &L, int lane)
{
    for (int i = lane; i < g.tree_nodes; i += NT) {
        L.t.ival[i] = 999; L.t.ilow[i] = 0; L.t.iknown[i] = 0;
        L.t.mval[i] = L.mval0[i]; L.t.mlow[i] = 0; L.t.mknown[i] = 0;
    }
    __syncthreads();
    if (lane < g.nbands) {
        const J2kBand &bd = g.bands[lane];
        for (int cy = 0; cy < bd.nch; cy++)
            for (int cx = 0; cx < bd.ncw; cx++)
                if (L.npass[bd.first_block + cy * bd.ncw + cx]) tree_setmin(bd, L.t.ival, cx, cy, 0);
    }
    __syncthreads();
}

// Per-pass rate / distortion tables of one frame.  k_rate walks them ~60 times per call, so they are staged in
// LDS when all code-blocks of the frame fit (10 bytes per coding pass); otherwise they are read from global
// memory as they are.
struct PassTab {
    const int *rates;              // [nblocks][kJ2kMaxPasses] of this frame (global memory)
    const double *disto;
    const __attribute__((address_space(3))) double *l_disto;   // LDS copies, code-block b at l_off[b] .. + totalpasses
    const __attribute__((address_space(3))) unsigned short *l_rate;   // (a code-block's bytes fit its 16 KB slot)
    const __attribute__((address_space(3))) int *l_off;
    bool lds;
    // (kept in their own address spaces: a pointer that may be either would turn every access into a flat load)
    template <bool LDS> __device__ int rate(int base, int p) const { if constexpr (LDS) return l_rate[base + p]; else return rates[base + p]; }
    template <bool LDS> __device__ double dist(int base, int p) const { if constexpr (LDS) return l_disto[base + p]; else return disto[base + p]; }
    template <bool LDS> __device__ int base_of(int b) const { if constexpr (LDS) return l_off[b]; else return b * kJ2kMaxPasses; }
};

// opj_tcd_makelayer for one quality layer
// opj_tcd_makelayer's greedy walk over the passes of code-block b: passes taken (count returned, bit mask in m0 / m1)
// `from` > 0: the walk is taken up at pass `from` with passes [0, from) decided already - their mask in m0 / m1 on entry,
// the last pass taken among them `n_in` (count, 0: none) - see the tail of k_rate's bisection
template <bool LDS>
__device__ inline int walk_block(const PassTab &pt, const int *totalpasses, int gid0, int b, double thresh, unsigned long long &m0,
                                 unsigned long long &m1, int from = 0, int n_in = 0)
{
    if (from == 0) m0 = m1 = 0;
    int tp;
    if constexpr (LDS) tp = pt.l_off[b + 1] - pt.l_off[b]; else tp = totalpasses[gid0 + b];
    const int base = pt.base_of<LDS>(b);
    int n = n_in;
    if (thresh < 0) return tp;
    int rbase = 0;                                                   // rate / distortion of the last pass taken
    double dbase = 0;
    if (n_in > 0) { rbase = pt.rate<LDS>(base, n_in - 1); dbase = pt.dist<LDS>(base, n_in - 1); }
    for (int p = from; p < tp; p++) {
        const int rp = pt.rate<LDS>(base, p);
        const double dp = pt.dist<LDS>(base, p);
        unsigned int dr;
        double dd;
        if (n == 0) { dr = (unsigned int) rp; dd = dp; }
        else { dr = (unsigned int) (rp - rbase); dd = dp - dbase; }
        // OpenJPEG's test is  thresh - fl(dd / dr) < DBL_EPSILON.  The outcome is already certain whenever dd and
        // thresh * dr differ by more than rounding can bridge (1e-11 relative, against 2^-52 per operation); only the
        // narrow band in between is divided.  (The reject shortcut also needs DBL_EPSILON to be negligible next to
        // thresh.)
        bool take;
        if (!dr) take = dd != 0;
        else {
            const double tdr = thresh * (double) dr;
            if (dd >= tdr * 1.00000000001) take = true;
            else if (thresh >= 1e-4 && dd <= tdr * 0.99999999999) take = false;
            else take = thresh - (dd / dr) < DBL_EPSILON;
        }
        if (take) {
            n = p + 1; rbase = rp; dbase = dp;
            if (p < 64) m0 |= 1ull << p; else m1 |= 1ull << (p - 64);
        }
    }
    return n;
}

// The thresholds at which a given greedy walk IS the walk: a pass with slope s = fl(dd / dr) is taken iff fl(thresh - s) <
// DBL_EPSILON, which holds up to a limit T(s) and fails from it on (the difference is monotone in thresh).  So the walk
// that takes exactly the passes of (m0, m1) from pass `from` on (the passes below it are given: last one taken n_in) is
// the walk at thresh iff A <= thresh < B, with B the smallest limit of a pass it takes and A the largest limit of a pass
// it leaves - two numbers per walk instead of a walk per threshold (the tail of k_rate's bisection).
__device__ inline double next_up(double x)       // the next double above a finite x
{
    long long u = __double_as_longlong(x);
    if (x == 0.0) return __longlong_as_double(1ll);
    u += x > 0.0 ? 1 : -1;
    return __longlong_as_double(u);
}
__device__ inline double next_down(double x) { return -next_up(-x); }
__device__ inline double take_limit(double s)
{
    double t = s + DBL_EPSILON;                                        // close to the limit; settle on it exactly
    for (int k = 0; k < 8 && (t - s) < DBL_EPSILON; k++) t = next_up(t);
    for (int k = 0; k < 8; k++) { const double d = next_down(t); if ((d - s) < DBL_EPSILON) break; t = d; }
    return t;
}
// [tlo, thi]: the bracket the thresholds to come lie in.  A pass whose slope is clearly outside it (by 1e-9 relative; only
// used when the bracket is far above DBL_EPSILON) either does not constrain them at all or rules the walk out for all of
// them - its exact limit is not needed, nor the division that gives its slope.
template <bool LDS>
__device__ inline void walk_interval(const PassTab &pt, const int *totalpasses, int gid0, int b, unsigned long long m0, unsigned long long m1,
                                     int from, int n_in, double tlo, double thi, double &A, double &B)
{
    int tp;
    if constexpr (LDS) tp = pt.l_off[b + 1] - pt.l_off[b]; else tp = totalpasses[gid0 + b];
    const int base = pt.base_of<LDS>(b);
    int n = n_in, rbase = 0;
    double dbase = 0;
    if (n_in > 0) { rbase = pt.rate<LDS>(base, n_in - 1); dbase = pt.dist<LDS>(base, n_in - 1); }
    A = -DBL_MAX; B = DBL_MAX;
    const bool coarse = tlo >= 1e-4;
    for (int p = from; p < tp; p++) {
        const int rp = pt.rate<LDS>(base, p);
        const double dp = pt.dist<LDS>(base, p);
        const unsigned int dr = n == 0 ? (unsigned int) rp : (unsigned int) (rp - rbase);
        const double dd = n == 0 ? dp : dp - dbase;
        const bool taken = p < 64 ? (m0 >> p) & 1ull : (m1 >> (p - 64)) & 1ull;
        if (dr) {
            const double fdr = (double) dr;
            if (coarse && dd > thi * fdr * 1.000000001) {                // limit above every threshold to come
                if (!taken) A = DBL_MAX;
            } else if (coarse && dd < tlo * fdr * 0.999999999) {         // limit below every threshold to come
                if (taken) B = -DBL_MAX;
            } else {
                const double lim = take_limit(dd / fdr);
                if (taken) B = lim < B ? lim : B; else A = lim > A ? lim : A;
            }
        } else if (taken != (dd != 0)) { A = DBL_MAX; B = -DBL_MAX; }    // (cannot happen for a walk that was walked: never matches)
        if (taken) { n = p + 1; rbase = rp; dbase = dp; }
    }
}

// `path` (optional): [2][nblocks] bit masks of the passes taken, i.e. the whole greedy walk and not just its end;
// `frozen` (optional): code-blocks whose walk cannot change any more (k_rate) keep their assignment
template <int NT, bool LDS>
__device__ void make_layer_impl(const J2kGeom &g, RateLds &L, const int *totalpasses, const PassTab &pt, int gid0, double thresh, int lane,
                                unsigned long long *path, const unsigned char *frozen)
{
    for (int b = lane; b < g.nblocks; b += NT) {
        if (frozen && frozen[b]) continue;
        unsigned long long m0, m1;
        const int n = walk_block<LDS>(pt, totalpasses, gid0, b, thresh, m0, m1);
        L.npass[b] = (short) n;
        if (path) { path[b] = m0; path[g.nblocks + b] = m1; }
    }
    __syncthreads();
}

template <int NT>
__device__ void make_layer(const J2kGeom &g, RateLds &L, const int *totalpasses, const PassTab &pt, int gid0, double thresh, int lane,
                           unsigned long long *path = nullptr, const unsigned char *frozen = nullptr)
{
    if (pt.lds) make_layer_impl<NT, true>(g, L, totalpasses, pt, gid0, thresh, lane, path, frozen);
    else make_layer_impl<NT, false>(g, L, totalpasses, pt, gid0, thresh, lane, path, frozen);
}

// total packet bytes of the current assignment (one lane per resolution)
template <int NT>
__device__ int layer_bytes(const J2kGeom &g, RateLds &L, const int *rates, int gid0, int lane, int *s_sum)
{
    trees_reset<NT>(g, L, lane);
    if (lane == 0) *s_sum = 0;
    __syncthreads();
    if (lane < kJ2kRes) {
        int body = 0;
        int hdr = packet_header(lane, g, L.t, L.npass, rates, gid0, nullptr, &body);
        atomicAdd(s_sum, hdr + body);
    }
    __syncthreads();
    return *s_sum;
}


// ------------------------------------------------------------------------------------------------
// Parallel evaluation of the packet sizes (what opj_t2_encode_packets(THRESH_CALC) returns).
// Tag-tree coding in closed form (B.10.2 with every leaf visited in raster order):
//   inclusion tree, threshold 1: a node emits ONE bit - "some leaf below me is included" - when its top-left
//     leaf is visited, provided every ancestor has an included leaf (otherwise the walk stopped above);
//   zero-bit-plane tree: a node emits (value - parent's value) zeros and a one when the FIRST INCLUDED leaf
//     below it is visited; node values are static minima over all leaves.
// So every code-block derives its header bits independently; offsets come from a scan, the bits are OR-ed
// into LDS, and the B.10.1 bit stuffing (7 bits after a 0xFF byte) is counted by a ballot search for the
// next 0xFF byte.  Byte-for-byte equal to the serial writer (packet_header) that k_write uses.
// ------------------------------------------------------------------------------------------------
__device__ inline void raw_put(unsigned int *raw, int pos, unsigned long long v, int n)
{
    // n <= 64 bits of v (right-aligned), MSB first, at bit position pos
    while (n > 0) {
        int take = n > 32 ? n - 32 : n;                              // leading chunk first
        if (n <= 32) take = n;
        unsigned int chunk = (unsigned int) ((v >> (n - take)) & (take == 32 ? 0xFFFFFFFFull : ((1ull << take) - 1)));
        int w = pos >> 5, sh = pos & 31;
        unsigned long long x = (unsigned long long) chunk << (64 - take - sh);
        unsigned int hi = (unsigned int) (x >> 32), lo = (unsigned int) x;
        if (hi) atomicOr(&raw[w], hi);
        if (lo) atomicOr(&raw[w + 1], lo);
        pos += take; n -= take;
    }
}

__device__ inline unsigned int raw_byte(const unsigned int *raw, int pos)
{
    int w = pos >> 5, sh = pos & 31;
    unsigned long long x = ((unsigned long long) raw[w] << 32) | raw[w + 1];
    return (unsigned int) ((x << sh) >> 56);
}

// hdr_out (optional, LDS): header bytes of every resolution's packet.  Leaves the unstuffed header bits in L.raw
// (resolution r at bit res_first[r] * 128) and every leaf's bit offset in L.leafbits for the writer.
template <int NT>
__device__ int layer_bytes_fast(const J2kGeom &g, RateLds &L, const int *rates, int gid0, int lane, int *s_sum, int *hdr_out = nullptr)
{
    const int INF = 0x7FFFFFFF;
    for (int i = lane; i < g.tree_nodes; i += NT) L.firstinc[i] = INF;
    for (int i = lane; i < g.nblocks * 4 + 4; i += NT) L.raw[i] = 0;
    if (lane == 0) *s_sum = 0;
    __syncthreads();
    // first included leaf below every node
    for (int b = lane; b < g.nblocks; b += NT) {
        if (!L.npass[b]) continue;
        // band of the block: bands are few, find by first_block
        int bi = 0;
        while (bi + 1 < g.nbands && g.bands[bi + 1].first_block <= b) bi++;
        const J2kBand &bd = g.bands[bi];
        const int k = b - bd.first_block, cy = k / bd.ncw, cx = k - cy * bd.ncw;
        for (int l = 0; l < bd.tree_levels; l++)
            atomicMin(&L.firstinc[bd.tree_off + bd.lvl_off[l] + (cy >> l) * bd.lvl_w[l] + (cx >> l)], k);
    }
    if (lane < kJ2kRes) atomicOr(&L.raw[(g.res_first[lane] * 128) >> 5], 0x80000000u);   // leading "packet present" bit
    __syncthreads();
    // per-leaf header bits: A = inclusion + zero-bit-plane bits, B = passes + Lblock comma code + length
    __shared__ int s_wave[NT / 64];
    int body = 0, running = 0;
    for (int base = 0; base < g.nblocks; base += NT) {
        const int b = base + lane;
        unsigned long long A = 0, B = 0;
        int na = 0, nbb = 0, res = 0;
        if (b < g.nblocks) {
            int bi = 0;
            while (bi + 1 < g.nbands && g.bands[bi + 1].first_block <= b) bi++;
            const J2kBand &bd = g.bands[bi];
            res = bd.res;
            const int k = b - bd.first_block, cy = k / bd.ncw, cx = k - cy * bd.ncw;
            bool alive = true;
            for (int l = bd.tree_levels - 1; l >= 0 && alive; l--) {
                const int node = bd.tree_off + bd.lvl_off[l] + (cy >> l) * bd.lvl_w[l] + (cx >> l);
                const bool has = L.firstinc[node] != INF;
                const int mask = (1 << l) - 1;
                if (((cx & mask) | (cy & mask)) == 0) { A = (A << 1) | (has ? 1ull : 0ull); na++; }
                if (!has) alive = false;
            }
            const int n = L.npass[b];
            if (n) {
                int prevm = 0;
                for (int l = bd.tree_levels - 1; l >= 0; l--) {
                    const int node = bd.tree_off + bd.lvl_off[l] + (cy >> l) * bd.lvl_w[l] + (cx >> l);
                    const int m = L.mval0[node];
                    if (L.firstinc[node] == k) { const int z = m - prevm; A = ((A << z) << 1) | 1ull; na += z + 1; }
                    prevm = m;
                }
                // number of passes (table B.4)
                if (n == 1) { B = 0; nbb = 1; }
                else if (n == 2) { B = 2; nbb = 2; }
                else if (n <= 5) { B = 0xCull | (unsigned) (n - 3); nbb = 4; }
                else if (n <= 36) { B = 0x1E0ull | (unsigned) (n - 6); nbb = 9; }
                else { B = 0xFF80ull | (unsigned) (n - 37); nbb = 16; }
                const int seglen = rates[(size_t) (gid0 + b) * kJ2kMaxPasses + n - 1];
                int inc = floorlog2d(seglen) + 1 - (3 + floorlog2d(n));
                if (inc < 0) inc = 0;
                B = (B << (inc + 1)) | (((1ull << inc) - 1) << 1);                 // inc ones, then a zero
                nbb += inc + 1;
                const int lb = 3 + inc + floorlog2d(n);
                B = (B << lb) | ((unsigned long long) seglen & ((1ull << lb) - 1));
                nbb += lb;
                body += seglen;
            }
        }
        // exclusive scan of the header bit counts over all leaves (packet order): L.leafbits[b] = bits before leaf b
        {
            const int v = b < g.nblocks ? na + nbb : 0;
            int x = v;
            for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if ((lane & 63) >= d) x += y; }
            if ((lane & 63) == 63) s_wave[lane >> 6] = x;
            __syncthreads();
            int before = running;
            for (int w = 0; w < (lane >> 6); w++) before += s_wave[w];
            if (b < g.nblocks) L.leafbits[b] = before + x - v;
            int tot = 0;
            for (int w = 0; w < NT / 64; w++) tot += s_wave[w];
            running += tot;
        }
        __syncthreads();
        // offsets inside the resolution: leaves are in packet order, resolutions are contiguous
        if (b < g.nblocks) {
            const int off = 1 + L.leafbits[b] - L.leafbits[g.res_first[res]];      // 1: the leading "packet present" bit
            const int region = g.res_first[res] * 128;
            raw_put(L.raw, region + off, A, na);
            raw_put(L.raw, region + off + na, B, nbb);
        }
        __syncthreads();
    }
    if (lane == 0) L.leafbits[g.nblocks] = running;
    __syncthreads();
    for (int d = 32; d >= 1; d >>= 1) body += __shfl_xor(body, d);
    // stuffing-aware byte count: one wave per resolution (the search for the next 0xFF byte is a chain)
    const int wl = lane & 63;
    int count = 0;
    for (int r = lane >> 6; r < kJ2kRes; r += NT / 64) {
        const int region = g.res_first[r] * 128;
        const int T = 1 + L.leafbits[g.res_first[r + 1]] - L.leafbits[g.res_first[r]];
        const int count_before = count;
        int p = 0;
        for (;;) {
            // first full byte == 0xFF at or after bit p (8-bit groups)
            const int groups = (T - p) >> 3;                                       // full bytes available
            int found = -1;
            for (int j0 = 0; j0 < groups && found < 0; j0 += 64) {
                const int j = j0 + wl;
                const bool hit = j < groups && raw_byte(L.raw, region + p + 8 * j) == 0xFFu;
                const unsigned long long m = __ballot(hit);
                if (m) found = j0 + (__ffsll((long long) m) - 1);
            }
            if (found < 0) { count += (T - p + 7) >> 3; break; }
            count += found + 1 + 1;                                                // bytes up to the FF, plus the 7-bit byte / flush byte after it
            p += 8 * (found + 1) + 7;
            if (p >= T) break;
        }
        if (hdr_out && wl == 0) hdr_out[r] = count - count_before;
    }
    if (wl == 0) atomicAdd(s_sum, body + count);
    __syncthreads();
    const int total = *s_sum;
    __syncthreads();
    return total;
}

// ================================================================================================
// rate allocation kernel: one workgroup (one wave) per frame
// ================================================================================================
__global__ __launch_bounds__(kRateThreads) void k_rate(const int *__restrict__ numbps, const int *__restrict__ totalpasses,
                                                        const int *__restrict__ rates, const double *__restrict__ disto,
                                                        int *__restrict__ npass_out, const J2kGeom *geom, J2kFrame *jf,
                                                        const FrameState *fs, const int *active, int pass_capacity,
                                                        int *__restrict__ path_bytes, int *__restrict__ path_n,
                                                        const float *cand_cr, int *cand_out, size_t cand_stride, const int *have_rate,
                                                        J2kBuffers::RateCache cache)
{
    extern __shared__ unsigned char lds_raw[];
    __shared__ int s_sum, s_changed;
#ifdef EBCC_RATE_PROFILE
    const long long t_start = wall_clock64();                        // 100 MHz
#endif
    __shared__ double s_min[kRateThreads / 64], s_max[kRateThreads / 64];
    __shared__ double s_td[5];                                        // state handed between the block-wide loop and its tail
    __shared__ int s_ti[5], s_nun, s_list[64];
    const int frame = blockIdx.x, lane = threadIdx.x;
    if ((active && !active[frame]) || fs[frame].const_field) return;
    // Candidate mode (cand_cr != null; launch_j2k_rate_candidates): the layer of a rate the search MAY ask for next, worked out
    // beside the decode of the current probe - blockIdx.y = which of the two candidates; results go to the candidate slots
    // (cand_out, npass_out + slot * cand_stride) and the trie of recorded bisection steps is only read.  Normal mode: frames
    // whose layer was taken over from a candidate (have_rate) are done already.
    const bool candidate = cand_cr != nullptr;
    const int slot = candidate ? (int) blockIdx.y : 0;
    const float cr_in = candidate ? cand_cr[2 * frame + slot] : jf[frame].cr;
    if (candidate ? !(cr_in > 0.0f) : (have_rate && have_rate[frame])) return;
    if (candidate) npass_out += (size_t) slot * cand_stride;
    const bool record = !candidate;
    const J2kGeom &g = j2k_frame_geom(geom, frame);
    const int gid0 = frame * g.stride;
    RateLds L;
    L.carve(lds_raw, g.nblocks, g.tree_nodes);
#ifdef EBCC_RATE_PROFILE
    long long t_s[5] = {0, 0, 0, 0, 0};
    t_s[0] = wall_clock64() - t_start;
#endif
    // Set-up that is the same at every call for this frame - the trees' static minima, the table offsets, the slope range,
    // the pass tables in the order they are staged - is worked out by the frame's first (recording) call and kept in
    // global memory (J2kBuffers::RateCache, reset by the analysis): 58 us of every later call had been this.
    // (the flag is read ONCE per workgroup and handed round through LDS: every thread takes the same barrier path even if a
    //  recording call of the frame were to publish while this one starts)
    __shared__ int s_cached;
    if (lane == 0) s_cached = __hip_atomic_load(&cache.ok[frame], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool cached = s_cached != 0;
    short *const c_mval = cache.mval + (size_t) frame * cache.nodes_cap;
    int *const c_off = cache.off + (size_t) frame * (size_t) (g.stride + 1);
    unsigned short *const c_rate = cache.crate + (size_t) frame * (size_t) cache.cap;
    double *const c_disto = cache.cdisto + (size_t) frame * (size_t) cache.cap;
    if (cached) {
        for (int i = lane; i < g.tree_nodes; i += kRateThreads) L.mval0[i] = c_mval[i];
        __syncthreads();
    } else {
        trees_static<kRateThreads>(g, L, numbps, gid0, lane);
        if (record) for (int i = lane; i < g.tree_nodes; i += kRateThreads) c_mval[i] = L.mval0[i];
    }
#ifdef EBCC_RATE_PROFILE
    t_s[1] = wall_clock64() - t_start;
#endif
    for (int b = lane; b < g.nblocks; b += kRateThreads) L.prev[b] = -1;
    // pass tables into LDS (behind the RateLds carve-up) when they fit
    __shared__ int s_fit;
    unsigned char *extra = lds_raw + ((rate_lds_bytes(g.nblocks, g.tree_nodes) + 15) & ~(size_t) 15);
    int *l_off = (int *) extra;
    // greedy walks (make_layer_impl `path`) at the current threshold and at the two ends of the bisection bracket,
    // and the code-blocks whose walk is settled
    unsigned long long *p_cur = (unsigned long long *) (extra + rate_off_bytes(g.nblocks));
    unsigned long long *p_lo = p_cur + 2 * g.nblocks, *p_hi = p_lo + 2 * g.nblocks;
    unsigned char *frozen = (unsigned char *) (p_hi + 2 * g.nblocks);
    double *l_disto = (double *) (frozen + (((size_t) g.nblocks + 7) & ~(size_t) 7));
    for (int b = lane; b < g.nblocks; b += kRateThreads) frozen[b] = 0;
    // table offsets: exclusive prefix sums of the pass counts (block scan: wave shuffles + one LDS round per 512 code-blocks;
    // a single lane walking the ~300 counts cost 8 us of every call)
    if (cached) {
        for (int b = lane; b <= g.nblocks; b += kRateThreads) l_off[b] = c_off[b];
        if (lane == 0) { const int running = c_off[g.nblocks]; s_fit = running <= pass_capacity ? running : -1; }
    } else {
        __shared__ int s_scan[kRateThreads / 64];
        int running = 0;
        for (int base = 0; base < g.nblocks; base += kRateThreads) {
            const int b = base + lane;
            const int tp = b < g.nblocks ? totalpasses[gid0 + b] : 0;
            int x = tp;
            for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if ((lane & 63) >= d) x += y; }
            if ((lane & 63) == 63) s_scan[lane >> 6] = x;
            __syncthreads();
            int before = running, tot = 0;
            for (int w = 0; w < kRateThreads / 64; w++) { const int v = s_scan[w]; if (w < (lane >> 6)) before += v; tot += v; }
            if (b < g.nblocks) l_off[b] = before + x - tp;
            running += tot;
            __syncthreads();
        }
        if (lane == 0) { l_off[g.nblocks] = running; s_fit = running <= pass_capacity ? running : -1; }
        __syncthreads();
        if (record) for (int b = lane; b <= g.nblocks; b += kRateThreads) c_off[b] = l_off[b];
    }
    __syncthreads();
#ifdef EBCC_RATE_PROFILE
    t_s[2] = wall_clock64() - t_start;
#endif
    const int n_entries = s_fit;
    unsigned short *l_rate = (unsigned short *) (l_disto + (n_entries > 0 ? n_entries : 0));
    typedef __attribute__((address_space(3))) double lds_double;
    typedef __attribute__((address_space(3))) int lds_int;
    typedef __attribute__((address_space(3))) unsigned short lds_ushort;
    PassTab pt{rates + (size_t) gid0 * kJ2kMaxPasses, disto + (size_t) gid0 * kJ2kMaxPasses, (const lds_double *) l_disto,
               (const lds_ushort *) l_rate, (const lds_int *) l_off, n_entries >= 0};
    if (pt.lds && cached) {
        // (packed by the first call: a straight copy)
        for (int i = lane; i < n_entries; i += kRateThreads) { l_rate[i] = c_rate[i]; l_disto[i] = c_disto[i]; }
        __syncthreads();
    } else if (pt.lds) {
        // all threads over all (code-block, pass) slots of the frame's tables: consecutive threads read consecutive
        // entries, nothing depends on the previous load (one thread per code-block copying its ~40 passes one after the
        // other was 30 us of every call)
        const int slots = g.nblocks * kJ2kMaxPasses;
#pragma unroll 4
        for (int idx = lane; idx < slots; idx += kRateThreads) {
            // (the loads do not wait for the test: slots past a code-block's last pass exist and are simply not kept)
            const int r_ = pt.rates[idx];
            const double d_ = pt.disto[idx];
            const int b = idx / kJ2kMaxPasses, p = idx - b * kJ2kMaxPasses;
            const int o = l_off[b];
            if (p < l_off[b + 1] - o) {
                l_rate[o + p] = (unsigned short) r_;
                l_disto[o + p] = d_;
                if (record) { c_rate[o + p] = (unsigned short) r_; c_disto[o + p] = d_; }
            }
        }
        __syncthreads();
    }
#ifdef EBCC_RATE_PROFILE
    t_s[3] = wall_clock64() - t_start;
#endif
    int prev_bytes = 0;
    // size of the current assignment; identical assignments (late bisection steps) reuse the previous result
    auto sized = [&]() -> int {
        if (lane == 0) s_changed = 0;
        __syncthreads();
        int ch = 0;
        for (int b = lane; b < g.nblocks; b += kRateThreads) { if (L.prev[b] != L.npass[b]) ch = 1; L.prev[b] = L.npass[b]; }
        if (ch) s_changed = 1;
        __syncthreads();
        if (s_changed) {
#ifdef EBCC_RATE_CHECK
            const int ref = layer_bytes<kRateThreads>(g, L, rates, gid0, lane, &s_sum);
            prev_bytes = layer_bytes_fast<kRateThreads>(g, L, rates, gid0, lane, &s_sum);
            if (ref != prev_bytes && lane == 0) { printf("rate check: frame %d fast %d serial %d\n", frame, prev_bytes, ref); }
            prev_bytes = ref;
#else
            prev_bytes = layer_bytes_fast<kRateThreads>(g, L, rates, gid0, lane, &s_sum);
#endif
        }
        return prev_bytes;
    };

    // opj_j2k_setup_encoder / opj_j2k_update_rates: byte budget of the single layer
    float rate = cr_in / 2;                                          // tcp_rates[0] = base_cr / 2, ebcc_codec.c:116
    if (rate <= 1.0f) rate = 0.0f;                                   // "force lossless"
    if (rate > 0.0f) {
        rate = (float) (((double) 16 * (double) g.W * (double) g.H) / ((double) rate * (double) 8)) - 0.0f;
        rate -= jf[frame].hdr_share > 0.0f ? jf[frame].hdr_share : (float) kMainHeaderBytes / 1.0f;
        if (rate < 30.0f) rate = 30.0f;
    }

    // slope range over consecutive passes (opj_tcd_rateallocate)
    double mn = DBL_MAX, mx = 0;
    if (cached) { mn = cache.mnmx[2 * frame]; mx = cache.mnmx[2 * frame + 1]; }
    else {
    for (int b = lane; b < g.nblocks; b += kRateThreads) {
        const int tp = l_off[b + 1] - l_off[b], o = l_off[b];
        int rprev = 0; double dprev = 0;
        for (int p = 0; p < tp; p++) {
            // (from the LDS copies when there are any: the same values)
            const int rp = pt.lds ? (int) l_rate[o + p] : pt.rates[(size_t) b * kJ2kMaxPasses + p];
            const double dp = pt.lds ? l_disto[o + p] : pt.disto[(size_t) b * kJ2kMaxPasses + p];
            const int dr = rp - rprev; const double dd = dp - dprev;
            rprev = rp; dprev = dp;
            if (dr == 0) continue;
            double sl = dd / dr;
            if (sl < mn) mn = sl;
            if (sl > mx) mx = sl;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const double omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d);
        mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx;
    }
    if ((lane & 63) == 0) { s_min[lane >> 6] = mn; s_max[lane >> 6] = mx; }
    __syncthreads();
    for (int i = 0; i < kRateThreads / 64; i++) { mn = s_min[i] < mn ? s_min[i] : mn; mx = s_max[i] > mx ? s_max[i] : mx; }
    // the first recording call leaves all of the above for the later ones: every thread's part of the cache is fenced, then
    // the flag is stored with release order
    if (record) {
        __threadfence();
        __syncthreads();
        if (lane == 0) {
            cache.mnmx[2 * frame] = mn; cache.mnmx[2 * frame + 1] = mx;
            if (pt.lds) { __threadfence(); __hip_atomic_store(&cache.ok[frame], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
        }
    }
    }

#ifdef EBCC_RATE_PROFILE
    long long t_setup = wall_clock64() - t_start, t_ml = 0, t_sz = 0, t_replay_end = 0, t_tail = 0, t_loop_end = 0; int n_it = 0, n_real = 0, n_replayed = 0, n_tail = 0;
#define RP_T0 const long long rp0 = wall_clock64()
#define RP_ADD(acc) acc += wall_clock64() - rp0
#else
#define RP_T0
#define RP_ADD(acc)
#endif
    double good = -1;                                                // rate 0: every pass
    bool converged = false;
    if (rate > 0.0f) {
        const long long maxlen = (long long) ceil((double) rate);
        double lo = mn, hi = mx, thresh = 0, stable = 0, prev = -1;
        bool lo_seen = false, hi_seen = false;
        int bytes_lo = 0, bytes_hi = 0;
        // The bisection starts from the same bracket [mn, mx] at every call for this frame, so the threshold of a step
        // is a function of the fits / does-not-fit outcomes before it.  The sizes found are kept across calls in a
        // binary trie over those outcomes (reset by the analysis): a call replays its steps from the trie for as long
        // as its budget leads it along recorded ones - the first ~25 always (everything fits while the threshold is
        // far above the slopes that matter; they are the expensive steps, no code-block is settled yet), and most of
        // the way for the later probes of a rate search, whose budgets differ little - and computes from there.
        //   node = {bytes, child after "does not fit", child after "fits"}; node 0 = the first step
        int *const trie = path_bytes + (size_t) frame * kRateTrieNodes * 3;
        const int known = path_n[frame];
        int n_nodes = known;
        int cur = known > 0 ? 0 : -1, parent = -1, pdir = 0;         // the node of this step (-1: not recorded yet) and its link
        bool lo_from_rec = false, hi_from_rec = false;
        for (int i = 0; i < 128; i++) {
            thresh = (lo + hi) / 2;
            if (i > 0 && thresh == prev) break;                      // the remaining iterations would repeat this one
            prev = thresh;
            if (cur >= 0) {
#ifdef EBCC_RATE_PROFILE
                n_replayed++;
#endif
                const int bytes = trie[3 * cur];
                const int f = (long long) bytes <= maxlen;
                if (f) { hi = thresh; stable = thresh; bytes_hi = bytes; hi_from_rec = true; hi_seen = false; }
                else { lo = thresh; bytes_lo = bytes; lo_from_rec = true; lo_seen = false; }
                parent = cur; pdir = f;
                const int nxt = trie[3 * cur + 1 + f];               // (written by earlier launches only: nodes made in
                cur = nxt > 0 ? nxt : -1;                            //  this one are never followed, see below)
                continue;
            }
#ifdef EBCC_RATE_PROFILE
            if (!t_replay_end) t_replay_end = wall_clock64();
#endif
            // the walks at bracket ends that came from the record (needed to tell settled code-blocks)
            if (hi_from_rec) { make_layer<kRateThreads>(g, L, totalpasses, pt, gid0, hi, lane, p_hi); hi_seen = true; hi_from_rec = false; }
            if (lo_from_rec) { make_layer<kRateThreads>(g, L, totalpasses, pt, gid0, lo, lane, p_lo); lo_seen = true; lo_from_rec = false; }
            { RP_T0; make_layer<kRateThreads>(g, L, totalpasses, pt, gid0, thresh, lane, p_cur, frozen); RP_ADD(t_ml); }
            if (lane == 0) s_nun = 0;       // (after the barrier make_layer ends with: every thread is done with the previous
                                            //  step's value; before the barriers below: ahead of this step's additions)
            // late iterations alternate between the assignments of the two bracket ends: their sizes are known
            int at_lo = lo_seen, at_hi = hi_seen;
            for (int b = lane; b < g.nblocks; b += kRateThreads) {
                if (frozen[b]) continue;
                if (p_cur[b] != p_lo[b] || p_cur[g.nblocks + b] != p_lo[g.nblocks + b]) at_lo = 0;
                if (p_cur[b] != p_hi[b] || p_cur[g.nblocks + b] != p_hi[g.nblocks + b]) at_hi = 0;
            }
            at_lo = __syncthreads_and(at_lo);
            at_hi = __syncthreads_and(at_hi);
            int bytes;
            if (at_lo) bytes = bytes_lo;
            else if (at_hi) bytes = bytes_hi;
            else { RP_T0; bytes = sized(); RP_ADD(t_sz); }
#ifdef EBCC_RATE_PROFILE
            n_it++; n_real += (at_lo || at_hi) ? 0 : s_changed;
#endif
            const bool fits = (long long) bytes <= maxlen;
            if (record && n_nodes < kRateTrieNodes && (parent >= 0 || n_nodes == 0)) {   // record this step and hang it under the previous one
                if (lane == 0) {
                    trie[3 * n_nodes] = bytes; trie[3 * n_nodes + 1] = 0; trie[3 * n_nodes + 2] = 0;
                    if (parent >= 0) trie[3 * parent + 1 + pdir] = n_nodes;
                    path_n[frame] = n_nodes + 1;
                }
                parent = n_nodes++; pdir = fits;
            } else {
                parent = -1;                                         // trie full: stop recording
            }
            if (fits) bytes_hi = bytes; else bytes_lo = bytes;
            if (fits) { hi = thresh; stable = thresh; hi_seen = true; } else { lo = thresh; lo_seen = true; }
            // A pass is taken iff fl(thresh - slope) < DBL_EPSILON, which is monotone in thresh: a code-block whose
            // greedy walks at the two ends of the bracket are the same walk takes that walk at every threshold in
            // between, i.e. for the rest of the bisection (frozen: make_layer skips it).  When that holds for all
            // of them, all remaining iterations see this assignment and this size, and whichever end they settle on
            // (`stable` = hi when it does not fit, the limit point when it does) the layer is the current one.
            unsigned long long *end = fits ? p_hi : p_lo;
            int same = lo_seen && hi_seen;
            for (int b = lane; b < g.nblocks; b += kRateThreads) {
                if (frozen[b]) continue;
                end[b] = p_cur[b]; end[g.nblocks + b] = p_cur[g.nblocks + b];
                if (lo_seen && hi_seen && p_lo[b] == p_hi[b] && p_lo[g.nblocks + b] == p_hi[g.nblocks + b]) frozen[b] = 1;
                else {
                    same = 0;
                    const int k = atomicAdd(&s_nun, 1);              // the code-blocks still open, for the tail below
                    if (k < 64) s_list[k] = b;
                }
            }
            if (__syncthreads_and(same)) { converged = true; break; }
            // Tail: once both ends of the bracket are known and at most 64 code-blocks are still open, the steps that
            // land on one of the two end assignments again (nearly all of the ~45 that remain: the bisection is
            // closing in on one slope) need no barrier and no sizing - one wave walks the open code-blocks, a lane
            // each, until the thresholds repeat or a step produces a third assignment, which goes back through the
            // block-wide path above.
            if (lo_seen && hi_seen && s_nun <= 64) {
                RP_T0;
                if (lane < 64) {
                    const int nb = g.nblocks, b = lane < s_nun ? s_list[lane] : -1;
                    double tlo = lo, thi = hi, tprev = prev, tstable = stable, tthresh = thresh;
                    int ti = i, tn = n_nodes, tparent = parent, tpdir = pdir, status = 0;
                    // The two end walks of an open code-block agree on the passes below the first one they differ in, and
                    // so does the walk at every threshold in between (the monotonicity argument above, applied to the
                    // code-block cut off at that pass): a step only has to walk from there - the last pass or two of
                    // ~40 - with the common part's mask and its last taken pass as the start state.
                    int from = 0, n_in = 0;
                    unsigned long long c0 = 0, c1 = 0;
                    if (b >= 0) {
                        const unsigned long long d0 = p_lo[b] ^ p_hi[b], d1 = p_lo[nb + b] ^ p_hi[nb + b];
                        from = d0 ? __builtin_ctzll(d0) : (d1 ? 64 + __builtin_ctzll(d1) : 0);
                        if (from > 0) {
                            c0 = from >= 64 ? p_lo[b] : p_lo[b] & ((1ull << from) - 1ull);
                            c1 = from > 64 ? p_lo[nb + b] & ((1ull << (from - 64)) - 1ull) : 0ull;
                            n_in = c1 ? 128 - __builtin_clzll(c1) : (c0 ? 64 - __builtin_clzll(c0) : 0);   // (1 + index of the last pass taken)
                        }
                    }
                    // ... and which thresholds give exactly the walk of the lower end, which exactly that of the upper
                    // end, is worked out once per code-block (walk_interval): a step is then two comparisons per lane.
                    double a_lo = -DBL_MAX, b_lo = DBL_MAX, a_hi = -DBL_MAX, b_hi = DBL_MAX;
                    if (b >= 0) {
                        if (pt.lds) {
                            walk_interval<true>(pt, totalpasses, gid0, b, p_lo[b], p_lo[nb + b], from, n_in, tlo, thi, a_lo, b_lo);
                            walk_interval<true>(pt, totalpasses, gid0, b, p_hi[b], p_hi[nb + b], from, n_in, tlo, thi, a_hi, b_hi);
                        } else {
                            walk_interval<false>(pt, totalpasses, gid0, b, p_lo[b], p_lo[nb + b], from, n_in, tlo, thi, a_lo, b_lo);
                            walk_interval<false>(pt, totalpasses, gid0, b, p_hi[b], p_hi[nb + b], from, n_in, tlo, thi, a_hi, b_hi);
                        }
                    }
                    (void) c0; (void) c1;
                    for (int j = i + 1; j < 128; j++) {
                        const double t = (tlo + thi) / 2;
                        if (t == tprev) { tthresh = t; break; }      // the bisection has ended
                        const int all_lo = __all(b < 0 || (t >= a_lo && t < b_lo));
                        const int all_hi = __all(b < 0 || (t >= a_hi && t < b_hi));
                        if (!all_lo && !all_hi) { status = 1; break; }   // a third assignment: not committed here
                        tthresh = t; tprev = t; ti = j;
                        const int bytes = all_lo ? bytes_lo : bytes_hi;
                        const bool fits = (long long) bytes <= maxlen;
                        if (record && tn < kRateTrieNodes && (tparent >= 0 || tn == 0)) {
                            if (lane == 0) {
                                trie[3 * tn] = bytes; trie[3 * tn + 1] = 0; trie[3 * tn + 2] = 0;
                                if (tparent >= 0) trie[3 * tparent + 1 + tpdir] = tn;
                                path_n[frame] = tn + 1;
                            }
                            tparent = tn++; tpdir = fits;
                        } else {
                            tparent = -1;
                        }
                        if (fits) { thi = t; tstable = t; } else { tlo = t; }
                    }
                    if (lane == 0) {
                        s_td[0] = tlo; s_td[1] = thi; s_td[2] = tprev; s_td[3] = tstable; s_td[4] = tthresh;
                        s_ti[0] = ti; s_ti[1] = tn; s_ti[2] = tparent; s_ti[3] = tpdir; s_ti[4] = status;
                    }
                }
                __syncthreads();
                lo = s_td[0]; hi = s_td[1]; prev = s_td[2]; stable = s_td[3]; thresh = s_td[4];
                i = s_ti[0]; n_nodes = s_ti[1]; parent = s_ti[2]; pdir = s_ti[3];
                const int status = s_ti[4];
                __syncthreads();                                     // (s_td / s_ti are written again on the next visit)
                RP_ADD(t_tail);
#ifdef EBCC_RATE_PROFILE
                n_tail++;
#endif
                if (status == 0) break;                              // ended in the tail: thresh, stable as the loop leaves them
            }
        }
        good = stable == 0 ? thresh : stable;
    }
#ifdef EBCC_RATE_PROFILE
    t_loop_end = wall_clock64();
#endif
    if (!converged) make_layer<kRateThreads>(g, L, totalpasses, pt, gid0, good, lane);
    const int body = sized();
    for (int b = lane; b < g.nblocks; b += kRateThreads) npass_out[gid0 + b] = L.npass[b];
    if (lane == 0 && candidate) {
        int *o = cand_out + (size_t) (2 * frame + slot) * 3;
        o[0] = body; o[1] = kMainHeaderBytes + 12 + 2 + body + 2; o[2] = (int) (rate > 0.0f ? ceil((double) rate) : 0);
    }
    if (lane == 0 && !candidate) {
        jf[frame].body_bytes = body;
        jf[frame].stream_bytes = kMainHeaderBytes + 12 + 2 + body + 2;
        jf[frame].maxlen = (int) (rate > 0.0f ? ceil((double) rate) : 0);
#ifdef EBCC_RATE_PROFILE
        if (frame == 0) printf("k_rate frame0 setup: entry %lld, trees %lld, offsets %lld, tables %lld us\n", t_s[0] / 100, t_s[1] / 100, t_s[2] / 100, t_s[3] / 100);
        if (frame == 0) printf("k_rate frame0: setup %lld us, %d replayed steps until %lld us, make_layer %lld us, sized %lld us, computed iterations %d (real sizings %d), tail %lld us in %d visits, loop ends at %lld us, total %lld us\n",
                               t_setup / 100, n_replayed, t_replay_end ? (t_replay_end - t_start) / 100 : -1, t_ml / 100, t_sz / 100, n_it, n_real, t_tail / 100, n_tail,
                               (t_loop_end - t_start) / 100, (wall_clock64() - t_start) / 100);
#endif
    }
}

// ================================================================================================
// codestream writer: main header (A.5.1, A.6.1, A.6.4, A.9.2), SOT/SOD, packets, EOC
// ================================================================================================
__device__ inline void put16(uint8_t *&p, unsigned v) { *p++ = (uint8_t) (v >> 8); *p++ = (uint8_t) v; }
__device__ inline void put32(uint8_t *&p, unsigned v) { put16(p, v >> 16); put16(p, v & 0xFFFFu); }

__global__ __launch_bounds__(kWriteThreads) void k_write(const int *__restrict__ numbps, const int *__restrict__ rates,
                                                         const int *__restrict__ npass_in, const uint8_t *__restrict__ cblk_bytes,
                                                         uint8_t *__restrict__ stream, size_t stream_cap, const J2kGeom *geom,
                                                         J2kFrame *jf, const FrameState *fs, const int *active)
{
    extern __shared__ unsigned char lds_raw[];
    __shared__ int s_hdr[kJ2kRes], s_body[kJ2kRes], s_off[kJ2kRes + 1];
    const int frame = blockIdx.x, lane = threadIdx.x;
    if ((active && !active[frame]) || fs[frame].const_field) return;
    const J2kGeom &g = j2k_frame_geom(geom, frame);
    const int gid0 = frame * g.stride;
    RateLds L;
    L.carve(lds_raw, g.nblocks, g.tree_nodes);
    trees_static<kWriteThreads>(g, L, numbps, gid0, lane);
    for (int b = lane; b < g.nblocks; b += kWriteThreads) L.npass[b] = (short) npass_in[gid0 + b];
    __syncthreads();
    uint8_t *base = stream + (size_t) frame * stream_cap;

    // Packet headers: the closed-form header bits of layer_bytes_fast (the evaluation k_rate sizes layers with), then
    // the bit stuffing of B.10.1 - a byte that follows 0xFF carries 7 bits - as one short serial walk per resolution.
    __shared__ int s_sum, s_T[kJ2kRes];
    layer_bytes_fast<kWriteThreads>(g, L, rates, gid0, lane, &s_sum, s_hdr);
    if (lane < kJ2kRes) {
        s_body[lane] = 0;
        s_T[lane] = 1 + L.leafbits[g.res_first[lane + 1]] - L.leafbits[g.res_first[lane]];
    }
    __syncthreads();
    // segment lengths: per resolution totals, and (below) an exclusive scan in packet order for the destinations
    for (int b = lane; b < g.nblocks; b += kWriteThreads) {
        const int n = L.npass[b];
        if (!n) continue;
        int bi = 0;
        while (bi + 1 < g.nbands && g.bands[bi + 1].first_block <= b) bi++;
        atomicAdd(&s_body[g.bands[bi].res], rates[(size_t) (gid0 + b) * kJ2kMaxPasses + n - 1]);
    }
    __syncthreads();
    if (lane == 0) {
        int off = kMainHeaderBytes + 14;
        for (int r = 0; r < kJ2kRes; r++) { s_off[r] = off; off += s_hdr[r] + s_body[r]; }
        s_off[kJ2kRes] = off;
    }
    __syncthreads();
    if (lane < kJ2kRes) {
        const int region = g.res_first[lane] * 128, T = s_T[lane];
        uint8_t *o = base + s_off[lane];
        int p = 0, nb = 0;
        unsigned int last = 0;
        while (p < T) {                                             // (bits past T are zero: the flush padding)
            const int take = last == 0xFFu ? 7 : 8;
            last = raw_byte(L.raw, region + p) >> (8 - take);
            o[nb++] = (uint8_t) last;
            p += take;
        }
        if (last == 0xFFu) o[nb++] = 0;                             // Bio::flush after a 0xFF byte
    }
    __syncthreads();
    {
        __shared__ int s_wave[kWriteThreads / 64];
        int running = 0;
        for (int base_b = 0; base_b < g.nblocks; base_b += kWriteThreads) {
            const int b = base_b + lane;
            int v = 0;
            if (b < g.nblocks && L.npass[b]) v = rates[(size_t) (gid0 + b) * kJ2kMaxPasses + L.npass[b] - 1];
            int x = v;
            for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if ((lane & 63) >= d) x += y; }
            if ((lane & 63) == 63) s_wave[lane >> 6] = x;
            __syncthreads();
            int before = running, tot = 0;
            for (int w = 0; w < kWriteThreads / 64; w++) { if (w < (lane >> 6)) before += s_wave[w]; tot += s_wave[w]; }
            if (b < g.nblocks) L.leafbits[b] = before + x - v;      // bytes of the segments before this one, all resolutions
            running += tot;
            __syncthreads();
        }
    }
    {
        const int wave = lane >> 6, wl = lane & 63;
        for (int blk = wave; blk < g.nblocks; blk += kWriteThreads / 64) {           // one wave per code-block segment
            const int n = L.npass[blk];
            if (!n) continue;
            int bi = 0;
            while (bi + 1 < g.nbands && g.bands[bi + 1].first_block <= blk) bi++;
            const int res = g.bands[bi].res;
            const int seglen = rates[(size_t) (gid0 + blk) * kJ2kMaxPasses + n - 1];
            const uint8_t *src = cblk_bytes + (size_t) (gid0 + blk) * kJ2kCblkBytes;
            uint8_t *dst = base + s_off[res] + s_hdr[res] + (L.leafbits[blk] - L.leafbits[g.res_first[res]]);
            for (int i = wl; i < seglen; i += 64) dst[i] = src[i];
        }
    }
    if (lane == 0) {
        uint8_t *p = base;
        put16(p, 0xFF4F);
        put16(p, 0xFF51); put16(p, 41); put16(p, 0);
        put32(p, g.W); put32(p, g.H); put32(p, 0); put32(p, 0); put32(p, g.W); put32(p, g.H); put32(p, 0); put32(p, 0);
        put16(p, 1); *p++ = 15; *p++ = 1; *p++ = 1;
        put16(p, 0xFF52); put16(p, 12); *p++ = 0; *p++ = 0; put16(p, 1); *p++ = 0; *p++ = kJ2kRes - 1; *p++ = 4; *p++ = 4; *p++ = 0; *p++ = 0;
        put16(p, 0xFF5C); put16(p, 3 + 2 * kJ2kBands); *p++ = (uint8_t) (2 + (2 << 5));
        for (int bi = 0; bi < kJ2kBands; bi++) put16(p, (unsigned) ((g.bands[bi].expn << 11) | g.bands[bi].mant));
        const char com[] = "Created by OpenJPEG version 2.4.0";
        put16(p, 0xFF64); put16(p, 4 + 33); put16(p, 1);
        for (int i = 0; i < 33; i++) *p++ = (uint8_t) com[i];
        const int body = s_off[kJ2kRes] - (kMainHeaderBytes + 14);
        put16(p, 0xFF90); put16(p, 10); put16(p, 0); put32(p, (unsigned) (12 + 2 + body)); *p++ = 0; *p++ = 1;
        put16(p, 0xFF93);
        uint8_t *e = base + s_off[kJ2kRes];
        put16(e, 0xFFD9);
        jf[frame].stream_bytes = s_off[kJ2kRes] + 2;
        jf[frame].body_bytes = body;
    }
}

// sums the per-tile statistics the last inverse column pass left (k_j2k_cols FIN, j2k_analysis.hip)
__global__ void k_finish_reduce(const double *partial, const unsigned long long *partial_u, J2kFrame *jf, int n, int n_partials,
                                const FrameState *fs, const int *active)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n || (active && !active[f]) || fs[f].const_field) return;
    jf[f].bad_seen = 0;                                                 // (the early-exit counter of the probe just finished)
    double s = 0;
    unsigned long long b = 0;
    for (int i = 0; i < n_partials; i++) { s += partial[(size_t) f * kPartials + i]; b += partial_u[(size_t) f * kPartials + i]; }
    jf[f].err_sum = s;
    jf[f].nbad = b;
}


// MQ state table in LDS (see t1_core.hpp ConstTable): filled by the first 47 lanes of the workgroup
struct LdsTable {
    const __attribute__((address_space(3))) uint32_t *t;
    __device__ uint32_t operator()(int i) const { return t[i]; }
};
#define EBCC_LDS_MQ_TABLE(name)                                                              \
    __shared__ uint32_t name##_store[48];                                                    \
    if (threadIdx.x < 47) name##_store[threadIdx.x] = t1::mq_entry((int) threadIdx.x);       \
    __syncthreads();                                                                         \
    LdsTable name{(const __attribute__((address_space(3))) uint32_t *) name##_store}

// ================================================================================================
// true decode: tier-1 MQ decoding, one code-block per lane, values scattered into V (half units)
// ================================================================================================
// context tables of the serial passes (t1_device.hpp: LdsCtx): in device memory once per device, copied to LDS by every workgroup
__device__ __attribute__((aligned(16))) uint8_t g_ctx_tables[LdsCtx::kBytes];
void ensure_ctx_tables()
{
    static std::once_flag once[64];
    int dev = 0;
    EBCC_HIP_CHECK(hipGetDevice(&dev));
    std::call_once(once[dev & 63], [] {
        uint8_t t[LdsCtx::kBytes];
        make_ctx_tables(t);
        EBCC_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_ctx_tables), t, sizeof t));
    });
}

struct DecStore {
    unsigned char *st; // group base of the state words (uniform: a wave stays inside one group of 64 code-blocks)
    uint32_t lane8;    // lane * 8
    int32_t *v;        // tile-buffer position of the block's (0,0)
    int W;
    __device__ unsigned long long &at(int row) const { return *(unsigned long long *) (st + ((uint32_t) row * 512u + lane8)); }
    __device__ unsigned long long &S(int y) { return at(y + 1); }
    __device__ unsigned long long &NEG(int y) { return at(66 + y); }
    __device__ unsigned long long &VIS(int y) { return at(130 + y); }
    __device__ unsigned long long &REF(int y) { return at(194 + y); }
    __device__ void set_sig(int x, int y, int neg, int plane)
    {
        int one = 1 << (plane + 1), val = one | (one >> 1);
        v[y * W + x] = neg ? -val : val;
    }
    __device__ void refine(int x, int y, int bit, int plane, int neg)
    {
        // the sign is known from the NEG row mask, so the update is a fire-and-forget atomic add (no load stall)
        const int half = 1 << plane;
        atomicAdd(&v[y * W + x], (bit ^ neg) ? half : -half);
    }
};
struct DecSrc {
    // byte source with an 8-byte register window (slots and stream slots are 8-byte aligned and padded)
    const uint8_t *p; int n;
    unsigned long long win = 0; int base = -16;
    __device__ uint32_t get(int i)
    {
        if (i >= n) return 0xFFu;
        const int b = i & ~7;
        if (b != base) {
            const uintptr_t a = (uintptr_t) (p + b);
            if ((a & 7) == 0) win = *reinterpret_cast<const unsigned long long *>(p + b);
            else { win = 0; for (int k = 0; k < 8; k++) win |= (unsigned long long) p[b + k] << (8 * k); }
            base = b;
        }
        return (uint32_t) (win >> (8 * (i & 7))) & 0xFFu;
    }
};

// Decodes the segments located by the host-side packet parser (dec_table): offset, length, numbps, passes.
// (Rate probes do not come through here: they restart from checkpoints, k_t1_resume below.)
// ---- the decoder with its state next to the SIMD: the four row-mask arrays of a code-block (S, NEG, VIS, REF: 258
// rows of 8 bytes) live in LDS, not in jb.T1S, and the segment's bytes come through a register window that is one
// 16-byte chunk ahead of the decoder.  A stripe of a coding pass starts with 16-20 row loads and ends with 12 row stores:
// from HBM/L2 every one of the ~500 stripe-passes of a code-block cost a round trip (and a load also waits for the value
// scatter before it - loads and stores share the counter); the decoder was waiting, not computing (30% of its VALU slots).
// The only global accesses left in the decision loops are the fire-and-forget value scatter and the byte prefetch.
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
struct DecStoreLds {
    uint32_t base;     // LDS byte address of this lane's row 0
    uint32_t stride;   // bytes between rows (8 * lanes per wave)
    int32_t *v;        // tile-buffer position of the block's (0,0)
    int W;
    __device__ lds_u64 &at(int row) const { return *(lds_u64 *) (uintptr_t) (base + (uint32_t) row * stride); }
    __device__ lds_u64 &S(int y) { return at(y + 1); }
    __device__ lds_u64 &NEG(int y) { return at(66 + y); }
    __device__ lds_u64 &VIS(int y) { return at(130 + y); }
    __device__ lds_u64 &REF(int y) { return at(194 + y); }
    __device__ void set_sig(int x, int y, int neg, int plane)
    {
        int one = 1 << (plane + 1), val = one | (one >> 1);
        v[y * W + x] = neg ? -val : val;
    }
    __device__ void refine(int x, int y, int bit, int plane, int neg)
    {
        const int half = 1 << plane;
        atomicAdd(&v[y * W + x], (bit ^ neg) ? half : -half);
    }
};
constexpr int kDecStateRows = 258;
struct DecSrcAhead {
    // 16-byte aligned chunks of the stream slot: `cur` holds chunk k, `nxt` chunk k + 1 (requested when the decoder
    // entered chunk k, ~100 decisions before its first byte is wanted).  Chunks without a byte of the segment are not read.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint8_t *base;           // 16-byte aligned address at or before the segment's first byte
    int a0, n, k = -2;             // offset of the first byte inside its chunk; segment bytes; chunk in `cur`
    u32x4 cur = {0, 0, 0, 0}, nxt = {0, 0, 0, 0};
    __device__ DecSrcAhead(const uint8_t *p, int len) : base((const uint8_t *) ((uintptr_t) p & ~(uintptr_t) 15)), a0((int) ((uintptr_t) p & 15)), n(len) {}
    __device__ u32x4 chunk(int c) const
    {
        if (c * 16 >= a0 + n) return u32x4{0, 0, 0, 0};
        return *(const __attribute__((address_space(1))) u32x4 *) (uintptr_t) (base + (size_t) c * 16);    // (a global, not a flat, load)
    }
    __device__ uint32_t get(int i)
    {
        if (i >= n) return 0xFFu;
        const int o = i + a0, c = o >> 4;
        if (c != k) {
            cur = c == k + 1 ? nxt : chunk(c);
            nxt = chunk(c + 1);
            k = c;
        }
        const uint32_t w = (o & 8) ? ((o & 4) ? cur.w : cur.z) : ((o & 4) ? cur.y : cur.x);
        return (w >> (8 * (o & 3))) & 0xFFu;
    }
};

// The same source for a decoder that walks the segment from its first byte (the true decode): BYTEIN looks at the byte it
// stands on and the one after it - both come out of two dword registers with one v_alignbyte - and steps forward by at most
// one; every fourth step the next dword of the 16-byte chunk registers moves in (bytes past the segment's end read as 0xFF).
// get() cost ~40 vector instructions per byte and BYTEIN asked twice: a third of the decoder's instructions.
struct DecSrcSeq {
    static constexpr bool kSequential = true;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint8_t *base;           // 16-byte aligned address at or before the segment's first byte
    int end;                       // offset from `base` one past the segment's last byte
    int o;                         // offset of the byte the decoder stands on
    uint32_t lo, hi;               // the dword that holds byte o, and the one after it
    u32x4 cur;                     // the chunk that holds `hi`'s dword (no chunk is requested ahead: the launch is bound by
                                   // vector issue, other waves cover the load, and four registers fewer end the spills)
    __device__ u32x4 chunk(int c) const
    {
        if (c * 16 >= end) return u32x4{0, 0, 0, 0};
        return *(const __attribute__((address_space(1))) u32x4 *) (uintptr_t) (base + (size_t) c * 16);
    }
    __device__ uint32_t dword(int off)                                   // off: a multiple of 4, asked for in rising order
    {
        if ((off & 15) == 0 && off > 0) cur = chunk(off >> 4);
        const int j = (off >> 2) & 3;
        uint32_t w = j & 2 ? (j & 1 ? cur.w : cur.z) : (j & 1 ? cur.y : cur.x);
        const int rem = end - off;
        if (rem < 4) w |= rem <= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * rem);
        return w;
    }
    __device__ DecSrcSeq(const uint8_t *p, int len)
        : base((const uint8_t *) ((uintptr_t) p & ~(uintptr_t) 15)), end((int) ((uintptr_t) p & 15) + len), o((int) ((uintptr_t) p & 15))
    {
        cur = chunk(0);
        lo = dword(o & ~3);
        hi = dword((o & ~3) + 4);
    }
    __device__ uint32_t pair() const { return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t) o & 3u); }   // bytes o, o + 1, .. from bit 0 up
    __device__ uint32_t at0() const { return pair() & 0xFFu; }
    __device__ uint32_t at1() const { return (pair() >> 8) & 0xFFu; }
    __device__ void step()
    {
        o++;
        if ((o & 3) == 0) { lo = hi; hi = dword(o + 4); }
    }
    __device__ uint32_t get(int) const { return 0xFFu; }                  // (not used by a sequential decoder)
};

// Longest first: a code-block is one serial chain (a large one ~10 ms of dependent instructions), so the launch ends
// when the last large block does.  The blocks are put in order of falling segment length (a counting sort over 64
// length classes, k_dec_hist / k_dec_offsets / k_dec_place) and the waves take them in that order: the long chains start
// at once, the short ones fill in behind them, and the lanes of a wave hold blocks of similar length.
constexpr int kDecClasses = 64;
constexpr int kDecTierDen0 = 0, kDecTierDen1 = 64, kDecTierDen2 = 1;     // tiers of k_t1_decode_lds (launch_j2k_decode): the longest 1/64 at 2 lanes, the rest at 4 (round 4, three alternating runs: 16.5-16.7 ms against 17.0-17.3 with 1/32)
__device__ inline int dec_class(const int *e) { return e[3] <= 0 || e[2] <= 0 ? 0 : min(kDecClasses - 1, 1 + (e[1] >> 6)); }
__global__ __launch_bounds__(256) void k_dec_hist(const int *dec_table, int *counters, int total)
{
    __shared__ int h[kDecClasses];                                      // (a histogram per workgroup: 64 global counters cannot take 10^5 atomics)
    if (threadIdx.x < kDecClasses) h[threadIdx.x] = 0;
    __syncthreads();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < total) atomicAdd(&h[dec_class(dec_table + (size_t) gid * 4)], 1);
    __syncthreads();
    if (threadIdx.x < kDecClasses && h[threadIdx.x]) atomicAdd(&counters[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_dec_offsets(int *counters)                         // [0,64) counts -> [64,128) first slot of every class
{
    const int c = threadIdx.x;
    int before = 0;
    for (int k = kDecClasses - 1; k > c; k--) before += counters[k];
    counters[kDecClasses + c] = before;
}
__global__ __launch_bounds__(256) void k_dec_place(const int *dec_table, int *counters, int *order, int total)
{
    __shared__ int h[kDecClasses], base[kDecClasses];
    if (threadIdx.x < kDecClasses) h[threadIdx.x] = 0;
    __syncthreads();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int cls = 0, rank = 0;
    if (gid < total) { cls = dec_class(dec_table + (size_t) gid * 4); rank = atomicAdd(&h[cls], 1); }
    __syncthreads();
    if (threadIdx.x < kDecClasses && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&counters[kDecClasses + threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (gid < total) order[base[cls] + rank] = gid;
}

// Lanes per wave by rank in the order: the first n[0] code-blocks (the longest chains) go lanes[0] to a wave, the next n[1]
// lanes[1] to a wave, the next n[2] lanes[2], the rest lanes[3] (every n[i] a multiple of lanes[i]).
struct DecTiers { int n[3]; int lanes[4]; };
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_t1_decode_lds(const uint8_t *bytes, size_t stream_cap, const int *dec_table, const int *order, int32_t *V,
                                                       const J2kGeom *geom, const J2kBlock *blocks, const FrameState *fs, int total, DecTiers tiers)
{
    extern __shared__ unsigned long long dec_state[];                  // [kDecStateRows][lanes of this wave]
    __shared__ __attribute__((aligned(16))) uint8_t ctx_store[LdsCtx::kBytes];
    copy_ctx_tables(ctx_store, g_ctx_tables, (int) threadIdx.x, 64);   // (EBCC_LDS_MQ_TABLE synchronises)
    EBCC_LDS_MQ_TABLE(tab);
    const LdsCtx ctxp{(uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) ctx_store, 0u};
    // the kernel is bound by vector issue slots, diverged lanes share most of their instructions (4 lanes: 1.8x fewer wave
    // instructions than 2), but a wave lasts as long as its lanes together: the longer a code-block's segment, the fewer
    // lanes its wave has (tiers by rank in the longest-first order), so that every wave carries about the same work
    int w = (int) blockIdx.x, slot0 = 0, lanes = tiers.lanes[3], end = total;
    {
        int t = 0;
        for (; t < 3; t++) {
            const int waves = tiers.n[t] / tiers.lanes[t];
            if (w < waves) { lanes = tiers.lanes[t]; end = slot0 + tiers.n[t]; break; }
            w -= waves; slot0 += tiers.n[t];
        }
    }
    if ((int) threadIdx.x >= lanes) return;                            // see t1_lanes_per_wave()
    const int slot = slot0 + w * lanes + (int) threadIdx.x;
    if (slot >= end || slot >= total) return;
    const int gid = order[slot];
    const int nb = geom->stride;
    const int frame = gid / nb, bi = gid - frame * nb;
    if (fs[frame].const_field) return;
    blocks = j2k_frame_blocks(geom, blocks, frame);
    geom = &j2k_frame_geom(geom, frame);
    const int *e = dec_table + (size_t) gid * 4;
    const int len = e[1], P = e[2], np = e[3];
    const uint8_t *src = bytes + (size_t) frame * stream_cap + e[0];
    if (np <= 0 || P <= 0) return;
    const J2kBlock blk = blocks[bi];
    DecStoreLds st{(uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned long long *) dec_state + threadIdx.x * 8u, (uint32_t) lanes * 8u,
                   V + (size_t) frame * geom->W * geom->H + (size_t) blk.y * geom->W + blk.x, geom->W};
    for (int r = 0; r < kDecStateRows; r++) st.at(r) = 0ull;
    t1::decode_block(st, DecSrcSeq(src, len), blk.w, blk.h, geom->bands[blk.band].orient, P, np, tab, ctxp);
}


// ================================================================================================
// rate-probe decode, restarted at the last coding pass the layer keeps
//   k_probe_plan : per code-block, the restart pass r (last kept pass, or an earlier one if the checkpoint there
//                  was taken after the decoder had already consumed a byte at/after the truncation point)
//   k_probe_init : per sample, the value the decoder holds at the start of pass r (passes < r are complete, so
//                  it follows from the quantised coefficient and the encoder's pass bookkeeping) - replaces
//                  zero-filling V
//   k_t1_resume  : one code-block per lane: state masks from the bit-plane / suffix-OR / visited masks, MQ
//                  registers from the checkpoint, then only passes r .. n-1 on the truncated bytes
// ================================================================================================
// EBCC_HIP_T1_STATS=1: which code-blocks a probe changes.  Per round of a search: the frames probed, by the coarsest
// resolution in which one of their code-blocks has other passes than in the frame's previous probe (0 = LL .. 5 = the finest
// detail bands; 6 = nothing changed), and the number of changed finest-level code-blocks of the frames in class 5.
constexpr int kProbeHistRounds = 96, kProbeHistFrames = 4096;
__device__ unsigned int g_probe_hist[kProbeHistRounds][8];
__device__ unsigned int g_probe_round;
__device__ int g_probe_minres[kProbeHistFrames];
__device__ unsigned int g_probe_fine_changed[kProbeHistFrames];
__device__ unsigned int g_probe_tiles[kProbeHistFrames][4];             // 12 x 8 tiles of 120 columns x H/8 rows a changed code-block reaches
__device__ unsigned long long g_probe_dirty[kProbeHistRounds];          // their number, summed over the round's frames
__global__ void k_probe_hist_round(int n_frames, const FrameState *fs, const int *active)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int round = min(g_probe_round, (unsigned int) kProbeHistRounds - 1);
    if (f < n_frames && f < kProbeHistFrames && !((active && !active[f]) || fs[f].const_field)) {
        const int m = g_probe_minres[f];
        atomicAdd(&g_probe_hist[round][m > 5 ? 6 : m], 1u);
        if (m == 5) atomicAdd(&g_probe_hist[round][7], g_probe_fine_changed[f]);
        atomicAdd(&g_probe_dirty[round], (unsigned long long) (__popc(g_probe_tiles[f][0]) + __popc(g_probe_tiles[f][1]) + __popc(g_probe_tiles[f][2])));
    }
    if (f < kProbeHistFrames) { g_probe_minres[f] = 99; g_probe_fine_changed[f] = 0; g_probe_tiles[f][0] = g_probe_tiles[f][1] = g_probe_tiles[f][2] = 0; }
}
__global__ void k_probe_hist_next() { g_probe_round++; }

__global__ void k_probe_plan(const int *__restrict__ numbps, const int *__restrict__ npass, const int *__restrict__ rates,
                             void *ckpt, int *__restrict__ rpass, int *__restrict__ lastnp, const J2kGeom *geom,
                             const J2kBlock *blocks, const FrameState *fs, const int *active, int total, int stats)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int frame = gid / geom->stride;
    if ((active && !active[frame]) || fs[frame].const_field) return;
    const int n = npass[gid], P = numbps[gid];
    // a code-block that keeps the passes it had in the frame's previous probe keeps its decoded values: V still
    // holds them (later bisection steps move few code-blocks)
    if (lastnp[gid] == n) { rpass[gid] = -2; return; }
    if (stats && frame < kProbeHistFrames && lastnp[gid] != -1) {       // (the first probe of a frame changes everything: not counted)
        const int res = j2k_frame_geom(geom, frame).bands[j2k_frame_blocks(geom, blocks, frame)[gid - frame * geom->stride].band].res;
        atomicMin(&g_probe_minres[frame], res);
        if (res == 5) atomicAdd(&g_probe_fine_changed[frame], 1u);
        // the samples the code-block reaches: its rectangle in the band, scaled to the frame, plus the synthesis filters'
        // reach (4 samples per level, i.e. < 5 x the scale in all)
        const J2kGeom &g = j2k_frame_geom(geom, frame);
        const J2kBlock &b = j2k_frame_blocks(geom, blocks, frame)[gid - frame * geom->stride];
        const J2kBand &bd = g.bands[b.band];
        const int sc = 1 << (res == 0 ? 5 : 6 - res), bx = b.x - bd.offx, by = b.y - bd.offy;
        const int x0 = max(0, (bx - 5) * sc), x1 = min(g.W - 1, (bx + b.w + 5) * sc), y0 = max(0, (by - 5) * sc), y1 = min(g.H - 1, (by + b.h + 5) * sc);
        const int th = (g.H + 7) / 8;
        for (int ty = y0 / th; ty <= y1 / th; ty++)
            for (int tx = x0 / 120; tx <= x1 / 120; tx++) { const int t = ty * 12 + tx; if (t < 96) atomicOr(&g_probe_tiles[frame][t >> 5], 1u << (t & 31)); }
    }
    lastnp[gid] = n;
    int plan = -1;
    if (n > 0 && P > 0) {
        const int len = rates[(size_t) gid * kJ2kMaxPasses + n - 1];
        const int nstr = (j2k_frame_blocks(geom, blocks, frame)[gid - frame * geom->stride].h + 3) >> 2;
        const J2kCkptView ck = J2kCkptView::of(ckpt, (size_t) gid);
        // latest checkpoint taken before the decoder touched a byte at/after the truncation point (pos is
        // non-decreasing in coding order); slot 0 is the initial state and always usable
        int lo = 0, hi = n * nstr - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            const int mp = mid / nstr, ms = mid - mp * nstr;
            if ((int) ck.slot((uint32_t) (mp * 16 + ms))[1] + 1 < len) lo = mid; else hi = mid - 1;
        }
        const int r = lo / nstr, st = lo - r * nstr;
        plan = r | (st << 8);
    }
    rpass[gid] = plan;
}

// One workgroup per (code-block, frame): a code-block that keeps its passes returns at once (round 2 walked every sample of
// the frame through the code-block map to find that out - 2 MB per frame and round), the others rewrite their own
// rectangle, 256-byte row segments.
__global__ __launch_bounds__(256) void k_probe_init(const int32_t *__restrict__ Q6, const int *__restrict__ rpass,
                                                     const int *__restrict__ numbps, const unsigned long long *__restrict__ SPS,
                                                     int32_t *__restrict__ V, const J2kGeom *geom, const J2kBlock *blocks,
                                                     const FrameState *fs, const int *active)
{
    const int frame = blockIdx.y, bi = blockIdx.x;
    if ((active && !active[frame]) || fs[frame].const_field) return;
    const int gid = frame * geom->stride + bi;
    const int plan = rpass[gid];
    if (plan == -2) return;                                             // unchanged code-block: V is up to date
    if (bi >= j2k_frame_geom(geom, frame).nblocks) return;              // (slots past this tile position's last code-block)
    const int W = geom->W;
    const size_t n_pix = (size_t) W * geom->H;
    const J2kBlock blk = j2k_frame_blocks(geom, blocks, frame)[bi];
    const int P = numbps[gid];
    const int x = threadIdx.x & 63;
    if (x >= blk.w) return;
    const int32_t *q = Q6 + (size_t) frame * n_pix + (size_t) blk.y * W + blk.x + x;
    int32_t *v = V + (size_t) frame * n_pix + (size_t) blk.y * W + blk.x + x;
    const unsigned long long *spsrow = SPS + ((size_t) (gid >> 6) * 64) * 64 + (gid & 63);
    for (int y = threadIdx.x >> 6; y < blk.h; y += 4) {
        const int q6 = q[(size_t) y * W];
        const unsigned int a = (unsigned int) (q6 < 0 ? -q6 : q6) >> 6;
        int out = 0;
        if (plan > 0 && a) {
            const int bs = 31 - __clz(a);
            // rows above the restart stripe have already been through pass r
            const int r = (plan & 0xFF) + (y < 4 * (plan >> 8) ? 1 : 0);
            const unsigned long long sps = spsrow[(size_t) y * 64];
            const int ps = bs == P - 1 ? 0 : 3 * (P - 1 - bs) - (((sps >> x) & 1ull) ? 2 : 0);
            if (ps < r) {
                out = 3 << bs;                                          // 1.5 * 2^bs in half units, then the refinements
                for (int pl = bs - 1; pl >= 0; pl--) {
                    if (3 * (P - 1 - pl) - 1 >= r) break;
                    out += ((a >> pl) & 1u) ? (1 << pl) : -(1 << pl);
                }
                if (q6 < 0) out = -out;
            }
        }
        v[(size_t) y * W] = out;
    }
}

__global__ __launch_bounds__(64) void k_t1_resume(unsigned long long *T1S, const unsigned long long *BP,
                                                   const unsigned long long *SUF, const unsigned long long *SGN,
                                                   const unsigned long long *SPS, const unsigned long long *VISP,
                                                   const uint8_t *cblk_bytes, const int *numbps, const int *npass,
                                                   const int *rates, const int *rpass, void *ckpt, int32_t *V,
                                                   const J2kGeom *geom, const J2kBlock *blocks, const FrameState *fs,
                                                   const int *active, int total, int lpw)
{
    extern __shared__ unsigned long long dec_state[];                  // [kDecStateRows][lpw]: the state row masks of this wave's code-blocks
    __shared__ __attribute__((aligned(16))) uint8_t ctx_store[LdsCtx::kBytes];
    EBCC_LDS_MQ_TABLE(tab);
    const int gid0 = blockIdx.x * lpw;                                 // lpw divides 64: the wave stays inside one group
    const int gid = gid0 + threadIdx.x;
    const int nb = geom->stride;
    int plan = -1, frame = 0, bi = 0;
    if ((int) threadIdx.x < lpw && gid < total) {                       // see t1_lanes_per_wave()
        frame = gid / nb; bi = gid - frame * nb;
        if (!((active && !active[frame]) || fs[frame].const_field)) plan = rpass[gid];
    }
    // most code-blocks keep their passes from one probe to the next: a workgroup (one wave) without a changed one leaves here
    if (!__any(plan >= 0)) return;
    copy_ctx_tables(ctx_store, g_ctx_tables, (int) threadIdx.x, 64);
    __syncthreads();
    if (plan < 0) return;
    const LdsCtx ctxp{(uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) ctx_store, 0u};
    blocks = j2k_frame_blocks(geom, blocks, frame);
    geom = &j2k_frame_geom(geom, frame);
    const int r = plan & 0xFF, stripe = plan >> 8;
    const int np = npass[gid], P = numbps[gid];
    const int len = rates[(size_t) gid * kJ2kMaxPasses + np - 1];
    const J2kBlock blk = blocks[bi];
    const size_t grp = (size_t) (gid0 >> 6);
    const int gl = (gid0 & 63) + threadIdx.x;
    DecStoreLds st{(uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned long long *) dec_state + threadIdx.x * 8u, (uint32_t) lpw * 8u,
                   V + (size_t) frame * geom->W * geom->H + (size_t) blk.y * geom->W + blk.x, geom->W};
    // Decoder state when pass r reaches the restart stripe: rows above it are as at the start of pass r + 1,
    // the rest as at the start of pass r.  Only the rows the remaining stripes can
    // see are needed when no further pass follows, but all 64 are cheap next to the decode.
    const unsigned long long *sg = SGN + grp * 64 * 64 + gl;
    const unsigned long long *sps = SPS + grp * 64 * 64 + gl;
    st.S(-1) = 0; st.S(64) = 0;                                         // (the guard rows: nothing else writes them on this path)
    // When no further pass follows (the usual case: the restart point lies in the last pass the layer keeps) the decoder only
    // looks at the rows from the one above the restart stripe on.  Four rows are requested together: taken one at a time every
    // row waited for its own loads (64 round trips to L2 ahead of a decode of one or two stripes).
    const int y_first = r == np - 1 ? max(0, 4 * stripe - 1) & ~3 : 0;
    for (int y0 = y_first; y0 < 64; y0 += 4) {
        unsigned long long s1[4], s2[4], bpv[4], spv[4], vv[4], sgv[4];
        int tt[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = y0 + k;
            const int q = r + (y < 4 * stripe ? 1 : 0);
            tt[k] = -1; s1[k] = s2[k] = bpv[k] = spv[k] = vv[k] = sgv[k] = 0ull;
            if (y < blk.h && q > 0) {
                const int p = t1::plane_of_pass(P, q), t = t1::type_of_pass(q);
                tt[k] = t;
                s1[k] = SUF[((grp * (kJ2kMaxPlanes + 2) + p + 1) * 64 + y) * 64 + gl];
                if (t >= 1) {
                    spv[k] = sps[(size_t) y * 64];
                    bpv[k] = BP[((grp * kJ2kMaxPlanes + p) * 64 + y) * 64 + gl];
                    vv[k] = VISP[((grp * kJ2kMaxPlanes + p) * 64 + y) * 64 + gl];
                }
                if (t != 2) s2[k] = SUF[((grp * (kJ2kMaxPlanes + 2) + p + 2) * 64 + y) * 64 + gl];
                sgv[k] = sg[(size_t) y * 64];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = y0 + k;
            unsigned long long S = 0, R = 0, Vv = 0;
            if (tt[k] >= 0) {
                S = s1[k];
                if (tt[k] >= 1) { S |= spv[k] & bpv[k] & ~s1[k]; Vv = vv[k]; }
                R = tt[k] == 2 ? s1[k] : s2[k];
            }
            st.S(y) = S;
            st.NEG(y) = sgv[k] & S;
            st.REF(y) = R;
            st.VIS(y) = Vv;
        }
    }
    const J2kCkptView cv = J2kCkptView::of(ckpt, (size_t) gid);
    const uint4 *rec = (const uint4 *) cv.slot((uint32_t) (r * 16 + stripe));
    const uint4 r0 = rec[0], r1 = rec[1];
    const uint32_t cxb[5] = {r0.z, r0.w, r1.x, r1.y, r1.z};                // record: { a | ct << 16, pos, five context words, c } (j2k.hpp)
    const t1::Contexts cxp = t1::Contexts::from_bytes(cxb);
    const t1::MqCheckpoint ck{r0.x & 0xFFFFu, r1.w, (int) (r0.x >> 16), (int) r0.y, cxp.w0, cxp.w1, cxp.w2};
    t1::decode_resume(st, DecSrc{cblk_bytes + (size_t) gid * kJ2kCblkBytes, len}, blk.w, blk.h, geom->bands[blk.band].orient, P,
                      np, r, stripe, ck, tab, ctxp);
}

// LDS carve-up of the rate / write kernels for the largest tile position of the context
size_t rate_lds(const J2kBuffers &jb)
{
    size_t n = 0;
    for (const J2kGeom &g : jb.geoms) n = std::max(n, rate_lds_bytes(g.nblocks, g.tree_nodes));
    return n;
}

}  // namespace

// ================================================================================================
static void rate_launch(const J2kBuffers &jb, int n_frames, const int *d_active, hipStream_t s, bool candidates, const int *have_rate)
{
    // dynamic LDS: the carve-up, the per-block table offsets, then as many (rate, distortion) entries as fit
    const size_t head = ((rate_lds(jb) + 15) & ~(size_t) 15) + rate_off_bytes(jb.geom.stride) + rate_path_bytes(jb.geom.stride);
    const size_t budget = 150 * 1024;
    size_t want = (size_t) jb.geom.stride * kJ2kMaxPasses;
    if (head + want * 10 + 16 > budget) want = head + 16 < budget ? (budget - head - 16) / 10 : 0;
    const size_t lds = head + want * 10 + 16;
    static std::once_flag once;
    std::call_once(once, [] {
        EBCC_HIP_CHECK(hipFuncSetAttribute((const void *) k_rate, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    });
    hipLaunchKernelGGL(k_rate, dim3(n_frames, candidates ? 2 : 1), dim3(kRateThreads), lds, s, jb.numbps, jb.totalpasses, jb.rates,
                       jb.disto, candidates ? jb.cand_npass : jb.npass, jb.d_geom, jb.jf, jb.fs, d_active, (int) want, jb.rate_path, jb.rate_path_n,
                       candidates ? jb.cand_cr : nullptr, jb.cand_out, (size_t) jb.max_frames * (size_t) jb.geom.stride, have_rate, jb.rate_cache);
    EBCC_HIP_LAUNCH_CHECK();
}

void launch_j2k_rate(const J2kBuffers &jb, int n_frames, const int *d_active, hipStream_t s, const int *have_rate)
{
    ScopedTiming t("rate_alloc", s);
    rate_launch(jb, n_frames, d_active, s, false, have_rate);
}

void launch_j2k_rate_candidates(const J2kBuffers &jb, int n_frames, const int *d_active, hipStream_t s)
{
    rate_launch(jb, n_frames, d_active, s, true, nullptr);
}

// takes the layer of candidate sel[f] (0 / 1; -1: none) over as the frame's current layer
__global__ void k_rate_publish(const int *sel, const int *cand_npass, const int *cand_out, size_t cand_stride, int *npass, J2kFrame *jf,
                               const J2kGeom *geom, int *have_rate)
{
    const int frame = blockIdx.x, c = sel[frame];
    if (threadIdx.x == 0) have_rate[frame] = c >= 0 ? 1 : 0;
    if (c < 0) return;
    const int nb = geom->stride;
    const int *src = cand_npass + (size_t) c * cand_stride + (size_t) frame * nb;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) npass[(size_t) frame * nb + b] = src[b];
    if (threadIdx.x == 0) {
        const int *o = cand_out + (size_t) (2 * frame + c) * 3;
        jf[frame].body_bytes = o[0]; jf[frame].stream_bytes = o[1]; jf[frame].maxlen = o[2];
    }
}
void launch_j2k_rate_publish(const J2kBuffers &jb, int n_frames, hipStream_t s)
{
    hipLaunchKernelGGL(k_rate_publish, dim3(n_frames), dim3(64), 0, s, jb.cand_sel, jb.cand_npass, jb.cand_out,
                       (size_t) jb.max_frames * (size_t) jb.geom.stride, jb.npass, jb.jf, jb.d_geom, jb.have_rate);
    EBCC_HIP_LAUNCH_CHECK();
}

void launch_j2k_write(const J2kBuffers &jb, int n_frames, const int *d_active, hipStream_t s)
{
    hipLaunchKernelGGL(k_write, dim3(n_frames), dim3(kWriteThreads), rate_lds(jb), s, jb.numbps, jb.rates, jb.npass,
                       jb.cblk_bytes, jb.stream, jb.stream_cap, jb.d_geom, jb.jf, jb.fs, d_active);
    EBCC_HIP_LAUNCH_CHECK();
}

static void decode_tail(const float *data, const J2kBuffers &jb, int n_frames, const int *d_active, bool stats, hipStream_t s,
                        int keep_field = 1)
{
    // dequantisation happens in the row passes, the mapping to the fp32 field and the statistics in the last column pass
    const int partials = j2k_inverse_dwt(jb.B, jb.V, stats ? data : nullptr, jb, n_frames, jb.fs, d_active, s, keep_field);
    if (stats)
        hipLaunchKernelGGL(k_finish_reduce, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, jb.partial, jb.partial_u, jb.jf,
                           n_frames, partials, jb.fs, d_active);
}

void launch_j2k_probe_decode(const float *data, const J2kBuffers &jb, int n_frames, const int *d_active, hipStream_t s, int keep_field)
{
    const int total = n_frames * jb.geom.stride;
    void *ck = jb.ckpt;
    static const bool stats = getenv("EBCC_HIP_T1_STATS") != nullptr;
    hipLaunchKernelGGL(k_probe_plan, dim3(ceil_div(total, 256)), dim3(256), 0, s, jb.numbps, jb.npass, jb.rates, ck, jb.qplane,
                       jb.lastnp, jb.d_geom, jb.d_blocks, jb.fs, d_active, total, stats ? 1 : 0);
    if (stats) {
        hipLaunchKernelGGL(k_probe_hist_round, dim3(ceil_div(std::max(n_frames, kProbeHistFrames), 256)), dim3(256), 0, s, n_frames, jb.fs, d_active);
        hipLaunchKernelGGL(k_probe_hist_next, dim3(1), dim3(1), 0, s);
    }
    hipLaunchKernelGGL(k_probe_init, dim3(jb.geom.stride, n_frames), dim3(256), 0, s, jb.Q6, jb.qplane, jb.numbps, jb.SPS, jb.V,
                       jb.d_geom, jb.d_blocks, jb.fs, d_active);
    timing_begin("t1_probe_decode", s);
    const int lpw = t1_lanes_per_wave(T1_RESUME);
    ensure_ctx_tables();
    hipLaunchKernelGGL(k_t1_resume, dim3((unsigned) ceil_div(total, lpw)), dim3(64), (size_t) kDecStateRows * lpw * 8, s, jb.T1S, jb.BP, jb.SUF, jb.SGN, jb.SPS,
                       jb.VISP, jb.cblk_bytes, jb.numbps, jb.npass, jb.rates, jb.qplane, ck, jb.V, jb.d_geom, jb.d_blocks, jb.fs,
                       d_active, total, lpw);
    timing_end("t1_probe_decode", s);
    decode_tail(data, jb, n_frames, d_active, true, s, keep_field);
    EBCC_HIP_LAUNCH_CHECK();
    EBCC_HIP_LAUNCH_CHECK();
}

// EBCC_HIP_T1_STATS=1: prints and resets the histogram above (the stream must be idle)
void j2k_probe_hist_dump(const char *what)
{
    unsigned int h[kProbeHistRounds][8], rounds = 0, zero = 0;
    unsigned long long dirty[kProbeHistRounds];
    EBCC_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_probe_hist), sizeof h));
    EBCC_HIP_CHECK(hipMemcpyFromSymbol(dirty, HIP_SYMBOL(g_probe_dirty), sizeof dirty));
    EBCC_HIP_CHECK(hipMemcpyFromSymbol(&rounds, HIP_SYMBOL(g_probe_round), sizeof rounds));
    fprintf(stderr, "ebcc-mi355x probes of %s, per round: frames by the coarsest resolution with a changed code-block [0 1 2 3 4 5 none] (changed finest code-blocks per frame of class 5)\n", what);
    for (unsigned int r = 0; r < std::min(rounds, (unsigned int) kProbeHistRounds); r++) {
        unsigned int n = 0;
        for (int k = 0; k < 7; k++) n += h[r][k];
        if (!n) continue;
        fprintf(stderr, "ebcc-mi355x   round %2u: %4u %4u %4u %4u %4u %4u %4u  (%.1f)  tiles reached %.1f %%\n", r, h[r][0], h[r][1], h[r][2], h[r][3], h[r][4], h[r][5], h[r][6],
                h[r][5] ? (double) h[r][7] / h[r][5] : 0.0, 100.0 * (double) dirty[r] / (96.0 * n));
    }
    static unsigned long long zd[kProbeHistRounds];
    EBCC_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_dirty), zd, sizeof zd));
    static unsigned int zeros[kProbeHistRounds][8];
    EBCC_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_hist), zeros, sizeof zeros));
    EBCC_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_round), &zero, sizeof zero));
}

// The lanes per wave of a launch from the lengths the host parsed out of the packet headers.  A wave lasts as long as its
// lanes together (a lone lane: kDecChainMsPerByte of its segment; a wave of two 0.875 of its two chains one after the other,
// a wave of four 0.68), and the launch cannot be shorter than its vector issue slots allow (kDecIssueMsPerByte[lanes] of all
// segments: lanes of a wave share part of their instructions).  With 256 frames the issue slots bind and four lanes are
// best (16.6 ms; 19.3 at two, 23.4 at one); a batch of 43 or 85 frames has issue slots to spare and ends soonest with
// every chain alone in its wave (tools/gpu/dec_tiers_sweep.sh, four lanes -> one: 43 frames 17.6 -> 7.2 ms, 85 frames
// 15.9 -> 9.1, 128 frames 16.4 -> 12.6, 11.7 at two).  The plan takes whichever of the three measured shapes the model
// puts first.  (Finer mixtures - every wave filled up to the launch's duration - were built and measured: the waves of
// the wider tiers start behind all narrower ones and last as long, 85 frames 10.7 - 12.4 ms.)
constexpr double kDecChainMsPerByte = 3.0e-3;                            // (a 2.3 KB segment alone in its wave: 7 ms)
constexpr double kDecIssueMsPerByte[3] = {1.0e-6, 0.82e-6, 0.706e-6};    // 1, 2, 4 lanes per wave: 23.4, 19.3, 16.6 ms for the 23.5 MB of 256 frames
static DecTiers plan_dec_tiers(const int *host_table, int total)
{
    long long cnt[kDecClasses] = {0};
    double all = 0;
    int longest = 0;
    for (int i = 0; i < total; i++) {
        const int *e = host_table + (size_t) i * 4;
        const int c = e[3] <= 0 || e[2] <= 0 ? 0 : std::min(kDecClasses - 1, 1 + (e[1] >> 6));    // == dec_class
        cnt[c]++;
        if (c) { all += e[1]; longest = std::max(longest, e[1]); }
    }
    // the longest segment a wave of four would hold with the longest 1/64 of the code-blocks in pairs (kDecTierDen1)
    int c64 = kDecClasses - 1;
    for (long long above = 0; c64 > 0 && above + cnt[c64] <= total / kDecTierDen1; c64--) above += cnt[c64];
    const double len64 = std::min((double) longest, 64.0 * c64);
    const double alone = std::max(kDecChainMsPerByte * longest, kDecIssueMsPerByte[0] * all);
    const double pairs = std::max(2 * 0.875 * kDecChainMsPerByte * longest, kDecIssueMsPerByte[1] * all);
    const double fours = std::max({2 * 0.875 * kDecChainMsPerByte * longest, 4 * 0.68 * kDecChainMsPerByte * len64, kDecIssueMsPerByte[2] * all});
    DecTiers tiers{{0, 0, 0}, {1, 2, 4, 4}};
    if (alone <= pairs && alone <= fours) tiers.lanes[3] = 1;
    else if (pairs <= fours) tiers.lanes[3] = 2;
    else tiers.n[1] = total / kDecTierDen1 / 2 * 2;
    return tiers;
}

void plan_decode_lanes(const int *host_table, int total, int out[4])
{
    const DecTiers t = plan_dec_tiers(host_table, total);
    for (int i = 0; i < 3; i++) out[i] = t.n[i];
    out[3] = t.lanes[3];
}

void launch_j2k_decode(const J2kBuffers &jb, int n_frames, hipStream_t s, const int *host_table)
{
    const size_t n_pix = (size_t) jb.geom.W * jb.geom.H;
    const int total = n_frames * jb.geom.stride;
    const size_t groups = ((size_t) total + 63) / 64;
    EBCC_HIP_CHECK(hipMemsetAsync(jb.V, 0, (size_t) n_frames * n_pix * sizeof(int32_t), s));
    int lpw = std::min(16, t1_lanes_per_wave(T1_DECODE));              // (the state row masks of a wave's code-blocks live in LDS: at most 16)
    // few code-blocks (a frame or a few decoded alone, e.g. from an HDF5 filter callback): there are wave slots to spare and
    // a wave per code-block ends soonest (one 721 x 1440 frame through ebcc_decode: 16.8 -> 8.6 ms)
    const bool few_blocks = !getenv("EBCC_T1_LPW") && total <= 4096;
    if (few_blocks) lpw = 1;
    if (getenv("EBCC_HIP_T1_STATS")) {                                   // diagnostics: the code-blocks with the longest segments
        std::vector<int> h((size_t) total * 4);
        EBCC_HIP_CHECK(hipMemcpyAsync(h.data(), jb.dec_table, h.size() * sizeof(int), hipMemcpyDeviceToHost, s));
        EBCC_HIP_CHECK(hipStreamSynchronize(s));
        std::vector<int> idx(total);
        for (int i = 0; i < total; i++) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](int a, int b) { return h[(size_t) a * 4 + 1] > h[(size_t) b * 4 + 1]; });
        long long sum = 0;
        for (int i = 0; i < total; i++) sum += h[(size_t) i * 4 + 1];
        fprintf(stderr, "ebcc-mi355x t1 decode: %d code-blocks, %lld bytes; longest:", total, sum);
        for (int i = 0; i < std::min(total, 12); i++) fprintf(stderr, " [blk %d len %d P %d np %d]", idx[i] % jb.geom.stride, h[(size_t) idx[i] * 4 + 1], h[(size_t) idx[i] * 4 + 2], h[(size_t) idx[i] * 4 + 3]);
        fprintf(stderr, "; percentiles of len:");
        for (int q : {50, 90, 99}) fprintf(stderr, " p%d %d", q, h[(size_t) idx[(size_t) total * (100 - q) / 100] * 4 + 1]);
        fprintf(stderr, "\n");
    }
    timing_begin("t1_decode", s);
    {
        int *counters = jb.dec_order + groups * 64;
        EBCC_HIP_CHECK(hipMemsetAsync(counters, 0, 2 * kDecClasses * sizeof(int), s));
        hipLaunchKernelGGL(k_dec_hist, dim3(ceil_div(total, 256)), dim3(256), 0, s, jb.dec_table, counters, total);
        hipLaunchKernelGGL(k_dec_offsets, dim3(1), dim3(kDecClasses), 0, s, counters);
        hipLaunchKernelGGL(k_dec_place, dim3(ceil_div(total, 256)), dim3(256), 0, s, jb.dec_table, counters, jb.dec_order, total);
        // Tiers by rank (EBCC_T1_DEC_TIERS = "d0,d1,d2[,L]": the first total/d0 code-blocks of the order at 1 lane per wave, up
        // to total/d1 at 2, up to total/d2 at 4, the rest at L = 4, 8 or 16; EBCC_T1_LPW = one number of lanes for all).
        // Measured on 256 frames of configs[1] (round 3): 1/32 at 2 lanes and the rest at 4: 18.5 ms; all at 4: 18.5; a tier
        // of single lanes for the longest 1/256 or 1/512: 18.5 (the launch is bound by vector issue, not by its longest
        // chain); an 8-lane tier for the shortest 2/3: 31.6 - and 29.7 with no wave using it: what costs is the LDS a
        // workgroup reserves (the state rows of its widest tier: 16.5 KB instead of 8), i.e. how many waves a CU holds.  With
        // the NEG and REF rows in device memory instead (half the LDS per code-block, so that 8 lanes reserve what 4 do now;
        // bit-exact, since removed): 18.9 ms at 4 lanes, 20.5 - 27.6 ms with 8-lane tiers - lanes of a wave take turns, a
        // wave of 8 lasts twice as long; and 96 VGPRs for a fifth wave per SIMD spill 62 registers: 20.1 - 20.8 ms.
        DecTiers tiers{{0, 0, 0}, {1, 2, 4, 4}};
        if (few_blocks) { tiers.lanes[3] = 1; }
        else if (getenv("EBCC_T1_LPW")) { tiers.lanes[3] = lpw; }
        else if (host_table && !getenv("EBCC_T1_DEC_TIERS")) tiers = plan_dec_tiers(host_table, total);
        else {
            int den[3] = {kDecTierDen0, kDecTierDen1, kDecTierDen2};
            if (const char *e = getenv("EBCC_T1_DEC_TIERS")) {
                int v[4];
                const int got = sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]);
                if (got >= 3) for (int i = 0; i < 3; i++) den[i] = v[i];
                if (got == 4 && (v[3] == 4 || v[3] == 8 || v[3] == 16)) tiers.lanes[3] = v[3];      // (lanes of the last tier)
            }
            int before = 0;
            for (int t = 0; t < 3; t++) {
                const int upto = den[t] > 0 ? total / den[t] : 0;                  // (den 0: no such tier)
                const int n = std::max(0, upto - before) / tiers.lanes[t] * tiers.lanes[t];
                tiers.n[t] = n; before += n;
            }
        }
        if (getenv("EBCC_HIP_T1_STATS")) fprintf(stderr, "ebcc-mi355x t1 decode tiers: %d code-blocks alone in their waves, %d in pairs, the rest %d to a wave\n", tiers.n[0], tiers.n[1], tiers.lanes[3]);
        unsigned waves = 0;
        int rest = total;
        for (int t = 0; t < 3; t++) { waves += (unsigned) (tiers.n[t] / tiers.lanes[t]); rest -= tiers.n[t]; }
        waves += (unsigned) ceil_div(rest, tiers.lanes[3]);
        const int max_lanes = std::max(std::max(tiers.n[0] ? tiers.lanes[0] : 1, tiers.n[1] ? tiers.lanes[1] : 1), std::max(tiers.n[2] ? tiers.lanes[2] : 1, rest > 0 ? tiers.lanes[3] : 1));
        ensure_ctx_tables();
        hipLaunchKernelGGL(k_t1_decode_lds, dim3(waves), dim3(64), (size_t) kDecStateRows * max_lanes * 8, s, jb.stream,
                           jb.stream_cap, jb.dec_table, jb.dec_order, jb.V, jb.d_geom, jb.d_blocks, jb.fs, total, tiers);
    }
    timing_end("t1_decode", s);
    decode_tail(nullptr, jb, n_frames, nullptr, false, s);
    EBCC_HIP_LAUNCH_CHECK();
}

}  // namespace ebcc
