// j2k_analysis.hip - rate-independent half of the JPEG 2000 encoder on gfx950:
//   input statistics, u16 scaling + DC level shift, forward 9/7 (OpenJPEG 2.4.0 arithmetic order),
//   quantisation to bit-plane row masks (wave ballots), tier-1 coding of every code-block (one
//   code-block per lane, t1_core.hpp) and the per-pass distortion tables for rate allocation.
// OpenJPEG runs all of this inside every opj_encode call; the reference calls opj_encode ~22 times per
// frame with different rates (/root/reference/src/ebcc_codec.c:545-596).  None of it depends on the
// rate, so here it runs once per frame.
#include <array>
#include <mutex>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "j2k.hpp"
#include "t1_core.hpp"
#include "t1_device.hpp"

namespace ebcc {

// ================================================================================================
// geometry (T.800 B.5-B.7 with default precincts), quantisation parameters (E.1)
// ================================================================================================
static int ceildivpow2(int a, int b) { return (int) (((long long) a + (1ll << b) - 1) >> b); }
static int floorlog2i(int a) { int l = 0; while (a > 1) { a >>= 1; l++; } return l; }

static const double kNormsReal[4][10] = {
    {1.000, 1.965, 4.177, 8.403, 16.90, 33.84, 67.69, 135.3, 270.6, 540.9},
    {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0},
    {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0},
    {2.080, 3.865, 8.307, 17.18, 34.71, 69.59, 139.3, 278.6, 557.2}};

int t1_lanes_per_wave(int kernel, int total_blocks)
{
    // EBCC_T1_LPW = "<n>" (all kernels) or "<decision pass>,<MQ pass>,<probe restart>,<decode>", each a power of two up to 64
    // (read at every launch: a tuning knob, results do not depend on it)
    int t[4] = {64, 64, 4, 2};                                            // (decode: 2 lanes 26.5 ms per 256 frames, 4 lanes 29, 8 lanes 37 - tools/gpu/r2_sweep.sh)
    if (const char *e = getenv("EBCC_T1_LPW")) {
        int v[4], n = sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]);
        for (int i = 0; i < 4; i++) {
            const int x = n == 1 ? v[0] : (i < n ? v[i] : 0);
            if (x == 1 || x == 2 || x == 4 || x == 8 || x == 16 || x == 32 || x == 64) t[i] = x;
        }
        return t[kernel & 3];
    }
    // The encoder kernels step all lanes of a wave together, so a wave lasts as long as the sum over its steps of the
    // slowest lane: with few code-blocks in the launch (a frame or two coded alone, e.g. from an HDF5 filter callback)
    // there are SIMDs to spare and fewer lanes per wave finish sooner (one 721 x 1440 frame: 46.6 ms at 64 lanes,
    // 36.3 ms at 4); large batches need the lanes (measured: 64 is best from 64 frames on).
    if (kernel <= T1_MQ && total_blocks > 0 && total_blocks <= 8192) {
        int lpw = 4;
        while (lpw < 64 && total_blocks / lpw > 1024) lpw <<= 1;
        return lpw;
    }
    return t[kernel & 3];
}

J2kGeom make_j2k_geom(int H, int W, std::vector<J2kBlock> &blocks, int ty0)
{
    J2kGeom g{};
    g.W = W; g.H = H; g.ty0 = ty0; g.period = 1;
    const int prec = 16, guard = 2, ty1 = ty0 + H;
    for (int r = 0; r < kJ2kRes; r++) {
        const int lv = kJ2kRes - 1 - r;
        g.rw[r] = ceildivpow2(W, lv);
        g.ry0[r] = ceildivpow2(ty0, lv);
        g.rh[r] = ceildivpow2(ty1, lv) - g.ry0[r];
    }
    blocks.clear();
    int bi = 0, nodes = 0;
    for (int r = 0; r < kJ2kRes; r++) {
        g.res_first[r] = (int) blocks.size();
        int lv = kJ2kRes - 1 - r;
        int nb = r == 0 ? 1 : 3;
        for (int b = 0; b < nb; b++, bi++) {
            J2kBand &bd = g.bands[bi];
            bd.res = r; bd.level = lv;
            if (r == 0) {
                bd.orient = 0;
                bd.x0 = 0; bd.y0 = g.ry0[0]; bd.x1 = g.rw[0]; bd.y1 = g.ry0[0] + g.rh[0];
                bd.offx = 0; bd.offy = 0;
            } else {
                bd.orient = b + 1;
                int xb = bd.orient & 1, yb = bd.orient >> 1;
                bd.x0 = 0;                                          // tiles span the image width
                bd.y0 = std::max(0, ceildivpow2(ty0 - (yb << lv), lv + 1));          // T.800 B-15
                bd.x1 = ceildivpow2(W - (xb << lv), lv + 1);
                bd.y1 = ceildivpow2(ty1 - (yb << lv), lv + 1);
                bd.offx = xb ? g.rw[r - 1] : 0;
                bd.offy = yb ? g.rh[r - 1] : 0;
            }
            // QCD entry: opj_dwt_calc_explicit_stepsizes + opj_dwt_encode_stepsize
            double stepsize = 1.0 / kNormsReal[bd.orient][lv];
            int v = (int) std::floor(stepsize * 8192.0);
            int p = floorlog2i(v) - 13, n = 11 - floorlog2i(v);
            bd.mant = (n < 0 ? v >> -n : v << n) & 0x7FF;
            bd.expn = prec - p;
            bd.numbps = bd.expn + guard - 1;
            int log2_gain = bd.orient == 0 ? 0 : (bd.orient == 3 ? 2 : 1);
            bd.step_enc = (float) ((1.0 + bd.mant / 2048.0) * std::pow(2.0, (double) (prec + log2_gain - bd.expn)));
            bd.step_dec = (float) ((1.0 + bd.mant / 2048.0) * std::pow(2.0, (double) (prec - bd.expn)));
            bd.norm = kNormsReal[bd.orient][lv];
            int bw = bd.x1 - bd.x0, bh = bd.y1 - bd.y0;
            bd.first_block = (int) blocks.size();
            if (bw <= 0 || bh <= 0) { bd.ncw = bd.nch = 0; bd.tree_levels = 0; bd.tree_off = nodes; continue; }
            // the code-block partition is anchored at multiples of 64 of the band coordinates (B.7)
            const int gx0 = bd.x0 / 64, gy0 = bd.y0 / 64;
            bd.ncw = (bd.x1 + 63) / 64 - gx0;
            bd.nch = (bd.y1 + 63) / 64 - gy0;
            for (int cy = 0; cy < bd.nch; cy++)
                for (int cx = 0; cx < bd.ncw; cx++) {
                    J2kBlock k{};
                    k.band = bi; k.cx = cx; k.cy = cy;
                    int x0 = std::max((gx0 + cx) * 64, bd.x0), y0 = std::max((gy0 + cy) * 64, bd.y0);
                    int x1 = std::min((gx0 + cx) * 64 + 64, bd.x1), y1 = std::min((gy0 + cy) * 64 + 64, bd.y1);
                    k.x = bd.offx + x0 - bd.x0; k.y = bd.offy + y0 - bd.y0; k.w = x1 - x0; k.h = y1 - y0;
                    blocks.push_back(k);
                }
            // tag-tree level layout (B.10.2)
            int lw = bd.ncw, lh = bd.nch, off = 0, L = 0;
            bd.tree_off = nodes;
            for (;;) {
                bd.lvl_w[L] = lw; bd.lvl_h[L] = lh; bd.lvl_off[L] = off;
                off += lw * lh; L++;
                if (lw * lh <= 1) break;
                lw = (lw + 1) / 2; lh = (lh + 1) / 2;
            }
            bd.tree_levels = L;
            nodes += off;
        }
    }
    g.nbands = bi;
    g.nblocks = (int) blocks.size();
    g.stride = g.nblocks;
    g.res_first[kJ2kRes] = g.nblocks;
    g.tree_nodes = nodes;
    return g;
}

namespace {

// ================================================================================================
// input statistics
// ================================================================================================
__global__ void k_in_init(FrameState *fs, int n)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) { fs[f].min_key = ~0ull; fs[f].max_key = 0ull; fs[f].has_nonfinite = 0; }
}

__global__ __launch_bounds__(256) void k_in_minmax(const float *__restrict__ data, size_t n_pix, FrameState *fs)
{
    const int frame = blockIdx.y;
    const float *x = data + (size_t) frame * n_pix;
    unsigned long long kmin = ~0ull, kmax = 0ull;
    int bad = 0;
    auto take = [&](float v, size_t i) {
        if (isnan(v) || isinf(v)) bad = 1;                            // check_nan_inf, ebcc_codec.c:598-605
        unsigned long long k = (unsigned long long) float_order_key(v) << 32;
        unsigned long long lo = k | (unsigned int) i, hi = k | (0xFFFFFFFFu - (unsigned int) i);
        kmin = lo < kmin ? lo : kmin;
        kmax = hi > kmax ? hi : kmax;
    };
    if ((n_pix & 3) == 0 && ((size_t) data & 15) == 0) {
        // 16 bytes per lane, two loads in flight (4 bytes per lane and one load in flight ran at 48 % of the copy rate)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
        const size_t n4 = n_pix >> 2, step = (size_t) gridDim.x * blockDim.x;
        size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + step < n4; i += 2 * step) {
            const f32x4 a = x4[i], b = x4[i + step];
            take(a.x, 4 * i); take(a.y, 4 * i + 1); take(a.z, 4 * i + 2); take(a.w, 4 * i + 3);
            take(b.x, 4 * (i + step)); take(b.y, 4 * (i + step) + 1); take(b.z, 4 * (i + step) + 2); take(b.w, 4 * (i + step) + 3);
        }
        if (i < n4) { const f32x4 a = x4[i]; take(a.x, 4 * i); take(a.y, 4 * i + 1); take(a.z, 4 * i + 2); take(a.w, 4 * i + 3); }
    } else {
        for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += (size_t) gridDim.x * blockDim.x) take(x[i], i);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(kmin, d); kmin = o < kmin ? o : kmin;
        o = __shfl_xor(kmax, d); kmax = o > kmax ? o : kmax;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&fs[frame].min_key, kmin);
        atomicMax(&fs[frame].max_key, kmax);
    }
    if (bad) fs[frame].has_nonfinite = 1;
}

__global__ void k_in_finish(const float *data, size_t n_pix, FrameState *fs, int n)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const float *x = data + (size_t) f * n_pix;
    float mn = x[(unsigned int) fs[f].min_key], mx = x[0xFFFFFFFFu - (unsigned int) fs[f].max_key];
    fs[f].minv = mn; fs[f].maxv = mx;
    fs[f].const_field = (mn == mx) ? 1 : 0;                           // ebcc_codec.c:678
}

// ================================================================================================
// scaling to u16 (ebcc_codec.c:688), DC level shift to float (opj_tcd_dc_level_shift_encode)
// ================================================================================================
__global__ __launch_bounds__(256) void k_scale_shift(const float *__restrict__ data, float *__restrict__ B, size_t n_pix,
                                                      const FrameState *fs)
{
    const int frame = blockIdx.y;
    if (fs[frame].const_field) return;
    const float *x = data + (size_t) frame * n_pix;
    float *b = B + (size_t) frame * n_pix;
    const float mn = fs[frame].minv, rng = fs[frame].maxv - fs[frame].minv;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += (size_t) gridDim.x * blockDim.x) {
        unsigned int u = __float2uint_rz(((x[i] - mn) / rng) * 65535.0f) & 0xFFFFu;
        b[i] = (float) ((int) u - 32768);
    }
}

// ================================================================================================
// 9/7 lifting with OpenJPEG's boundary handling and arithmetic order
// ================================================================================================
__device__ constexpr float kA = -1.586134342f, kB = -0.052980118f, kG = 0.882911075f, kD = 0.443506852f;
__device__ constexpr float kK = 1.230174105f, kTwoInvK = 1.625732422f;

// high[i] += (low[i] + low[i+1]) * c, last high of an even-length line uses 2*low[i]   (opj_dwt_encode_step2)
template <typename Idx>
__device__ inline void fstep_hi(float *E, float *O, int sn, int dn, int lines, Idx at, float c, int tid, int nt)
{
    for (int t = tid; t < dn * lines; t += nt) {
        int i = t / lines, l = t - i * lines;
        float o = O[at(i, l)];
        if (i + 1 < sn) o = o + ((E[at(i, l)] + E[at(i + 1, l)]) * c);
        else            o = o + ((2 * E[at(i, l)]) * c);
        O[at(i, l)] = o;
    }
    __syncthreads();
}
// low[i] += (high[i-1] + high[i]) * c with high[-1] := high[0]; last low of an odd-length line uses 2*high[i-1]
template <typename Idx>
__device__ inline void fstep_lo(float *E, float *O, int sn, int dn, int lines, Idx at, float c, int tid, int nt)
{
    for (int t = tid; t < sn * lines; t += nt) {
        int i = t / lines, l = t - i * lines;
        float e = E[at(i, l)];
        if (i < dn) {
            float hl = i == 0 ? O[at(0, l)] : O[at(i - 1, l)];
            e = e + ((hl + O[at(i, l)]) * c);
        } else {
            e = e + ((2 * O[at(i - 1, l)]) * c);
        }
        E[at(i, l)] = e;
    }
    __syncthreads();
}
// decoder forms (opj_v8dwt_decode_step2): the boundary term is x + h * (c + c)
template <typename Idx>
__device__ inline void istep_lo(float *E, float *O, int sn, int dn, int lines, Idx at, float c, int tid, int nt)
{
    for (int t = tid; t < sn * lines; t += nt) {
        int i = t / lines, l = t - i * lines;
        float e = E[at(i, l)];
        if (i < dn) {
            float hl = i == 0 ? O[at(0, l)] : O[at(i - 1, l)];
            e = e + ((hl + O[at(i, l)]) * c);
        } else {
            e = e + (O[at(i - 1, l)] * (c + c));
        }
        E[at(i, l)] = e;
    }
    __syncthreads();
}
template <typename Idx>
__device__ inline void istep_hi(float *E, float *O, int sn, int dn, int lines, Idx at, float c, int tid, int nt)
{
    for (int t = tid; t < dn * lines; t += nt) {
        int i = t / lines, l = t - i * lines;
        float o = O[at(i, l)];
        if (i + 1 < sn) o = o + ((E[at(i, l)] + E[at(i + 1, l)]) * c);
        else            o = o + (E[at(i, l)] * (c + c));
        O[at(i, l)] = o;
    }
    __syncthreads();
}

// E = low-pass samples, O = high-pass samples of a line that starts with a low-pass sample (cas = 0: rows, and the
// columns of a tile whose first row at this resolution is even) or with a high-pass one (cas = 1: the neighbour
// relations of the two lifting steps swap; opj_dwt_encode_1_real / opj_v8dwt_decode with cas)
template <typename Idx>
__device__ inline void fdwt_tile(float *E, float *O, int sn, int dn, int lines, Idx at, int tid, int nt, int cas = 0)
{
    const float invK = (float) (1.0 / 1.230174105);
    if (cas == 0) {
        fstep_hi(E, O, sn, dn, lines, at, kA, tid, nt);
        fstep_lo(E, O, sn, dn, lines, at, kB, tid, nt);
        fstep_hi(E, O, sn, dn, lines, at, kG, tid, nt);
        fstep_lo(E, O, sn, dn, lines, at, kD, tid, nt);
    } else {
        fstep_lo(O, E, dn, sn, lines, at, kA, tid, nt);
        fstep_hi(O, E, dn, sn, lines, at, kB, tid, nt);
        fstep_lo(O, E, dn, sn, lines, at, kG, tid, nt);
        fstep_hi(O, E, dn, sn, lines, at, kD, tid, nt);
    }
    for (int t = tid; t < sn * lines; t += nt) { int i = t / lines, l = t - i * lines; E[at(i, l)] *= invK; }
    for (int t = tid; t < dn * lines; t += nt) { int i = t / lines, l = t - i * lines; O[at(i, l)] *= kK; }
    __syncthreads();
}
template <typename Idx>
__device__ inline void idwt_tile(float *E, float *O, int sn, int dn, int lines, Idx at, int tid, int nt, int cas = 0)
{
    for (int t = tid; t < sn * lines; t += nt) { int i = t / lines, l = t - i * lines; E[at(i, l)] = E[at(i, l)] * kK; }
    for (int t = tid; t < dn * lines; t += nt) { int i = t / lines, l = t - i * lines; O[at(i, l)] = O[at(i, l)] * kTwoInvK; }
    __syncthreads();
    if (cas == 0) {
        istep_lo(E, O, sn, dn, lines, at, -kD, tid, nt);
        istep_hi(E, O, sn, dn, lines, at, -kG, tid, nt);
        istep_lo(E, O, sn, dn, lines, at, -kB, tid, nt);
        istep_hi(E, O, sn, dn, lines, at, -kA, tid, nt);
    } else {
        istep_hi(O, E, dn, sn, lines, at, -kD, tid, nt);
        istep_lo(O, E, dn, sn, lines, at, -kG, tid, nt);
        istep_hi(O, E, dn, sn, lines, at, -kB, tid, nt);
        istep_lo(O, E, dn, sn, lines, at, -kA, tid, nt);
    }
}

struct RowIdx { __device__ int operator()(int k, int) const { return k; } };
template <int CW> struct ColIdx { __device__ int operator()(int k, int l) const { return k * CW + l; } };

constexpr int kRowT = 256, kColT = 1024;

// rows of the region [0,n) x [0,rows) of the tile buffer at resolution r, in place (the horizontal extents are the
// same for every tile: tiles span the image width).  The inverse transform can take the three detail bands of the
// level (and at r == 1 the LL band too) straight from the tier-1 decoder's output V (half-step units, same layout
// as B), dequantising on the way: (float) v * (0.5f * step) as opj_t1_decode_cblk does; the LL quadrant of r > 1
// is the previous level's result in B.
template <bool FWD>
__global__ __launch_bounds__(kRowT) void k_j2k_rows(float *__restrict__ B, const int32_t *__restrict__ V, const J2kGeom *geom, int r,
                                                     const FrameState *fs, const int *active)
{
    extern __shared__ float sm[];
    const int frame = blockIdx.y;
    if ((active && !active[frame]) || (fs && fs[frame].const_field)) return;
    const J2kGeom &g = j2k_frame_geom(geom, frame);
    const int W = g.W, n = g.rw[r], sn = g.rw[r - 1], rows = g.rh[r];
    const int dn = n - sn, tid = threadIdx.x;
    float *E = sm, *O = sm + sn;
    float *buf = B + (size_t) frame * ((size_t) W * g.H);
    const int32_t *vbuf = V ? V + (size_t) frame * ((size_t) W * g.H) : nullptr;
    const int rows_lo = g.rh[r - 1];                                     // rows of the LL / HL bands; LH / HH below
    const float s_ll = 0.5f * g.bands[0].step_dec, s_hl = 0.5f * g.bands[3 * (r - 1) + 1].step_dec,
                s_lh = 0.5f * g.bands[3 * (r - 1) + 2].step_dec, s_hh = 0.5f * g.bands[3 * (r - 1) + 3].step_dec;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        float *line = buf + (size_t) row * W;
        if (FWD) {
            for (int i = tid; i < n; i += kRowT) { float v = line[i]; ((i & 1) ? O : E)[i >> 1] = v; }
        } else if (vbuf) {
            const int32_t *vline = vbuf + (size_t) row * W;
            const bool top = row < rows_lo;
            const float s_lo = top ? s_ll : s_lh, s_hi = top ? s_hl : s_hh;
            const bool lo_from_v = !top || r == 1;
            for (int i = tid; i < sn; i += kRowT) E[i] = lo_from_v ? (float) vline[i] * s_lo : line[i];
            for (int i = tid; i < dn; i += kRowT) O[i] = (float) vline[sn + i] * s_hi;
        } else {
            for (int i = tid; i < sn; i += kRowT) E[i] = line[i];
            for (int i = tid; i < dn; i += kRowT) O[i] = line[sn + i];
        }
        __syncthreads();
        if (FWD) {
            fdwt_tile(E, O, sn, dn, 1, RowIdx(), tid, kRowT);
            for (int i = tid; i < sn; i += kRowT) line[i] = E[i];
            for (int i = tid; i < dn; i += kRowT) line[sn + i] = O[i];
        } else {
            idwt_tile(E, O, sn, dn, 1, RowIdx(), tid, kRowT);
            for (int i = tid; i < n; i += kRowT) line[i] = ((i & 1) ? O : E)[i >> 1];
        }
        __syncthreads();
    }
}

// What the last inverse column pass does with its result instead of storing it (FIN): DC level shift + rounding +
// clamp (opj_tcd_dc_level_shift_decode), u16 -> fp32 (ebcc_codec.c:1130) into DEC, and with `data` the error
// statistics against the input (get_mean_error :494-501, get_error_target_quantile :503-513) as one partial sum per
// column tile.
struct J2kFinish {
    const float *data;             // nullptr: no statistics
    float *DEC;
    J2kFrame *jf;
    double *partial;               // [frames][kPartials]
    unsigned long long *partial_u;
    int per_frame;                 // DEC is written only for frames with jf[frame].keep set
    __device__ float *field(int frame, size_t n_pix) const
    {
        return DEC && (!per_frame || jf[frame].keep) ? DEC + (size_t) frame * n_pix : nullptr;      // (null: statistics only)
    }
};

// columns of the region [0,cols) x [0,n) at resolution r, CW columns per tile staged through LDS, in place.  The
// number of rows, of low-pass rows and the parity of the first row belong to the frame's tile position.
template <bool FWD, int CW, bool FIN = false>
__global__ __launch_bounds__(kColT) void k_j2k_cols(float *__restrict__ B, const J2kGeom *geom, int r, const FrameState *fs,
                                                     const int *active, J2kFinish fin = J2kFinish{})
{
    extern __shared__ float sm[];
    const int frame = blockIdx.y;
    if ((active && !active[frame]) || (fs && fs[frame].const_field)) return;
    const J2kGeom &g = j2k_frame_geom(geom, frame);
    const int W = g.W, n = g.rh[r], sn = g.rh[r - 1], cols = g.rw[r], cas = g.ry0[r] & 1;
    if (n <= 1) return;
    const int dn = n - sn, tid = threadIdx.x;
    float *E = sm, *O = sm + (size_t) sn * CW;
    float *buf = B + (size_t) frame * ((size_t) W * g.H);
    const int ntiles = (cols + CW - 1) / CW;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int x0 = tile * CW, w = min(CW, cols - x0);
        for (int t = tid; t < n * CW; t += kColT) {
            int y = t / CW, c = t - y * CW;
            float v = c < w ? buf[(size_t) y * W + x0 + c] : 0.0f;
            if (FWD) (((y ^ cas) & 1) ? O : E)[(y >> 1) * CW + c] = v;
            else     (y < sn ? E : O)[(y < sn ? y : y - sn) * CW + c] = v;
        }
        __syncthreads();
        if (FWD) fdwt_tile(E, O, sn, dn, CW, ColIdx<CW>(), tid, kColT, cas);
        else     idwt_tile(E, O, sn, dn, CW, ColIdx<CW>(), tid, kColT, cas);
        if constexpr (FIN) {
            __shared__ double red[kColT / 64];
            __shared__ unsigned int redu[kColT / 64];
            const size_t n_pix = (size_t) W * g.H;
            const float *x = fin.data ? fin.data + (size_t) frame * n_pix : nullptr;
            float *d = fin.field(frame, n_pix);
            const float mn = fs[frame].minv, rng = fs[frame].maxv - fs[frame].minv;
            const float target = x ? fin.jf[frame].target : 0.0f;
            double acc = 0;
            unsigned int bad = 0;
            for (int t = tid; t < n * CW; t += kColT) {
                int y = t / CW, c = t - y * CW;
                if (c >= w) continue;
                const float v = (((y ^ cas) & 1) ? O : E)[(y >> 1) * CW + c];
                long long q;
                if (v > 2147483647.0f) q = 65535;
                else if (v < -2147483648.0f) q = 0;
                else {
                    q = (long long) __float2int_rn(v) + 32768;
                    q = q < 0 ? 0 : (q > 65535 ? 65535 : q);
                }
                const float dv = ((float) (int) q / 65535.0f) * rng + mn;
                const size_t i = (size_t) y * W + x0 + c;
                if (d) d[i] = dv;
                if (x) {
                    const float e = x[i] - (dv + 0.0f);
                    acc += (double) e;
                    if (fabsf(e) > target) bad++;
                }
            }
            if (x) {
                for (int k = 32; k >= 1; k >>= 1) { acc += __shfl_xor(acc, k); bad += __shfl_xor(bad, k); }
                if ((tid & 63) == 0) { red[tid >> 6] = acc; redu[tid >> 6] = bad; }
                __syncthreads();
                if (tid == 0) {
                    double a = 0;
                    unsigned long long b = 0;
                    for (int k = 0; k < kColT / 64; k++) { a += red[k]; b += redu[k]; }
                    fin.partial[(size_t) frame * kPartials + tile] = a;
                    fin.partial_u[(size_t) frame * kPartials + tile] = b;
                }
            }
        } else {
            for (int t = tid; t < n * CW; t += kColT) {
                int y = t / CW, c = t - y * CW;
                if (c < w) {
                    float v;
                    if (FWD) v = (y < sn ? E : O)[(y < sn ? y : y - sn) * CW + c];
                    else     v = (((y ^ cas) & 1) ? O : E)[(y >> 1) * CW + c];
                    buf[(size_t) y * W + x0 + c] = v;
                }
            }
        }
        __syncthreads();
    }
}

// The last inverse column pass as a STREAM: one thread per column walks down the rows with the four lifting steps as a
// register pipeline two samples deep, so every global access is a full row segment (64 columns = 256 B per wave) and
// nothing goes through LDS.  Possible because this pass does not write in place (its result becomes the fp32 field
// or only the statistics); the expressions, their order and the boundary forms are those of idwt_tile for a line that
// starts with a low-pass sample (cas == 0: every frame that is not a tile at an odd offset).
__device__ inline float fin_map(float v, float rng, float mn)
{
    long long q;
    if (v > 2147483647.0f) q = 65535;
    else if (v < -2147483648.0f) q = 0;
    else {
        q = (long long) __float2int_rn(v) + 32768;
        q = q < 0 ? 0 : (q > 65535 ? 65535 : q);
    }
    return ((float) (int) q / 65535.0f) * rng + mn;
}
constexpr int kFinT = 128;           // columns per workgroup
__global__ __launch_bounds__(kFinT) void k_j2k_cols_fin(const float *__restrict__ B, const J2kGeom *geom, int r, const FrameState *fs,
                                                         const int *active, J2kFinish fin)
{
    const int frame = blockIdx.y;
    if ((active && !active[frame]) || (fs && fs[frame].const_field)) return;
    const J2kGeom &g = j2k_frame_geom(geom, frame);
    const int W = g.W, n = g.rh[r], sn = g.rh[r - 1], cols = g.rw[r], dn = n - sn;
    const size_t n_pix = (size_t) W * g.H;
    const int col = blockIdx.x * kFinT + threadIdx.x;
    const bool live = col < cols;
    const float *lo = B + (size_t) frame * n_pix + (live ? col : 0), *hi = lo + (size_t) sn * W;
    const float *x = fin.data ? fin.data + (size_t) frame * n_pix + (live ? col : 0) : nullptr;
    float *d = fin.field(frame, n_pix);
    if (d && live) d += col;
    const float mn = fs[frame].minv, rng = fs[frame].maxv - fs[frame].minv;
    const float target = x ? fin.jf[frame].target : 0.0f;
    const float c1 = -kD, c2 = -kG, c3 = -kB, c4 = -kA;
    double acc = 0;
    unsigned int bad = 0;
    auto out = [&](int y, float v) {
        const float dv = fin_map(v, rng, mn);
        if (d) d[(size_t) y * W] = dv;
        if (x) {
            const float e = x[(size_t) y * W] - (dv + 0.0f);
            acc += (double) e;
            if (fabsf(e) > target) bad++;
        }
    };
    if (live) {
        // e0/o0: scaled inputs; e1, o1, e2, o2: after lifting steps 1..4 (istep_lo -kD, istep_hi -kG, istep_lo -kB, istep_hi -kA)
        float o0_prev = 0, e1_prev = 0, o1_prev = 0, o1_prev2 = 0, e2_prev = 0;
        // j runs two past the end: position j enters the pipeline, position j - 2 leaves it
        for (int j = 0; j < sn + 2; j++) {
            float e1 = 0, o0 = 0;
            if (j < sn) {
                const float e0 = lo[(size_t) j * W] * kK;
                if (j < dn) o0 = hi[(size_t) j * W] * kTwoInvK;
                if (j < dn) e1 = e0 + (((j == 0 ? o0 : o0_prev) + o0) * c1);
                else        e1 = e0 + (o0_prev * (c1 + c1));
            }
            // o1[j - 1] = o0[j - 1] + (e1[j - 1] + e1[j]) * c2, boundary form when e1[j] does not exist
            float o1 = 0;
            const int i1 = j - 1;
            if (i1 >= 0 && i1 < dn) o1 = (i1 + 1 < sn) ? o0_prev + ((e1_prev + e1) * c2) : o0_prev + (e1_prev * (c2 + c2));
            // e2[j - 1] = e1[j - 1] + (o1[j - 2] + o1[j - 1]) * c3   (o1[-1] := o1[0])
            float e2 = 0;
            if (i1 >= 0 && i1 < sn) e2 = (i1 < dn) ? e1_prev + (((i1 == 0 ? o1 : o1_prev) + o1) * c3) : e1_prev + (o1_prev * (c3 + c3));
            // o2[j - 2] = o1[j - 2] + (e2[j - 2] + e2[j - 1]) * c4
            const int i2 = j - 2;
            if (i2 >= 0) {
                if (i2 < sn) out(2 * i2, e2_prev);
                if (i2 < dn) out(2 * i2 + 1, (i2 + 1 < sn) ? o1_prev + ((e2_prev + e2) * c4) : o1_prev + (e2_prev * (c4 + c4)));
            }
            o0_prev = o0; e1_prev = e1; o1_prev2 = o1_prev; o1_prev = o1; e2_prev = e2;
        }
        (void) o1_prev2;
    }
    if (x) {
        __shared__ double red[kFinT / 64];
        __shared__ unsigned int redu[kFinT / 64];
        for (int k = 32; k >= 1; k >>= 1) { acc += __shfl_xor(acc, k); bad += __shfl_xor(bad, k); }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = acc; redu[threadIdx.x >> 6] = bad; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0;
            unsigned long long b = 0;
            for (int k = 0; k < kFinT / 64; k++) { a += red[k]; b += redu[k]; }
            fin.partial[(size_t) frame * kPartials + blockIdx.x] = a;
            fin.partial_u[(size_t) frame * kPartials + blockIdx.x] = b;
        }
    }
}

// The vertical half of the last inverse level as a two-deep register pipeline (see k_j2k_cols_fin): step(j) takes the low- and
// high-pass samples of vertical position j and hands out the finished samples of position j - 2.
struct ColPipe {
    float o0_prev = 0, e1_prev = 0, o1_prev = 0, e2_prev = 0;
    template <class Out>
    __device__ void step(int j, int sn, int dn, float lo_raw, float hi_raw, Out out)
    {
        const float c1 = -kD, c2 = -kG, c3 = -kB, c4 = -kA;
        float e1 = 0, o0 = 0;
        if (j < sn) {
            const float e0 = lo_raw * kK;
            if (j < dn) o0 = hi_raw * kTwoInvK;
            if (j < dn) e1 = e0 + (((j == 0 ? o0 : o0_prev) + o0) * c1);
            else        e1 = e0 + (o0_prev * (c1 + c1));
        }
        float o1 = 0;
        const int i1 = j - 1;
        if (i1 >= 0 && i1 < dn) o1 = (i1 + 1 < sn) ? o0_prev + ((e1_prev + e1) * c2) : o0_prev + (e1_prev * (c2 + c2));
        float e2 = 0;
        if (i1 >= 0 && i1 < sn) e2 = (i1 < dn) ? e1_prev + (((i1 == 0 ? o1 : o1_prev) + o1) * c3) : e1_prev + (o1_prev * (c3 + c3));
        const int i2 = j - 2;
        if (i2 >= 0) {
            if (i2 < sn) out(2 * i2, e2_prev);
            if (i2 < dn) out(2 * i2 + 1, (i2 + 1 < sn) ? o1_prev + ((e2_prev + e2) * c4) : o1_prev + (e2_prev * (c4 + c4)));
        }
        o0_prev = o0; e1_prev = e1; o1_prev = o1; e2_prev = e2;
    }
};

// The WHOLE last inverse level (horizontal then vertical, opj_dwt_decode_tile_97 at the top resolution) with
// dequantisation, field mapping and statistics in one pass over the data: a wave owns 60 sample pairs of the width
// (+ 2 pairs of halo on either side), one pair per lane; for every vertical position it synthesises the low-pass and
// the high-pass row horizontally in registers (neighbours through DPP wave shifts) and feeds two ColPipes, one per
// column of the pair.  Nothing is written but the fp32 field (if wanted) and the statistics: against the separate
// row and column passes this saves one write and one read of the frame per probe.  cas == 0 frames only.
//
// Round 3.  The kernel runs 34 times per frame and was thought to be HBM-bound with a 2x over-fetch; it was bound by
// VALU issue (~600 wave-instructions per iteration of 240 samples: boundary tests on every step, 26 LDS-crossbar
// shuffles, four correctly rounded divisions by 65535, four fp64 accumulations) AND moved 1.7-2x its bytes (the strips of
// a frame are 240-byte column ranges with halo, not multiples of a 128-byte line: the lines two strips share were fetched
// once per strip, because neighbouring strips ran as independent workgroups on different XCDs or microseconds apart).
// Now: (1) neighbouring strips of a (frame, piece) tile are waves of one workgroup, four at a time - same CU, same
// instruction stream, started together - so a line two of them share is fetched once; (2) the vertical positions away
// from the top and bottom edge take a step without any boundary test (ColPipe::step_interior), (3) neighbours come
// through v_mov_dpp wave shifts, (4) s / 65535.0f is fmaf(s, K_hi, s * K_lo) - equal to the correctly rounded quotient for
// every integer s in [0, 65535] (div65535_exact; checked exhaustively by ebcc_hip_selfcheck and tests/test_boundary.py),
// (5) frame samples and field leave / arrive as 8-byte pairs.
constexpr int kL5Pairs = 60;
constexpr int kL5MaxWaves = 16;       // strips (waves) of one workgroup
// FIN = false: the same pass for a lower level r - the synthesised samples go to `out` (pitch out_pitch) as they are, the
// next level's low-pass band.  Out of place: the level's output region covers its own LL input, so the levels alternate
// between two buffers.  LL comes from `ll` (pitch ll_pitch), or for level 1 from the decoder's output like the other bands.
struct J2kLevelIO {
    const float *ll; int ll_pitch; size_t ll_frame;      // low-pass input (null: from V, level 1)
    float *out; int out_pitch; size_t out_frame;         // FIN = false only
    int r;
};
// s / 65535.0f for an integer-valued s in [0, 65535], without the division: K_hi + K_lo = 1 / 65535 to 2^-49, one rounding
// at the end; the quotient's binary expansion repeats with period 16, so it is never within 2^-40 of a rounding boundary.
__host__ __device__ inline float div65535_exact(float s)
{
    const float k_hi = 1.5259021893143654e-05f, k_lo = 3.552767888909125e-15f;
#if defined(__HIP_DEVICE_COMPILE__)
    return __fmaf_rn(s, k_hi, __fmul_rn(s, k_lo));
#else
    return std::fma(s, k_hi, s * k_lo);
#endif
}
// fin_map without 64-bit integers: rint, level shift, clamp in fp32 (exact: below 2^23 every step is; beyond, the sum only
// has to stay outside [0, 65535], which it does), then the map to the field
__device__ inline float fin_map_fast(float v, float rng, float mn)
{
    const float q = __builtin_amdgcn_fmed3f(__builtin_rintf(v) + 32768.0f, 0.0f, 65535.0f);
    return div65535_exact(q) * rng + mn;
}
// the interior step of ColPipe: 2 <= j < dn (every neighbour exists); hands out the samples of position j - 2
struct ColPipe2 {
    float o0_prev = 0, e1_prev = 0, o1_prev = 0, e2_prev = 0;
    __device__ void interior(float lo_raw, float hi_raw, float &even, float &odd)
    {
        const float c1 = -kD, c2 = -kG, c3 = -kB, c4 = -kA;
        const float e0 = lo_raw * kK, o0 = hi_raw * kTwoInvK;
        const float e1 = e0 + ((o0_prev + o0) * c1);
        const float o1 = o0_prev + ((e1_prev + e1) * c2);
        const float e2 = e1_prev + ((o1_prev + o1) * c3);
        even = e2_prev;
        odd = o1_prev + ((e2_prev + e2) * c4);
        o0_prev = o0; e1_prev = e1; o1_prev = o1; e2_prev = e2;
    }
    // any position (the edges): the forms of ColPipe::step; has_even / has_odd say which of the two samples exist
    __device__ void edge(int j, int sn, int dn, float lo_raw, float hi_raw, float &even, float &odd, bool &has_even, bool &has_odd)
    {
        const float c1 = -kD, c2 = -kG, c3 = -kB, c4 = -kA;
        float e1 = 0, o0 = 0;
        if (j < sn) {
            const float e0 = lo_raw * kK;
            if (j < dn) o0 = hi_raw * kTwoInvK;
            if (j < dn) e1 = e0 + (((j == 0 ? o0 : o0_prev) + o0) * c1);
            else        e1 = e0 + (o0_prev * (c1 + c1));
        }
        float o1 = 0;
        const int i1 = j - 1;
        if (i1 >= 0 && i1 < dn) o1 = (i1 + 1 < sn) ? o0_prev + ((e1_prev + e1) * c2) : o0_prev + (e1_prev * (c2 + c2));
        float e2 = 0;
        if (i1 >= 0 && i1 < sn) e2 = (i1 < dn) ? e1_prev + (((i1 == 0 ? o1 : o1_prev) + o1) * c3) : e1_prev + (o1_prev * (c3 + c3));
        const int i2 = j - 2;
        has_even = i2 >= 0 && i2 < sn; has_odd = i2 >= 0 && i2 < dn;
        even = e2_prev;
        odd = (i2 + 1 < sn) ? o1_prev + ((e2_prev + e2) * c4) : o1_prev + (e2_prev * (c4 + c4));
        o0_prev = o0; e1_prev = e1; o1_prev = o1; e2_prev = e2;
    }
};
template <bool FIN>
__global__ __launch_bounds__(64 * kL5MaxWaves) void k_j2k_level5_fin(J2kLevelIO io, const int32_t *__restrict__ V, const J2kGeom *geom,
                                                        const FrameState *fs, const int *active, J2kFinish fin, int strips, int n_frames, int pieces)
{
    // workgroup = the strips (up to 16, one per wave) of tile blockIdx.y = piece * n_frames + frame: piece-major, so that
    // every frame's first pieces are dispatched first (the early exit below relies on that order for speed only)
    const int lane = (int) threadIdx.x & 63, strip = (int) blockIdx.x * ((int) blockDim.x >> 6) + ((int) threadIdx.x >> 6);
    const int frame = (int) blockIdx.y % n_frames, piece = (int) blockIdx.y / n_frames;
    if (strip >= strips) return;                                        // (no barrier anywhere below)
    if ((active && !active[frame]) || (fs && fs[frame].const_field)) return;
    const J2kGeom &g = j2k_frame_geom(geom, frame);
    const int r = FIN ? kJ2kRes - 1 : io.r;
    const int W = g.W, nh = g.rw[r], snh = g.rw[r - 1], dnh = nh - snh;          // horizontal: samples, low-pass, high-pass
    const int nv = g.rh[r], snv = g.rh[r - 1], dnv = nv - snv;                   // vertical
    const size_t n_pix = (size_t) W * g.H;
    const int i = strip * kL5Pairs + lane - 2;                                   // this lane's pair
    const bool has_e = i >= 0 && i < snh, has_o = i >= 0 && i < dnh;
    const bool owner = lane >= 2 && lane < 2 + kL5Pairs && has_e;                // (halo lanes compute, owners put out)
    const float *b = io.ll ? io.ll + (size_t) frame * io.ll_frame : nullptr;
    const unsigned lp = (unsigned) io.ll_pitch;
    const int32_t *v = V + (size_t) frame * n_pix;
    const float s_ll = 0.5f * g.bands[0].step_dec, s_hl = 0.5f * g.bands[3 * (r - 1) + 1].step_dec, s_lh = 0.5f * g.bands[3 * (r - 1) + 2].step_dec,
                s_hh = 0.5f * g.bands[3 * (r - 1) + 3].step_dec;
    const float *x = FIN && fin.data ? fin.data + (size_t) frame * n_pix : nullptr;
    float *d = FIN ? fin.field(frame, n_pix) : nullptr;
    float *o = FIN ? nullptr : io.out + (size_t) frame * io.out_frame;
    const float mn = fs[frame].minv, rng = fs[frame].maxv - fs[frame].minv;
    const float target = x ? fin.jf[frame].target : 0.0f;
    const float c1 = -kD, c2 = -kG, c3 = -kB, c4 = -kA;
    const bool first = i == 0, no_o = !(i < dnh), last_o = !(i + 1 < snh);
    // horizontal synthesis of one row: the pair's low-/high-pass inputs -> its two output samples (idwt_tile, cas 0)
    auto hsynth = [&](float e_raw, float o_raw, float &even, float &odd) {
        const float e0 = has_e ? e_raw * kK : 0.0f, o0 = has_o ? o_raw * kTwoInvK : 0.0f;
        const float o0l = lane_below(o0);
        const float e1 = no_o ? e0 + (o0l * (c1 + c1)) : e0 + (((first ? o0 : o0l) + o0) * c1);
        const float e1r = lane_above(e1);
        const float o1 = last_o ? o0 + (e1 * (c2 + c2)) : o0 + ((e1 + e1r) * c2);
        const float o1l = lane_below(o1);
        const float e2 = no_o ? e1 + (o1l * (c3 + c3)) : e1 + (((first ? o1 : o1l) + o1) * c3);
        const float e2r = lane_above(e2);
        even = e2;
        odd = last_o ? o1 + (e2 * (c4 + c4)) : o1 + ((e2 + e2r) * c4);
    };
    const int per = ceil_div(snv, pieces), ja = piece * per, jb = min(snv, ja + per);
    const size_t part = (size_t) frame * kPartials + (size_t) (strip + strips * piece);   // this wave's partial sums
    if (ja >= jb) {                                                     // (more pieces than positions: nothing to put out)
        if (FIN && x && lane == 0) { fin.partial[part] = 0.0; fin.partial_u[part] = 0; }
        return;
    }
    const int jstart = max(ja - 2, 0);
    // statistics-only probes of the rate search stop once the frame has gathered bad_limit samples above the target
    // (J2kFrame::bad_limit): the pieces are dispatched piece-major over all frames, so the later pieces of a frame that
    // is clearly infeasible at this rate leave without reading their strip.
    unsigned int limit = 0;
    // (the error sum is taken by every probe: a search's result often rests on a probe that was made to steer it - the final
    //  probe of the pure base-layer search is normally one already on record)
    const bool need_sum = true;
    if constexpr (FIN) {
        if (x && !d) {
            limit = fin.jf[frame].bad_limit;
            if (limit && __hip_atomic_load(&fin.jf[frame].bad_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= limit) {
                if (lane == 0) { fin.partial[part] = 0.0; fin.partial_u[part] = 0; }
                return;
            }
        }
    }
    double acc = 0;
    unsigned int bad = 0;
    ColPipe2 p0, p1;
    // every lane loads every step (halo and out-of-range lanes from clamped positions, their results are dropped), and
    // the inputs of step j + 1 are requested before step j is computed: no divergent control flow around the memory
    // accesses, one row of latency in flight.  32-bit offsets from uniform bases.
    const unsigned c_lo = (unsigned) min(max(i, 0), snh - 1), c_hi = (unsigned) snh + (unsigned) min(max(i, 0), max(dnh - 1, 0));
    const unsigned c0 = (unsigned) min(max(2 * i, 0), nh - 1), c1x = (unsigned) min(max(2 * i + 1, 0), nh - 1);
    // the 8-byte form reads columns (cp, cp + 1) of a row: the PAIR index is clamped, so that cp + 1 <= nh - 1 (nh is even
    // there) - clamping the column instead would let an out-of-range halo lane read one sample past the end of the row,
    // and on the last row of the last frame past the end of the caller's buffer
    const unsigned cp = 2u * (unsigned) min(max(i, 0), snh - 1);
    // 8-byte accesses of the output pair (and of the frame's samples beside it): even widths and 8-byte aligned bases
    const bool pair_io = (W & 1) == 0 && (nh & 1) == 0 && (n_pix & 1) == 0 && (((size_t) x | (size_t) d | (size_t) o) & 7) == 0 &&
                         (!FIN ? (io.out_pitch & 1) == 0 && (io.out_frame & 1) == 0 : true);
    float in_ll, in_hl, in_lh, in_hh;
    auto fetch = [&](int j, float &ll, float &hl, float &lh, float &hh) {
        const unsigned jl = (unsigned) min(j, snv - 1), jh = (unsigned) (snv + min(j, max(dnv - 1, 0)));
        const unsigned rl = jl * (unsigned) W, rh = jh * (unsigned) W;
        ll = b ? b[jl * lp + c_lo] : (float) v[rl + c_lo] * s_ll;
        hl = (float) v[rl + c_hi] * s_hl;
        lh = (float) v[rh + c_lo] * s_lh;
        hh = (float) v[rh + c_hi] * s_hh;
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // what happens to the two finished samples (columns 2i, 2i + 1) of output row y
    auto put_row = [&](bool emit, int y, float ev, float od, float xe, float xo, bool has_odd_col) {
        if constexpr (!FIN) {
            if (!emit) return;
            float *q = o + (size_t) ((unsigned) y * (unsigned) io.out_pitch + 2u * (unsigned) i);
            if (pair_io) *reinterpret_cast<f32x2 *>(q) = f32x2{ev, od};
            else { q[0] = ev; if (has_odd_col) q[1] = od; }
            return;
        } else {
            const float de = fin_map_fast(ev, rng, mn), dq = fin_map_fast(od, rng, mn);
            if (d && emit) {
                float *q = d + ((unsigned) y * (unsigned) W + 2u * (unsigned) i);
                if (pair_io) *reinterpret_cast<f32x2 *>(q) = f32x2{de, dq};
                else { q[0] = de; if (has_odd_col) q[1] = dq; }
            }
            if (x) {
                const float e0 = xe - (de + 0.0f), e1 = xo - (dq + 0.0f);
                const bool m0 = emit, m1 = emit && has_odd_col;
                bad += (m0 && fabsf(e0) > target) ? 1u : 0u;
                bad += (m1 && fabsf(e1) > target) ? 1u : 0u;
                if (need_sum) { acc += m0 ? (double) e0 : 0.0; acc += m1 ? (double) e1 : 0.0; }
            }
        }
    };
    fetch(jstart, in_ll, in_hl, in_lh, in_hh);
    for (int j = jstart; j < jb + 2; j++) {
        float n_ll, n_hl, n_lh, n_hh;
        fetch(j + 1, n_ll, n_hl, n_lh, n_hh);
        // the frame's samples at the four positions this step puts out (rows 2 (j - 2) and the next, this pair's columns)
        float x00 = 0, x01 = 0, x10 = 0, x11 = 0;
        if (x) {
            const unsigned y0 = (unsigned) min(max(2 * (j - 2), 0), nv - 1), y1 = (unsigned) min(max(2 * (j - 2) + 1, 0), nv - 1);
            if (pair_io) {
                const f32x2 a = *reinterpret_cast<const f32x2 *>(x + (y0 * (unsigned) W + cp)), c = *reinterpret_cast<const f32x2 *>(x + (y1 * (unsigned) W + cp));
                x00 = a.x; x01 = a.y; x10 = c.x; x11 = c.y;
            } else {
                x00 = x[y0 * (unsigned) W + c0]; x01 = x[y0 * (unsigned) W + c1x]; x10 = x[y1 * (unsigned) W + c0]; x11 = x[y1 * (unsigned) W + c1x];
            }
        }
        const bool emit = owner && j - 2 >= ja;                          // (positions before the piece belong to its neighbour)
        float lo_even = 0, lo_odd = 0, hi_even = 0, hi_odd = 0;
        if (j >= 2 && j < dnv) {                                         // (uniform) every vertical neighbour exists: no boundary forms
            hsynth(in_ll, in_hl, lo_even, lo_odd);                       // low-pass row j: LL from the previous level, HL from the decoder
            hsynth(in_lh, in_hh, hi_even, hi_odd);                       // high-pass row j: LH, HH
            float a0, a1, b0, b1;
            p0.interior(lo_even, hi_even, a0, a1);                       // column 2i: rows 2 (j - 2), 2 (j - 2) + 1
            p1.interior(lo_odd, hi_odd, b0, b1);                         // column 2i + 1
            put_row(emit, 2 * (j - 2), a0, b0, x00, x01, has_o);
            put_row(emit, 2 * (j - 2) + 1, a1, b1, x10, x11, has_o);
        } else {
            if (j < snv) hsynth(in_ll, in_hl, lo_even, lo_odd);
            if (j < dnv) hsynth(in_lh, in_hh, hi_even, hi_odd);
            float a0, a1, b0, b1;
            bool he, ho, he1, ho1;
            p0.edge(j, snv, dnv, lo_even, hi_even, a0, a1, he, ho);
            p1.edge(j, snv, dnv, lo_odd, hi_odd, b0, b1, he1, ho1);
            // (he / ho are uniform: whether rows 2 (j - 2) and 2 (j - 2) + 1 exist)
            if (he) put_row(emit, 2 * (j - 2), a0, b0, x00, x01, has_o);
            if (ho) put_row(emit, 2 * (j - 2) + 1, a1, b1, x10, x11, has_o);
            (void) he1; (void) ho1;
        }
        in_ll = n_ll; in_hl = n_hl; in_lh = n_lh; in_hh = n_hh;
    }
    if (x) {
        for (int k = 32; k >= 1; k >>= 1) { acc += __shfl_xor(acc, k); bad += __shfl_xor(bad, k); }
        if (lane == 0) {
            fin.partial[part] = acc;
            fin.partial_u[part] = bad;
            if (limit && bad) atomicAdd(&fin.jf[frame].bad_seen, bad);
        }
    }
}

template <typename K>
void big_lds(K k, size_t bytes)
{
    if (bytes > 48 * 1024)
        EBCC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int) bytes));
}

// most rows any tile position has at resolution r
static int max_rows(const J2kBuffers &jb, int r)
{
    int n = 0;
    for (const J2kGeom &t : jb.geoms) n = std::max(n, t.rh[r]);
    return n;
}

// returns the number of column tiles (= statistics partials per frame when FIN)
template <bool FWD, bool FIN = false>
int dwt_cols(float *B, const J2kBuffers &jb, int r, int n_frames, const FrameState *fs, const int *active, hipStream_t s,
             J2kFinish fin = J2kFinish{})
{
    const int n = max_rows(jb, r), cols = jb.geom.rw[r];
    if (n <= 1) return 0;
    if constexpr (FIN && !FWD) {
        // plain frames (every column starts with a low-pass sample): the streaming form
        if (jb.geom.period == 1 && jb.geom.ry0[r] % 2 == 0) {
            hipLaunchKernelGGL(k_j2k_cols_fin, dim3(ceil_div(cols, kFinT), n_frames), dim3(kFinT), 0, s, B, jb.d_geom, r, fs, active, fin);
            return ceil_div(cols, kFinT);
        }
    }
    if ((size_t) n * 32 * 4 <= 156 * 1024) {
        size_t lds = (size_t) n * 32 * 4;
        auto k = k_j2k_cols<FWD, 32, FIN>;
        big_lds(k, lds);
        hipLaunchKernelGGL(k, dim3(ceil_div(cols, 32), n_frames), dim3(kColT), lds, s, B, jb.d_geom, r, fs, active, fin);
        return ceil_div(cols, 32);
    }
    size_t lds = (size_t) n * 16 * 4;
    auto k = k_j2k_cols<FWD, 16, FIN>;
    big_lds(k, lds);
    hipLaunchKernelGGL(k, dim3(ceil_div(cols, 16), n_frames), dim3(kColT), lds, s, B, jb.d_geom, r, fs, active, fin);
    return ceil_div(cols, 16);
}
template <bool FWD>
void dwt_rows(float *B, const int32_t *V, const J2kBuffers &jb, int r, int n_frames, const FrameState *fs, const int *active, hipStream_t s)
{
    const int n = jb.geom.rw[r], rows = max_rows(jb, r);
    hipLaunchKernelGGL(k_j2k_rows<FWD>, dim3(min(rows, 96), n_frames), dim3(kRowT), (size_t) n * 4, s, B, V, jb.d_geom, r, fs, active);
}

// ================================================================================================
// quantisation -> Q6 + bit-plane row masks.  One workgroup per GROUP of 64 code-blocks (the unit the masks are
// interleaved in): for every row index y the waves ballot the y-th row of each code-block of the group into LDS, and
// the 64 masks of a (plane, y) - one per code-block, 512 contiguous bytes - leave as one full line.
// ================================================================================================
constexpr int kQuantMasks = kJ2kMaxPlanes + 1 + (kJ2kMaxPlanes + 2);     // bit-planes, sign, suffix-ORs
__global__ __launch_bounds__(256) void k_quantize(const float *__restrict__ B, int32_t *__restrict__ Q6,
                                                   unsigned long long *__restrict__ BP, unsigned long long *__restrict__ SGN,
                                                   unsigned long long *__restrict__ SUF, int *__restrict__ blkmax, const J2kGeom *geom, const J2kBlock *blocks,
                                                   const FrameState *fs, int total)
{
    __shared__ __attribute__((aligned(16))) unsigned long long m[kQuantMasks][64];   // [mask][code-block of the group]
    __shared__ int smax[64];
    // the 64 code-blocks of the group: element offset of (0, 0) in B / Q6 (-1: none), extent, row pitch, step size
    __shared__ long long s_base[64];
    __shared__ int s_w[64], s_h[64], s_pitch[64];
    __shared__ float s_step[64];
    const size_t grp = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = geom->stride;
    if (threadIdx.x < 64) {
        const int k = threadIdx.x, gid = (int) (grp * 64) + k;
        smax[k] = 0;
        s_base[k] = -1; s_w[k] = 0; s_h[k] = 0; s_pitch[k] = 0; s_step[k] = 1.0f;
        if (gid < total) {
            const int frame = gid / nb;
            if (!fs[frame].const_field) {
                const J2kBlock blk = j2k_frame_blocks(geom, blocks, frame)[gid - frame * nb];   // (slots past the tile's last code-block hold empty ones)
                const J2kGeom &g = j2k_frame_geom(geom, frame);
                s_base[k] = (long long) ((size_t) frame * ((size_t) g.W * g.H) + (size_t) blk.y * g.W + blk.x);
                s_w[k] = blk.w; s_h[k] = blk.h; s_pitch[k] = g.W; s_step[k] = g.bands[blk.band].step_enc;
            }
        }
    }
    __syncthreads();
    unsigned long long *bp = BP + grp * kJ2kMaxPlanes * 64 * 64, *sg = SGN + grp * 64 * 64, *su = SUF + grp * (kJ2kMaxPlanes + 2) * 64 * 64;
    // a wave owns code-blocks wave, wave + 4, ...: their 16 row segments of the next row are requested before the current
    // row is worked on (the kernel used to wait for every 256-byte segment in turn: 14 % of the HBM roof)
    constexpr int kPer = 16;
    float cur[kPer], nxt[kPer];
    auto fetch = [&](int row, float (&v)[kPer]) {
#pragma unroll
        for (int jj = 0; jj < kPer; jj++) {
            const int k = wave + 4 * jj;
            const bool in = s_base[k] >= 0 && row < s_h[k] && lane < s_w[k];
            v[jj] = in ? B[s_base[k] + (long long) row * s_pitch[k] + lane] : 0.0f;
        }
    };
    fetch(0, cur);
    for (int row = 0; row < 64; row++) {
        if (row + 1 < 64) fetch(row + 1, nxt);
#pragma unroll
        for (int jj = 0; jj < kPer; jj++) {
            const int k = wave + 4 * jj;
            const bool in = s_base[k] >= 0 && row < s_h[k] && lane < s_w[k];
            int q6 = 0;
            if (in) {
                q6 = __float2int_rn((cur[jj] / s_step[k]) * 64.0f);     // lrintf((c / stepsize) * 64), opj_t1_encode_cblks
                Q6[s_base[k] + (long long) row * s_pitch[k] + lane] = q6;
            }
            const int a6 = q6 < 0 ? -q6 : q6;
            const int mx = wave_max_nonneg(a6);                          // (DPP: six shuffles through the LDS crossbar before)
            const unsigned long long sgn = __ballot(q6 < 0);
            const int a = a6 >> 6;
            // planes above the row's top magnitude bit are empty: no ballots for them
            const int top = mx >> 6, p_top = top ? 32 - __builtin_clz(top) : 0;
            if (lane < kJ2kMaxPlanes && lane >= p_top) { m[lane][k] = 0; m[kJ2kMaxPlanes + 1 + lane][k] = 0; }
            unsigned long long suf = 0;                                  // OR of the planes >= p: "some bit at or above p"
            if (lane == 0) {
                if (mx > smax[k]) smax[k] = mx;                          // (this wave owns code-block k)
                m[kJ2kMaxPlanes][k] = sgn;
                m[kJ2kMaxPlanes + 1 + kJ2kMaxPlanes + 1][k] = 0;
                m[kJ2kMaxPlanes + 1 + kJ2kMaxPlanes][k] = 0;
            }
            for (int p = p_top - 1; p >= 0; p--) {
                const unsigned long long bits = __ballot((a >> p) & 1);
                suf |= bits;
                if (lane == 0) { m[p][k] = bits; m[kJ2kMaxPlanes + 1 + p][k] = suf; }
            }
        }
        __syncthreads();
        // the 55 lines of this row index leave as 16-byte pieces: 32 lanes per line of 512 bytes
        for (int t = threadIdx.x; t < kQuantMasks * 32; t += 256) {
            const int i = t >> 5, k2 = (t & 31) * 2;
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            const u64x2 v = *reinterpret_cast<const u64x2 *>(&m[i][k2]);
            unsigned long long *line = i < kJ2kMaxPlanes ? bp + ((size_t) i * 64 + row) * 64
                                     : (i == kJ2kMaxPlanes ? sg + (size_t) row * 64 : su + ((size_t) (i - kJ2kMaxPlanes - 1) * 64 + row) * 64);
            *reinterpret_cast<u64x2 *>(line + k2) = v;
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < kPer; jj++) cur[jj] = nxt[jj];
    }
    if (threadIdx.x < 64) {
        const int gid = (int) (grp * 64) + (int) threadIdx.x;
        if (gid < total) blkmax[gid] = smax[threadIdx.x];
    }
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
// tier-1 encoder: one code-block per lane (t1_core.hpp)
// ================================================================================================
// All per-code-block arrays are lane-interleaved inside a group of 64 code-blocks, and a wave never straddles
// two groups, so every access is (group base: uniform, lives in SGPRs) + (32-bit lane offset: one VGPR) -
// the saddr + voffset form of the global instructions - instead of a 64-bit pointer per lane and array.
struct DevStore {
    unsigned char *st;            // group base of the state words
    const unsigned char *bp;      // group base of the bit-plane masks
    const unsigned char *sg;
    unsigned char *sps;           // group base of the propagation-significance masks
    uint32_t lane8;               // lane * 8
    __device__ unsigned long long &at(unsigned char *b, int row) const { return *(unsigned long long *) (b + ((uint32_t) row * 512u + lane8)); }
    __device__ unsigned long long ld(const unsigned char *b, int row) const { return *(const unsigned long long *) (b + ((uint32_t) row * 512u + lane8)); }
    __device__ unsigned long long &S(int y) { return at(st, y + 1); }
    __device__ unsigned long long &NEG(int y) { return at(st, 66 + y); }
    __device__ unsigned long long &VIS(int y) { return at(st, 130 + y); }
    __device__ unsigned long long &REF(int y) { return at(st, 194 + y); }
    __device__ unsigned long long &SPS(int y) { return at(sps, y); }
    __device__ unsigned long long SGN(int y) const { return y < 64 ? ld(sg, y) : 0ull; }
    __device__ unsigned long long BP(int plane, int y) const { return y < 64 ? ld(bp, plane * 64 + y) : 0ull; }
};
struct DevSink {
    uint8_t *p; int cap; int *overflow;
    __device__ void put(int i, uint8_t b) { if (i < cap) p[i] = b; else *overflow = 1; }
};
struct DevAt {
    const uint8_t *p; int cap;
    __device__ uint8_t operator()(int i) const { return i < cap ? p[i] : (uint8_t) 0; }
};

// MQ-decoder checkpoints for the rate probes: the decoder registers at the start of every stripe of every
// coding pass (and the visited masks after every propagation pass) so that a probe restarts a stripe or two
// before the point where its truncated segment ends (j2k_rate.hip).  They are derived from the encoder's own registers - see t1::finalize_checkpoints - so no
// decode of the segment is needed.
struct CkObserver {
    J2kCkptView ck;                // checkpoint slots of this code-block (slot = pass * 16 + stripe)
    unsigned char *visp;           // group base of the per-plane visited masks
    uint32_t lane8;                // lane * 8
    int cur = 0;
    template <class Mq>
    __device__ void pass_start(int p, const Mq &) { cur = p * 16; }
    template <class Mq>
    __device__ void stripe_start(int y0, const Mq &m)                       // (single-phase encoder: packed contexts)
    {
        uint32_t x[5];
        m.cx.to_bytes(x);
        store(0, cur + (y0 >> 2), m.a, m.c & 0xFFFFu, m.shifts, x);
    }
    // CkArray of the MQ pass (interval chain: a, shifts, contexts; code chain: the low half of C) and of
    // t1::finalize_checkpoints.  Record: { a | ct << 16, shifts -> pos, five context words, c16 -> c } (j2k.hpp)
    __device__ void store(int p, int s, uint32_t a, uint32_t c16, uint32_t shifts, const uint32_t x[5])
    {
        uint4 *r = (uint4 *) ck.slot((uint32_t) (p * 16 + s));
        r[0] = make_uint4(a, shifts, x[0], x[1]);
        r[1] = make_uint4(x[2], x[3], x[4], c16);
    }
    __device__ void store_interval(int p, int s, uint32_t a, uint32_t shifts, const uint32_t x[5])
    {
        uint32_t *r = ck.slot((uint32_t) (p * 16 + s));
        *(uint4 *) r = make_uint4(a, shifts, x[0], x[1]);
        r[4] = x[2]; r[5] = x[3]; r[6] = x[4];
    }
    __device__ void store_code(int p, int s, uint32_t c16) { ck.slot((uint32_t) (p * 16 + s))[7] = c16; }
    __device__ uint32_t shifts(int p, int s) const { return ck.slot((uint32_t) (p * 16 + s))[1]; }
    __device__ uint32_t c16(int p, int s) const { return ck.slot((uint32_t) (p * 16 + s))[7]; }
    __device__ void finish(int p, int s, uint32_t c, int ct, int pos)
    {
        uint32_t *r = ck.slot((uint32_t) (p * 16 + s));
        r[0] = (r[0] & 0xFFFFu) | ((uint32_t) ct << 16); r[1] = (uint32_t) pos; r[7] = c;
    }
    template <class Store>
    __device__ void sigprop_done(int bp, Store &st)
    {
        for (int y = 0; y < 64; y++) *(unsigned long long *) (visp + ((uint32_t) (bp * 64 + y) * 512u + lane8)) = st.VIS(y);
    }
};
struct CkSrc {
    // byte source with an 8-byte register window: one aligned load per 8 bytes instead of two byte loads per BYTEIN
    const uint8_t *p; int n;
    unsigned long long win = 0; int base = -16;
    __device__ uint32_t get(int i)
    {
        if (i >= n) return 0xFFu;
        const int b = i & ~7;
        if (b != base) { win = *reinterpret_cast<const unsigned long long *>(p + b); base = b; }
        return (uint32_t) (win >> (8 * (i & 7))) & 0xFFu;
    }
};

// ---- single-phase encoder: passes and MQ coder in one kernel
__global__ __launch_bounds__(64) void k_t1_encode(unsigned long long *T1S, const unsigned long long *BP,
                                                   const unsigned long long *SGN, unsigned long long *SPS, const int *blkmax, int *numbps,
                                                   int *totalpasses, int *cblk_len, int *rates, uint8_t *cblk_bytes,
                                                   void *ckpt, unsigned long long *VISP,
                                                   const J2kGeom *geom, const J2kBlock *blocks, const FrameState *fs,
                                                   J2kFrame *jf, int total, int lpw)
{
    EBCC_LDS_MQ_TABLE(tab);
    if ((int) threadIdx.x >= lpw) return;                              // see t1_lanes_per_wave()
    const int gid0 = blockIdx.x * lpw;                                 // lpw divides 64: the wave stays inside one group
    const int gid = gid0 + threadIdx.x;
    if (gid >= total) return;
    const int nb = geom->stride;
    const int frame = gid / nb, bi = gid - frame * nb;
    if (fs[frame].const_field) return;
    blocks = j2k_frame_blocks(geom, blocks, frame);
    geom = &j2k_frame_geom(geom, frame);
    const J2kBlock blk = blocks[bi];
    const int orient = geom->bands[blk.band].orient;
    const int m = blkmax[gid];
    int P = m ? (31 - __clz(m)) + 1 - 6 : 0;                         // opj_t1_encode_cblk: numbps
    numbps[gid] = P;
    if (P <= 0) { totalpasses[gid] = 0; cblk_len[gid] = 0; return; }
    const size_t grp = (size_t) (gid0 >> 6);
    const uint32_t lane8 = (uint32_t) ((gid0 & 63) + threadIdx.x) * 8u;
    DevStore st{(unsigned char *) (T1S + grp * kT1StateWords * 64), (const unsigned char *) (BP + grp * kJ2kMaxPlanes * 64 * 64),
                (const unsigned char *) (SGN + grp * 64 * 64), (unsigned char *) (SPS + grp * 64 * 64), lane8};
    uint8_t *out = cblk_bytes + (size_t) gid * kJ2kCblkBytes;
    CkObserver obs{J2kCkptView::of(ckpt, (size_t) gid), (unsigned char *) (VISP + grp * kJ2kMaxPlanes * 64 * 64), lane8};
    t1::EncodeResult r = t1::encode_block_observed(st, DevSink{out, kJ2kCblkBytes, &jf[frame].overflow}, DevAt{out, kJ2kCblkBytes},
                                                   blk.w, blk.h, orient, P, rates + (size_t) gid * kJ2kMaxPasses, obs, tab);
    totalpasses[gid] = r.totalpasses;
    cblk_len[gid] = r.length;
    __threadfence();                                                    // the sweep below re-reads this lane's own bytes
    t1::finalize_checkpoints(obs, r.totalpasses, (blk.h + 3) >> 2, CkSrc{out, r.length < kJ2kCblkBytes ? r.length : kJ2kCblkBytes});
}

// ================================================================================================
// segmented two-phase encoder (t1_core.hpp): masks + counts, segment offsets, decisions, arithmetic coding
// ================================================================================================
// the encoder's row masks of one code-block (lane-interleaved inside its group)
struct DevMasks {
    const unsigned char *bp, *suf, *sg;    // group bases
    unsigned char *visp;
    uint32_t lane8;
    __device__ unsigned long long ld(const unsigned char *b, int row) const { return *(const unsigned long long *) (b + ((uint32_t) row * 512u + lane8)); }
    __device__ unsigned long long BP(int p, int y) const { return ld(bp, p * 64 + y); }
    __device__ unsigned long long SUF(int p, int y) const { return ld(suf, p * 64 + y); }
    __device__ unsigned long long SGN(int y) const { return ld(sg, y); }
    __device__ unsigned long long VISP(int p, int y) const { return ld(visp, p * 64 + y); }
};
struct Tier1Lane {                 // what every tier-1 kernel derives for its lane's code-block
    int gid, frame, P, w, h, orient;
    bool live;
};
__device__ inline Tier1Lane tier1_lane(int gid, int total, const int *blkmax, const J2kGeom *geom, const J2kBlock *blocks, const FrameState *fs)
{
    Tier1Lane l{gid, 0, 0, 0, 0, 0, false};
    if (gid >= total) return l;
    const int nb = geom->stride;
    l.frame = gid / nb;
    if (fs[l.frame].const_field) return l;
    const J2kBlock blk = j2k_frame_blocks(geom, blocks, l.frame)[gid - l.frame * nb];
    const int m = blkmax[gid];
    l.P = m ? (31 - __clz(m)) + 1 - 6 : 0;                             // opj_t1_encode_cblk: numbps
    l.w = blk.w; l.h = blk.h;
    l.orient = j2k_frame_geom(geom, l.frame).bands[blk.band].orient;
    l.live = l.P > 0;
    return l;
}

// ---- masks and counts: one code-block per lane, planes from the top; mask arithmetic only
struct ScanOutDev {
    unsigned char *vp, *sp;        // group bases of VISP and SPS
    std::uint16_t *lens;           // this lane's column of the group's segment lengths
    uint32_t lane8;
    __device__ void visp_row(unsigned char *b, int row, unsigned long long v) const { *(unsigned long long *) (b + ((uint32_t) row * 512u + lane8)) = v; }
    __device__ void visp(int p, int y, unsigned long long m) const { visp_row(vp, p * 64 + y, m); }
    __device__ void sps_or(int y, unsigned long long n) const { *(unsigned long long *) (sp + ((uint32_t) y * 512u + lane8)) |= n; }
    __device__ void len(int seg, uint32_t n) const { lens[(uint32_t) seg * 64u] = (std::uint16_t) n; }
};
__global__ __launch_bounds__(64) void k_t1_scan(const unsigned long long *BP, const unsigned long long *SUF, const unsigned long long *SGN,
                                                 unsigned long long *SPS, unsigned long long *VISP, const int *blkmax, int *numbps,
                                                 int *totalpasses, int *cblk_len, std::uint16_t *seglen, const J2kGeom *geom,
                                                 const J2kBlock *blocks, const FrameState *fs, int total)
{
    const size_t grp = blockIdx.x;
    const int gid = (int) (grp * 64) + (int) threadIdx.x;
    const Tier1Lane l = tier1_lane(gid, total, blkmax, geom, blocks, fs);
    if (gid < total) { numbps[gid] = l.P; totalpasses[gid] = l.live ? 3 * l.P - 2 : 0; cblk_len[gid] = 0; }
    if (!l.live) return;
    const uint32_t lane8 = threadIdx.x * 8u;
    DevMasks M{(const unsigned char *) (BP + grp * kJ2kMaxPlanes * 64 * 64), (const unsigned char *) (SUF + grp * (kJ2kMaxPlanes + 2) * 64 * 64),
               (const unsigned char *) (SGN + grp * 64 * 64), (unsigned char *) (VISP + grp * kJ2kMaxPlanes * 64 * 64), lane8};
    ScanOutDev out{M.visp, (unsigned char *) (SPS + grp * 64 * 64), seglen + grp * kJ2kSegCount * 64 + threadIdx.x, lane8};
    t1::scan_block(M, out, l.P, l.w, l.h, l.orient);
}

// ---- row offsets: every segment of a code-block takes seg_rows(decisions) 16-byte rows of the block's stream, in coding
// order; the counts k_t1_scan left become the segments' first rows (in place), the total goes to lanerows
__global__ __launch_bounds__(64) void k_t1_rowoffs(std::uint16_t *seglen, uint32_t *lanerows, const int *blkmax, J2kFrame *jf,
                                                    const J2kGeom *geom, const J2kBlock *blocks, const FrameState *fs, int total,
                                                    uint32_t sym_rows)
{
    const size_t grp = blockIdx.x;
    const int gid = (int) (grp * 64) + (int) threadIdx.x;
    const Tier1Lane l = tier1_lane(gid, total, blkmax, geom, blocks, fs);
    std::uint16_t *L = seglen + grp * kJ2kSegCount * 64 + threadIdx.x;
    uint32_t off = 0;
    if (l.live) {
        const int nstr = (l.h + 3) >> 2;
        // sixteen counts (the stripes of one pass) are requested together, then turned into offsets: the counts do not depend
        // on the running offset, but read and written in place one at a time every load waited for the store before it
        for (int seg0 = t1::seg_index(l.P - 1, 2, 0) & ~15; seg0 < kJ2kSegCount; seg0 += 16) {
            if (!t1::seg_valid(l.P, nstr, t1::seg_plane(seg0), t1::seg_type(seg0), 0)) continue;      // (a pass the code-block does not have)
            uint32_t n[16];
#pragma unroll
            for (int k = 0; k < 16; k++) n[k] = k < nstr ? L[(uint32_t) (seg0 + k) * 64u] : 0u;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < nstr) { L[(uint32_t) (seg0 + k) * 64u] = (std::uint16_t) off; off += t1::seg_rows(n[k]); }
        }
    }
    if (gid < total) lanerows[gid] = off;
    if (off > sym_rows) atomicOr(&jf[l.frame].overflow, 2);             // (cannot happen for sym_rows = kJ2kSymRows: the host would re-run tier-1 with the single-kernel encoder)
}

// ---- decisions: one wave per (group of 64 code-blocks, bit-plane), a code-block per lane; the decisions of a segment
// go through a per-lane byte ring in LDS and leave as the 16-byte rows of the block's stream: row r of lane l of a
// group at ((r * 64) + l) * 16, so that the MQ pass reads one contiguous KB per step
// Rows are stored in PAIRS: rows 2j and 2j + 1 of lane l at (j * 64 + l) * 32 (+ 16 for the odd row), so that the two rows a
// lane writes one after the other fill one 32-byte sector (16-byte pieces at 1 KB stride were partial-sector writes: the emit
// wrote 23.9 MB per frame for 11 MB of rows) and the MQ pass, which takes two rows per step, reads 32 contiguous bytes per lane.
__device__ inline size_t sym_row_offset(uint32_t row) { return (size_t) (row >> 1) * 2048 + (size_t) (row & 1u) * 16; }   // + lane * 32
__host__ __device__ inline size_t sym_group_bytes(uint32_t sym_rows) { return (size_t) ((sym_rows + 1u) & ~1u) * 1024; }
struct SegEmDev {
    unsigned char *ring;           // LDS: this lane's 64 bytes (lane stride 68 bytes = 17 banks: lanes at the same offset never conflict)
    uint8_t *sym;                  // the group's rows + lane * 32
    const std::uint16_t *rowoff;   // this lane's column of the segments' first rows
    uint32_t cap;                  // rows the lane may write
    uint32_t cnt = 0, fl = 0;      // decisions of the segment so far, rows written
    uint32_t row0 = 0;
    __device__ void begin(int seg) { cnt = 0; fl = 0; row0 = rowoff[(uint32_t) seg * 64u]; }
    __device__ void emit_if(bool on, uint32_t ctx, uint32_t d)
    {
        ring[cnt & 63u] = (unsigned char) (ctx | (d << 5));             // (overwritten by the next decision unless `on`)
        cnt += on ? 1u : 0u;
    }
    __device__ void piece(uint32_t keep)                                 // keep: decisions in this row (the rest is padding)
    {
        const uint32_t *q = (const uint32_t *) (ring + ((fl & 3u) << 4));
        uint32_t w[4] = {q[0], q[1], q[2], q[3]};
        if (keep < 16u) {
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int nv = min(max((int) keep - 4 * d, 0), 4);
                const uint32_t m = nv ? 0xFFFFFFFFu >> (32 - 8 * nv) : 0u;
                w[d] = (w[d] & m) | ((t1::kRowPad * 0x01010101u) & ~m);
            }
        }
        if (fl == 0) w[0] |= t1::kRowStart;
        if (row0 + fl < cap) *(uint4 *) (sym + sym_row_offset(row0 + fl)) = make_uint4(w[0], w[1], w[2], w[3]);
        fl++;
    }
    __device__ void column_end() { if ((cnt >> 4) != fl) piece(16u); }  // (a column adds at most 11 decisions)
    __device__ void end()
    {
        if ((cnt >> 4) != fl) piece(16u);
        if ((cnt & 15u) || cnt == 0) piece(cnt & 15u);
    }
};

__global__ __launch_bounds__(64) void k_t1_emit(const unsigned long long *BP, const unsigned long long *SUF, const unsigned long long *SGN,
                                                 unsigned long long *VISP, const int *blkmax, uint8_t *SYM, const std::uint16_t *seglen,
                                                 const uint32_t *lanerows, const J2kGeom *geom, const J2kBlock *blocks,
                                                 const FrameState *fs, int total, uint32_t sym_rows)
{
    __shared__ uint32_t ring[17 * 64];
    const size_t grp = blockIdx.x;
    const int p = (int) blockIdx.y;
    const int gid = (int) (grp * 64) + (int) threadIdx.x;
    const Tier1Lane l = tier1_lane(gid, total, blkmax, geom, blocks, fs);
    if (!l.live || p >= l.P || lanerows[gid] > sym_rows) return;
    const uint32_t lane8 = threadIdx.x * 8u;
    DevMasks M{(const unsigned char *) (BP + grp * kJ2kMaxPlanes * 64 * 64), (const unsigned char *) (SUF + grp * (kJ2kMaxPlanes + 2) * 64 * 64),
               (const unsigned char *) (SGN + grp * 64 * 64), (unsigned char *) (VISP + grp * kJ2kMaxPlanes * 64 * 64), lane8};
    SegEmDev em{(unsigned char *) ring + threadIdx.x * 68u, SYM + grp * sym_group_bytes(sym_rows) + threadIdx.x * 32u,
                seglen + grp * kJ2kSegCount * 64 + threadIdx.x, sym_rows};
    const int nstr = (l.h + 3) >> 2;
    for (int s = 0; s < nstr; s++) t1::emit_stripe_segments(M, em, l.P, p, s, l.w, l.h, l.orient);
}

// ---- arithmetic coding of the row streams: one wave per group, a code-block per lane (t1::mq_encode_rows)
// The MQ pass runs as TWO waves per group of 64 code-blocks (t1_core.hpp: mq_rows_interval / MqCodeChain):
//   wave 0 - the interval chain: reads the decision rows (from LDS), keeps the context states, hands a 16-bit word
//            per decision over through LDS, stores the interval half of the checkpoints;
//   wave 1 - copies the rows from HBM into LDS kRowChunk rows ahead, and runs the code chain kRowChunk rows behind
//            wave 0: output bytes, pass rates, the code half of the checkpoints.
// One workgroup barrier per chunk of kRowChunk rows keeps the three stages (load chunk q + 1 | interval chunk q | code
// chunk q - 1) in step.  Wave 0 never waits for vector memory: its global accesses are stores only (on gfx9 a load's data
// would wait for every older store of the wave: one counter, in issue order).
constexpr int kRowChunk = 2;
struct RowSrcDev {
    uint32_t buf;                  // LDS byte address of this lane's 16 bytes of row 0 of buffer 0 (row stride 1 KB, buffers kRowChunk KB apart)
    uint32_t n, wn;                // rows of this lane, of the longest lane of the wave
    __device__ uint32_t rows() const { return n; }
    __device__ uint32_t wave_rows() const { return wn; }
    __device__ void sync(uint32_t row) const { if (row % kRowChunk == 0) __syncthreads(); }
    __device__ void finish() const { __syncthreads(); __syncthreads(); }                 // (the code chain is two chunks behind)
    __device__ void load(uint32_t row, uint32_t w[4]) const
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = *(const __attribute__((address_space(3))) u32x4 *) (uintptr_t) (buf + (row % (2 * kRowChunk)) * 1024u);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    }
};
struct HandDev {
    uint32_t buf;                  // LDS byte address of this lane's word 0 of row 0 of buffer 0: [row % (2 kRowChunk)][j][lane] 32-bit words
    __device__ void put(uint32_t row, int j, uint32_t word) const
    {
        *(__attribute__((address_space(3))) uint32_t *) (uintptr_t) (buf + ((row % (2 * kRowChunk)) * 16u + (uint32_t) j) * 256u) = word;
    }
    __device__ uint32_t get(uint32_t row, int j) const
    {
        return *(const __attribute__((address_space(3))) uint32_t *) (uintptr_t) (buf + ((row % (2 * kRowChunk)) * 16u + (uint32_t) j) * 256u);
    }
};
struct MqSinkLds {
    // coded bytes go into a 64-byte ring per lane in LDS (lane stride 68 bytes: lanes at the same offset never
    // conflict) and leave as whole 32-byte sectors at the uniform points of the loop (every 16 decisions: at most 30
    // new bytes on top of at most 31 waiting ones)
    unsigned char *ring;           // LDS, this lane's 64 bytes
    uint8_t *out; int *overflow;
    int fl = 0;                    // bytes written out (multiple of 32)
    __device__ void put(int i, uint32_t b) { ring[(uint32_t) i & 63u] = (unsigned char) b; }    // (i == -1 lands in byte 63 and is overwritten later)
    __device__ void sector()
    {
        const uint32_t *q = (const uint32_t *) (ring + ((uint32_t) fl & 63u));
        if (fl + 32 <= kJ2kCblkBytes) {
            uint4 *o = (uint4 *) (out + fl);
            o[0] = make_uint4(q[0], q[1], q[2], q[3]);
            o[1] = make_uint4(q[4], q[5], q[6], q[7]);
        } else *overflow = 1;
        fl += 32;
    }
    __device__ void row_end(int n) { while (fl + 32 <= n) sector(); }
    __device__ void finish(int n) { row_end(n); if (fl < n) sector(); }
};

// The 64 code-blocks of a wave are coded in lock-step, row by row: the wave lasts as long as its code-block with the most
// rows (the mean over the waves of that maximum was 25 % above the mean row count with the code-blocks in their natural
// order - a quarter of the pass's instructions coded padding).  So the waves take the code-blocks in order of falling row
// count (a counting sort over 128 classes of 32 rows: k_mq_hist / k_mq_offsets / k_mq_place, as the decoder orders its
// chains): lanes of a wave hold code-blocks of nearly the same length, the longest chains start first, and the code-blocks
// without rows gather in the last waves, which leave at once.  Results are per code-block: nothing depends on the order.
constexpr int kMqClasses = 128;
__device__ inline int mq_class(uint32_t rows, uint32_t sym_rows) { return rows == 0 || rows > sym_rows ? 0 : min(kMqClasses - 1, 1 + (int) (rows >> 5)); }
__global__ __launch_bounds__(256) void k_mq_hist(const uint32_t *lanerows, int *counters, int total, uint32_t sym_rows)
{
    __shared__ int h[kMqClasses];
    if (threadIdx.x < kMqClasses) h[threadIdx.x] = 0;
    __syncthreads();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < total) atomicAdd(&h[mq_class(lanerows[gid], sym_rows)], 1);
    __syncthreads();
    if (threadIdx.x < kMqClasses && h[threadIdx.x]) atomicAdd(&counters[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_mq_offsets(int *counters)                           // [0,128) counts -> [128,256) first slot of every class (longest first)
{
    const int c = threadIdx.x;
    int before = 0;
    for (int k = kMqClasses - 1; k > c; k--) before += counters[k];
    counters[kMqClasses + c] = before;
}
__global__ __launch_bounds__(256) void k_mq_place(const uint32_t *lanerows, int *counters, int *order, int total, int padded, uint32_t sym_rows)
{
    __shared__ int h[kMqClasses], base[kMqClasses];
    if (threadIdx.x < kMqClasses) h[threadIdx.x] = 0;
    __syncthreads();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int cls = 0, rank = 0;
    if (gid < total) { cls = mq_class(lanerows[gid], sym_rows); rank = atomicAdd(&h[cls], 1); }
    __syncthreads();
    if (threadIdx.x < kMqClasses && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&counters[kMqClasses + threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (gid < total) order[base[cls] + rank] = gid;
    else if (gid < padded) order[gid] = -1;                            // (the last wave's lanes without a code-block)
}

__global__ __launch_bounds__(128) void k_t1_mqrows(const uint8_t *SYM, const uint32_t *lanerows, const int *blkmax, int *cblk_len,
                                                    int *rates, uint8_t *cblk_bytes, void *ckpt, const J2kGeom *geom,
                                                    const J2kBlock *blocks, const FrameState *fs, J2kFrame *jf, int total, uint32_t sym_rows,
                                                    const int *order)
{
    __shared__ uint2 tab_store[128];
    __shared__ uint32_t ctxw[CtxSlotsLds::kBytes / 4];
    __shared__ uint32_t bring[17 * 64];
    __shared__ uint4 rowbuf[2 * kRowChunk * 64];
    __shared__ uint32_t handbuf[2 * kRowChunk * 16 * 64];
    __shared__ uint32_t a_end[64];
    const int lane = (int) threadIdx.x & 63;
    const bool code_wave = threadIdx.x >= 64;
    fill_mq_table_next(tab_store);
    const LdsTableNext tab{(uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint2 *) tab_store};
    // this lane's code-block: the order's entry (null: natural order); its decision rows lie where the emit put them - in the
    // rows of ITS group of 64 (gid >> 6), at ITS lane's place (gid & 63)
    const int slot = (int) (blockIdx.x * 64) + lane;
    const int pick = order ? order[slot] : slot;
    const int gid = pick < 0 ? total : pick;
    const size_t grp = (size_t) (pick < 0 ? 0 : pick) >> 6;
    const uint32_t home = (uint32_t) (pick < 0 ? 0 : pick) & 63u;
    const Tier1Lane l = tier1_lane(gid, total, blkmax, geom, blocks, fs);
    uint32_t nrows = l.live ? lanerows[gid] : 0u;
    if (nrows > sym_rows) nrows = 0;                                     // (overflow: the host retries, see k_t1_rowoffs)
    uint32_t wrows = nrows;
    for (int d = 32; d >= 1; d >>= 1) wrows = max(wrows, (uint32_t) __shfl_xor((int) wrows, d));
    wrows = (uint32_t) __builtin_amdgcn_readfirstlane((int) wrows);      // (both waves of the workgroup compute the same value)
    if (wrows == 0) return;
    // (lanes without a code-block run along with no rows: nothing is coded or stored for them)
    const bool on = nrows > 0;
    const int P = on ? l.P : 0, nstr = (l.h + 3) >> 2;
    CkObserver ck{J2kCkptView::of(ckpt, (size_t) gid), nullptr, (uint32_t) lane * 8u};
    const HandDev hand{(uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint32_t *) handbuf + (uint32_t) lane * 4u};
    const uint32_t nchunks = (wrows + kRowChunk - 1) / kRowChunk;
    if (!code_wave) {
        RowSrcDev src{(uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint4 *) rowbuf + (uint32_t) lane * 16u, nrows, wrows};
        const uint32_t ctx_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint32_t *) ctxw + (uint32_t) lane * 4u;
        a_end[lane] = t1::mq_rows_interval(src, CtxSlotsLds{ctx_base}, P, nstr, hand, ck, tab);
        __syncthreads();                                                 // (a_end, and this wave's checkpoint stores, before the other wave's epilogue)
        return;
    }
    // ---- wave 1: loader + code chain.  Barrier k hands chunk k of the rows to wave 0 and chunk k - 1 of the hand-over words to this wave.
    const uint8_t *sym = SYM + grp * sym_group_bytes(sym_rows) + (size_t) home * 32u;
    uint8_t *out = cblk_bytes + (size_t) (on ? gid : 0) * kJ2kCblkBytes;
    int *myrates = rates + (size_t) (on ? gid : 0) * kJ2kMaxPasses;
    MqSinkLds sink{(unsigned char *) bring + lane * 68, out, &jf[l.frame].overflow};
    t1::MqCodeChain chain;
    auto load_chunk = [&](uint32_t q, uint4 v[kRowChunk]) {
#pragma unroll
        for (int i = 0; i < kRowChunk; i++) {
            const uint32_t row = q * kRowChunk + (uint32_t) i;
            v[i] = row < wrows ? *(const uint4 *) (sym + sym_row_offset(row)) : make_uint4(0, 0, 0, 0);
        }
    };
    auto put_chunk = [&](uint32_t q, const uint4 v[kRowChunk]) {
#pragma unroll
        for (int i = 0; i < kRowChunk; i++) rowbuf[((q & 1u) * kRowChunk + (uint32_t) i) * 64 + (uint32_t) lane] = v[i];
    };
    {
        uint4 v[kRowChunk];
        load_chunk(0, v);
        put_chunk(0, v);
    }
    __syncthreads();                                                     // barrier 0: the table and chunk 0 are in LDS
    for (uint32_t q = 0; q <= nchunks; q++) {
        uint4 v[kRowChunk];
        if (q + 1 < nchunks) load_chunk(q + 1, v);                       // (in flight while the code chain works)
        if (q >= 1) {                                                    // the hand-over words of chunk q - 1
            for (uint32_t row = (q - 1) * kRowChunk; row < min(q * kRowChunk, wrows); row++) {
                if (row < nrows) {
                    uint32_t hw[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) hw[j] = hand.get(row, j);
                    chain.row(hw, nstr, myrates, sink, ck, [](bool b) { return __any(b) != 0; });
                }
            }
        }
        if (q + 1 < nchunks) put_chunk(q + 1, v);
        __syncthreads();                                                 // barrier q + 1
    }
    __syncthreads();                                                     // (matches wave 0's last one: a_end and its checkpoint stores are visible)
    if (P <= 0) return;
    __threadfence();
    t1::EncodeResult r = chain.finish(a_end[lane], 3 * P - 2, myrates, sink, DevAt{out, kJ2kCblkBytes});
    cblk_len[gid] = r.length;
    __threadfence();                                                    // the sweep below re-reads this lane's own bytes
    t1::finalize_checkpoints(ck, r.totalpasses, nstr, CkSrc{out, r.length < kJ2kCblkBytes ? r.length : kJ2kCblkBytes});
}

// ================================================================================================
// distortion tables: per-pass nmsedec sums are order-independent integers, so they are accumulated in
// parallel from (q6, msb, "significant in a propagation pass") instead of inside the serial coder
// ================================================================================================
__global__ __launch_bounds__(256) void k_distortion(const int32_t *__restrict__ Q6, const unsigned long long *__restrict__ SPS,
                                                     const int *__restrict__ numbps, const int *__restrict__ totalpasses,
                                                     double *__restrict__ disto, const short *__restrict__ luts,
                                                     const J2kGeom *geom, const J2kBlock *blocks, const FrameState *fs)
{
    // 32 copies of every pass's sum, a lane adds to copy lane % 32 (one LDS bank each): the samples of a wave mostly hit the
    // same few passes, and 64 atomics on one address are served one after the other (5.8 % of the HBM roof before)
    constexpr int kCopies = 32;
    __shared__ int nms[kJ2kMaxPasses * kCopies];
    __shared__ short lut[4 * 128];
    const int frame = blockIdx.y, bi = blockIdx.x;
    if (fs[frame].const_field) return;
    const int gid = frame * geom->stride + bi;
    blocks = j2k_frame_blocks(geom, blocks, frame);
    geom = &j2k_frame_geom(geom, frame);
    const int P = numbps[gid], np = totalpasses[gid];
    if (np <= 0) return;
    for (int i = threadIdx.x; i < kJ2kMaxPasses * kCopies; i += 256) nms[i] = 0;
    for (int i = threadIdx.x; i < 4 * 128; i += 256) lut[i] = luts[i];
    __syncthreads();
    const J2kBlock blk = blocks[bi];
    const int W = geom->W;
    const int32_t *q = Q6 + (size_t) frame * W * geom->H;
    const size_t grp = (size_t) (gid >> 6);
    const int gl = gid & 63;
    const unsigned long long *sps = SPS + grp * 64 * 64 + gl;
    const int copy = threadIdx.x & (kCopies - 1);
    for (int t = threadIdx.x; t < blk.w * blk.h; t += 256) {
        int y = t / blk.w, x = t - y * blk.w;
        int q6 = q[(size_t) (blk.y + y) * W + blk.x + x];
        unsigned int a6 = (unsigned int) (q6 < 0 ? -q6 : q6), a = a6 >> 6;
        if (!a) continue;
        int bs = 31 - __clz(a);
        int from_sp = (int) ((sps[(size_t) y * 64] >> x) & 1ull);
        int ps = bs == P - 1 ? 0 : 3 * (P - 1 - bs) - (from_sp ? 2 : 0);
        // opj_t1_getnmsedec_sig / _ref: 7-bit index around the coded bit, separate tables for plane 0
        int v = bs > 0 ? lut[0 * 128 + ((a6 >> bs) & 127)] : lut[1 * 128 + (a6 & 127)];
        atomicAdd(&nms[ps * kCopies + copy], v);
        for (int b = bs - 1; b >= 0; b--) {
            int r = b > 0 ? lut[2 * 128 + ((a6 >> b) & 127)] : lut[3 * 128 + (a6 & 127)];
            atomicAdd(&nms[(3 * (P - 1 - b) - 1) * kCopies + copy], r);
        }
    }
    __syncthreads();
    if (threadIdx.x < kJ2kMaxPasses) {                                  // (integers: the order of the additions does not matter)
        int sum = 0;
        for (int c = 0; c < kCopies; c++) sum += nms[threadIdx.x * kCopies + ((c + threadIdx.x) & (kCopies - 1))];
        nms[threadIdx.x * kCopies] = sum;
    }
    __syncthreads();
    // every pass's weighted distortion on its own thread (the same double expressions), then one thread adds them up in pass
    // order - the running sum's roundings are the reference's - and all threads store: the tail used to be one thread's
    // loop of ~40 passes with three dependent double operations and a store each
    __shared__ double wms[kJ2kMaxPasses];
    if ((int) threadIdx.x < np) {
        const int p = (int) threadIdx.x;
        const J2kBand &bd = geom->bands[blk.band];
        const int log2_gain = bd.orient == 0 ? 0 : (bd.orient == 3 ? 2 : 1);
        const double st = (double) bd.step_enc / (double) (1 << log2_gain);   // opj_t1_getwmsedec: step without the gain
        const int bp = t1::plane_of_pass(P, p);
        double wm = ((1.0 * bd.norm) * st) * (double) (1 << bp);
        wm = wm * ((wm * (double) nms[p * kCopies]) / 8192.0);
        wms[p] = wm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double cum = 0;
        for (int p = 0; p < np; p++) { cum += wms[p]; wms[p] = cum; }
    }
    __syncthreads();
    if ((int) threadIdx.x < np) disto[(size_t) gid * kJ2kMaxPasses + threadIdx.x] = wms[threadIdx.x];
}


}  // namespace

// ================================================================================================
void launch_input_stats(const float *data, int n_frames, size_t n_pix, FrameState *fs, hipStream_t s)
{
    hipLaunchKernelGGL(k_in_init, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, fs, n_frames);
    hipLaunchKernelGGL(k_in_minmax, dim3(64, n_frames), dim3(256), 0, s, data, n_pix, fs);
    hipLaunchKernelGGL(k_in_finish, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, data, n_pix, fs, n_frames);
    EBCC_HIP_LAUNCH_CHECK();
}

static short *g_luts_dev[64] = {nullptr};   // one copy per device
static const short *nmsedec_luts(hipStream_t s)
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    EBCC_HIP_CHECK(hipGetDevice(&dev));
    short *&g_luts = g_luts_dev[dev & 63];
    if (!g_luts) {
        short h[4 * 128];
        for (int i = 0; i < 128; i++) {                                // t1_generate_luts: T.800 J.14.4 estimates
            double t = i / std::pow(2, 6), u, v;
            int x;
            u = t; v = t - 1.5;
            x = (int) (std::floor((u * u - v * v) * std::pow(2, 6) + 0.5) / std::pow(2, 6) * 8192.0);
            h[0 * 128 + i] = (short) std::max(0, x);
            x = (int) (std::floor((u * u) * std::pow(2, 6) + 0.5) / std::pow(2, 6) * 8192.0);
            h[1 * 128 + i] = (short) std::max(0, x);
            u = t - 1.0;
            v = (i & 64) ? t - 1.5 : t - 0.5;
            x = (int) (std::floor((u * u - v * v) * std::pow(2, 6) + 0.5) / std::pow(2, 6) * 8192.0);
            h[2 * 128 + i] = (short) std::max(0, x);
            x = (int) (std::floor((u * u) * std::pow(2, 6) + 0.5) / std::pow(2, 6) * 8192.0);
            h[3 * 128 + i] = (short) std::max(0, x);
        }
        EBCC_HIP_CHECK(device_malloc((void **) &g_luts, sizeof h));
        EBCC_HIP_CHECK(hipMemcpyAsync(g_luts, h, sizeof h, hipMemcpyHostToDevice, s));
        wait_stream(s);
    }
    return g_luts;
}

__global__ void k_jf_reset(J2kFrame *jf, int n)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) jf[f].overflow = 0;
}

// tier-1 of every code-block of the batch from the masks k_quantize left.  Default: the segmented two-phase encoder;
// `single_kernel` (EBCC_T1_TWO_PHASE=0, or the retry after a group's decisions outgrew its rows of SYM - jf.overflow
// bit 1) runs passes and MQ coder in one kernel instead.  Same bytes, rates, checkpoints and masks either way.
void launch_j2k_tier1(const J2kBuffers &jb, int n_frames, hipStream_t s, bool single_kernel)
{
    const J2kGeom &g = jb.geom;
    const FrameState *fs = jb.fs;
    const int total = n_frames * g.stride;
    const size_t groups = ((size_t) total + 63) / 64;
    EBCC_HIP_CHECK(hipMemsetAsync(jb.SPS, 0, groups * 64 * 64 * sizeof(unsigned long long), s));
    timing_begin("t1_encode", s);
    if (!jb.SYM || single_kernel) {
        EBCC_HIP_CHECK(hipMemsetAsync(jb.T1S, 0, groups * kT1StateWords * 64 * sizeof(unsigned long long), s));
        int lpw = getenv("EBCC_T1_LPW") ? t1_lanes_per_wave(T1_ENCODE) : 32;    // (the single-kernel encoder is at its best with 32)
        unsigned t1_grid = (unsigned) ceil_div(total, lpw);
        hipLaunchKernelGGL(k_t1_encode, dim3(t1_grid), dim3(64), 0, s, jb.T1S, jb.BP, jb.SGN, jb.SPS, jb.blkmax, jb.numbps,
                           jb.totalpasses, jb.cblk_len, jb.rates, jb.cblk_bytes, jb.ckpt, jb.VISP, jb.d_geom,
                           jb.d_blocks, fs, jb.jf, total, lpw);
    } else {
        EBCC_HIP_CHECK(hipMemsetAsync(jb.seglen, 0, groups * (size_t) kJ2kSegCount * 64 * sizeof(std::uint16_t), s));
        timing_begin("t1_symbols", s);
        hipLaunchKernelGGL(k_t1_scan, dim3((unsigned) groups), dim3(64), 0, s, jb.BP, jb.SUF, jb.SGN, jb.SPS, jb.VISP, jb.blkmax, jb.numbps,
                           jb.totalpasses, jb.cblk_len, jb.seglen, jb.d_geom, jb.d_blocks, fs, total);
        hipLaunchKernelGGL(k_t1_rowoffs, dim3((unsigned) groups), dim3(64), 0, s, jb.seglen, jb.lanerows, jb.blkmax, jb.jf, jb.d_geom, jb.d_blocks,
                           fs, total, (uint32_t) jb.sym_rows);
        hipLaunchKernelGGL(k_t1_emit, dim3((unsigned) groups, kJ2kMaxPlanes), dim3(64), 0, s, jb.BP, jb.SUF, jb.SGN, jb.VISP, jb.blkmax, jb.SYM,
                           jb.seglen, jb.lanerows, jb.d_geom, jb.d_blocks, fs, total, (uint32_t) jb.sym_rows);
        timing_end("t1_symbols", s);
        if (getenv("EBCC_HIP_T1_STATS")) {                               // diagnostics: rows of 16 decisions per code-block (a wave of the MQ pass walks its longest lane's)
            std::vector<uint32_t> tot((size_t) total);
            EBCC_HIP_CHECK(hipMemcpyAsync(tot.data(), jb.lanerows, sizeof(uint32_t) * (size_t) total, hipMemcpyDeviceToHost, s));
            wait_stream(s);
            uint32_t mx = 0; double sum = 0, wsum = 0;
            for (size_t g0 = 0; g0 < (size_t) total; g0 += 64) {
                uint32_t wm = 0;
                for (size_t i = g0; i < std::min((size_t) total, g0 + 64); i++) { wm = std::max(wm, tot[i]); sum += tot[i]; }
                wsum += wm; mx = std::max(mx, wm);
            }
            fprintf(stderr, "ebcc-mi355x tier-1: %zu groups, rows of 16 decisions: per code-block mean %.0f, per wave mean %.0f max %u (cap %d)\n", groups,
                    sum / (double) total, wsum / (double) groups, mx, jb.sym_rows);
        }
        timing_begin("t1_mq", s);
        const int *order = nullptr;
        if (!getenv("EBCC_HIP_MQ_NATURAL_ORDER")) {                      // (cross-check: the code-blocks in their natural order)
            int *counters = jb.mq_order + groups * 64;
            EBCC_HIP_CHECK(hipMemsetAsync(counters, 0, 2 * kMqClasses * sizeof(int), s));
            hipLaunchKernelGGL(k_mq_hist, dim3(ceil_div(total, 256)), dim3(256), 0, s, jb.lanerows, counters, total, (uint32_t) jb.sym_rows);
            hipLaunchKernelGGL(k_mq_offsets, dim3(1), dim3(kMqClasses), 0, s, counters);
            hipLaunchKernelGGL(k_mq_place, dim3(ceil_div((int) (groups * 64), 256)), dim3(256), 0, s, jb.lanerows, counters, jb.mq_order, total, (int) (groups * 64),
                               (uint32_t) jb.sym_rows);
            order = jb.mq_order;
        }
        hipLaunchKernelGGL(k_t1_mqrows, dim3((unsigned) groups), dim3(128), 0, s, jb.SYM, jb.lanerows, jb.blkmax, jb.cblk_len, jb.rates,
                           jb.cblk_bytes, jb.ckpt, jb.d_geom, jb.d_blocks, fs, jb.jf, total, (uint32_t) jb.sym_rows, order);
        timing_end("t1_mq", s);
    }
    timing_end("t1_encode", s);
    hipLaunchKernelGGL(k_distortion, dim3(g.stride, n_frames), dim3(256), 0, s, jb.Q6, jb.SPS, jb.numbps, jb.totalpasses,
                       jb.disto, nmsedec_luts(s), jb.d_geom, jb.d_blocks, fs);
    EBCC_HIP_LAUNCH_CHECK();
}

bool j2k_tier1_retry(const J2kBuffers &jb, int n_frames, const J2kFrame *host_jf, hipStream_t s)
{
    bool need = false;
    for (int f = 0; f < n_frames; f++) need |= (host_jf[f].overflow & 2) != 0;
    if (!need) return false;
    hipLaunchKernelGGL(k_jf_reset, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, jb.jf, n_frames);
    launch_j2k_tier1(jb, n_frames, s, true);
    return true;
}

// The top level's forward COLUMN pass as a stream, fused with the u16 scaling and the DC shift (k_scale_shift): a thread
// per column reads the frame's rows in order, carries the four lifting steps in a register pipeline two positions deep
// (fdwt_tile's expressions, order and boundary forms, cas == 0) and writes the low-pass row j to row j and the high-pass
// row j to row sn + j of the tile buffer.  Out of place by nature (frame -> jb.B), so the level needs no LDS staging,
// every access is a full row segment, and the shifted samples never exist in memory.  The column is cut into gridDim.z
// pieces; a piece starts two positions early and ends two late (an output depends on the inputs of positions i - 2 .. i + 2).
constexpr int kFwdT = 64;
__global__ __launch_bounds__(kFwdT) void k_j2k_cols_fwd_top(const float *__restrict__ data, float *__restrict__ B, const J2kGeom *geom,
                                                             const FrameState *fs)
{
    const int frame = blockIdx.y;
    if (fs[frame].const_field) return;
    const J2kGeom &g = j2k_frame_geom(geom, frame);
    constexpr int r = kJ2kRes - 1;
    const int W = g.W, nv = g.rh[r], sn = g.rh[r - 1], dn = nv - sn;
    const int col = blockIdx.x * kFwdT + threadIdx.x;
    if (col >= g.rw[r]) return;
    const size_t n_pix = (size_t) W * g.H;
    const float *x = data + (size_t) frame * n_pix + col;
    float *b = B + (size_t) frame * n_pix + col;
    const float mn = fs[frame].minv, rng = fs[frame].maxv - fs[frame].minv;
    auto sample = [&](int y) {                                          // k_scale_shift: u16 scaling (:686-689) and OpenJPEG's DC shift
        const unsigned int u = __float2uint_rz(((x[(size_t) y * W] - mn) / rng) * 65535.0f) & 0xFFFFu;
        return (float) ((int) u - 32768);
    };
    const float invK = (float) (1.0 / 1.230174105);
    const int per = ceil_div(sn, (int) gridDim.z), ka = (int) blockIdx.z * per, kb = min(sn, ka + per);
    const int j0 = max(ka - 2, 0);
    float E_1 = 0, O_1 = 0;        // inputs of position j - 1
    float o1_2 = 0, e1_2 = 0;      // after steps 1 and 2: position j - 2
    float o2_3 = 0;                // after step 3: position j - 3
    float e_in = sample(min(2 * j0, nv - 1)), o_in = sample(min(2 * j0 + 1, nv - 1));
    for (int j = j0; j < kb + 2; j++) {
        // the next position's rows are requested before this one is worked on
        const int jn = j + 1;
        const float e_nx = sample(min(2 * jn, nv - 1)), o_nx = sample(min(2 * jn + 1, nv - 1));
        const float E0 = e_in, O0 = o_in;                                // E[j] (j < sn), O[j] (j < dn)
        int i = j - 1;
        float o1 = 0, e1 = 0;
        if (i >= 0 && i < dn) o1 = (i + 1 < sn) ? O_1 + ((E_1 + E0) * kA) : O_1 + ((2 * E_1) * kA);
        if (i >= 0 && i < sn) e1 = (i < dn) ? E_1 + (((i == 0 ? o1 : o1_2) + o1) * kB) : E_1 + ((2 * o1_2) * kB);
        i = j - 2;
        float o2 = 0;
        if (i >= 0 && i < dn) o2 = (i + 1 < sn) ? o1_2 + ((e1_2 + e1) * kG) : o1_2 + ((2 * e1_2) * kG);
        if (i >= ka && i < kb) {
            const float e2 = (i < dn) ? e1_2 + (((i == 0 ? o2 : o2_3) + o2) * kD) : e1_2 + ((2 * o2_3) * kD);
            b[(size_t) i * W] = e2 * invK;
            if (i < dn) b[(size_t) (sn + i) * W] = o2 * kK;
        }
        E_1 = E0; O_1 = O0;
        o1_2 = o1; e1_2 = e1;
        o2_3 = o2;
        e_in = e_nx; o_in = o_nx;
    }
}

void launch_j2k_analysis(const float *data, const J2kBuffers &jb, int n_frames, hipStream_t s)
{
    const FrameState *fs = jb.fs;
    const J2kGeom &g = jb.geom;
    const size_t n_pix = (size_t) g.W * g.H;
    const int total = n_frames * g.stride;
    hipLaunchKernelGGL(k_jf_reset, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, jb.jf, n_frames);
    // plain frames: scaling, DC shift and the top level's column pass in one streaming kernel
    const bool fwd_top = g.period == 1 && g.ry0[kJ2kRes - 1] % 2 == 0 && g.rh[kJ2kRes - 2] >= 2 &&
                         g.rh[kJ2kRes - 1] - g.rh[kJ2kRes - 2] >= 1;
    if (!fwd_top) hipLaunchKernelGGL(k_scale_shift, dim3(128, n_frames), dim3(256), 0, s, data, jb.B, n_pix, fs);
    EBCC_HIP_CHECK(hipMemsetAsync(jb.rate_path_n, 0, sizeof(int) * (size_t) n_frames, s));   // new pass tables: k_rate's record starts over
    EBCC_HIP_CHECK(hipMemsetAsync(jb.rate_cache.ok, 0, sizeof(int) * (size_t) n_frames, s));  // ... and so does what it keeps of its set-up
    EBCC_HIP_CHECK(hipMemsetAsync(jb.lastnp, 0xFF, sizeof(int) * (size_t) total, s));         // and so does the probe decode's
    timing_begin("j2k_dwt_fwd", s);
    for (int r = kJ2kRes - 1; r >= 1; r--) {                           // opj_dwt_encode_procedure: vertical, then horizontal
        if (fwd_top && r == kJ2kRes - 1) {
            const int pieces = std::max(1, std::min(8, g.rh[r - 1] / 16));
            hipLaunchKernelGGL(k_j2k_cols_fwd_top, dim3(ceil_div(g.rw[r], kFwdT), n_frames, pieces), dim3(kFwdT), 0, s, data, jb.B, jb.d_geom, fs);
        } else
        dwt_cols<true>(jb.B, jb, r, n_frames, fs, nullptr, s);
        if (g.rw[r] > 1) dwt_rows<true>(jb.B, nullptr, jb, r, n_frames, fs, nullptr, s);
    }
    timing_end("j2k_dwt_fwd", s);
    hipLaunchKernelGGL(k_quantize, dim3((unsigned) (((size_t) total + 63) / 64)), dim3(256), 0, s, jb.B, jb.Q6, jb.BP, jb.SGN, jb.SUF, jb.blkmax,
                       jb.d_geom, jb.d_blocks, fs, total);
    launch_j2k_tier1(jb, n_frames, s, false);
    EBCC_HIP_LAUNCH_CHECK();
}

// dequantisation + inverse transform of the tier-1 decoder's output V through the tile buffers B to the decoded
// field jb.DEC, with the error statistics against `data` (if given) left as partial sums per frame; used by both
// decode flavours (j2k_rate.hip).  Returns the number of partials per frame.
int j2k_inverse_dwt(float *B, const int32_t *V, const float *data, const J2kBuffers &jb, int n_frames, const FrameState *fs,
                    const int *active, hipStream_t s, int keep_field)
{
    int partials = 0;
    // plain frames: every level in one fused pass (k_j2k_level5_fin: horizontal synthesis in registers, vertical register
    // pipeline), alternating between the tile buffer and jb.B2 from the first level that is large enough; the levels
    // below it and tiles at odd offsets use the separate LDS-staged row / column passes in place.
    const J2kGeom &g = jb.geom;
    const size_t n_pix = (size_t) g.W * g.H;
    auto fusable = [&](int r) {
        return V && g.period == 1 && g.ry0[r] % 2 == 0 && g.rw[r - 1] >= 2 && g.rh[r - 1] >= 1 &&
               (r == kJ2kRes - 1 ? ceil_div(g.rw[r - 1], kL5Pairs) <= kPartials : jb.B2 != nullptr);
    };
    int first_fused = kJ2kRes;                                          // levels first_fused .. top are fused (sizes grow with r)
    for (int r = kJ2kRes - 1; r >= 1 && fusable(r); r--) first_fused = r;
    const float *ll = nullptr;                                          // low-pass input of the next fused level (null: from V)
    float *spare = jb.B2;
    for (int r = 1; r < kJ2kRes; r++) {                                // opj_dwt_decode_tile_97: horizontal, then vertical
        if (r >= first_fused) {
            const int strips = ceil_div(g.rw[r - 1], kL5Pairs);
            // (the top level in more, shorter pieces: a piece is what an infeasible probe can skip)
            int pieces = std::max(1, std::min({r == kJ2kRes - 1 ? 8 : 4, kPartials / strips, g.rh[r - 1] / 16}));
            while (pieces > 1 && (long long) n_frames * pieces > 65535) pieces--;            // (grid y)
            // neighbouring strips of a tile as the waves of one workgroup, four at a time (measured per 128-frame probe round,
            // tools/gpu/kstat.sh: 1 wave 231 us, 4 waves 220, 6 waves 258, all 12 strips 264 - a large workgroup needs all
            // its wave slots free on one CU at once)
            const int wave_cap = 4;
            const int wg = std::min(strips, wave_cap), groups = ceil_div(strips, wg);
            if (r > 1 && !ll) ll = B;                                   // (the separate passes below left their result in B)
            J2kLevelIO io{ll, g.W, n_pix, nullptr, g.W, n_pix, r};
            if (r == kJ2kRes - 1) {
                partials = strips * pieces;
                hipLaunchKernelGGL(k_j2k_level5_fin<true>, dim3((unsigned) groups, (unsigned) (n_frames * pieces)), dim3(64 * wg), 0, s, io, V, jb.d_geom, fs, active,
                                   J2kFinish{data, keep_field ? jb.DEC : nullptr, jb.jf, jb.partial, jb.partial_u, keep_field == 2}, strips, n_frames, pieces);
            } else {
                io.out = ll == spare ? B : spare;                         // never the buffer the level reads from
                hipLaunchKernelGGL(k_j2k_level5_fin<false>, dim3((unsigned) groups, (unsigned) (n_frames * pieces)), dim3(64 * wg), 0, s, io, V, jb.d_geom, fs, active, J2kFinish{}, strips, n_frames, pieces);
                ll = io.out;
            }
            continue;
        }
        dwt_rows<false>(B, V, jb, r, n_frames, fs, active, s);
        if (r + 1 < kJ2kRes) dwt_cols<false>(B, jb, r, n_frames, fs, active, s);
        else partials = dwt_cols<false, true>(B, jb, r, n_frames, fs, active, s, J2kFinish{data, keep_field ? jb.DEC : nullptr, jb.jf, jb.partial, jb.partial_u, keep_field == 2});
    }
    EBCC_HIP_LAUNCH_CHECK();
    return partials;
}

// host check of div65535_exact against the division it replaces, for every value it is used on (ebcc_hip_selfcheck)
int j2k_selfcheck_div65535()
{
    int bad = 0;
    for (int q = 0; q <= 65535; q++) {
        const volatile float s = (float) q;
        if (div65535_exact(s) != s / 65535.0f) bad++;
    }
    return bad;
}

}  // namespace ebcc
