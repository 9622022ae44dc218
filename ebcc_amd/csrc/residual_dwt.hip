// residual_dwt.hip - pad / DC / CDF 9/7 lifting / truncation / error statistics kernels of the residual
// layer.  gfx950 only, compiled with -ffp-contract=off: every expression below is the same sequence of
// IEEE fp32 operations the reference executes on x86-64 (no FMA), see SURVEY.md Appendix B.
//
// Lifting layout: a line of n samples is held de-interleaved in LDS as E[k] = s[2k], O[k] = s[2k+1].
// The four lifting steps are data-parallel sweeps separated by barriers; each output element is the
// same 2-3 operand fp32 expression as in the reference's sequential loops
// (reference src/spiht/dwt.h:87-112 rows, :142-167 columns, :114-140 / :169-272 inverse).
// Rows: one row per workgroup iteration, HBM access is full contiguous rows (float2 per lane).
// Columns: a tile of CW adjacent columns x full height is staged through LDS so that every HBM
// access is a CW*4-byte row segment; the strided walk happens only in LDS.
#include <cstring>
#include <cmath>
#include <mutex>
#include "residual.hpp"
#include "residual_device.hpp"

namespace ebcc {

namespace {

__device__ constexpr float kAlpha = -1.586134342f;
__device__ constexpr float kBeta  = -0.05298011854f;
__device__ constexpr float kGamma = 0.8829110762f;
__device__ constexpr float kDelta = 0.44355068522f;
__device__ constexpr float kXi    = 1.149604398f;

constexpr int kRowThreads = 256;
constexpr int kColThreads = 1024;

// ------------------------------------------------------------------------------------------------
// forward / inverse lifting sweeps on an LDS tile: `lines` lines, element (k, line) at [k*ls + line*ks]
// (rows: ks = half-padded stride, ls = 1 ... we simply pass index functors)
// ------------------------------------------------------------------------------------------------
template <typename Idx>
__device__ inline void lift_forward_tile(float *E, float *O, int half, int lines, Idx at, int tid, int nt)
{
    const int total = half * lines;
    // step 1: O[k] = O[k] + alpha*(E[k] + E[k+1]);  last: O[h-1] + (2*alpha)*E[h-1]      dwt.h:92-94
    for (int i = tid; i < total; i += nt) {
        int k = i / lines, l = i - k * lines;
        float o = O[at(k, l)];
        if (k + 1 < half) o = o + kAlpha * (E[at(k, l)] + E[at(k + 1, l)]);
        else              o = o + (2 * kAlpha) * E[at(k, l)];
        O[at(k, l)] = o;
    }
    __syncthreads();
    // step 2: E[k] = E[k] + beta*(O[k] + O[k-1]);  first: beta*(O[0] + O[1])             dwt.h:96-98
    for (int i = tid; i < total; i += nt) {
        int k = i / lines, l = i - k * lines;
        float e = E[at(k, l)];
        if (k > 0) e = e + kBeta * (O[at(k, l)] + O[at(k - 1, l)]);
        else       e = e + kBeta * (O[at(0, l)] + O[at(1, l)]);
        E[at(k, l)] = e;
    }
    __syncthreads();
    // step 3: O[k] += gamma*(E[k] + E[k+1]);  last: gamma*(E[h-1] + E[h-2])               dwt.h:100-102
    for (int i = tid; i < total; i += nt) {
        int k = i / lines, l = i - k * lines;
        float o = O[at(k, l)];
        if (k + 1 < half) o = o + kGamma * (E[at(k, l)] + E[at(k + 1, l)]);
        else              o = o + kGamma * (E[at(k, l)] + E[at(k - 1, l)]);
        O[at(k, l)] = o;
    }
    __syncthreads();
    // step 4: E[k] += delta*(O[k] + O[k-1]);  first: delta*(O[0] + O[1])                  dwt.h:104-106
    for (int i = tid; i < total; i += nt) {
        int k = i / lines, l = i - k * lines;
        float e = E[at(k, l)];
        if (k > 0) e = e + kDelta * (O[at(k, l)] + O[at(k - 1, l)]);
        else       e = e + kDelta * (O[at(0, l)] + O[at(1, l)]);
        E[at(k, l)] = e;
    }
    __syncthreads();
}

// On entry E = low band, O = high band (unscaled, as stored).  On exit E[k] = x[2k], O[k] = x[2k+1].
template <typename Idx>
__device__ inline void lift_inverse_tile(float *E, float *O, int half, int lines, Idx at, int tid, int nt)
{
    const int total = half * lines;
    // un-scale                                                                              dwt.h:120-123
    for (int i = tid; i < total; i += nt) {
        int k = i / lines, l = i - k * lines;
        E[at(k, l)] = E[at(k, l)] / kXi;
        O[at(k, l)] = O[at(k, l)] * kXi;
    }
    __syncthreads();
    // E[k] -= delta*(O[k] + O[k-1]); first uses O[0] + O[1]                                 dwt.h:125-127
    for (int i = tid; i < total; i += nt) {
        int k = i / lines, l = i - k * lines;
        float e = E[at(k, l)];
        if (k > 0) e = e - kDelta * (O[at(k, l)] + O[at(k - 1, l)]);
        else       e = e - kDelta * (O[at(0, l)] + O[at(1, l)]);
        E[at(k, l)] = e;
    }
    __syncthreads();
    // O[k] -= gamma*(E[k] + E[k+1]); last uses E[h-1] + E[h-2]                              dwt.h:129-131
    for (int i = tid; i < total; i += nt) {
        int k = i / lines, l = i - k * lines;
        float o = O[at(k, l)];
        if (k + 1 < half) o = o - kGamma * (E[at(k, l)] + E[at(k + 1, l)]);
        else              o = o - kGamma * (E[at(k, l)] + E[at(k - 1, l)]);
        O[at(k, l)] = o;
    }
    __syncthreads();
    // x[2k] = E[k] - beta*(O[k] + O[k-1]); first uses O[0] + O[1]                           dwt.h:133-135
    for (int i = tid; i < total; i += nt) {
        int k = i / lines, l = i - k * lines;
        float e = E[at(k, l)];
        if (k > 0) e = e - kBeta * (O[at(k, l)] + O[at(k - 1, l)]);
        else       e = e - kBeta * (O[at(0, l)] + O[at(1, l)]);
        E[at(k, l)] = e;
    }
    __syncthreads();
    // x[2k+1] = O[k] - alpha*(x[2k] + x[2k+2]); last: O[h-1] - (2*alpha)*x[n-2]             dwt.h:137-139
    for (int i = tid; i < total; i += nt) {
        int k = i / lines, l = i - k * lines;
        float o = O[at(k, l)];
        if (k + 1 < half) o = o - kAlpha * (E[at(k, l)] + E[at(k + 1, l)]);
        else              o = o - (2 * kAlpha) * E[at(k, l)];
        O[at(k, l)] = o;
    }
    __syncthreads();
}

struct RowIdx { __device__ int operator()(int k, int) const { return k; } };
template <int CW> struct ColIdx { __device__ int operator()(int k, int l) const { return k * CW + l; } };

// ------------------------------------------------------------------------------------------------
// row kernels: grid (blocks, frames)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRowThreads) void k_rows_fwd(const float *__restrict__ src, float *__restrict__ dst,
                                                           int stride, size_t frame_stride, int n, int rows,
                                                           const FrameState *fs, int sub_dc, const int *active)
{
    extern __shared__ float sm[];
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    const int half = n >> 1;
    float *E = sm, *O = sm + half;
    src += (size_t) frame * frame_stride;
    dst += (size_t) frame * frame_stride;
    const float dc = sub_dc ? fs[frame].dc : 0.0f;
    const int tid = threadIdx.x;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const float2 *s2 = reinterpret_cast<const float2 *>(src + (size_t) row * stride);
        for (int k = tid; k < half; k += kRowThreads) {
            float2 v = s2[k];
            // sub_dc (dwt.h:330-332): data -= dc.  x - 0.0f == x bit-for-bit, so the no-DC levels share the path.
            E[k] = v.x - dc;
            O[k] = v.y - dc;
        }
        __syncthreads();
        lift_forward_tile(E, O, half, 1, RowIdx(), tid, kRowThreads);
        float *d = dst + (size_t) row * stride;
        for (int k = tid; k < half; k += kRowThreads) {
            d[k] = E[k] * kXi;                    // dwt.h:109
            d[half + k] = O[k] / kXi;             // dwt.h:110
        }
        __syncthreads();
    }
}

// residual pixel from the synthesised grid: add_dc (dwt.h:345-350), /255 (spiht_re.c:514),
// de-normalisation (ebcc_codec.c:752)
__device__ inline float residual_value(float a, float dc, float rmin, float rng)
{
    float v = floorf(a + dc);
    v = v > 255.0f ? 255.0f : (v < 0.0f ? 0.0f : v);
    float rn = v / 255.0f;
    return rn * rng + rmin;
}

// What the last synthesis pass does with its rows instead of storing the grid (k_rows_inv_use):
//   statistics (data != nullptr): max |x - (decoded + r)| and the sum of x - (decoded + r) per frame
//                                 (ebcc_codec.c:481,498), one partial sum per workgroup;
//   addition   (data == nullptr): out += r (ebcc_codec.c:1307).
struct RowUse {
    const float *data;
    const float *decoded;          // statistics: the base layer's field; addition: unused
    float *out;                    // addition: the field the residual is added to
    int size_x, size_y;            // the image inside the padded grid
    size_t n_pix;
    FrameState *fs;
    double *partial;               // [frames][kPartials]
};

__global__ __launch_bounds__(kRowThreads) void k_rows_inv_use(const float *__restrict__ src, int stride, size_t frame_stride, int n,
                                                               int rows, const int *active, RowUse u)
{
    extern __shared__ float sm[];
    __shared__ double red[kRowThreads / 64];
    __shared__ float redm[kRowThreads / 64];
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    const int half = n >> 1;
    float *E = sm, *O = sm + half;
    src += (size_t) frame * frame_stride;
    const int tid = threadIdx.x;
    const bool stats = u.data != nullptr;
    const float *x = stats ? u.data + (size_t) frame * u.n_pix : nullptr;
    const float *d = stats ? u.decoded + (size_t) frame * u.n_pix : nullptr;
    float *o = stats ? nullptr : u.out + (size_t) frame * u.n_pix;
    const float dc = stats ? u.fs[frame].dc : (float) u.fs[frame].dec_dc, rmin = u.fs[frame].rmin, rng = u.fs[frame].rmax - u.fs[frame].rmin;
    double acc = 0;
    float mx = 0;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const float *s = src + (size_t) row * stride;
        for (int k = tid; k < half; k += kRowThreads) {
            E[k] = s[k];
            O[k] = s[half + k];
        }
        __syncthreads();
        lift_inverse_tile(E, O, half, 1, RowIdx(), tid, kRowThreads);
        if (row < u.size_y) {
            for (int xx = tid; xx < u.size_x; xx += kRowThreads) {
                const float r = residual_value((xx & 1) ? O[xx >> 1] : E[xx >> 1], dc, rmin, rng);
                const size_t i = (size_t) row * u.size_x + xx;
                if (stats) {
                    const float t = x[i] - (d[i] + r);
                    acc += (double) t;
                    const float e = fabsf(t);
                    mx = e > mx ? e : mx;
                } else {
                    o[i] = o[i] + r;
                }
            }
        }
        __syncthreads();
    }
    if (!stats) return;
    for (int k = 32; k >= 1; k >>= 1) { acc += __shfl_xor(acc, k); mx = fmaxf(mx, __shfl_xor(mx, k)); }
    if ((tid & 63) == 0) { red[tid >> 6] = acc; redm[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        double a = 0;
        float m = 0;
        for (int k = 0; k < kRowThreads / 64; k++) { a += red[k]; m = fmaxf(m, redm[k]); }
        u.partial[(size_t) frame * kPartials + blockIdx.x] = a;
        atomicMax(&u.fs[frame].maxerr_bits, __float_as_uint(m));
    }
}

__global__ __launch_bounds__(kRowThreads) void k_rows_inv(const float *__restrict__ src, float *__restrict__ dst,
                                                           int stride, size_t frame_stride, int n, int rows,
                                                           const int *active)
{
    extern __shared__ float sm[];
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    const int half = n >> 1;
    float *E = sm, *O = sm + half;
    src += (size_t) frame * frame_stride;
    dst += (size_t) frame * frame_stride;
    const int tid = threadIdx.x;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const float *s = src + (size_t) row * stride;
        for (int k = tid; k < half; k += kRowThreads) {
            E[k] = s[k];
            O[k] = s[half + k];
        }
        __syncthreads();
        lift_inverse_tile(E, O, half, 1, RowIdx(), tid, kRowThreads);
        float2 *d2 = reinterpret_cast<float2 *>(dst + (size_t) row * stride);
        for (int k = tid; k < half; k += kRowThreads) d2[k] = make_float2(E[k], O[k]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// column kernels: grid (tiles, frames); tile = CW columns x n rows in LDS
// ------------------------------------------------------------------------------------------------
template <int CW>
__global__ __launch_bounds__(kColThreads) void k_cols_fwd(const float *__restrict__ src, float *__restrict__ dst,
                                                           int stride, size_t frame_stride, int n, int cols,
                                                           const int *active)
{
    extern __shared__ float sm[];
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    const int half = n >> 1;
    float *E = sm, *O = sm + (size_t) half * CW;
    src += (size_t) frame * frame_stride;
    dst += (size_t) frame * frame_stride;
    const int tid = threadIdx.x;
    const int ntiles = (cols + CW - 1) / CW;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int x0 = tile * CW;
        const int w = min(CW, cols - x0);
        for (int i = tid; i < n * CW; i += kColThreads) {
            int y = i / CW, c = i - y * CW;
            float v = (c < w) ? src[(size_t) y * stride + x0 + c] : 0.0f;
            ((y & 1) ? O : E)[(y >> 1) * CW + c] = v;
        }
        __syncthreads();
        lift_forward_tile(E, O, half, CW, ColIdx<CW>(), tid, kColThreads);
        for (int i = tid; i < half * CW; i += kColThreads) {
            int k = i / CW, c = i - k * CW;
            if (c < w) {
                dst[(size_t) k * stride + x0 + c] = E[i] * kXi;             // dwt.h:164
                dst[(size_t) (half + k) * stride + x0 + c] = O[i] / kXi;    // dwt.h:165
            }
        }
        __syncthreads();
    }
}

template <int CW>
__global__ __launch_bounds__(kColThreads) void k_cols_inv(const float *__restrict__ src, float *__restrict__ dst,
                                                           int stride, size_t frame_stride, int n, int cols,
                                                           const int *active)
{
    extern __shared__ float sm[];
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    const int half = n >> 1;
    float *E = sm, *O = sm + (size_t) half * CW;
    src += (size_t) frame * frame_stride;
    dst += (size_t) frame * frame_stride;
    const int tid = threadIdx.x;
    const int ntiles = (cols + CW - 1) / CW;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int x0 = tile * CW;
        const int w = min(CW, cols - x0);
        for (int i = tid; i < half * CW; i += kColThreads) {
            int k = i / CW, c = i - k * CW;
            E[i] = (c < w) ? src[(size_t) k * stride + x0 + c] : 0.0f;
            O[i] = (c < w) ? src[(size_t) (half + k) * stride + x0 + c] : 0.0f;
        }
        __syncthreads();
        lift_inverse_tile(E, O, half, CW, ColIdx<CW>(), tid, kColThreads);
        for (int i = tid; i < n * CW; i += kColThreads) {
            int y = i / CW, c = i - y * CW;
            if (c < w) dst[(size_t) y * stride + x0 + c] = ((y & 1) ? O : E)[(y >> 1) * CW + c];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The finest level's inverse column pass as a STREAM (one thread per column, registers only - cf. k_j2k_cols_fin of
// the base layer), for the probes of the truncation search: its three detail bands are not read from a reconstructed
// grid but computed on the fly from the encoder's bookkeeping (prefix_value), the LL quadrant comes from `src` (the
// coarser levels' result); the result goes to `dst`.  lift_inverse_tile's expressions in its order; the reference's
// boundary taps (dwt.h:125-139: first low-pass uses O[0] + O[1], last high-pass of step 2 E[h-1] + E[h-2], of step 4
// (2 alpha) x[n-2]) make the pipeline four positions deep.
// ------------------------------------------------------------------------------------------------
// x / kXi without the division sequence (v_div_scale x 2, v_rcp, five fmas, v_div_fmas, v_div_fixup): q0 = x * z with
// z = RN(1 / kXi), one exact remainder, one correction - the correctly rounded quotient for EVERY fp32 x with
// |x| >= 2^-96 (checked over all 2^23 significands by ebcc_hip_selfcheck; below that the remainder is subnormal and the
// division itself is used), and x itself for a zero of either sign.
__host__ __device__ inline float div_xi(float x)
{
    const float z = 0.869864404f;                                      // RN(1 / 1.149604398f)
#if defined(__HIP_DEVICE_COMPILE__)
    const float q0 = __fmul_rn(x, z);
    const float q1 = __fmaf_rn(__fmaf_rn(-q0, kXi, x), z, q0);
    const float ax = fabsf(x);
    float q = x == 0.0f ? x : q1;
    if (__builtin_expect(ax < 1.2621774e-29f && ax != 0.0f, 0)) q = x / kXi;
    return q;
#else
    const float q0 = x * z;
    const float q1 = std::fma(std::fma(-q0, 1.149604398f, x), z, q0);
    const float ax = std::fabs(x);
    if (x == 0.0f) return x;
    if (ax < 1.2621774e-29f) return x / 1.149604398f;
    return q1;
#endif
}
// v / 255.0f for an integer-valued v in [0, 255] (add_dc's clamp): fmaf(v, K_hi, v * K_lo), equal to the quotient for all 256
__host__ __device__ inline float div255_exact(float v)
{
    const float k_hi = 0.0039215688593685627f, k_lo = -2.3191758236063009e-10f;
#if defined(__HIP_DEVICE_COMPILE__)
    return __fmaf_rn(v, k_hi, __fmul_rn(v, k_lo));
#else
    return std::fma(v, k_hi, v * k_lo);
#endif
}
// residual_value with the clamp as one median and the division replaced
__device__ inline float residual_value_fast(float a, float dc, float rmin, float rng)
{
    const float v = __builtin_amdgcn_fmed3f(floorf(a + dc), 0.0f, 255.0f);
    return div255_exact(v) * rng + rmin;
}

struct RPipe {
    float E_1 = 0, O_1 = 0, O_2 = 0;       // inputs (scaled) of positions j - 1, j - 1, j - 2
    float a1 = 0, a2 = 0;                  // after step 1: positions j - 2, j - 3
    float b1 = 0, b2 = 0;                  // after step 2: positions j - 3, j - 4
    float c1 = 0;                          // after step 3: position j - 4
    // inputs of position j in; finished samples 2 (j - 4), 2 (j - 4) + 1 out (valid when j - 4 is in [0, half))
    __device__ void step(int j, int half, float Ein, float Oin, float &x_even, float &x_odd)
    {
        const float E0 = div_xi(Ein), O0 = Oin * kXi;
        int k = j - 1;                                                  // step 1
        float e1 = 0;
        if (k >= 0 && k < half) e1 = E_1 - kDelta * (O_1 + (k > 0 ? O_2 : O0));
        k = j - 2;                                                      // step 2
        float o1 = 0;
        if (k >= 0 && k < half) o1 = O_2 - kGamma * (a1 + (k + 1 < half ? e1 : a2));
        k = j - 3;                                                      // step 3
        float e2 = 0;
        if (k >= 0 && k < half) e2 = a2 - kBeta * (b1 + (k > 0 ? b2 : o1));
        k = j - 4;                                                      // step 4
        x_even = c1;
        x_odd = (k + 1 < half) ? b2 - kAlpha * (c1 + e2) : b2 - (2 * kAlpha) * c1;
        O_2 = O_1; O_1 = O0; E_1 = E0;
        a2 = a1; a1 = e1;
        b2 = b1; b1 = o1;
        c1 = e2;
    }
    // the same for 4 <= j < half: every neighbour exists, no boundary form, no test
    __device__ void interior(float Ein, float Oin, float &x_even, float &x_odd)
    {
        const float E0 = div_xi(Ein), O0 = Oin * kXi;
        const float e1 = E_1 - kDelta * (O_1 + O_2);
        const float o1 = O_2 - kGamma * (a1 + e1);
        const float e2 = a2 - kBeta * (b1 + b2);
        x_even = c1;
        x_odd = b2 - kAlpha * (c1 + e2);
        O_2 = O_1; O_1 = O0; E_1 = E0;
        a2 = a1; a1 = e1;
        b2 = b1; b1 = o1;
        c1 = e2;
    }
};

constexpr int kStreamT = 128;
__global__ __launch_bounds__(kStreamT) void k_cols_inv_stream(const float *__restrict__ src, float *__restrict__ dst, Grid g, size_t np,
                                                               const int32_t *__restrict__ Cb, const uint32_t *__restrict__ sigordb,
                                                               const uint32_t *__restrict__ lspidxb, const FrameState *fsb,
                                                               const unsigned long long *trunc_bits, const int *active)
{
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    const FrameState &fs = fsb[frame];
    unsigned long long nb = trunc_bits[frame], bits0 = fs.budget + 128;   // spiht_decode: num_bits = min(num_bits, bits0) - 128 (spiht_re.c:495-500)
    if (nb > bits0) nb = bits0;
    const unsigned long long B = nb - 128;
    __shared__ unsigned int rbase[32], rreach[32];
    if (threadIdx.x < 32) { rbase[threadIdx.x] = fs.refine_base[threadIdx.x]; rreach[threadIdx.x] = fs.step_reached[threadIdx.x]; }
    __syncthreads();
    const int nx = g.nx, ny = g.ny, half = ny >> 1, hx = nx >> 1;
    const int col = blockIdx.x * kStreamT + threadIdx.x;
    if (col >= nx) return;
    const float *a = src + (size_t) frame * np;
    float *out = dst + (size_t) frame * np;
    const int32_t *C = Cb + (size_t) frame * np;
    const uint32_t *so = sigordb + (size_t) frame * np, *li = lspidxb + (size_t) frame * np;
    // the column is cut into gridDim.z pieces; a piece starts its pipeline two positions early (an output of position
    // k depends on the inputs of positions k - 2 .. k + 2)
    const int per = (half + (int) gridDim.z - 1) / (int) gridDim.z, ka = (int) blockIdx.z * per, kb = min(half, ka + per);
    const int jstart = max(ka - 2, 0);
    // raw inputs of a position, requested one step before the position enters the pipeline.  The ordinal of the
    // significance bit is always read; coefficient and LSP slot only where that bit lies inside the prefix (at the usual
    // cut points most coefficients are still insignificant: the frame is HBM-bound on these probes, not latency-bound)
    struct Raw { float ll; uint32_t eo, el; int ec; uint32_t oo, ol; int oc; };
    const bool left = col < hx;
    auto fetch = [&](int j, Raw &r) {
        const int jj = min(j, half - 1);
        const size_t ie = (size_t) jj * nx + col, io = (size_t) (half + jj) * nx + col;
        r.ll = left ? a[ie] : 0.0f;                                      // (rows above `half` of the left half: the LL quadrant)
        r.eo = left ? 0xFFFFFFFFu : so[ie]; r.ec = 0; r.el = 0;
        if (r.eo != 0xFFFFFFFFu && (unsigned long long) r.eo <= B) { r.ec = C[ie]; r.el = li[ie]; }
        r.oo = so[io]; r.oc = 0; r.ol = 0;
        if (r.oo != 0xFFFFFFFFu && (unsigned long long) r.oo <= B) { r.oc = C[io]; r.ol = li[io]; }
    };
    auto value = [&](const Raw &r, float &e, float &o) {
        e = left ? r.ll : prefix_value_of(r.eo, r.ec, r.el, B, rbase, rreach);
        o = prefix_value_of(r.oo, r.oc, r.ol, B, rbase, rreach);
    };
    RPipe p;
    Raw cur;
    fetch(jstart, cur);
    for (int j = jstart; j < kb + 4; j++) {
        Raw nxt;
        fetch(j + 1, nxt);
        float e_in, o_in, xe, xo;
        value(cur, e_in, o_in);
        p.step(j, half, e_in, o_in, xe, xo);
        const int k = j - 4;
        if (k >= ka && k < kb) {
            out[(size_t) (2 * k) * nx + col] = xe;
            out[(size_t) (2 * k + 1) * nx + col] = xo;
        }
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------
// The finest level of the probes WHOLE - vertical synthesis (two RPipes per lane: low-pass column k and high-pass column
// hx + k, inputs as in k_cols_inv_stream), then the horizontal synthesis of the two finished rows in registers
// (neighbours through wave shuffles: 60 pairs + 2 of halo either side per wave, lift_inverse_tile's expressions and
// boundary forms), then what k_rows_inv_use does with a row (residual_value, statistics against the frame).  Neither the
// column pass's result nor the grid is written: one write and one read of the padded frame less per probe.
// ------------------------------------------------------------------------------------------------
constexpr int kFusePairs = 60;
constexpr int kFuseMaxWaves = 16;
__global__ __launch_bounds__(64 * kFuseMaxWaves) void k_finest_inv_use(const float *__restrict__ src, Grid g, size_t np, const int32_t *__restrict__ Cb,
                                                        const uint32_t *__restrict__ sigordb, const uint32_t *__restrict__ lspidxb,
                                                        const unsigned long long *trunc_bits, const int *active, RowUse u, int strips, int n_frames, int pieces,
                                                        const int *frame_of, size_t src_stride)
{
    // workgroup = `blockDim.x / 64` neighbouring strips (one per wave) of tile blockIdx.y = piece * n_frames + frame;
    // one by default (launch_prefix_synthesis_stats has the measurements).
    // frame_of (cut slots, residual.hpp): "frame" is a slot - its state, cut, mask, LL input (src) and statistics are the
    // slot's own; the bookkeeping, the frame and the base layer are those of frame frame_of[slot].
    struct { int strip, frame, piece; } tb{(int) blockIdx.x * ((int) blockDim.x >> 6) + ((int) threadIdx.x >> 6), (int) blockIdx.y % n_frames, (int) blockIdx.y / n_frames};
    const int slot = tb.frame;
    if (active && !active[slot]) return;
    const int frame = frame_of ? frame_of[slot] : slot;
    const FrameState &fs = u.fs[slot];
    unsigned long long nb = trunc_bits[slot], bits0 = fs.budget + 128;   // spiht_decode: num_bits = min(num_bits, bits0) - 128 (spiht_re.c:495-500)
    if (nb > bits0) nb = bits0;
    const unsigned long long B = nb - 128;
    __shared__ unsigned int rbase[32], rreach[32];
    if (threadIdx.x < 32) { rbase[threadIdx.x] = fs.refine_base[threadIdx.x]; rreach[threadIdx.x] = fs.step_reached[threadIdx.x]; }
    __syncthreads();
    if (tb.strip >= strips) return;                                     // (after the only barrier)
    const int lane = (int) threadIdx.x & 63;
    const int nx = g.nx, ny = g.ny, half = ny >> 1, hx = nx >> 1;
    const int k = tb.strip * kFusePairs + lane - 2;                         // this lane's pair of output columns (2k, 2k + 1)
    const bool in_range = k >= 0 && k < hx;
    const bool owner = lane >= 2 && lane < 2 + kFusePairs && in_range;
    const int kc = min(max(k, 0), hx - 1);
    const float *a = src + (size_t) slot * src_stride;
    const int32_t *C = Cb + (size_t) frame * np;
    const uint32_t *so = sigordb + (size_t) frame * np, *li = lspidxb + (size_t) frame * np;
    const float *x = u.data + (size_t) frame * u.n_pix, *d = u.decoded + (size_t) frame * u.n_pix;
    const float dc = fs.dc, rmin = fs.rmin, rng = fs.rmax - fs.rmin;
    const int per = (half + pieces - 1) / pieces, ka = tb.piece * per, kb = min(half, ka + per);
    const size_t part = (size_t) slot * kPartials + (size_t) (tb.strip + strips * tb.piece);
    const int jstart = max(ka - 2, 0);
    // A probe that only has to answer "is the maximum error above the target?" (the truncation bisection's rounds,
    // /root/reference/src/ebcc_codec.c:788: cur > target moves trunc_lo and uses nothing else of the probe) is decided as
    // soon as one wave has seen such a sample: the waves that start after that leave at once.  Pieces are dispatched
    // piece-major over all frames, so most of an infeasible probe's waves never read their strip.
    const float exit_above = fs.exit_above;
    if (exit_above > 0.0f && __uint_as_float(__hip_atomic_load(&u.fs[slot].maxerr_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) > exit_above) {
        if (lane == 0) u.partial[part] = 0.0;
        return;
    }
    // the three detail coefficients of a vertical position (ordinal always, value and slot only inside the prefix) + LL
    struct Det { uint32_t o, l; int c; };
    struct Raw { float ll; Det lh, hl, hh; };
    auto det = [&](size_t i, Det &t) {
        t.o = so[i]; t.c = 0; t.l = 0;
        if (t.o != 0xFFFFFFFFu && (unsigned long long) t.o <= B) { t.c = C[i]; t.l = li[i]; }
    };
    auto fetch = [&](int j, Raw &r) {
        const int jj = min(j, half - 1);
        const size_t top = (size_t) jj * nx, bot = (size_t) (half + jj) * nx;
        r.ll = a[top + kc];
        det(bot + kc, r.lh);
        det(top + hx + kc, r.hl);
        det(bot + hx + kc, r.hh);
    };
    auto val = [&](const Det &t) { return prefix_value_of(t.o, t.c, t.l, B, rbase, rreach); };
    // horizontal synthesis of one finished row: this pair's low-/high-pass samples -> its two output samples (neighbours
    // through DPP wave shifts; the reference's boundary taps at the first and the last pair)
    const bool first = !(k > 0), last = !(k + 1 < hx);
    auto hsynth = [&](float e_raw, float o_raw, float &even, float &odd) {
        const float E0 = div_xi(e_raw), O0 = o_raw * kXi;
        const float Ol = lane_below(O0), Or = lane_above(O0);
        const float e1 = E0 - kDelta * (O0 + (first ? Or : Ol));
        const float e1l = lane_below(e1), e1r = lane_above(e1);
        const float o1 = O0 - kGamma * (e1 + (last ? e1l : e1r));
        const float o1l = lane_below(o1), o1r = lane_above(o1);
        const float e2 = e1 - kBeta * (o1 + (first ? o1r : o1l));
        const float e2r = lane_above(e2);
        even = e2;
        odd = last ? o1 - (2 * kAlpha) * e2 : o1 - kAlpha * (e2 + e2r);
    };
    double acc = 0;
    float mx = 0;
    // the two finished samples (columns 2k, 2k + 1) of output row y against the frame
    auto use_row = [&](bool mine, int y, float s0, float s1, float x0, float x1, float d0, float d1) {
        const bool m0 = mine && y < u.size_y && 2 * k < u.size_x, m1 = mine && y < u.size_y && 2 * k + 1 < u.size_x;
        const float t0 = x0 - (d0 + residual_value_fast(s0, dc, rmin, rng)), t1 = x1 - (d1 + residual_value_fast(s1, dc, rmin, rng));
        acc += m0 ? (double) t0 : 0.0;
        acc += m1 ? (double) t1 : 0.0;
        mx = m0 ? fmaxf(mx, fabsf(t0)) : mx;
        mx = m1 ? fmaxf(mx, fabsf(t1)) : mx;
    };
    RPipe pl, ph;
    Raw cur;
    fetch(jstart, cur);
    const int c0 = min(2 * kc, u.size_x - 1), c1 = min(2 * kc + 1, u.size_x - 1);
    // 8-byte reads of the frame and the base layer (columns cp, cp + 1): even width, 8-byte aligned bases; the PAIR index is
    // clamped (a clamped column would let a halo lane read one sample past the row, and past the buffer on the last one)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bool pair_io = (u.size_x & 1) == 0 && (u.n_pix & 1) == 0 && (((size_t) x | (size_t) d) & 7) == 0;
    const unsigned cp = 2u * (unsigned) min(kc, max((u.size_x >> 1) - 1, 0));
    for (int j = jstart; j < kb + 4; j++) {
        Raw nxt;
        fetch(j + 1, nxt);
        // the frame's and the base layer's samples at the four positions this step finishes (clamped: dropped when outside)
        const int kk = j - 4;
        const unsigned y0 = (unsigned) min(max(2 * kk, 0), u.size_y - 1), y1 = (unsigned) min(max(2 * kk + 1, 0), u.size_y - 1);
        float x00, x01, x10, x11, d00, d01, d10, d11;
        if (pair_io) {
            const unsigned i0 = y0 * (unsigned) u.size_x + cp, i1 = y1 * (unsigned) u.size_x + cp;
            const f32x2 xa = *reinterpret_cast<const f32x2 *>(x + i0), xb = *reinterpret_cast<const f32x2 *>(x + i1);
            const f32x2 da = *reinterpret_cast<const f32x2 *>(d + i0), db = *reinterpret_cast<const f32x2 *>(d + i1);
            x00 = xa.x; x01 = xa.y; x10 = xb.x; x11 = xb.y; d00 = da.x; d01 = da.y; d10 = db.x; d11 = db.y;
        } else {
            const unsigned i00 = y0 * (unsigned) u.size_x + c0, i01 = y0 * (unsigned) u.size_x + c1, i10 = y1 * (unsigned) u.size_x + c0, i11 = y1 * (unsigned) u.size_x + c1;
            x00 = x[i00]; x01 = x[i01]; x10 = x[i10]; x11 = x[i11]; d00 = d[i00]; d01 = d[i01]; d10 = d[i10]; d11 = d[i11];
        }
        float le, lo, he, ho;
        if (j >= 4 && j < half) {                                        // (uniform) no boundary form anywhere in the step
            pl.interior(cur.ll, val(cur.lh), le, lo);                    // low-pass column k: rows 2 kk, 2 kk + 1
            ph.interior(val(cur.hl), val(cur.hh), he, ho);               // high-pass column hx + k
        } else {
            pl.step(j, half, cur.ll, val(cur.lh), le, lo);
            ph.step(j, half, val(cur.hl), val(cur.hh), he, ho);
        }
        const bool mine = owner && kk >= ka && kk < kb;
        if (kk >= ka) {                                                  // (uniform: the warm-up steps of a piece put nothing out)
            float s00, s01, s10, s11;
            hsynth(le, he, s00, s01);
            hsynth(lo, ho, s10, s11);
            use_row(mine, 2 * kk, s00, s01, x00, x01, d00, d01);
            use_row(mine, 2 * kk + 1, s10, s11, x10, x11, d10, d11);
        }
        cur = nxt;
    }
    for (int q = 32; q >= 1; q >>= 1) { acc += __shfl_xor(acc, q); mx = fmaxf(mx, __shfl_xor(mx, q)); }
    if (lane == 0) {
        u.partial[part] = acc;
        atomicMax(&u.fs[slot].maxerr_bits, __float_as_uint(mx));
    }
}

// ------------------------------------------------------------------------------------------------
// A COARSER level of the probes whole, the same way: the level's four bands are floats of the grid (LL from `ll`, the
// three detail bands from `det` - the coarse reconstruction; the two differ from the second level on, whose LL is the
// level before's result), the ny x nx synthesised samples go to `dst` as the next level's LL.  Replaces the
// LDS-staged column pass + row pass of the level (two launches and a trip of the level through memory between them; at
// these sizes - 360 x 182 and 720 x 364 of a 1440 x 728 grid - the passes were bound by workgroup latency, not by bytes).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_level_inv_fused(const float *__restrict__ ll, const float *__restrict__ det, float *__restrict__ dst,
                                                         int stride, size_t np, int nx, int ny, const int *active, int strips, int n_frames, int pieces)
{
    const int strip = (int) blockIdx.x, frame = (int) blockIdx.y % n_frames, piece = (int) blockIdx.y / n_frames;
    if (active && !active[frame]) return;
    const int lane = (int) threadIdx.x;
    const int half = ny >> 1, hx = nx >> 1;
    const int k = strip * kFusePairs + lane - 2;                          // this lane's pair of output columns (2k, 2k + 1)
    const bool owner = lane >= 2 && lane < 2 + kFusePairs && k >= 0 && k < hx;
    const int kc = min(max(k, 0), hx - 1);
    const float *a = ll + (size_t) frame * np, *b = det + (size_t) frame * np;
    float *out = dst + (size_t) frame * np;
    const int per = (half + pieces - 1) / pieces, ka = piece * per, kb = min(half, ka + per);
    if (ka >= kb) return;
    const int jstart = max(ka - 2, 0);
    struct Raw { float ll, lh, hl, hh; };
    auto fetch = [&](int j, Raw &r) {
        const int jj = min(j, half - 1);
        const size_t top = (size_t) jj * stride, bot = (size_t) (half + jj) * stride;
        r.ll = a[top + kc];
        r.lh = b[bot + kc];
        r.hl = b[top + hx + kc];
        r.hh = b[bot + hx + kc];
    };
    const bool first = !(k > 0), last = !(k + 1 < hx);
    auto hsynth = [&](float e_raw, float o_raw, float &even, float &odd) {   // (k_finest_inv_use: lift_inverse_tile's expressions and boundary forms)
        const float E0 = div_xi(e_raw), O0 = o_raw * kXi;
        const float Ol = lane_below(O0), Or = lane_above(O0);
        const float e1 = E0 - kDelta * (O0 + (first ? Or : Ol));
        const float e1l = lane_below(e1), e1r = lane_above(e1);
        const float o1 = O0 - kGamma * (e1 + (last ? e1l : e1r));
        const float o1l = lane_below(o1), o1r = lane_above(o1);
        const float e2 = e1 - kBeta * (o1 + (first ? o1r : o1l));
        const float e2r = lane_above(e2);
        even = e2;
        odd = last ? o1 - (2 * kAlpha) * e2 : o1 - kAlpha * (e2 + e2r);
    };
    RPipe pl, ph;
    Raw cur;
    fetch(jstart, cur);
    for (int j = jstart; j < kb + 4; j++) {
        Raw nxt;
        fetch(j + 1, nxt);
        const int kk = j - 4;
        float le, lo, he, ho;
        if (j >= 4 && j < half) {
            pl.interior(cur.ll, cur.lh, le, lo);
            ph.interior(cur.hl, cur.hh, he, ho);
        } else {
            pl.step(j, half, cur.ll, cur.lh, le, lo);
            ph.step(j, half, cur.hl, cur.hh, he, ho);
        }
        if (kk >= ka) {                                                  // (uniform: the warm-up steps of a piece put nothing out)
            float s00, s01, s10, s11;
            hsynth(le, he, s00, s01);
            hsynth(lo, ho, s10, s11);
            if (owner && kk < kb) {
                *reinterpret_cast<float2 *>(out + (size_t) (2 * kk) * stride + 2 * k) = make_float2(s00, s01);
                *reinterpret_cast<float2 *>(out + (size_t) (2 * kk + 1) * stride + 2 * k) = make_float2(s10, s11);
            }
        }
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------
// residual min/max with "first occurrence wins" tie-break (reference findMinMaxf, ebcc_codec.c:515-533)
// ------------------------------------------------------------------------------------------------
__global__ void k_minmax_init(FrameState *fs, int n_frames)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_frames) {
        fs[f].rmin_key = ~0ull;
        fs[f].rmax_key = 0ull;
    }
}

__device__ inline unsigned long long wave_min_u64(unsigned long long v)
{
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(v, d);
        v = o < v ? o : v;
    }
    return v;
}
__device__ inline unsigned long long wave_max_u64(unsigned long long v)
{
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(v, d);
        v = o > v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_residual_minmax(const float *__restrict__ data,
                                                          const float *__restrict__ decoded, size_t n_pix,
                                                          FrameState *fs)
{
    const int frame = blockIdx.y;
    const float *x = data + (size_t) frame * n_pix;
    const float *d = decoded + (size_t) frame * n_pix;
    unsigned long long kmin = ~0ull, kmax = 0ull;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += (size_t) gridDim.x * blockDim.x) {
        float r = x[i] - d[i];                                   // ebcc_codec.c:713
        unsigned long long k = (unsigned long long) float_order_key(r) << 32;
        unsigned long long lo = k | (unsigned int) i, hi = k | (0xFFFFFFFFu - (unsigned int) i);
        kmin = lo < kmin ? lo : kmin;
        kmax = hi > kmax ? hi : kmax;
    }
    kmin = wave_min_u64(kmin);
    kmax = wave_max_u64(kmax);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&fs[frame].rmin_key, kmin);
        atomicMax(&fs[frame].rmax_key, kmax);
    }
}

__global__ void k_residual_minmax_finish(const float *data, const float *decoded, size_t n_pix, FrameState *fs,
                                         int n_frames)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    size_t imin = (unsigned int) fs[f].rmin_key;
    size_t imax = 0xFFFFFFFFu - (unsigned int) fs[f].rmax_key;
    const float *x = data + (size_t) f * n_pix;
    const float *d = decoded + (size_t) f * n_pix;
    fs[f].rmin = x[imin] - d[imin];
    fs[f].rmax = x[imax] - d[imax];
}

// ------------------------------------------------------------------------------------------------
// load_image (+ residual normalisation) and deterministic DC sum
// ------------------------------------------------------------------------------------------------
template <bool FROM_IMAGE>
__global__ __launch_bounds__(256) void k_pad_load(const float *__restrict__ data, const float *__restrict__ decoded,
                                                   float *__restrict__ A, Grid g, size_t n_pix, size_t np,
                                                   const FrameState *fs, double *partial, const int *active)
{
    __shared__ double red[256];
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    const float *x = data + (size_t) frame * n_pix;
    const float *d = FROM_IMAGE ? nullptr : decoded + (size_t) frame * n_pix;
    float *a = A + (size_t) frame * np;
    float rmin = 0, rng = 1;
    if (!FROM_IMAGE) {
        rmin = fs[frame].rmin;
        rng = fs[frame].rmax - fs[frame].rmin;
    }
    // fixed partition of the padded grid: block b owns [b*chunk, (b+1)*chunk)
    const size_t chunk = (np + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t) blockIdx.x * chunk;
    const size_t hi = lo + chunk < np ? lo + chunk : np;
    double acc = 0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        int y = (int) (i / g.nx), xx = (int) (i - (size_t) y * g.nx);
        float v;
        if (y >= g.size_y && xx >= g.size_x) {
            v = 0.0f;                                                   // dwt.h:74-76
        } else {
            int sy = y < g.size_y ? y : 2 * g.size_y - 1 - y;           // dwt.h:71-73
            int sx = xx < g.size_x ? xx : 2 * g.size_x - 1 - xx;        // dwt.h:68-70
            size_t si = (size_t) sy * g.size_x + sx;
            float rn;
            if (FROM_IMAGE) rn = x[si];
            else rn = ((x[si] - d[si]) - rmin) / rng;                   // ebcc_codec.c:731,745
            v = rn * 255.0f;                                            // dwt.h:65
        }
        a[i] = v;
        acc += (double) v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int) threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t) frame * kPartials + blockIdx.x] = red[0];
}

__global__ void k_dc_finish(const double *partial, int n_partials, size_t np, FrameState *fs, int n_frames, const int *active)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames || (active && !active[f])) return;
    double s = 0;
    for (int i = 0; i < n_partials; i++) s += partial[(size_t) f * kPartials + i];
    // The reference sums sequentially in double (dwt.h:324-329); only floor(mean) survives.  Both
    // orders are within eps of the exact sum; if the floor could differ, fall back to the exact order.
    const double eps = 0.0625 + (double) np * 1e-9;
    double m = floor(s / (double) np);
    double m_lo = floor((s - eps) / (double) np), m_hi = floor((s + eps) / (double) np);
    fs[f].dc_sum = s;
    fs[f].dc = (float) m;
    fs[f].dc_uncertain = (m_lo != m_hi) ? 1 : 0;
}

// rare fallback: exact reference order, one lane per frame
__global__ void k_dc_sequential(const float *A, size_t np, FrameState *fs, int n_frames, const int *active)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames || (active && !active[f]) || !fs[f].dc_uncertain) return;
    const float *a = A + (size_t) f * np;
    double s = 0;
    for (size_t i = 0; i < np; i++) s += a[i];
    s /= (double) np;
    fs[f].dc = (float) floor(s);
    fs[f].dc_uncertain = 0;
}

// ------------------------------------------------------------------------------------------------
// normalize (dwt.h:355-368) -> integer coefficients, running max (spiht_re.c:54-59)
// ------------------------------------------------------------------------------------------------
__global__ void k_cmax_init(FrameState *fs, int n_frames)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_frames) fs[f].cmax = 2;                              // "max = 2.0", spiht_re.c:32
}

__global__ __launch_bounds__(256) void k_truncate(const float *__restrict__ A, int32_t *__restrict__ C, size_t np,
                                                   FrameState *fs, const int *active)
{
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    const float *a = A + (size_t) frame * np;
    int32_t *c = C + (size_t) frame * np;
    int m = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (size_t) gridDim.x * blockDim.x) {
        float v = a[i];
        int q = (int) v;                                            // truncation toward zero == dwt.h:362-366
        c[i] = q;
        int aq = q < 0 ? -q : q;
        m = aq > m ? aq : m;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        int o = __shfl_xor(m, d);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > 2) atomicMax(&fs[frame].cmax, m);
}

// spiht_re.c:60: step = floor(log(max)/log(2.0)).  The quotient is evaluated by the host libm for every
// power of two (step_of_pow2[k]); for other integers < 2^24 the floor is unambiguous.
__global__ void k_top_step(FrameState *fs, int n_frames, const int *step_of_pow2)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    int m = fs[f].cmax;
    int k = 31 - __clz(m);
    fs[f].top_step = (m == (1 << k)) ? step_of_pow2[k] : k;
}

// ------------------------------------------------------------------------------------------------
// descendant maxima (replaces the recursion of spiht_re.c:160-206)
//   D[p] = max |c| over all descendants of p,  G[p] = max over descendants at depth >= 2
// ------------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(256) void k_descmax(const int32_t *__restrict__ C, int32_t *__restrict__ D,
                                                  int32_t *__restrict__ G, Grid g, size_t np, int rx, int ry,
                                                  int ex, int ey, int child_has_d, const int *active)
{
    const int frame = blockIdx.y;
    if (active && !active[frame]) return;
    const int32_t *c = C + (size_t) frame * np;
    int32_t *dd = D + (size_t) frame * np;
    int32_t *gg = G + (size_t) frame * np;
    const int total = rx * ry;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int y = i / rx, x = i - y * rx;
        if (x < ex && y < ey) continue;
        int ch = first_child(g, x, y);
        int dmax = 0, gmax = 0;
        if (ch >= 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int q = ch + (k & 1) + (k >> 1) * g.nx;
                int v = c[q];
                v = v < 0 ? -v : v;
                int dv = child_has_d ? dd[q] : 0;
                dmax = max(dmax, max(v, dv));
                gmax = max(gmax, dv);
            }
        }
        dd[x + y * g.nx] = dmax;
        gg[x + y * g.nx] = gmax;
    }
}

// ------------------------------------------------------------------------------------------------
// probe statistics / final combination
// ------------------------------------------------------------------------------------------------
__global__ void k_probe_init(FrameState *fs, int n_frames, const int *active)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_frames && (!active || active[f])) {
        fs[f].maxerr_bits = 0;
        fs[f].err_sum = 0;
    }
}

__global__ void k_probe_finish(const double *partial, int n_partials, FrameState *fs, int n_frames, const int *active)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames || (active && !active[f])) return;
    double s = 0;
    for (int i = 0; i < n_partials; i++) s += partial[(size_t) f * kPartials + i];
    fs[f].err_sum = s;
}

__global__ __launch_bounds__(256) void k_emit_image(float *__restrict__ out, const float *__restrict__ A, Grid g,
                                                     size_t n_pix, size_t np, const FrameState *fs)
{
    const int frame = blockIdx.y;
    float *o = out + (size_t) frame * n_pix;
    const float *a = A + (size_t) frame * np;
    const float dc = (float) fs[frame].dec_dc;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += (size_t) gridDim.x * blockDim.x) {
        int y = (int) (i / g.size_x), xx = (int) (i - (size_t) y * g.size_x);
        float v = floorf(a[(size_t) y * g.nx + xx] + dc);
        v = v > 255.0f ? 255.0f : (v < 0.0f ? 0.0f : v);
        o[i] = v / 255.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
template <typename K>
void allow_big_lds(K kernel, size_t bytes)
{
    if (bytes > 48 * 1024)
        EBCC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
}

void rows_fwd(const float *src, float *dst, const ResidualBuffers &rb, int n, int rows, int n_frames, int sub_dc,
              const int *active, hipStream_t s)
{
    dim3 grid(min(rows, 96), n_frames);
    size_t lds = (size_t) n * sizeof(float);
    hipLaunchKernelGGL(k_rows_fwd, grid, dim3(kRowThreads), lds, s, src, dst, rb.g.nx, rb.np, n, rows, rb.fs, sub_dc,
                       active);
}
void rows_inv(const float *src, float *dst, const ResidualBuffers &rb, int n, int rows, int n_frames,
              const int *active, hipStream_t s)
{
    dim3 grid(min(rows, 96), n_frames);
    size_t lds = (size_t) n * sizeof(float);
    hipLaunchKernelGGL(k_rows_inv, grid, dim3(kRowThreads), lds, s, src, dst, rb.g.nx, rb.np, n, rows, active);
}
template <bool FWD>
void cols_pass(const float *src, float *dst, const ResidualBuffers &rb, int n, int cols, int n_frames,
               const int *active, hipStream_t s)
{
    const size_t lds_limit = 156 * 1024;
    if ((size_t) n * 32 * sizeof(float) <= lds_limit) {
        size_t lds = (size_t) n * 32 * sizeof(float);
        auto k = FWD ? k_cols_fwd<32> : k_cols_inv<32>;
        allow_big_lds(k, lds);
        dim3 grid(ceil_div(cols, 32), n_frames);
        hipLaunchKernelGGL(k, grid, dim3(kColThreads), lds, s, src, dst, rb.g.nx, rb.np, n, cols, active);
    } else {
        size_t lds = (size_t) n * 16 * sizeof(float);
        auto k = FWD ? k_cols_fwd<16> : k_cols_inv<16>;
        allow_big_lds(k, lds);
        dim3 grid(ceil_div(cols, 16), n_frames);
        hipLaunchKernelGGL(k, grid, dim3(kColThreads), lds, s, src, dst, rb.g.nx, rb.np, n, cols, active);
    }
}

}  // namespace

// ================================================================================================
void launch_residual_minmax(const float *data, const float *decoded, int n_frames, size_t n_pix, FrameState *fs,
                            hipStream_t s)
{
    hipLaunchKernelGGL(k_minmax_init, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, fs, n_frames);
    hipLaunchKernelGGL(k_residual_minmax, dim3(64, n_frames), dim3(256), 0, s, data, decoded, n_pix, fs);
    hipLaunchKernelGGL(k_residual_minmax_finish, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, data, decoded, n_pix,
                       fs, n_frames);
    EBCC_HIP_LAUNCH_CHECK();
}

static void finish_dc(const ResidualBuffers &rb, int n_frames, const int *active, hipStream_t s)
{
    hipLaunchKernelGGL(k_dc_finish, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, rb.partial, kPartials, rb.np, rb.fs,
                       n_frames, active);
    hipLaunchKernelGGL(k_dc_sequential, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, rb.A, rb.np, rb.fs, n_frames, active);
}

void launch_pad_and_dc(const float *data, const float *decoded, const ResidualBuffers &rb, int n_frames,
                       const int *d_active, hipStream_t s)
{
    size_t n_pix = (size_t) rb.g.size_x * rb.g.size_y;
    hipLaunchKernelGGL(k_pad_load<false>, dim3(kPartials, n_frames), dim3(256), 0, s, data, decoded, rb.A, rb.g, n_pix,
                       rb.np, rb.fs, rb.partial, d_active);
    finish_dc(rb, n_frames, d_active, s);
    EBCC_HIP_LAUNCH_CHECK();
}

void launch_pad_and_dc_from_image(const float *image, const ResidualBuffers &rb, int n_frames, hipStream_t s)
{
    size_t n_pix = (size_t) rb.g.size_x * rb.g.size_y;
    hipLaunchKernelGGL(k_pad_load<true>, dim3(kPartials, n_frames), dim3(256), 0, s, image, (const float *) nullptr,
                       rb.A, rb.g, n_pix, rb.np, rb.fs, rb.partial, (const int *) nullptr);
    finish_dc(rb, n_frames, nullptr, s);
    EBCC_HIP_LAUNCH_CHECK();
}

static int *g_step_table_dev[64] = {nullptr};   // device copies (one per device) of floor(log(2^k)/log(2.0)) evaluated by the host libm

static const int *step_table(hipStream_t s)
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    EBCC_HIP_CHECK(hipGetDevice(&dev));
    int *&g_step_table = g_step_table_dev[dev & 63];
    if (!g_step_table) {
        int h[32];
        for (int k = 0; k < 32; k++) {
            float m = (float) (1u << k);
            h[k] = (int) floor(log((double) m) / log(2.0));        // spiht_re.c:60 with max = 2^k
        }
        EBCC_HIP_CHECK(device_malloc((void **) &g_step_table, sizeof h));
        EBCC_HIP_CHECK(hipMemcpyAsync(g_step_table, h, sizeof h, hipMemcpyHostToDevice, s));
        wait_stream(s);
    }
    return g_step_table;
}

void launch_analysis(const ResidualBuffers &rb, int n_frames, const int *d_active, hipStream_t s)
{
    const Grid &g = rb.g;
    int nx = g.nx, ny = g.ny;
    for (int lv = 0; lv < g.stages; lv++) {                         // dwt.h:297-301
        rows_fwd(rb.A, rb.T, rb, nx, ny, n_frames, lv == 0, d_active, s);
        cols_pass<true>(rb.T, rb.A, rb, ny, nx, n_frames, d_active, s);
        nx >>= 1; ny >>= 1;
    }
    hipLaunchKernelGGL(k_cmax_init, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, rb.fs, n_frames);
    hipLaunchKernelGGL(k_truncate, dim3(128, n_frames), dim3(256), 0, s, rb.A, rb.C, rb.np, rb.fs, d_active);
    hipLaunchKernelGGL(k_top_step, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, rb.fs, n_frames, step_table(s));
    for (int lv = 1; lv < g.stages; lv++) {
        int rx = g.nx >> lv, ry = g.ny >> lv;
        hipLaunchKernelGGL(k_descmax, dim3(ceil_div(rx * ry, 256), n_frames), dim3(256), 0, s, rb.C, rb.D, rb.G, g,
                           rb.np, rx, ry, rx >> 1, ry >> 1, lv > 1 ? 1 : 0, d_active);
    }
    hipLaunchKernelGGL(k_descmax, dim3(ceil_div(g.lx * g.ly, 256), n_frames), dim3(256), 0, s, rb.C, rb.D, rb.G, g,
                       rb.np, g.lx, g.ly, 0, 0, g.stages > 1 ? 1 : 0, d_active);
    EBCC_HIP_LAUNCH_CHECK();
    EBCC_HIP_LAUNCH_CHECK();
}

void launch_synthesis(const ResidualBuffers &rb, int n_frames, const int *d_active, hipStream_t s)
{
    const Grid &g = rb.g;
    for (int lv = g.stages - 1; lv >= 0; lv--) {                    // dwt.h:309-315
        int nx = g.nx >> lv, ny = g.ny >> lv;
        cols_pass<false>(rb.A, rb.T, rb, ny, nx, n_frames, d_active, s);
        rows_inv(rb.T, rb.A, rb, nx, ny, n_frames, d_active, s);
    }
    EBCC_HIP_LAUNCH_CHECK();
}

// the synthesis whose last row pass consumes the rows (RowUse) instead of storing the grid; returns the workgroups
// per frame of that pass (= partial sums per frame)
// every pass of the synthesis but the last row pass (result in rb.T)
void launch_synthesis_head(const ResidualBuffers &rb, int n_frames, const int *d_active, hipStream_t s)
{
    const Grid &g = rb.g;
    for (int lv = g.stages - 1; lv >= 1; lv--) {
        int nx = g.nx >> lv, ny = g.ny >> lv;
        cols_pass<false>(rb.A, rb.T, rb, ny, nx, n_frames, d_active, s);
        rows_inv(rb.T, rb.A, rb, nx, ny, n_frames, d_active, s);
    }
    cols_pass<false>(rb.A, rb.T, rb, g.ny, g.nx, n_frames, d_active, s);
    EBCC_HIP_LAUNCH_CHECK();
}

static int synthesis_tail(const ResidualBuffers &rb, int n_frames, const int *d_active, hipStream_t s, RowUse u)
{
    const Grid &g = rb.g;
    const int blocks = min(g.ny, 96);
    u.size_x = g.size_x; u.size_y = g.size_y; u.n_pix = (size_t) g.size_x * g.size_y; u.fs = rb.fs; u.partial = rb.partial;
    hipLaunchKernelGGL(k_rows_inv_use, dim3(blocks, n_frames), dim3(kRowThreads), (size_t) g.nx * sizeof(float), s, rb.T, g.nx, rb.np,
                       g.nx, g.ny, d_active, u);
    return blocks;
}

// synthesis + statistics of data - (decoded + residual) in one go (the grid itself is not kept)
void launch_synthesis_stats(const float *data, const float *decoded, const ResidualBuffers &rb, int n_frames, const int *d_active,
                            hipStream_t s)
{
    hipLaunchKernelGGL(k_probe_init, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, rb.fs, n_frames, d_active);
    RowUse u{};
    u.data = data; u.decoded = decoded;
    launch_synthesis_head(rb, n_frames, d_active, s);
    const int partials = synthesis_tail(rb, n_frames, d_active, s, u);
    hipLaunchKernelGGL(k_probe_finish, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, rb.partial, partials, rb.fs, n_frames, d_active);
    EBCC_HIP_LAUNCH_CHECK();
}

// launch_reconstruct + launch_synthesis_stats for the probes of the truncation search, without the finest level's
// detour through the grid: coarse quadrant reconstructed, coarser levels as usual, then k_cols_inv_stream and the
// consuming row pass.
void launch_prefix_synthesis_stats(const float *data, const float *decoded, const ResidualBuffers &rb, int n_frames,
                                   const unsigned long long *d_trunc_bits, const int *d_active, hipStream_t s)
{
    const Grid &g = rb.g;
    if (g.stages < 2 || g.ny < 32) {
        launch_reconstruct(rb, n_frames, d_trunc_bits, d_active, s);
        launch_synthesis_stats(data, decoded, rb, n_frames, d_active, s);
        return;
    }
    hipLaunchKernelGGL(k_probe_init, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, rb.fs, n_frames, d_active);
    launch_reconstruct_coarse(rb, n_frames, d_trunc_bits, d_active, s);
    // the coarser levels, each whole in one launch (k_level_inv_fused): the detail bands stay where the reconstruction put
    // them (rb.A), a level's result - the next level's LL - alternates between rb.T and the descendant-maxima grid, which
    // nothing reads between the SPIHT encoder and the next batch's analysis (A/B against the LDS-staged column + row passes,
    // three alternating runs each: the residual layer + truncation phase of a slice 18.2 - 18.7 ms against 18.4 - 19.2)
    const float *fine_ll = rb.A;
    const bool finest_fused = ceil_div(g.nx >> 1, kFusePairs) * 8 <= kPartials && (g.nx >> 1) >= 2;   // (the stream form of the finest level reads rb.A and writes rb.T)
    if (finest_fused && (long long) n_frames * 8 <= 65535) {
        float *scratch[2] = {rb.T, reinterpret_cast<float *>(rb.D)};
        const float *ll = rb.A;
        int turn = 0;
        for (int lv = g.stages - 1; lv >= 1; lv--) {
            const int nx = g.nx >> lv, ny = g.ny >> lv;
            const int lstrips = ceil_div(nx >> 1, kFusePairs), lpieces = std::max(1, std::min(8, (ny >> 1) / 16));
            hipLaunchKernelGGL(k_level_inv_fused, dim3((unsigned) lstrips, (unsigned) (n_frames * lpieces)), dim3(64), 0, s, ll, rb.A, scratch[turn], g.nx, rb.np,
                               nx, ny, d_active, lstrips, n_frames, lpieces);
            ll = scratch[turn];
            turn ^= 1;
        }
        fine_ll = ll;
    } else {
        for (int lv = g.stages - 1; lv >= 1; lv--) {
            int nx = g.nx >> lv, ny = g.ny >> lv;
            cols_pass<false>(rb.A, rb.T, rb, ny, nx, n_frames, d_active, s);
            rows_inv(rb.T, rb.A, rb, nx, ny, n_frames, d_active, s);
        }
    }
    int pieces = std::max(1, std::min(8, (g.ny >> 1) / 16));
    while (pieces > 1 && (long long) n_frames * pieces > 65535) pieces--;                // (grid y)
    RowUse u{};
    u.data = data; u.decoded = decoded;
    int partials;
    const int strips = ceil_div(g.nx >> 1, kFusePairs);
    if (strips * pieces <= kPartials && (g.nx >> 1) >= 2) {
        u.size_x = g.size_x; u.size_y = g.size_y; u.n_pix = (size_t) g.size_x * g.size_y; u.fs = rb.fs; u.partial = rb.partial;
        // (one strip per workgroup: with several strips as the waves of one workgroup this kernel got slower - 564 us per
        //  probe round with 1, 571 with 4, 788 with 6, 898 with all 12, tools/gpu/kstat.sh)
        const int wave_cap = 1;
        const int wg = std::min(strips, wave_cap);
        hipLaunchKernelGGL(k_finest_inv_use, dim3((unsigned) ceil_div(strips, wg), (unsigned) (n_frames * pieces)), dim3(64 * wg), 0, s, fine_ll, g, rb.np, rb.C, rb.sigord, rb.lspidx,
                           d_trunc_bits, d_active, u, strips, n_frames, pieces, (const int *) nullptr, rb.np);
        partials = strips * pieces;
    } else {
        hipLaunchKernelGGL(k_cols_inv_stream, dim3(ceil_div(g.nx, kStreamT), n_frames, pieces), dim3(kStreamT), 0, s, rb.A, rb.T, g, rb.np, rb.C,
                           rb.sigord, rb.lspidx, rb.fs, d_trunc_bits, d_active);
        partials = synthesis_tail(rb, n_frames, d_active, s, u);
    }
    hipLaunchKernelGGL(k_probe_finish, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, rb.partial, partials, rb.fs, n_frames, d_active);
    EBCC_HIP_LAUNCH_CHECK();
}

// ---- per-frame parameters of the residual layer set where they are used (no trip of the frame states through the host)
__global__ void k_residual_budget(FrameState *fs, const unsigned long long *bits0, int n_frames, const int *active)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames || (active && !active[f])) return;
    fs[f].budget = bits0[f] - 128;                                        // trunc_bits (src/spiht/spiht_re.c:433-446: bits0 = trunc_bits + 128)
    fs[f].exit_above = 0.0f;                                              // (probes are exact unless a search says otherwise)
}
__global__ void k_whole_stream_cut(FrameState *fs, unsigned long long *trunc_bits, int n_frames, const int *active)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames || (active && !active[f])) return;
    trunc_bits[f] = (unsigned long long) fs[f].stream_bytes * 8ull;       // src/ebcc_codec.c:749: decode of everything that was written
    fs[f].dec_dc = (int) fs[f].dc;
}
void launch_residual_budget(const ResidualBuffers &rb, int n_frames, const unsigned long long *d_bits0, const int *d_active, hipStream_t s)
{
    hipLaunchKernelGGL(k_residual_budget, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, rb.fs, d_bits0, n_frames, d_active);
    EBCC_HIP_LAUNCH_CHECK();
}
void launch_whole_stream_cut(const ResidualBuffers &rb, int n_frames, unsigned long long *d_trunc_bits, const int *d_active, hipStream_t s)
{
    hipLaunchKernelGGL(k_whole_stream_cut, dim3(ceil_div(n_frames, 64)), dim3(64), 0, s, rb.fs, d_trunc_bits, n_frames, d_active);
    EBCC_HIP_LAUNCH_CHECK();
}

// ---- cut slots (residual.hpp): the same probe for slot v = cut cs.bits[v] of frame cs.frame_of[v]
void launch_reconstruct_coarse_slots(const ResidualBuffers &rb, const CutSlots &cs, int n_slots, hipStream_t s);   // residual_spiht.hip
__global__ void k_slots_setup(FrameState *vfs, const FrameState *fs, const int *frame_of, int n_slots)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n_slots) vfs[v] = fs[frame_of[v]];
}
bool prefix_slots_supported(const ResidualBuffers &rb, int n_slots)
{
    const Grid &g = rb.g;
    if (g.stages < 2 || g.ny < 32 || (g.nx >> 1) < 2) return false;
    const int pieces = std::max(1, std::min(8, (g.ny >> 1) / 16));
    return ceil_div(g.nx >> 1, kFusePairs) * 8 <= kPartials && (long long) n_slots * 8 <= 65535 && ceil_div(g.nx >> 1, kFusePairs) * pieces <= kPartials;
}
void launch_slots_setup(const ResidualBuffers &rb, const CutSlots &cs, int n_slots, hipStream_t s)
{
    hipLaunchKernelGGL(k_slots_setup, dim3(ceil_div(n_slots, 64)), dim3(64), 0, s, cs.fs, rb.fs, cs.frame_of, n_slots);
    EBCC_HIP_LAUNCH_CHECK();
}
void launch_prefix_synthesis_slots(const float *data, const float *decoded, const ResidualBuffers &rb, const CutSlots &cs, int n_slots, hipStream_t s)
{
    const Grid &g = rb.g;
    hipLaunchKernelGGL(k_probe_init, dim3(ceil_div(n_slots, 64)), dim3(64), 0, s, cs.fs, n_slots, cs.active);
    launch_reconstruct_coarse_slots(rb, cs, n_slots, s);
    float *scratch[2] = {cs.T, cs.D};
    const float *ll = cs.A;
    int turn = 0;
    for (int lv = g.stages - 1; lv >= 1; lv--) {
        const int nx = g.nx >> lv, ny = g.ny >> lv;
        const int lstrips = ceil_div(nx >> 1, kFusePairs), lpieces = std::max(1, std::min(8, (ny >> 1) / 16));
        hipLaunchKernelGGL(k_level_inv_fused, dim3((unsigned) lstrips, (unsigned) (n_slots * lpieces)), dim3(64), 0, s, ll, cs.A, scratch[turn], g.nx, cs.stride,
                           nx, ny, cs.active, lstrips, n_slots, lpieces);
        ll = scratch[turn];
        turn ^= 1;
    }
    int pieces = std::max(1, std::min(8, (g.ny >> 1) / 16));
    while (pieces > 1 && (long long) n_slots * pieces > 65535) pieces--;
    RowUse u{};
    u.data = data; u.decoded = decoded;
    u.size_x = g.size_x; u.size_y = g.size_y; u.n_pix = (size_t) g.size_x * g.size_y; u.fs = cs.fs; u.partial = cs.partial;
    const int strips = ceil_div(g.nx >> 1, kFusePairs);
    hipLaunchKernelGGL(k_finest_inv_use, dim3((unsigned) strips, (unsigned) (n_slots * pieces)), dim3(64), 0, s, ll, g, rb.np, rb.C, rb.sigord, rb.lspidx,
                       cs.bits, cs.active, u, strips, n_slots, pieces, cs.frame_of, cs.stride);
    hipLaunchKernelGGL(k_probe_finish, dim3(ceil_div(n_slots, 64)), dim3(64), 0, s, cs.partial, strips * pieces, cs.fs, n_slots, cs.active);
    EBCC_HIP_LAUNCH_CHECK();
}

// after launch_synthesis_head: the last row pass with out += residual (ebcc_codec.c:1307) instead of a stored grid
void launch_synthesis_tail_add(float *out, const ResidualBuffers &rb, int n_frames, const int *d_active, hipStream_t s)
{
    RowUse u{};
    u.out = out;
    synthesis_tail(rb, n_frames, d_active, s, u);
    EBCC_HIP_LAUNCH_CHECK();
}

void launch_emit_image(float *image_out, const ResidualBuffers &rb, int n_frames, hipStream_t s)
{
    size_t n_pix = (size_t) rb.g.size_x * rb.g.size_y;
    hipLaunchKernelGGL(k_emit_image, dim3(128, n_frames), dim3(256), 0, s, image_out, rb.A, rb.g, n_pix, rb.np, rb.fs);
    EBCC_HIP_LAUNCH_CHECK();
}

// host check of the division-free forms above against the divisions they replace (ebcc_hip_selfcheck): v / 255 for the 256
// values it is used on, x / kXi for every significand at several exponents, zeros of both signs, and a sweep of all
// exponents (the subnormal-remainder range takes the real division)
int residual_selfcheck_divisions()
{
    int bad = 0;
    for (int q = 0; q <= 255; q++) { const volatile float v = (float) q; if (div255_exact(v) != v / 255.0f) bad++; }
    auto as_float = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    auto same = [&](float x) { const volatile float want = x / 1.149604398f; const float got = div_xi(x); return memcmp(&got, (const void *) &want, 4) == 0; };
    for (uint32_t e : {127u, 132u, 110u, 150u})
        for (uint32_t m = 0; m < (1u << 23); m++) if (!same(as_float((e << 23) | m))) bad++;
    for (uint32_t e = 0; e < 255; e++)
        for (uint32_t m = 0; m < (1u << 23); m += 4099) { if (!same(as_float((e << 23) | m))) bad++; if (!same(as_float(0x80000000u | (e << 23) | m))) bad++; }
    if (!same(0.0f) || !same(-0.0f)) bad++;
    return bad;
}

}  // namespace ebcc
