/*
 * codec_oracle.c - CPU restatement of the EBCC frame codec orchestration.  TEST INFRASTRUCTURE
 * ONLY (see oracle.h).  Follows /root/reference/src/ebcc_codec.c; every block cites the lines.
 *
 * Third-party pieces the reference links but does not vendor:
 *   - OpenJPEG (J2K base layer): either j2k_oracle.c (backend 0) or the image's
 *     libopenjp2.so.7 = 2.4.0 via dlopen (backend 1, built when the header is available).
 *   - zstd: dlopen'd (conda 1.4.9 preferred so level-22 bytes match the reference build).
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <assert.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define WAVELET_LEVELS 3                       /* ebcc_codec.c:28 */
#define DIM_MIN 32                             /* ebcc_codec.h:16 */
#define DIM_MAX 2047                           /* ebcc_codec.h:17 */

static int g_backend = 0;
static orc_trace_t g_trace;
void orc_set_j2k_backend(int b) { g_backend = b; }
int orc_get_j2k_backend(void) { return g_backend; }
void orc_last_trace(orc_trace_t *t) { *t = g_trace; }
void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------ zstd via dlopen */
typedef size_t (*zstd_bound_fn)(size_t);
typedef size_t (*zstd_comp_fn)(void *, size_t, const void *, size_t, int);
typedef size_t (*zstd_decomp_fn)(void *, size_t, const void *, size_t);
typedef unsigned (*zstd_iserr_fn)(size_t);
static zstd_bound_fn z_bound; static zstd_comp_fn z_comp; static zstd_decomp_fn z_decomp; static zstd_iserr_fn z_iserr;

static int zstd_load(void)
{
    if (z_comp) return 1;
    const char *names[] = { "/opt/conda/lib/libzstd.so.1", "libzstd.so.1", "libzstd.so", NULL };
    void *h = NULL;
    for (int i = 0; names[i] && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);   // DEEPBIND: never mix with another zstd already in the process
    if (!h) { fprintf(stderr, "oracle: libzstd not found\n"); return 0; }
    z_bound = (zstd_bound_fn) dlsym(h, "ZSTD_compressBound");
    z_comp = (zstd_comp_fn) dlsym(h, "ZSTD_compress");
    z_decomp = (zstd_decomp_fn) dlsym(h, "ZSTD_decompress");
    z_iserr = (zstd_iserr_fn) dlsym(h, "ZSTD_isError");
    return z_comp && z_decomp && z_bound;
}

/* ------------------------------------------------------------------ J2K back-ends */
#ifdef ORC_HAVE_OPENJPEG
size_t orc_opj_encode_tiled(const uint16_t *img, size_t tiles, size_t height, size_t width, float base_cr, uint8_t **out);
size_t orc_opj_decode(const uint8_t *cs, size_t n, int32_t **samples, size_t *h, size_t *w);
#endif

/* j2k_encode_internal, ebcc_codec.c:105-180: a chunk of several frames is one image with one tile per frame */
static size_t j2k_enc(const uint16_t *img, size_t tiles, size_t frame_h, size_t w, float cr, uint8_t **out)
{
    g_trace.n_j2k_encodes++;
#ifdef ORC_HAVE_OPENJPEG
    if (g_backend == 1) return orc_opj_encode_tiled(img, tiles, frame_h, w, cr, out);
#endif
    return orc_j2k_encode_tiled(img, tiles, frame_h, w, cr, out);
}

/* j2k_decode_internal, ebcc_codec.c:1092-1136: samples -> (s/65535)*(max-min)+min */
static size_t j2k_dec(const uint8_t *cs, size_t n, float minv, float maxv, float **data, size_t *h, size_t *w)
{
    g_trace.n_j2k_decodes++;
    int32_t *s = NULL; size_t hh = 0, ww = 0, npx;
#ifdef ORC_HAVE_OPENJPEG
    if (g_backend == 1) npx = orc_opj_decode(cs, n, &s, &hh, &ww); else
#endif
    npx = orc_j2k_decode(cs, n, &s, &hh, &ww);
    if (!npx) return 0;
    if (!*data) *data = (float *) malloc(npx * sizeof(float));
    for (size_t i = 0; i < npx; i++)
        (*data)[i] = ((float) s[i] / (uint16_t) -1) * (maxv - minv) + minv;      /* :1130 */
    free(s);
    if (h) *h = hh;
    if (w) *w = ww;
    return npx;
}

/* ------------------------------------------------------------------ error statistics, :477-533 */
static float max_abs_error(const float *x, const float *d, const float *r, size_t n)
{
    float m = 0;
    for (size_t i = 0; i < n; i++) {
        float rv = r ? r[i] : 0;
        float e = fabsf(x[i] - (d[i] + rv));
        if (e > m) m = e;
    }
    return m;
}
static double mean_error(const float *x, const float *d, const float *r, size_t n)
{
    double s = 0;
    for (size_t i = 0; i < n; i++) {
        float rv = r ? r[i] : 0;
        s += x[i] - (d[i] + rv);
    }
    return s / n;
}
static double feasible_fraction(const float *x, const float *d, size_t n, float target)
{
    size_t bad = 0;
    for (size_t i = 0; i < n; i++)
        if (fabsf(x[i] - (d[i] + 0.0f)) > target) bad++;
    return 1. - ((double) bad / n);
}
static void min_max(const float *a, size_t n, float *mn, float *mx)
{
    float lo = a[0], hi = a[0];
    for (size_t i = 0; i < n; i++) {
        if (a[i] < lo) lo = a[i];
        if (a[i] > hi) hi = a[i];
    }
    *mn = lo; *mx = hi;
}
static float data_range(const float *a, size_t n)
{
    float lo, hi;
    min_max(a, n, &lo, &hi);
    return hi - lo;
}

/* ------------------------------------------------------------------ rate search, :535-596 */
typedef struct {
    const uint16_t *scaled; size_t tiles, frame_h, h, w, n;
    const float *data; float minv, maxv, target;
    uint8_t *cs; size_t cs_len;          /* last codestream produced */
    float *decoded;
} search_t;

static double probe(search_t *s, float cr)
{
    free(s->cs); s->cs = NULL;
    s->cs_len = j2k_enc(s->scaled, s->tiles, s->frame_h, s->w, cr, &s->cs);
    j2k_dec(s->cs, s->cs_len, s->minv, s->maxv, &s->decoded, NULL, NULL);
    return feasible_fraction(s->data, s->decoded, s->n, s->target);
}

static float rate_search(search_t *s, float cr, double q_target)
{
    float lo = cr, hi = cr;
    double q = feasible_fraction(s->data, s->decoded, s->n, s->target);
    double q0 = q;
    const double eps = 1e-8;
    while (q < q_target && lo >= 1. / 2) { lo /= 2; q = probe(s, lo); }        /* :559-563 */
    q = q0;
    while (q >= q_target && hi <= 1000) { hi *= 2; q = probe(s, hi); }         /* :565-569 */
    if (q >= q_target) return hi;                                              /* :571-574 */
    q = q0;
    while ((fabs(q - q_target) > eps || q == 1.0) && hi - lo > 1.) {           /* :579-588 */
        cr = (lo + hi) / 2;
        q = probe(s, cr);
        if (q < q_target) hi = cr; else lo = cr;
    }
    probe(s, lo);                                                              /* :590 */
    return lo;
}

/* ------------------------------------------------------------------ stream header, :190-202 */
#pragma pack(push, 1)
typedef struct {
    uint8_t magic[4]; uint8_t version; uint8_t flags; uint16_t reserved;
    uint32_t minval_bits, maxval_bits; uint64_t coeffs_size;
    uint32_t rmin_bits, rmax_bits; uint64_t compressed_size; uint64_t tail_size;
} frame_hdr_t;
typedef struct {
    uint8_t magic[4]; uint32_t version, ndims, reserved;
    uint64_t dims[3], chunk_dims[3], num_chunks, chunk_size;
} chunk_hdr_t;
#pragma pack(pop)
_Static_assert(sizeof(frame_hdr_t) == 48, "frame header");
_Static_assert(sizeof(chunk_hdr_t) == 80, "chunk header");

static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static int dims_ok(const size_t d[3])
{
    /* :286-297 */
    if (d[0] == 0 || d[1] == 0) return 0;
    size_t hh = d[0] * d[1];
    if (d[0] != 0 && hh / d[0] != d[1]) return 0;
    return hh >= DIM_MIN && hh <= DIM_MAX && d[2] >= DIM_MIN && d[2] <= DIM_MAX;
}

size_t orc_ebcc_encode(const float *data, const orc_config_t *cfg, uint8_t **out)
{
    memset(&g_trace, 0, sizeof g_trace);
    if (!dims_ok(cfg->dims)) return 0;                                         /* :613-617 */

    /* env switches, :634-650 */
    double base_error_quantile = 1e-6;
    const char *e;
    if ((e = getenv("EBCC_INIT_BASE_ERROR_QUANTILE"))) base_error_quantile = strtod(e, NULL);
    int no_fallback = getenv("EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK") != NULL;
    int no_consistency = getenv("EBCC_DISABLE_PURE_BASE_COMPRESSION_FALLBACK_CONSISTENCY") != NULL;
    int no_mean_adjust = getenv("EBCC_DISABLE_MEAN_ADJUSTMENT") != NULL;
    double q_target = 1 - base_error_quantile;

    size_t H = cfg->dims[0] * cfg->dims[1], W = cfg->dims[2];
    size_t frame_h = cfg->dims[1];
    size_t n = (H / frame_h) * (frame_h * W);                                  /* :668-672 */
    for (size_t i = 0; i < n; i++)
        if (isnan(data[i]) || isinf(data[i])) { fprintf(stderr, "oracle: NaN/Inf in input\n"); exit(1); }

    float minv, maxv;
    min_max(data, n, &minv, &maxv);
    int const_field = minv == maxv;                                            /* :678 */

    int mode = cfg->residual_compression_type;
    uint16_t *scaled = NULL;
    uint8_t *tail = NULL; size_t tail_len = 0;
    uint8_t *zbytes = NULL; size_t zlen = 0;
    size_t coeffs_size = 0;
    float rmin = 0, rmax = 0;
    double mean_err = 0;
    search_t S; memset(&S, 0, sizeof S);

    if (!const_field) {
        scaled = (uint16_t *) malloc(n * sizeof(uint16_t));
        for (size_t i = 0; i < n; i++)
            scaled[i] = ((data[i] - minv) / (maxv - minv)) * (uint16_t) -1;    /* :688 */
        S.scaled = scaled; S.tiles = H / frame_h; S.frame_h = frame_h; S.h = H; S.w = W; S.n = n; S.data = data; S.minv = minv; S.maxv = maxv;
        S.cs_len = j2k_enc(scaled, S.tiles, frame_h, W, cfg->base_cr, &S.cs);                 /* :693 */
        if (mode == ORC_NONE) { tail = S.cs; tail_len = S.cs_len; S.cs = NULL; }
    }

    if (mode != ORC_NONE && !const_field) {
        float *residual = (float *) malloc(n * sizeof(float));
        float *rnorm = (float *) malloc(n * sizeof(float));
        S.decoded = (float *) malloc(n * sizeof(float));
        j2k_dec(S.cs, S.cs_len, minv, maxv, &S.decoded, NULL, NULL);           /* :707 */
        mean_err = mean_error(data, S.decoded, NULL, n);                       /* :709 */
        for (size_t i = 0; i < n; i++) residual[i] = data[i] - S.decoded[i];
        min_max(residual, n, &rmin, &rmax);                                    /* :716 */

        float target = cfg->error, cr = cfg->base_cr;
        if (mode == ORC_RELATIVE_ERROR) target *= data_range(data, n);         /* :724-726 */
        S.target = target;
        uint8_t *coeffs = NULL; size_t coeffs_orig = 0;
        int need_pure = 0, pure_done = 0;
        float cur, best_err; int skip;

        if (mode != ORC_MAX_ERROR && mode != ORC_RELATIVE_ERROR) goto after_search;   /* :721 (quirk Q2) */
        cr = rate_search(&S, cr, q_target);                                    /* :728 */
        for (size_t i = 0; i < n; i++) residual[i] = data[i] - S.decoded[i];
        min_max(residual, n, &rmin, &rmax);
        cur = fmaxf(fabsf(rmin), fabsf(rmax));
        best_err = -1;
        skip = cur <= target;                                              /* :737 */
        pure_done = q_target == 1.0;                                           /* :738 */

        if (!skip) {
            for (size_t i = 0; i < n; i++) rnorm[i] = (residual[i] - rmin) / (rmax - rmin);   /* :745 */
            size_t trunc_bits = S.cs_len * 8;                                  /* :747 */
            orc_spiht_encode(rnorm, H, W, &coeffs, &coeffs_orig, trunc_bits, WAVELET_LEVELS);
            orc_spiht_decode(coeffs, coeffs_orig, rnorm, H, W, coeffs_orig * 8);
            g_trace.n_spiht_decodes++;
            coeffs_size = coeffs_orig;
            for (size_t i = 0; i < n; i++) residual[i] = rnorm[i] * (rmax - rmin) + rmin;     /* :752 */
            cur = max_abs_error(data, S.decoded, residual, n);
            if (cur > target) { skip = 1; need_pure = 1; }                     /* :755-759 */
            else { best_err = cur; mean_err = mean_error(data, S.decoded, residual, n); }
        }
        if (!skip) {
            /* truncation bisection, :765-796 */
            double hi = (double) coeffs_size * 8, lo = 112.0, best = hi;
            const double eps = 1e-8;
            while (((target - best_err) / target > eps) && (hi - lo > 8 * 4)) {
                size_t tb = ((size_t) ceill((hi + lo) / 2 / 8)) * 8;
                orc_spiht_decode(coeffs, tb / 8, rnorm, H, W, tb);
                g_trace.n_spiht_decodes++;
                for (size_t i = 0; i < n; i++) residual[i] = rnorm[i] * (rmax - rmin) + rmin;
                cur = max_abs_error(data, S.decoded, residual, n);
                if (cur > target) lo = (double) tb;
                else {
                    hi = (double) tb;
                    if (cur >= best_err) {
                        best_err = cur; best = (double) tb;
                        mean_err = mean_error(data, S.decoded, residual, n);
                    }
                }
            }
            coeffs_size = (size_t) (best / 8.);
        }
after_search:
        if (coeffs_size <= 16) coeffs_size = 0;                                /* :811 */
        if (coeffs_size > 0) {
            if (!zstd_load()) exit(1);
            zlen = z_bound(coeffs_size);
            zbytes = (uint8_t *) malloc(zlen);
            zlen = z_comp(zbytes, zlen, coeffs, coeffs_size, 22);              /* :816 */
        }

        /* pure-J2K fallback, :819-854 */
        tail_len = S.cs_len;
        tail = (uint8_t *) malloc(tail_len ? tail_len : 1);
        memcpy(tail, S.cs, tail_len);
        if (!pure_done && !no_fallback && (mode == ORC_MAX_ERROR || mode == ORC_RELATIVE_ERROR)) {
            if (!no_consistency) {
                free(S.cs); S.cs = NULL;
                S.cs_len = j2k_enc(scaled, S.tiles, frame_h, W, cfg->base_cr, &S.cs);         /* :830 */
                j2k_dec(S.cs, S.cs_len, minv, maxv, &S.decoded, NULL, NULL);
                cr = cfg->base_cr;
            }
            rate_search(&S, cr, 1.0);                                          /* :836 */
            if (getenv("ORC_DUMP_DIR") && coeffs_size > 0) {                   /* study hook (tools/zstd_bound_study.py): the inputs of the :838 comparison */
                char path[512];
                snprintf(path, sizeof path, "%s/dump_%d_%zu.bin", getenv("ORC_DUMP_DIR"), (int) getpid(), (size_t) g_trace.n_j2k_encodes);
                FILE *fp = fopen(path, "wb");
                if (fp) {
                    uint64_t hdr[4] = { coeffs_size, zlen, tail_len, S.cs_len };
                    fwrite(hdr, 8, 4, fp); fwrite(coeffs, 1, coeffs_size, fp); fclose(fp);
                }
            }
            if (S.cs_len < zlen + tail_len || need_pure) {
                mean_err = mean_error(data, S.decoded, NULL, n);               /* :843 */
                zlen = 0; coeffs_size = 0;
                free(tail);
                tail_len = S.cs_len;
                tail = (uint8_t *) malloc(tail_len);
                memcpy(tail, S.cs, tail_len);
            }
        }
        g_trace.final_cr = cr;
        free(coeffs); free(residual); free(rnorm); free(S.decoded);
    }
    free(S.cs);
    free(scaled);

    if (!no_mean_adjust && fabs(mean_err) > 1e-18) {                           /* :864-868 */
        minv += mean_err;
        maxv += mean_err;
    }

    size_t codec_size = const_field ? sizeof(uint64_t) : tail_len;
    size_t total = sizeof(frame_hdr_t) + zlen + codec_size;
    uint8_t *o = (uint8_t *) malloc(total), *p = o;
    frame_hdr_t hd; memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, "EBCC", 4);
    hd.version = 1;
    if (const_field) hd.flags |= 1;
    hd.minval_bits = f2u(minv); hd.maxval_bits = f2u(maxv);
    hd.coeffs_size = coeffs_size;
    hd.rmin_bits = f2u(rmin); hd.rmax_bits = f2u(rmax);
    hd.compressed_size = zlen; hd.tail_size = codec_size;
    memcpy(p, &hd, sizeof hd); p += sizeof hd;
    if (zlen) { memcpy(p, zbytes, zlen); p += zlen; }
    if (const_field) { uint64_t cnt = n; memcpy(p, &cnt, 8); p += 8; }
    else { memcpy(p, tail, tail_len); p += tail_len; }
    free(zbytes); free(tail);
    g_trace.coeffs_size = coeffs_size; g_trace.compressed_size = zlen; g_trace.tail_size = codec_size;
    *out = o;
    return total;
}

/* ebcc_decode_legacy, :1147-1213 (header-less streams) */
static size_t decode_legacy(const uint8_t *d, size_t n, float **out)
{
    const uint8_t *p = d, *end = d + n;
    float minv, maxv, rmin, rmax; uint64_t coeffs_size, zlen;
    if (n < 32) return 0;
    memcpy(&minv, p, 4); p += 4; memcpy(&maxv, p, 4); p += 4;
    memcpy(&coeffs_size, p, 8); p += 8;
    memcpy(&rmin, p, 4); p += 4; memcpy(&rmax, p, 4); p += 4;
    memcpy(&zlen, p, 8); p += 8;
    if ((size_t) (end - p) < zlen) return 0;
    const uint8_t *z = p; p += zlen;
    size_t h = 0, w = 0, tot;
    int const_field = minv == maxv;
    if (const_field) {
        uint64_t cnt;
        if ((size_t) (end - p) < 8) return 0;
        memcpy(&cnt, p, 8);
        tot = cnt;
        *out = (float *) malloc(tot * sizeof(float));
        for (size_t i = 0; i < tot; i++) (*out)[i] = minv;
    } else {
        tot = j2k_dec(p, (size_t) (end - p), minv, maxv, out, &h, &w);
    }
    if (zlen > 0 && coeffs_size > 0) {
        if (const_field || !zstd_load()) return 0;
        uint8_t *c = (uint8_t *) calloc(coeffs_size, 1);
        float *r = (float *) calloc(tot, sizeof(float));
        z_decomp(c, coeffs_size, z, zlen);
        orc_spiht_decode(c, coeffs_size, r, h, w, coeffs_size * 8);
        for (size_t i = 0; i < tot; i++) (*out)[i] += r[i] * (rmax - rmin) + rmin;
        free(c); free(r);
    }
    return tot;
}

size_t orc_ebcc_decode(const uint8_t *d, size_t n, float **out)
{
    /* :1215-1320 */
    if (n < sizeof(frame_hdr_t) || memcmp(d, "EBCC", 4) != 0) return decode_legacy(d, n, out);
    frame_hdr_t hd;
    memcpy(&hd, d, sizeof hd);
    if (hd.version != 1) return 0;
    float minv = u2f(hd.minval_bits), maxv = u2f(hd.maxval_bits);
    float rmin = u2f(hd.rmin_bits), rmax = u2f(hd.rmax_bits);
    size_t used = sizeof hd;
    if (hd.compressed_size > n - used) return 0;
    used += hd.compressed_size;
    if (hd.tail_size > n - used) return 0;
    used += hd.tail_size;
    const uint8_t *z = d + sizeof hd, *p = z + hd.compressed_size;
    size_t h = 0, w = 0, tot = 0;
    int const_field = hd.flags & 1;
    if (const_field) {
        if (hd.tail_size != 8) return 0;
        uint64_t cnt; memcpy(&cnt, p, 8);
        tot = cnt; p += 8;
        *out = (float *) malloc(tot * sizeof(float));
        for (size_t i = 0; i < tot; i++) (*out)[i] = minv;
    } else {
        tot = j2k_dec(p, hd.tail_size, minv, maxv, out, &h, &w);
        p += hd.tail_size;
    }
    if (hd.compressed_size > 0 && hd.coeffs_size > 0) {
        if (const_field || !zstd_load()) return 0;
        uint8_t *c = (uint8_t *) calloc(hd.coeffs_size, 1);
        float *r = (float *) calloc(tot, sizeof(float));
        z_decomp(c, hd.coeffs_size, z, hd.compressed_size);
        orc_spiht_decode(c, hd.coeffs_size, r, h, w, hd.coeffs_size * 8);      /* :1304 */
        for (size_t i = 0; i < tot; i++) (*out)[i] += r[i] * (rmax - rmin) + rmin;   /* :1307 */
        free(c); free(r);
    }
    if ((size_t) (p - d) != used) return 0;                                    /* :1314-1317 */
    return tot;
}

/* ------------------------------------------------------------------ EBCK container, :920-1090,1322-1449 */
static size_t cdiv(size_t a, size_t b) { return a / b + (a % b != 0); }

size_t orc_ebcc_encode_chunking(const float *data, const orc_config_t *cfg, uint8_t **out)
{
    size_t cd[3]; int all_zero = 1;
    for (int i = 0; i < 3; i++) { cd[i] = cfg->chunk_dims[i]; if (cd[i]) all_zero = 0; }
    if (all_zero) for (int i = 0; i < 3; i++) cd[i] = cfg->dims[i];
    if (!dims_ok(cd)) return 0;
    size_t cnt[3];
    for (int i = 0; i < 3; i++) { if (!cfg->dims[i] || !cd[i]) return 0; cnt[i] = cdiv(cfg->dims[i], cd[i]); }
    size_t csize = cd[0] * cd[1] * cd[2], nchunks = cnt[0] * cnt[1] * cnt[2];
    int slabs = cd[1] == cfg->dims[1] && cd[2] == cfg->dims[2];

    size_t cap = 1024, len = 0;
    uint8_t *o = (uint8_t *) malloc(cap);
#define APPEND(src, nb) do { while (len + (nb) > cap) { cap *= 2; o = (uint8_t *) realloc(o, cap); } \
                             memcpy(o + len, (src), (nb)); len += (nb); } while (0)
    chunk_hdr_t hd; memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, "EBCK", 4); hd.version = 1; hd.ndims = 3;
    for (int i = 0; i < 3; i++) { hd.dims[i] = cfg->dims[i]; hd.chunk_dims[i] = cd[i]; }
    hd.num_chunks = nchunks; hd.chunk_size = csize;
    APPEND(&hd, sizeof hd);

    float *cb = (float *) malloc(csize * sizeof(float));
    orc_config_t cc = *cfg;
    for (int i = 0; i < 3; i++) { cc.dims[i] = cd[i]; cc.chunk_dims[i] = 0; }
    for (size_t cl = 0; cl < nchunks; cl++) {
        size_t org[3], t = cl;
        for (int d = 3; d-- > 0;) { org[d] = (t % cnt[d]) * cd[d]; t /= cnt[d]; }      /* :311-318 */
        int inb = 1;
        for (int d = 0; d < 3; d++) if (org[d] > cfg->dims[d] || cd[d] > cfg->dims[d] - org[d]) inb = 0;
        const float *src;
        if (slabs && inb) {
            src = data + (org[0] * cfg->dims[1] + org[1]) * cfg->dims[2] + org[2];
        } else {
            for (size_t li = 0; li < csize; li++) {                                    /* :339-351 */
                size_t rem = li, idx[3];
                for (int d = 3; d-- > 0;) {
                    size_t k = org[d] + rem % cd[d];
                    idx[d] = k < cfg->dims[d] - 1 ? k : cfg->dims[d] - 1;
                    rem /= cd[d];
                }
                cb[li] = data[(idx[0] * cfg->dims[1] + idx[1]) * cfg->dims[2] + idx[2]];
            }
            src = cb;
        }
        uint8_t *cs = NULL;
        size_t nb = orc_ebcc_encode(src, &cc, &cs);
        if (!nb) { free(cb); free(o); return 0; }
        uint64_t nb64 = nb;
        APPEND(&nb64, 8);
        APPEND(cs, nb);
        free(cs);
    }
#undef APPEND
    free(cb);
    *out = o;
    return len;
}

size_t orc_ebcc_encode_chunking_compat(const float *data, const orc_config_t *cfg, uint8_t **out)
{
    /* :1054-1090 */
    orc_config_t c = *cfg;
    if (!c.chunk_dims[0] && !c.chunk_dims[1] && !c.chunk_dims[2]) {
        c.chunk_dims[0] = 1;
        c.chunk_dims[1] = c.dims[1] > DIM_MAX ? 1024 : c.dims[1];
        c.chunk_dims[2] = c.dims[2] > DIM_MAX ? 1024 : c.dims[2];
    }
    if (c.residual_compression_type == ORC_RELATIVE_ERROR) {
        size_t tot = c.dims[0] * c.dims[1] * c.dims[2];
        if (!tot) return 0;
        c.error *= data_range(data, tot);
        c.residual_compression_type = ORC_MAX_ERROR;
    }
    return orc_ebcc_encode_chunking(data, &c, out);
}

size_t orc_ebcc_decode_chunking(const uint8_t *d, size_t n, float **out)
{
    if (n < sizeof(chunk_hdr_t) || memcmp(d, "EBCK", 4) != 0) return orc_ebcc_decode(d, n, out);
    chunk_hdr_t hd; memcpy(&hd, d, sizeof hd);
    if (hd.version != 1 || hd.ndims != 3) return 0;
    size_t dims[3], cd[3], cnt[3];
    for (int i = 0; i < 3; i++) { dims[i] = hd.dims[i]; cd[i] = hd.chunk_dims[i]; }
    if (!dims_ok(cd)) return 0;
    for (int i = 0; i < 3; i++) { if (!dims[i] || !cd[i]) return 0; cnt[i] = cdiv(dims[i], cd[i]); }
    size_t csize = cd[0] * cd[1] * cd[2], nchunks = cnt[0] * cnt[1] * cnt[2], tot = dims[0] * dims[1] * dims[2];
    if (hd.chunk_size != csize || hd.num_chunks != nchunks) return 0;
    int slabs = cd[1] == dims[1] && cd[2] == dims[2];
    float *o = (float *) malloc(tot * sizeof(float));
    const uint8_t *p = d + sizeof hd, *end = d + n;
    for (size_t cl = 0; cl < nchunks; cl++) {
        uint64_t nb;
        if ((size_t) (end - p) < 8) { free(o); return 0; }
        memcpy(&nb, p, 8); p += 8;
        if (nb > (size_t) (end - p)) { free(o); return 0; }
        float *cb = NULL;
        size_t got = orc_ebcc_decode(p, nb, &cb);
        if (got != csize || !cb) { free(cb); free(o); return 0; }
        size_t org[3], t = cl;
        for (int k = 3; k-- > 0;) { org[k] = (t % cnt[k]) * cd[k]; t /= cnt[k]; }
        int inb = 1;
        for (int k = 0; k < 3; k++) if (org[k] > dims[k] || cd[k] > dims[k] - org[k]) inb = 0;
        if (slabs && inb) {
            memcpy(o + (org[0] * dims[1] + org[1]) * dims[2] + org[2], cb, csize * sizeof(float));
        } else {
            for (size_t li = 0; li < csize; li++) {                                    /* :353-370 */
                size_t rem = li, idx[3]; int ok = 1;
                for (int k = 3; k-- > 0;) { idx[k] = org[k] + rem % cd[k]; if (idx[k] >= dims[k]) ok = 0; rem /= cd[k]; }
                if (ok) o[(idx[0] * dims[1] + idx[1]) * dims[2] + idx[2]] = cb[li];
            }
        }
        free(cb);
        p += nb;
    }
    if (p != end) { free(o); return 0; }
    *out = o;
    return tot;
}
