/*
 * spiht_oracle.c - CPU restatement of the EBCC residual coder.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h).  Sequential and deliberately simple; pinned bit-exactly against the reference
 * build in oracle/_ref/ and the fixtures in tests/golden/.
 *
 * Follows (paths under /root/reference):
 *   src/spiht/dwt.h      - padding, DC removal, CDF 9/7 lifting, truncation
 *   src/spiht/spiht_re.c - SPIHT list coder and "IMS" header
 *   src/spiht/ml.h       - list semantics (append-only + tombstones + stable compaction)
 *   src/spiht/bitio.h    - MSB-first bit packing; reads past the end return 0
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- lifting constants, dwt.h:3-7 */
static const float K_ALPHA = -1.586134342;
static const float K_BETA  = -0.05298011854;
static const float K_GAMMA = 0.8829110762;
static const float K_DELTA = 0.44355068522;
static const float K_XI    = 1.149604398;

void orc_grid_init(orc_grid_t *g, size_t height, size_t width, size_t stages)
{
    /* dwt.h:43-59: pad each axis up to a multiple of 2^(stages+1) */
    uint32_t unit = 1u << (stages + 1);
    g->size_x = (uint32_t) width;
    g->size_y = (uint32_t) height;
    g->extra_x = (unit - (uint32_t) width % unit) % unit;
    g->extra_y = (unit - (uint32_t) height % unit) % unit;
    g->stride = g->size_x + g->extra_x;
    g->stages = (uint32_t) stages;
}

/* One forward lifting pass over n samples read with stride ss from src, written de-interleaved
 * (low half then high half) with stride ds to dst.  dwt.h:87-112 (rows) / :142-167 (columns). */
static void lift_forward(const float *src, size_t ss, float *dst, size_t ds, size_t n)
{
    size_t half = n / 2;
    float *lo = dst, *hi = dst + half * ds;
    for (size_t k = 0; k + 1 < half; k++)
        hi[k * ds] = src[(2 * k + 1) * ss] + K_ALPHA * (src[(2 * k) * ss] + src[(2 * k + 2) * ss]);
    hi[(half - 1) * ds] = src[(n - 1) * ss] + 2 * K_ALPHA * src[(n - 2) * ss];          /* :94 */

    lo[0] = src[0] + K_BETA * (hi[0] + hi[ds]);                                           /* :96 */
    for (size_t k = 1; k < half; k++)
        lo[k * ds] = src[(2 * k) * ss] + K_BETA * (hi[k * ds] + hi[(k - 1) * ds]);

    for (size_t k = 0; k + 1 < half; k++)
        hi[k * ds] += K_GAMMA * (lo[k * ds] + lo[(k + 1) * ds]);
    hi[(half - 1) * ds] += K_GAMMA * (lo[(half - 1) * ds] + lo[(half - 2) * ds]);         /* :102 */

    lo[0] += K_DELTA * (hi[0] + hi[ds]);                                                  /* :104 */
    for (size_t k = 1; k < half; k++)
        lo[k * ds] += K_DELTA * (hi[k * ds] + hi[(k - 1) * ds]);

    for (size_t k = 0; k < half; k++) {                                                   /* :108-111 */
        lo[k * ds] *= K_XI;
        hi[k * ds] /= K_XI;
    }
}

/* Inverse pass: src holds low|high halves (modified in place as scratch, exactly like the
 * reference), dst receives the interleaved samples.  dwt.h:114-140 / :169-194. */
static void lift_inverse(float *src, size_t ss, float *dst, size_t ds, size_t n)
{
    size_t half = n / 2;
    float *lo = src, *hi = src + half * ss;
    for (size_t k = 0; k < half; k++) {
        lo[k * ss] /= K_XI;
        hi[k * ss] *= K_XI;
    }
    for (size_t k = 1; k < half; k++)
        lo[k * ss] -= K_DELTA * (hi[k * ss] + hi[(k - 1) * ss]);
    lo[0] -= K_DELTA * (hi[0] + hi[ss]);

    hi[(half - 1) * ss] -= K_GAMMA * (lo[(half - 1) * ss] + lo[(half - 2) * ss]);
    for (size_t k = 0; k + 1 < half; k++)
        hi[k * ss] -= K_GAMMA * (lo[k * ss] + lo[(k + 1) * ss]);

    for (size_t k = 1; k < half; k++)
        dst[(2 * k) * ds] = lo[k * ss] - K_BETA * (hi[k * ss] + hi[(k - 1) * ss]);
    dst[0] = lo[0] - K_BETA * (hi[0] + hi[ss]);

    dst[(n - 1) * ds] = hi[(half - 1) * ss] - 2 * K_ALPHA * dst[(n - 2) * ds];
    for (size_t k = 0; k + 1 < half; k++)
        dst[(2 * k + 1) * ds] = hi[k * ss] - K_ALPHA * (dst[(2 * k) * ds] + dst[(2 * k + 2) * ds]);
}

/* dwt2full, dwt.h:293-303: per level rows data->temp then columns temp->data */
static void dwt_forward_all(float *data, float *temp, const orc_grid_t *g)
{
    size_t nx = g->size_x + g->extra_x, ny = g->size_y + g->extra_y, st = g->stride;
    for (uint32_t lv = 0; lv < g->stages; lv++) {
        for (size_t y = 0; y < ny; y++) lift_forward(data + y * st, 1, temp + y * st, 1, nx);
        for (size_t x = 0; x < nx; x++) lift_forward(temp + x, st, data + x, st, ny);
        nx /= 2; ny /= 2;
    }
}

/* idwt2full, dwt.h:305-317: per level columns data->temp then rows temp->data, coarse to fine */
static void dwt_inverse_all(float *data, float *temp, const orc_grid_t *g)
{
    size_t nx = (g->size_x + g->extra_x) >> (g->stages - 1);
    size_t ny = (g->size_y + g->extra_y) >> (g->stages - 1);
    size_t st = g->stride;
    for (uint32_t lv = 0; lv < g->stages; lv++) {
        for (size_t x = 0; x < nx; x++) lift_inverse(data + x, st, temp + x, st, ny);
        for (size_t y = 0; y < ny; y++) lift_inverse(temp + y * st, 1, data + y * st, 1, nx);
        nx *= 2; ny *= 2;
    }
}

int orc_spiht_analysis(const float *image, size_t height, size_t width, size_t stages, float *coeffs)
{
    orc_grid_t g;
    orc_grid_init(&g, height, width, stages);
    size_t nx = g.size_x + g.extra_x, ny = g.size_y + g.extra_y, st = g.stride;
    float *temp = (float *) calloc(nx * ny, sizeof(float));
    memset(coeffs, 0, nx * ny * sizeof(float));

    /* load_image, dwt.h:63-76: scale by 255, mirror right edge, mirror bottom edge, zero corner */
    const float scale = 255;
    for (size_t y = 0; y < height; y++)
        for (size_t x = 0; x < width; x++)
            coeffs[x + y * st] = image[y * width + x] * scale;
    for (size_t y = 0; y < height; y++)
        for (size_t x = 0; x < g.extra_x; x++)
            coeffs[width + x + y * st] = coeffs[width - x - 1 + y * st];
    for (size_t x = 0; x < width; x++)
        for (size_t y = 0; y < g.extra_y; y++)
            coeffs[x + (height + y) * st] = coeffs[x + (height - y - 1) * st];

    /* sub_dc, dwt.h:319-334: sequential double sum, floor of the mean */
    double dc = 0;
    for (size_t y = 0; y < ny; y++)
        for (size_t x = 0; x < nx; x++)
            dc += coeffs[x + y * st];
    dc /= (double) (nx * ny);
    dc = floor(dc);
    for (size_t i = 0; i < nx * ny; i++)
        coeffs[i] = (float) ((double) coeffs[i] - dc);

    dwt_forward_all(coeffs, temp, &g);

    /* normalize, dwt.h:355-368: truncate toward zero */
    for (size_t i = 0; i < nx * ny; i++)
        coeffs[i] = (coeffs[i] >= 0) ? floorf(coeffs[i]) : -floorf(fabsf(coeffs[i]));

    free(temp);
    return (int) (uint8_t) (float) dc;
}

void orc_spiht_synthesis(float *coeffs, size_t height, size_t width, size_t stages, int dc, float *image_out)
{
    orc_grid_t g;
    orc_grid_init(&g, height, width, stages);
    size_t nx = g.size_x + g.extra_x, ny = g.size_y + g.extra_y, st = g.stride;
    float *temp = (float *) calloc(nx * ny, sizeof(float));
    dwt_inverse_all(coeffs, temp, &g);
    /* add_dc, dwt.h:336-353 then crop and /255, spiht_re.c:512-516 */
    float dcf = (float) dc;
    for (size_t y = 0; y < height; y++)
        for (size_t x = 0; x < width; x++) {
            float v = floorf(coeffs[x + y * st] + dcf);
            if (v > 255) v = 255; else if (v < 0) v = 0;
            image_out[x + y * width] = v / 255.0f;
        }
    free(temp);
}

/* ---------------------------------------------------------------- bit I/O, bitio.h */
typedef struct {
    uint8_t *buf;
    size_t cap;      /* bytes available          */
    size_t nbits;    /* bits written / read so far */
} bits_t;

static void put_bit(bits_t *b, unsigned bit)
{
    size_t byte = b->nbits >> 3;
    if (byte < b->cap && (bit & 1)) b->buf[byte] |= (uint8_t) (0x80u >> (b->nbits & 7));
    b->nbits++;
}
static void put_bits(bits_t *b, uint64_t v, unsigned n)
{
    for (unsigned i = n; i-- > 0;) put_bit(b, (unsigned) (v >> i) & 1);
}
static unsigned get_bit(bits_t *b)
{
    size_t byte = b->nbits >> 3;
    unsigned r = 0;
    /* bitio.h:60-63: whole bytes are fetched lazily, past the end reads as 0 (and does not advance
     * a byte counter, which is unobservable) */
    if (byte < b->cap) r = (b->buf[byte] >> (7 - (b->nbits & 7))) & 1;
    b->nbits++;
    return r;
}
static uint64_t get_bits(bits_t *b, unsigned n)
{
    uint64_t v = 0;
    for (unsigned i = 0; i < n; i++) v = (v << 1) | get_bit(b);
    return v;
}

/* ---------------------------------------------------------------- lists, ml.h */
typedef struct {
    int32_t *v;
    uint8_t *dead;
    size_t n, cap;
} list_t;

static void list_init(list_t *l, size_t cap)
{
    l->v = (int32_t *) malloc(cap * sizeof(int32_t));
    l->dead = (uint8_t *) calloc(cap, 1);
    l->n = 0; l->cap = cap;
}
static void list_free(list_t *l) { free(l->v); free(l->dead); }
static void list_push(list_t *l, int32_t x)
{
    if (l->n == l->cap) {
        size_t nc = l->cap * 2 + 16;
        l->v = (int32_t *) realloc(l->v, nc * sizeof(int32_t));
        l->dead = (uint8_t *) realloc(l->dead, nc);
        memset(l->dead + l->cap, 0, nc - l->cap);
        l->cap = nc;
    }
    l->v[l->n++] = x;
}
/* ml_consolidate, ml.h:52-66: stable compaction of live entries */
static void list_compact(list_t *l)
{
    size_t j = 0;
    for (size_t i = 0; i < l->n; i++) {
        if (!l->dead[i]) l->v[j++] = l->v[i];
        l->dead[i] = 0;
    }
    l->n = j;
}

/* ---------------------------------------------------------------- tree geometry, spiht_re.c:127-158 */
typedef struct {
    const orc_grid_t *g;
    int32_t nx, ny, lx, ly, st;
} tree_t;

static void tree_init(tree_t *t, const orc_grid_t *g)
{
    t->g = g;
    t->nx = (int32_t) (g->size_x + g->extra_x);
    t->ny = (int32_t) (g->size_y + g->extra_y);
    t->lx = t->nx >> g->stages;
    t->ly = t->ny >> g->stages;
    t->st = (int32_t) g->stride;
}

/* top-left child of (x,y), or -1 */
static int32_t first_child(const tree_t *t, int32_t x, int32_t y)
{
    int32_t cx, cy;
    if (x < t->lx && y < t->ly) {
        cx = (x & 1) ? x + t->lx - 1 : x;
        cy = (y & 1) ? y + t->ly - 1 : y;
        if (cx == x && cy == y) return -1;
    } else {
        cx = 2 * x; cy = 2 * y;
        if (cx >= t->nx || cy >= t->ny) return -1;
    }
    return cx + cy * t->st;
}

static int sig_px(int step, float v)
{
    /* spiht_re.c:124 */
    int64_t iv = (int64_t) v;
    if (iv < 0) iv = -iv;
    return iv >= ((int64_t) 1 << step);
}

/* any descendant at depth >= mindepth significant?  spiht_re.c:160-206 (A: mindepth 1, B: 2) */
static int sig_set(const tree_t *t, const float *c, int step, int32_t px, int depth, int mindepth)
{
    if (depth >= mindepth && sig_px(step, c[px])) return 1;
    int32_t ch = first_child(t, px % t->st, px / t->st);
    if (ch < 0) return 0;
    return sig_set(t, c, step, ch, depth + 1, mindepth) ||
           sig_set(t, c, step, ch + 1, depth + 1, mindepth) ||
           sig_set(t, c, step, ch + t->st, depth + 1, mindepth) ||
           sig_set(t, c, step, ch + t->st + 1, depth + 1, mindepth);
}

static void seed_lists(const tree_t *t, list_t *lip, list_t *lis)
{
    /* spiht_re.c:65-77 */
    for (int32_t y = 0; y < t->ly; y++)
        for (int32_t x = 0; x < t->lx; x++) {
            int32_t px = x + y * t->st;
            list_push(lip, px);
            if ((x & 1) || (y & 1)) list_push(lis, px + 1);      /* +(p+1) = type A, -(p+1) = type B */
        }
}

/* spiht_encode_process, spiht_re.c:208-317.  Returns when bit number budget+1 has been written. */
static void encode_passes(bits_t *bio, const tree_t *t, const float *c, int top_step, size_t budget)
{
    list_t lip, lsp, lis;
    size_t npx = (size_t) t->nx * t->ny;
    list_init(&lip, npx); list_init(&lsp, npx); list_init(&lis, npx);
    seed_lists(t, &lip, &lis);
    size_t cnt = 0;
#define EMIT(b) do { put_bit(bio, (b)); if (++cnt > budget) goto done; } while (0)
    for (int step = top_step; step >= 0; step--) {
        for (size_t i = 0; i < lip.n; i++) {
            int32_t px = lip.v[i];
            int s = sig_px(step, c[px]);
            EMIT(s);
            if (s) {
                list_push(&lsp, px);
                EMIT(c[px] > 0 ? 0 : 1);
                lip.dead[i] = 1;
            }
        }
        list_compact(&lip);

        for (size_t i = 0; i < lis.n; i++) {            /* lis.n grows inside the loop */
            int32_t e = lis.v[i];
            if (e > 0) {
                int32_t px = e - 1;
                int s = sig_set(t, c, step, px, 1, 2);
                EMIT(s);
                if (s) {
                    int32_t ch = first_child(t, px % t->st, px / t->st);
                    for (int dy = 0; dy < 2; dy++)
                        for (int dx = 0; dx < 2; dx++) {
                            int32_t q = ch + dx + dy * t->st;
                            int sq = sig_px(step, c[q]);
                            EMIT(sq);
                            if (sq) {
                                list_push(&lsp, q);
                                EMIT(c[q] > 0 ? 0 : 1);
                            } else {
                                list_push(&lip, q);
                            }
                        }
                    if (first_child(t, ch % t->st, ch / t->st) >= 0) list_push(&lis, -(px + 1));
                    lis.dead[i] = 1;
                }
            } else {
                int32_t px = -e - 1;
                int s = sig_set(t, c, step, px, 1, 3);
                EMIT(s);
                if (s) {
                    int32_t ch = first_child(t, px % t->st, px / t->st);
                    list_push(&lis, ch + 1);
                    list_push(&lis, ch + 1 + 1);
                    list_push(&lis, ch + t->st + 1);
                    list_push(&lis, ch + t->st + 1 + 1);
                    lis.dead[i] = 1;
                }
            }
        }
        list_compact(&lis);

        for (size_t i = 0; i < lsp.n; i++) {
            float v = c[lsp.v[i]];
            if (sig_px(step + 1, v)) {
                int64_t iv = (int64_t) v;
                if (iv < 0) iv = -iv;
                EMIT((unsigned) ((iv >> step) & 1));
            }
        }
    }
#undef EMIT
done:
    list_free(&lip); list_free(&lsp); list_free(&lis);
}

/* spiht_decode_process, spiht_re.c:319-430 */
static void decode_passes(bits_t *bio, const tree_t *t, float *c, int top_step, size_t budget)
{
    list_t lip, lsp, lis;
    size_t npx = (size_t) t->nx * t->ny;
    list_init(&lip, npx); list_init(&lsp, npx); list_init(&lis, npx);
    seed_lists(t, &lip, &lis);
    size_t cnt = 0;
#define TICK() do { if (++cnt > budget) goto done; } while (0)
    for (int step = top_step; step >= 0; step--) {
        for (size_t i = 0; i < lip.n; i++) {
            int32_t px = lip.v[i];
            unsigned s = get_bit(bio);
            TICK();
            if (s) {
                list_push(&lsp, px);
                c[px] = (float) ((get_bit(bio) ? -1 : 1) * (1 << step));
                TICK();
                lip.dead[i] = 1;
            }
        }
        list_compact(&lip);

        for (size_t i = 0; i < lis.n; i++) {
            int32_t e = lis.v[i];
            if (e > 0) {
                int32_t px = e - 1;
                unsigned s = get_bit(bio);
                TICK();
                if (s) {
                    int32_t ch = first_child(t, px % t->st, px / t->st);
                    for (int dy = 0; dy < 2; dy++)
                        for (int dx = 0; dx < 2; dx++) {
                            int32_t q = ch + dx + dy * t->st;
                            unsigned sq = get_bit(bio);
                            TICK();
                            if (sq) {
                                list_push(&lsp, q);
                                c[q] = (float) ((get_bit(bio) ? -1 : 1) * (1 << step));
                                TICK();
                            } else {
                                list_push(&lip, q);
                            }
                        }
                    if (first_child(t, ch % t->st, ch / t->st) >= 0) list_push(&lis, -(px + 1));
                    lis.dead[i] = 1;
                }
            } else {
                int32_t px = -e - 1;
                unsigned s = get_bit(bio);
                TICK();
                if (s) {
                    int32_t ch = first_child(t, px % t->st, px / t->st);
                    list_push(&lis, ch + 1);
                    list_push(&lis, ch + 1 + 1);
                    list_push(&lis, ch + t->st + 1);
                    list_push(&lis, ch + t->st + 1 + 1);
                    lis.dead[i] = 1;
                }
            }
        }
        list_compact(&lis);

        for (size_t i = 0; i < lsp.n; i++) {
            int32_t px = lsp.v[i];
            float v = c[px];
            if (sig_px(step + 1, v)) {
                int64_t iv = (int64_t) v;
                if (get_bit(bio)) {
                    if (iv >= 0) c[px] = (float) (iv | ((int64_t) 1 << step));
                    else c[px] = (float) (-((-iv) | ((int64_t) 1 << step)));
                } else {
                    c[px] = (float) (iv & ~((int64_t) 1 << step));
                }
                TICK();
            }
        }
    }
#undef TICK
done:
    list_free(&lip); list_free(&lsp); list_free(&lis);
}

void orc_spiht_encode(const float *image, size_t height, size_t width, uint8_t **out, size_t *out_size,
                      size_t trunc_bits, size_t stages)
{
    /* spiht_re.c:432-475 */
    size_t cap = (trunc_bits == 0) ? height * width * sizeof(float) : trunc_bits / sizeof(uint8_t) + 1;  /* sic: bytes = bits + 1 */
    if (cap < 16) cap = 16;
    orc_grid_t g;
    orc_grid_init(&g, height, width, stages);
    size_t npx = (size_t) (g.size_x + g.extra_x) * (g.size_y + g.extra_y);
    float *c = (float *) malloc(npx * sizeof(float));
    int dc = orc_spiht_analysis(image, height, width, stages, c);

    bits_t bio = { (uint8_t *) calloc(cap, 1), cap, 0 };
    put_bits(&bio, 'I', 8); put_bits(&bio, 'M', 8); put_bits(&bio, 'S', 8);
    put_bits(&bio, stages, 6);
    put_bits(&bio, g.size_x, 12); put_bits(&bio, g.size_y, 12);
    put_bits(&bio, g.extra_x, 10); put_bits(&bio, g.extra_y, 10);
    put_bit(&bio, 0);
    size_t bits0 = (trunc_bits == 0) ? ((size_t) 1 << 28) : trunc_bits + 128;
    put_bits(&bio, bits0, 29);
    put_bits(&bio, (uint64_t) dc, 8);

    /* spiht_encode_init, spiht_re.c:54-63 */
    float mx = 2.0f;
    for (size_t i = 0; i < npx; i++) {
        float a = fabsf(c[i]);
        if (a > mx) mx = a;
    }
    int top = (int) floor(log(mx) / log(2.0));
    put_bits(&bio, (uint64_t) top, 8);

    tree_t t;
    tree_init(&t, &g);
    encode_passes(&bio, &t, c, top, bits0 - 128);

    *out = bio.buf;
    *out_size = (bio.nbits + 7) / 8;            /* bitio_flush, bitio.h:78-88 */
    free(c);
}

int orc_spiht_decode_coeffs(const uint8_t *in, size_t in_size, size_t num_bits, orc_grid_t *g, float **coeffs)
{
    /* spiht_re.c:477-507 */
    bits_t bio = { (uint8_t *) in, in_size, 0 };
    (void) get_bits(&bio, 24);                               /* 'I' 'M' 'S' */
    size_t stages = get_bits(&bio, 6);
    g->size_x = (uint32_t) get_bits(&bio, 12);
    g->size_y = (uint32_t) get_bits(&bio, 12);
    g->extra_x = (uint32_t) get_bits(&bio, 10);
    g->extra_y = (uint32_t) get_bits(&bio, 10);
    g->stride = g->size_x + g->extra_x;
    g->stages = (uint32_t) stages;
    (void) get_bit(&bio);
    size_t bits0 = get_bits(&bio, 29);
    if (num_bits > bits0) num_bits = bits0;
    num_bits -= 128;
    int dc = (int) get_bits(&bio, 8);
    int top = (int) get_bits(&bio, 8);

    size_t npx = (size_t) (g->size_x + g->extra_x) * (g->size_y + g->extra_y);
    float *c = (float *) calloc(npx, sizeof(float));
    tree_t t;
    tree_init(&t, g);
    decode_passes(&bio, &t, c, top, num_bits);
    *coeffs = c;
    return dc;
}

void orc_spiht_decode(const uint8_t *in, size_t in_size, float *image_out, size_t height, size_t width,
                      size_t num_bits)
{
    orc_grid_t g;
    float *c = NULL;
    int dc = orc_spiht_decode_coeffs(in, in_size, num_bits, &g, &c);
    /* the reference crops with the caller's height/width but the stream's stride (spiht_re.c:511-516);
     * both agree for every stream the codec produces */
    orc_spiht_synthesis(c, g.size_y, g.size_x, g.stages, dc, image_out);
    (void) height; (void) width;
    free(c);
}
