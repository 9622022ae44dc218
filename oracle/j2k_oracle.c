/*
 * j2k_oracle.c - CPU restatement of the JPEG 2000 base layer the reference obtains from OpenJPEG.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference calls OpenJPEG (an un-vendored submodule, source absent from /root/reference) at
 * /root/reference/src/ebcc_codec.c:105-180 (encode: 1 component, 16-bit unsigned, irreversible 9/7,
 * one quality layer with tcp_rates[0] = base_cr/2, every other parameter left at the library default:
 * 6 resolutions, 64x64 code-blocks, LRCP, one tile, no precinct partition, cblksty 0, 2 guard bits)
 * and :1092-1136 (decode).  Parity target = OpenJPEG 2.4.0 (the image's /opt/conda/lib/libopenjp2.so.7);
 * the algorithms restated here are ITU-T T.800 (JPEG 2000 part 1): Annex B (codestream/packets),
 * C (MQ coder), D (coefficient bit modelling), E (quantisation), F (9/7 wavelet), J.14 (PCRD), written
 * to reproduce OpenJPEG 2.4.0's arithmetic order.  Pinned against golden vectors generated with that
 * library through opj_backend.c: tests/golden/j2k_openjpeg.json + j2k_inputs.npz (oracle/make_golden_j2k.py; 37
 * codestreams of 8 images at rates 1 ... 900 with their decoded samples), checked on any box by
 * tests/test_oracle_golden.py::test_j2k_restatement_against_openjpeg_fixtures, and live against the library where it
 * exists (test_j2k_restatement_matches_openjpeg_live); the whole-frame fixtures pin it once more through the frame codec.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ================================================================================================
 * geometry
 * ============================================================================================== */
#define J2K_NRES 6
#define J2K_CBLK 64
#define J2K_MAXPASSES 100

typedef struct {
    int x0, y0, x1, y1;           /* band-domain coordinates */
    int numbps;                   /* non-zero bit planes (decoder: Mb - zero bit planes) */
    int npasses;                  /* passes included / produced */
    uint8_t *data;                /* compressed bytes */
    int len;
    /* encoder */
    int totalpasses;
    int rate[J2K_MAXPASSES];
    double disto[J2K_MAXPASSES];
    int numlenbits;
} cblk_t;

typedef struct {
    int x0, y0, x1, y1;
    int orient;                   /* 0 LL, 1 HL, 2 LH, 3 HH */
    int level;                    /* numresolutions - 1 - resno */
    int ncw, nch;                 /* code-block grid */
    cblk_t *cblks;
    int expn, mant, numbps;
    float stepsize;
    int offx, offy;               /* position of the band inside the tile buffer */
} band_t;

typedef struct {
    int x0, y0, x1, y1;
    int nbands;
    band_t bands[3];
} res_t;

typedef struct {
    int W, H;
    int ty0;                      /* row of the tile's first line in the image (tiles are stacked along y) */
    res_t res[J2K_NRES];
} tile_t;

static int ceildivpow2(int a, int b) { return (int) (((int64_t) a + ((int64_t) 1 << b) - 1) >> b); }
static int floorlog2(int a) { int l = 0; while (a > 1) { a >>= 1; l++; } return l; }

/* resolutions / bands / code-blocks of the tile whose rows are [ty0, ty0 + H) of the image (T.800 B.5-B.7): a tile
 * away from the origin has its own sub-band extents, low/high parity and code-block partition (the partition is
 * anchored at multiples of 64 of the band coordinates) */
static void tile_init_at(tile_t *t, int W, int H, int ty0)
{
    memset(t, 0, sizeof *t);
    t->W = W; t->H = H; t->ty0 = ty0;
    const int ty1 = ty0 + H;
    for (int r = 0; r < J2K_NRES; r++) {
        int lv = J2K_NRES - 1 - r;
        res_t *rs = &t->res[r];
        rs->x0 = 0; rs->y0 = ceildivpow2(ty0, lv);
        rs->x1 = ceildivpow2(W, lv); rs->y1 = ceildivpow2(ty1, lv);
        rs->nbands = r == 0 ? 1 : 3;
        for (int b = 0; b < rs->nbands; b++) {
            band_t *bd = &rs->bands[b];
            bd->level = lv;
            if (r == 0) {
                bd->orient = 0;
                bd->x0 = 0; bd->y0 = rs->y0; bd->x1 = rs->x1; bd->y1 = rs->y1;
                bd->offx = 0; bd->offy = 0;
            } else {
                bd->orient = b + 1;
                int xb = bd->orient & 1, yb = bd->orient >> 1;
                /* T.800 B-15: band bounds from the tile-component bounds */
                bd->x0 = ceildivpow2(0 - (xb << lv), lv + 1) < 0 ? 0 : ceildivpow2(0 - (xb << lv), lv + 1);
                bd->y0 = ceildivpow2(ty0 - (yb << lv), lv + 1) < 0 ? 0 : ceildivpow2(ty0 - (yb << lv), lv + 1);
                bd->x1 = ceildivpow2(W - (xb << lv), lv + 1);
                bd->y1 = ceildivpow2(ty1 - (yb << lv), lv + 1);
                bd->offx = xb ? t->res[r - 1].x1 : 0;
                bd->offy = yb ? t->res[r - 1].y1 - t->res[r - 1].y0 : 0;
            }
            int bw = bd->x1 - bd->x0, bh = bd->y1 - bd->y0;
            if (bw <= 0 || bh <= 0) { bd->ncw = bd->nch = 0; continue; }
            bd->ncw = (bd->x1 + J2K_CBLK - 1) / J2K_CBLK - bd->x0 / J2K_CBLK;
            bd->nch = (bd->y1 + J2K_CBLK - 1) / J2K_CBLK - bd->y0 / J2K_CBLK;
            bd->cblks = (cblk_t *) calloc((size_t) bd->ncw * bd->nch, sizeof(cblk_t));
            for (int cy = 0; cy < bd->nch; cy++)
                for (int cx = 0; cx < bd->ncw; cx++) {
                    cblk_t *c = &bd->cblks[cy * bd->ncw + cx];
                    int gx = (bd->x0 / J2K_CBLK + cx) * J2K_CBLK, gy = (bd->y0 / J2K_CBLK + cy) * J2K_CBLK;
                    c->x0 = gx > bd->x0 ? gx : bd->x0;
                    c->y0 = gy > bd->y0 ? gy : bd->y0;
                    c->x1 = gx + J2K_CBLK < bd->x1 ? gx + J2K_CBLK : bd->x1;
                    c->y1 = gy + J2K_CBLK < bd->y1 ? gy + J2K_CBLK : bd->y1;
                }
        }
    }
}


static void tile_free(tile_t *t)
{
    for (int r = 0; r < J2K_NRES; r++)
        for (int b = 0; b < t->res[r].nbands; b++) {
            band_t *bd = &t->res[r].bands[b];
            for (int i = 0; i < bd->ncw * bd->nch; i++) free(bd->cblks[i].data);
            free(bd->cblks);
        }
}

/* band step size from the QCD (expn, mant): T.800 E-3; Mb = expn + guard - 1 (E-2) */
static void band_set_quant(band_t *bd, int expn, int mant, int prec, int guard)
{
    bd->expn = expn; bd->mant = mant;
    bd->numbps = expn + guard - 1;
    bd->stepsize = (float) ((1.0 + mant / 2048.0) * pow(2.0, (double) (prec - expn)));
}

/* ================================================================================================
 * MQ coder (T.800 Annex C), state table C-2
 * ============================================================================================== */
typedef struct { uint16_t qe; uint8_t nmps, nlps, sw; } mqstate_t;
static const mqstate_t MQ[47] = {
    {0x5601, 1, 1, 1},   {0x3401, 2, 6, 0},   {0x1801, 3, 9, 0},   {0x0AC1, 4, 12, 0},  {0x0521, 5, 29, 0},
    {0x0221, 38, 33, 0}, {0x5601, 7, 6, 1},   {0x5401, 8, 14, 0},  {0x4801, 9, 14, 0},  {0x3801, 10, 14, 0},
    {0x3001, 11, 17, 0}, {0x2401, 12, 18, 0}, {0x1C01, 13, 20, 0}, {0x1601, 29, 21, 0}, {0x5601, 15, 14, 1},
    {0x5401, 16, 14, 0}, {0x5101, 17, 15, 0}, {0x4801, 18, 16, 0}, {0x3801, 19, 17, 0}, {0x3401, 20, 18, 0},
    {0x3001, 21, 19, 0}, {0x2801, 22, 19, 0}, {0x2401, 23, 20, 0}, {0x2201, 24, 21, 0}, {0x1C01, 25, 22, 0},
    {0x1801, 26, 23, 0}, {0x1601, 27, 24, 0}, {0x1401, 28, 25, 0}, {0x1201, 29, 26, 0}, {0x1101, 30, 27, 0},
    {0x0AC1, 31, 28, 0}, {0x09C1, 32, 29, 0}, {0x08A1, 33, 30, 0}, {0x0521, 34, 31, 0}, {0x0441, 35, 32, 0},
    {0x02A1, 36, 33, 0}, {0x0221, 37, 34, 0}, {0x0141, 38, 35, 0}, {0x0111, 39, 36, 0}, {0x0085, 40, 37, 0},
    {0x0049, 41, 38, 0}, {0x0025, 42, 39, 0}, {0x0015, 43, 40, 0}, {0x0009, 44, 41, 0}, {0x0005, 45, 42, 0},
    {0x0001, 45, 43, 0}, {0x5601, 46, 46, 0}};

enum { CTX_ZC0 = 0, CTX_SC0 = 9, CTX_MAG0 = 14, CTX_AGG = 17, CTX_UNI = 18, NCTX = 19 };

typedef struct {
    uint32_t a, c;
    int ct;
    const uint8_t *bp, *end;       /* decoder */
    uint8_t *wp, *start;           /* encoder */
    uint8_t st[NCTX], mps[NCTX];
} mq_t;

static void mq_reset(mq_t *m)
{
    memset(m->st, 0, sizeof m->st);
    memset(m->mps, 0, sizeof m->mps);
    m->st[CTX_UNI] = 46; m->st[CTX_AGG] = 3; m->st[CTX_ZC0] = 4;       /* T.800 table D-7 */
}

/* ---- decoder (C.3) with the 0xFF 0xFF artificial-marker convention past the end */
static void mqd_bytein(mq_t *m)
{
    uint32_t cur = m->bp < m->end ? *m->bp : 0xFF;
    uint32_t nxt = m->bp + 1 < m->end ? m->bp[1] : 0xFF;
    if (cur == 0xFF) {
        if (nxt > 0x8F) { m->c += 0xFF00; m->ct = 8; }
        else { m->bp++; m->c += nxt << 9; m->ct = 7; }
    } else {
        m->bp++;
        m->c += nxt << 8;
        m->ct = 8;
    }
}
static void mqd_init(mq_t *m, const uint8_t *data, int len)
{
    mq_reset(m);
    m->bp = data; m->end = data + len;
    m->c = (len > 0 ? (uint32_t) data[0] : 0xFFu) << 16;
    mqd_bytein(m);
    m->c <<= 7;
    m->ct -= 7;
    m->a = 0x8000;
}
static int mqd_decode(mq_t *m, int cx)
{
    const mqstate_t *s = &MQ[m->st[cx]];
    int d;
    m->a -= s->qe;
    if ((m->c >> 16) < s->qe) {
        if (m->a < s->qe) { d = m->mps[cx]; m->st[cx] = s->nmps; }
        else { d = 1 - m->mps[cx]; if (s->sw) m->mps[cx] ^= 1; m->st[cx] = s->nlps; }
        m->a = s->qe;
        do { if (m->ct == 0) mqd_bytein(m); m->a <<= 1; m->c <<= 1; m->ct--; } while ((m->a & 0x8000) == 0);
    } else {
        m->c -= (uint32_t) s->qe << 16;
        if ((m->a & 0x8000) == 0) {
            if (m->a < s->qe) { d = 1 - m->mps[cx]; if (s->sw) m->mps[cx] ^= 1; m->st[cx] = s->nlps; }
            else { d = m->mps[cx]; m->st[cx] = s->nmps; }
            do { if (m->ct == 0) mqd_bytein(m); m->a <<= 1; m->c <<= 1; m->ct--; } while ((m->a & 0x8000) == 0);
        } else {
            d = m->mps[cx];
        }
    }
    return d;
}

/* ---- encoder (C.2) */
static void mqe_init(mq_t *m, uint8_t *buf)
{
    mq_reset(m);
    m->a = 0x8000; m->c = 0; m->ct = 12;
    m->start = buf;            /* buf[-1] must be writable and != 0xFF: callers pass buf = storage + 1 */
    m->wp = buf - 1;
}
static void mqe_byteout(mq_t *m)
{
    if (*m->wp == 0xFF) {
        m->wp++; *m->wp = (uint8_t) (m->c >> 20); m->c &= 0xFFFFF; m->ct = 7;
    } else if ((m->c & 0x8000000) == 0) {
        m->wp++; *m->wp = (uint8_t) (m->c >> 19); m->c &= 0x7FFFF; m->ct = 8;
    } else {
        (*m->wp)++;
        if (*m->wp == 0xFF) {
            m->c &= 0x7FFFFFF;
            m->wp++; *m->wp = (uint8_t) (m->c >> 20); m->c &= 0xFFFFF; m->ct = 7;
        } else {
            m->wp++; *m->wp = (uint8_t) (m->c >> 19); m->c &= 0x7FFFF; m->ct = 8;
        }
    }
}
static void mqe_renorm(mq_t *m)
{
    do { m->a <<= 1; m->c <<= 1; m->ct--; if (m->ct == 0) mqe_byteout(m); } while ((m->a & 0x8000) == 0);
}
static void mqe_encode(mq_t *m, int cx, int d)
{
    const mqstate_t *s = &MQ[m->st[cx]];
    if (d == m->mps[cx]) {                                    /* CODEMPS */
        m->a -= s->qe;
        if ((m->a & 0x8000) == 0) {
            if (m->a < s->qe) m->a = s->qe; else m->c += s->qe;
            m->st[cx] = s->nmps;
            mqe_renorm(m);
        } else {
            m->c += s->qe;
        }
    } else {                                                  /* CODELPS */
        m->a -= s->qe;
        if (m->a < s->qe) m->c += s->qe; else m->a = s->qe;
        if (s->sw) m->mps[cx] ^= 1;
        m->st[cx] = s->nlps;
        mqe_renorm(m);
    }
}
static int mqe_numbytes(const mq_t *m) { return (int) (m->wp - m->start); }   /* bytes completed, excluding the one being formed */
static void mqe_flush(mq_t *m)
{
    /* SETBITS + two BYTEOUTs, C.2.9 */
    uint32_t tempc = m->c + m->a;
    m->c |= 0xFFFF;
    if (m->c >= tempc) m->c -= 0x8000;
    m->c <<= m->ct; mqe_byteout(m);
    m->c <<= m->ct; mqe_byteout(m);
    if (*m->wp != 0xFF) m->wp++;                              /* the byte after the last one is not part of the segment unless FF */
}

/* ================================================================================================
 * T1 context modelling (T.800 Annex D)
 * ============================================================================================== */
#define F_SIG 1
#define F_VISIT 2
#define F_REFINED 4
#define F_NEG 8

typedef struct {
    int w, h, fs;                 /* fs = flag stride = w + 2 */
    uint8_t *flags;               /* (w+2) x (h+2) */
    int32_t *data;                /* w x h */
} t1_t;

static void t1_alloc(t1_t *t, int w, int h)
{
    t->w = w; t->h = h; t->fs = w + 2;
    t->flags = (uint8_t *) calloc((size_t) (w + 2) * (h + 2), 1);
    t->data = (int32_t *) calloc((size_t) w * h, sizeof(int32_t));
}
static void t1_free(t1_t *t) { free(t->flags); free(t->data); }
#define FL(t, x, y) ((t)->flags[((y) + 1) * (t)->fs + (x) + 1])

/* zero-coding context, table D-1 */
static int ctx_zc(const t1_t *t, int x, int y, int orient)
{
    int h = (FL(t, x - 1, y) & F_SIG) + (FL(t, x + 1, y) & F_SIG);
    int v = (FL(t, x, y - 1) & F_SIG) + (FL(t, x, y + 1) & F_SIG);
    int d = (FL(t, x - 1, y - 1) & F_SIG) + (FL(t, x + 1, y - 1) & F_SIG) + (FL(t, x - 1, y + 1) & F_SIG) +
            (FL(t, x + 1, y + 1) & F_SIG);
    int n;
    if (orient == 1) { int s = h; h = v; v = s; }             /* HL: swap roles */
    if (orient == 3) {                                        /* HH */
        int hv = h + v;
        if (d == 0) n = hv == 0 ? 0 : (hv == 1 ? 1 : 2);
        else if (d == 1) n = hv == 0 ? 3 : (hv == 1 ? 4 : 5);
        else if (d == 2) n = hv == 0 ? 6 : 7;
        else n = 8;
    } else {                                                  /* LL, LH (and HL after the swap) */
        if (h == 0) {
            if (v == 0) n = d == 0 ? 0 : (d == 1 ? 1 : 2);
            else if (v == 1) n = 3;
            else n = 4;
        } else if (h == 1) {
            if (v == 0) n = d == 0 ? 5 : 6;
            else n = 7;
        } else n = 8;
    }
    return CTX_ZC0 + n;
}

/* sign-coding context and XOR bit, tables D-2 / D-3 */
static int ctx_sc(const t1_t *t, int x, int y, int *xorbit)
{
    int hc = 0, vc = 0, f;
    f = FL(t, x - 1, y); if (f & F_SIG) hc += (f & F_NEG) ? -1 : 1;
    f = FL(t, x + 1, y); if (f & F_SIG) hc += (f & F_NEG) ? -1 : 1;
    f = FL(t, x, y - 1); if (f & F_SIG) vc += (f & F_NEG) ? -1 : 1;
    f = FL(t, x, y + 1); if (f & F_SIG) vc += (f & F_NEG) ? -1 : 1;
    hc = hc > 1 ? 1 : (hc < -1 ? -1 : hc);
    vc = vc > 1 ? 1 : (vc < -1 ? -1 : vc);
    int n, xb = 0;
    if (hc == 1) { n = vc == 1 ? 4 : (vc == 0 ? 3 : 2); }
    else if (hc == 0) { if (vc == 1) n = 1; else if (vc == 0) n = 0; else { n = 1; xb = 1; } }
    else { xb = 1; n = vc == 1 ? 2 : (vc == 0 ? 3 : 4); }
    *xorbit = xb;
    return CTX_SC0 + n;
}

/* magnitude-refinement context, table D-4 */
static int ctx_mag(const t1_t *t, int x, int y)
{
    if (FL(t, x, y) & F_REFINED) return CTX_MAG0 + 2;
    int any = (FL(t, x - 1, y) | FL(t, x + 1, y) | FL(t, x, y - 1) | FL(t, x, y + 1) | FL(t, x - 1, y - 1) |
               FL(t, x + 1, y - 1) | FL(t, x - 1, y + 1) | FL(t, x + 1, y + 1)) & F_SIG;
    return CTX_MAG0 + (any ? 1 : 0);
}

static int has_sig_neighbour(const t1_t *t, int x, int y)
{
    return ((FL(t, x - 1, y) | FL(t, x + 1, y) | FL(t, x, y - 1) | FL(t, x, y + 1) | FL(t, x - 1, y - 1) |
             FL(t, x + 1, y - 1) | FL(t, x - 1, y + 1) | FL(t, x + 1, y + 1)) & F_SIG) != 0;
}

/* ---------------------------------------------------------------- T1 decoder (D.3), OpenJPEG value convention:
 * data carries one fractional bit; a coefficient found significant in plane p becomes 1.5*2^p and each
 * refinement moves it by +-2^(p-1) (mid-point reconstruction). */
/* study hook (tools/t1_decode_stats.cpp): every code-block segment the decoder is handed, before it is decoded */
static void (*g_dec_sink)(const uint8_t *data, int len, int numbps, int npasses, int w, int h, int orient, void *user) = NULL;
static void *g_dec_sink_user = NULL;
void orc_j2k_set_decode_sink(void (*fn)(const uint8_t *, int, int, int, int, int, int, void *), void *user) { g_dec_sink = fn; g_dec_sink_user = user; }

static void t1_decode_cblk(t1_t *t, const cblk_t *cb, int orient)
{
    if (g_dec_sink) g_dec_sink(cb->data, cb->len, cb->numbps, cb->npasses, t->w, t->h, orient, g_dec_sink_user);
    mq_t mq;
    mqd_init(&mq, cb->data, cb->len);
    int bp = cb->numbps - 1, passtype = 2;
    for (int pass = 0; pass < cb->npasses && bp >= 0; pass++) {
        int one = 1 << (bp + 1), half = one >> 1, oneplushalf = one | half;
        for (int y0 = 0; y0 < t->h; y0 += 4) {
            if (passtype == 2) {
                for (int x = 0; x < t->w; x++) {
                    int y = y0, runlen = 0, agg = 0;
                    if (y0 + 3 < t->h) {
                        agg = 1;
                        for (int k = 0; k < 4; k++)
                            if ((FL(t, x, y0 + k) & (F_SIG | F_VISIT)) || has_sig_neighbour(t, x, y0 + k)) { agg = 0; break; }
                    }
                    int partial = 0;
                    if (agg) {
                        if (!mqd_decode(&mq, CTX_AGG)) continue;
                        runlen = mqd_decode(&mq, CTX_UNI);
                        runlen = (runlen << 1) | mqd_decode(&mq, CTX_UNI);
                        y = y0 + runlen;
                        partial = 1;
                    }
                    for (; y < y0 + 4 && y < t->h; y++) {
                        if (!partial && (FL(t, x, y) & (F_SIG | F_VISIT))) continue;
                        int sig = 1;
                        if (!partial) sig = mqd_decode(&mq, ctx_zc(t, x, y, orient));
                        partial = 0;
                        if (sig) {
                            int xb, cx = ctx_sc(t, x, y, &xb);
                            int neg = mqd_decode(&mq, cx) ^ xb;
                            t->data[y * t->w + x] = neg ? -oneplushalf : oneplushalf;
                            FL(t, x, y) |= F_SIG | (neg ? F_NEG : 0);
                        }
                    }
                }
            } else if (passtype == 0) {
                for (int x = 0; x < t->w; x++)
                    for (int y = y0; y < y0 + 4 && y < t->h; y++) {
                        if ((FL(t, x, y) & F_SIG) || !has_sig_neighbour(t, x, y)) continue;
                        if (mqd_decode(&mq, ctx_zc(t, x, y, orient))) {
                            int xb, cx = ctx_sc(t, x, y, &xb);
                            int neg = mqd_decode(&mq, cx) ^ xb;
                            t->data[y * t->w + x] = neg ? -oneplushalf : oneplushalf;
                            FL(t, x, y) |= F_SIG | (neg ? F_NEG : 0);
                        }
                        FL(t, x, y) |= F_VISIT;
                    }
            } else {
                for (int x = 0; x < t->w; x++)
                    for (int y = y0; y < y0 + 4 && y < t->h; y++) {
                        if ((FL(t, x, y) & (F_SIG | F_VISIT)) != F_SIG) continue;
                        int v = mqd_decode(&mq, ctx_mag(t, x, y));
                        int32_t *d = &t->data[y * t->w + x];
                        *d += (v ^ (*d < 0)) ? half : -half;
                        FL(t, x, y) |= F_REFINED;
                    }
            }
        }
        if (passtype == 2) {
            for (int y = 0; y < t->h; y++)
                for (int x = 0; x < t->w; x++) FL(t, x, y) &= (uint8_t) ~F_VISIT;
        }
        if (++passtype == 3) { passtype = 0; bp--; }
    }
}

/* ================================================================================================
 * bit reader for packet headers (B.10.1 bit stuffing) and tag trees (B.10.2)
 * ============================================================================================== */
typedef struct { const uint8_t *p, *end; uint32_t buf; int ct; } bior_t;
static void bior_init(bior_t *b, const uint8_t *p, const uint8_t *end) { b->p = p; b->end = end; b->buf = 0; b->ct = 0; }
static int bior_bit(bior_t *b)
{
    if (b->ct == 0) {
        b->buf = (b->buf << 8) & 0xFFFF;
        b->ct = b->buf == 0xFF00 ? 7 : 8;
        if (b->p < b->end) b->buf |= *b->p++;
    }
    b->ct--;
    return (b->buf >> b->ct) & 1;
}
static int bior_bits(bior_t *b, int n) { int v = 0; for (int i = n - 1; i >= 0; i--) v |= bior_bit(b) << i; return v; }
static void bior_align(bior_t *b)
{
    if ((b->buf & 0xFF) == 0xFF) { if (b->p < b->end) b->p++; }   /* a stuffed byte follows a trailing FF */
    b->ct = 0;
}

typedef struct tgnode { struct tgnode *parent; int value, low, known; } tgnode_t;
typedef struct { int nleafh, nleafv, nnodes; tgnode_t *nodes; } tgt_t;

static tgt_t *tgt_create(int nh, int nv)
{
    tgt_t *t = (tgt_t *) calloc(1, sizeof *t);
    int nplh[32], nplv[32], lv = 0, n;
    t->nleafh = nh; t->nleafv = nv;
    nplh[0] = nh; nplv[0] = nv;
    t->nnodes = 0;
    do {
        n = nplh[lv] * nplv[lv];
        nplh[lv + 1] = (nplh[lv] + 1) / 2;
        nplv[lv + 1] = (nplv[lv] + 1) / 2;
        t->nnodes += n;
        lv++;
    } while (n > 1);
    t->nodes = (tgnode_t *) calloc((size_t) t->nnodes, sizeof(tgnode_t));
    tgnode_t *node = t->nodes, *parent = &t->nodes[nh * nv], *parent0 = parent;
    for (int i = 0; i < lv - 1; i++) {
        for (int j = 0; j < nplv[i]; j++) {
            int k = nplh[i];
            while (--k >= 0) {
                node->parent = parent; node++;
                if (--k >= 0) { node->parent = parent; node++; }
                parent++;
            }
            if ((j & 1) || j == nplv[i] - 1) parent0 = parent;
            else { parent = parent0; parent0 += nplh[i]; }
        }
    }
    node->parent = NULL;
    for (int i = 0; i < t->nnodes; i++) { t->nodes[i].value = 999; t->nodes[i].low = 0; t->nodes[i].known = 0; }
    return t;
}
static void tgt_free(tgt_t *t) { if (t) { free(t->nodes); free(t); } }
static int tgt_decode(bior_t *b, tgt_t *t, int leaf, int threshold)
{
    tgnode_t *stk[32], **sp = stk, *node = &t->nodes[leaf];
    while (node->parent) { *sp++ = node; node = node->parent; }
    int low = 0;
    for (;;) {
        if (low > node->low) node->low = low; else low = node->low;
        while (low < threshold && low < node->value) {
            if (bior_bit(b)) node->value = low; else ++low;
        }
        node->low = low;
        if (sp == stk) break;
        node = *--sp;
    }
    return node->value < threshold;
}

/* ================================================================================================
 * inverse 9/7 (T.800 F.3.8.2 with OpenJPEG's scaling: low*K, high*2/K) - one line, in place, interleaved
 * ============================================================================================== */
static const float DWT_ALPHA = -1.586134342f, DWT_BETA = -0.052980118f, DWT_GAMMA = 0.882911075f,
                   DWT_DELTA = 0.443506852f, DWT_K = 1.230174105f, DWT_TWO_INVK = 1.625732422f;

/* One synthesis lifting step on an interleaved line (opj_v8dwt_decode_step2 / opj_dwt_decode_1_real): the `cnt`
 * samples at x[pa], x[pa + 2], ... are updated from their neighbours of the other parity x[pb + 2i - 1 ...];
 * m = number of samples that have both neighbours. */
static void idwt_step(float *x, int pa, int cnt, int m, float c)
{
    /* sample i sits at x[pa + 2 i]; its neighbours are x[pa + 2 i - 1] and x[pa + 2 i + 1]; at the left edge the
     * missing neighbour is mirrored (x[pa + 1]), at the right edge the last sample sees its left neighbour twice */
    int lim = cnt < m ? cnt : m;
    for (int i = 0; i < lim; i++) {
        float left = (pa + 2 * i - 1) >= 0 ? x[pa + 2 * i - 1] : x[pa + 2 * i + 1];
        x[pa + 2 * i] = x[pa + 2 * i] + ((left + x[pa + 2 * i + 1]) * c);
    }
    if (m < cnt) { float c2 = c + c; x[pa + 2 * m] = x[pa + 2 * m] + x[pa + 2 * m - 1] * c2; }
}

/* x: interleaved samples; sn lows, dn highs; cas = parity of the first sample's coordinate (0: it is a low) */
static void idwt97_line(float *x, int sn, int dn, int cas)
{
    int a, b;
    if (cas == 0) { if (!(dn > 0 || sn > 1)) return; a = 0; b = 1; }
    else { if (!(sn > 0 || dn > 1)) return; a = 1; b = 0; }
    /* lows live at x[a + 2i], highs at x[b + 2i] */
    for (int i = 0; i < sn; i++) x[a + 2 * i] = x[a + 2 * i] * DWT_K;
    for (int i = 0; i < dn; i++) x[b + 2 * i] = x[b + 2 * i] * DWT_TWO_INVK;
    idwt_step(x, a, sn, sn < dn - a ? sn : dn - a, -DWT_DELTA);
    idwt_step(x, b, dn, dn < sn - b ? dn : sn - b, -DWT_GAMMA);
    idwt_step(x, a, sn, sn < dn - a ? sn : dn - a, -DWT_BETA);
    idwt_step(x, b, dn, dn < sn - b ? dn : sn - b, -DWT_ALPHA);
}

static void idwt97_tile(float *buf, const tile_t *t)
{
    int W = t->W;
    float *line = (float *) malloc((size_t) (t->W > t->H ? t->W : t->H) * sizeof(float) + 64);
    for (int r = 1; r < J2K_NRES; r++) {
        int rw = t->res[r].x1 - t->res[r].x0, rh = t->res[r].y1 - t->res[r].y0;
        int sn = t->res[r - 1].x1 - t->res[r - 1].x0, dn = rw - sn, cas = t->res[r].x0 & 1;
        /* horizontal pass first (opj_dwt_decode_tile_97) */
        for (int y = 0; y < rh; y++) {
            float *row = buf + (size_t) y * W;
            for (int i = 0; i < sn; i++) line[cas + 2 * i] = row[i];
            for (int i = 0; i < dn; i++) line[1 - cas + 2 * i] = row[sn + i];
            idwt97_line(line, sn, dn, cas);
            memcpy(row, line, (size_t) rw * sizeof(float));
        }
        sn = t->res[r - 1].y1 - t->res[r - 1].y0; dn = rh - sn; cas = t->res[r].y0 & 1;
        for (int x = 0; x < rw; x++) {
            for (int i = 0; i < sn; i++) line[cas + 2 * i] = buf[(size_t) i * W + x];
            for (int i = 0; i < dn; i++) line[1 - cas + 2 * i] = buf[(size_t) (sn + i) * W + x];
            idwt97_line(line, sn, dn, cas);
            for (int i = 0; i < rh; i++) buf[(size_t) i * W + x] = line[i];
        }
    }
    free(line);
}

/* ================================================================================================
 * decoder
 * ============================================================================================== */
static uint32_t be16(const uint8_t *p) { return ((uint32_t) p[0] << 8) | p[1]; }
static uint32_t be32(const uint8_t *p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }

/* decodes one tile-part (packets at p .. tile_end) of the tile whose rows are [ty0, ty0 + H) into out[H * W] */
static void decode_tile(const uint8_t *p, const uint8_t *tile_end, int W, int H, int ty0, const int *expn, const int *mant,
                        int prec, int guard, int32_t *out)
{
    tile_t t;
    tile_init_at(&t, W, H, ty0);
    for (int r = 0; r < J2K_NRES; r++)
        for (int b = 0; b < t.res[r].nbands; b++) {
            int bi = r == 0 ? 0 : 3 * (r - 1) + b + 1;
            band_set_quant(&t.res[r].bands[b], expn[bi], mant[bi], prec, guard);
        }

    float *buf = (float *) calloc((size_t) W * H, sizeof(float));
    /* LRCP, one layer, one component, one precinct per resolution: packets in resolution order (B.12.1.1) */
    for (int r = 0; r < J2K_NRES; r++) {
        res_t *rs = &t.res[r];
        bior_t bio;
        bior_init(&bio, p, tile_end);
        int present = bior_bit(&bio);
        tgt_t *incl[3] = {0}, *imsb[3] = {0};
        if (present) {
            for (int b = 0; b < rs->nbands; b++) {
                band_t *bd = &rs->bands[b];
                if (bd->ncw * bd->nch == 0) continue;
                incl[b] = tgt_create(bd->ncw, bd->nch);
                imsb[b] = tgt_create(bd->ncw, bd->nch);
                for (int ci = 0; ci < bd->ncw * bd->nch; ci++) {
                    cblk_t *c = &bd->cblks[ci];
                    if (!tgt_decode(&bio, incl[b], ci, 1)) continue;
                    int i = 1;
                    while (!tgt_decode(&bio, imsb[b], ci, i)) i++;
                    c->numbps = bd->numbps + 1 - i;
                    /* number of passes, table B.4 */
                    int np;
                    if (!bior_bit(&bio)) np = 1;
                    else if (!bior_bit(&bio)) np = 2;
                    else { int v = bior_bits(&bio, 2); if (v != 3) np = 3 + v; else { v = bior_bits(&bio, 5); if (v != 31) np = 6 + v; else np = 37 + bior_bits(&bio, 7); } }
                    c->npasses = np;
                    int lblock = 3;
                    while (bior_bit(&bio)) lblock++;
                    c->len = bior_bits(&bio, lblock + floorlog2(np));
                }
            }
        }
        bior_align(&bio);
        p = bio.p;
        if (present) {
            for (int b = 0; b < rs->nbands; b++) {
                band_t *bd = &rs->bands[b];
                for (int ci = 0; ci < bd->ncw * bd->nch; ci++) {
                    cblk_t *c = &bd->cblks[ci];
                    if (!c->npasses) continue;
                    c->data = (uint8_t *) malloc((size_t) c->len + 2);
                    memcpy(c->data, p, (size_t) c->len);
                    p += c->len;
                    if (getenv("ORC_J2K_DUMP")) {
                        unsigned hsh = 0;
                        for (int k = 0; k < c->len; k++) hsh = hsh * 131 + c->data[k];
                        fprintf(stderr, "r%d b%d c%d numbps %d passes %d len %d hash %08x first %02x %02x last %02x\n", r, b, ci,
                                c->numbps, c->npasses, c->len, hsh, c->data[0], c->len > 1 ? c->data[1] : 0, c->data[c->len - 1]);
                    }
                    /* tier-1 + dequantisation straight into the tile buffer */
                    t1_t t1;
                    t1_alloc(&t1, c->x1 - c->x0, c->y1 - c->y0);
                    t1_decode_cblk(&t1, c, bd->orient);
                    const float step = 0.5f * bd->stepsize;
                    for (int y = 0; y < t1.h; y++)
                        for (int x = 0; x < t1.w; x++)
                            buf[(size_t) (bd->offy + c->y0 - bd->y0 + y) * W + bd->offx + c->x0 - bd->x0 + x] =
                                (float) t1.data[y * t1.w + x] * step;
                    t1_free(&t1);
                }
                tgt_free(incl[b]); tgt_free(imsb[b]);
            }
        }
    }

    idwt97_tile(buf, &t);

    /* DC level shift + rounding + clamp (unsigned prec bits) */
    const int64_t shift = (int64_t) 1 << (prec - 1), vmax = ((int64_t) 1 << prec) - 1;
    for (size_t i = 0; i < (size_t) W * H; i++) {
        int64_t v = (int64_t) lrintf(buf[i]) + shift;
        out[i] = (int32_t) (v < 0 ? 0 : (v > vmax ? vmax : v));
    }
    free(buf);
    tile_free(&t);
}


size_t orc_j2k_decode(const uint8_t *cs, size_t n, int32_t **samples, size_t *height, size_t *width)
{
    if (n < 4 || be16(cs) != 0xFF4F) return 0;
    size_t pos = 2;
    int W = 0, H = 0, TW = 0, TH = 0, prec = 16, guard = 2, nres = 0, qsty = 0;
    int expn[3 * J2K_NRES], mant[3 * J2K_NRES];
    memset(expn, 0, sizeof expn); memset(mant, 0, sizeof mant);
    int32_t *out = NULL;
    int tiles_seen = 0;
    while (pos + 4 <= n) {
        uint32_t mk = be16(cs + pos), len = be16(cs + pos + 2);
        const uint8_t *p = cs + pos + 4;
        if (mk == 0xFFD9) break;                    /* EOC */
        if (mk == 0xFF51) {                         /* SIZ, A.5.1 */
            W = (int) (be32(p + 2) - be32(p + 10));
            H = (int) (be32(p + 6) - be32(p + 14));
            TW = (int) be32(p + 18); TH = (int) be32(p + 22);
            prec = (p[36] & 0x7F) + 1;
        } else if (mk == 0xFF52) {                  /* COD, A.6.1 */
            nres = p[5] + 1;
        } else if (mk == 0xFF5C) {                  /* QCD, A.6.4 */
            qsty = p[0] & 0x1F; guard = p[0] >> 5;
            int nb = (int) (len - 3) / 2;
            for (int i = 0; i < nb && i < 3 * J2K_NRES; i++) {
                uint32_t v = be16(p + 1 + 2 * i);
                expn[i] = (int) (v >> 11); mant[i] = (int) (v & 0x7FF);
            }
        } else if (mk == 0xFF90) {                  /* SOT, A.4.2: one tile-part per tile, tiles stacked along y */
            if (nres != J2K_NRES || qsty != 2 || W <= 0 || H <= 0 || TW < W || TH <= 0) { free(out); return 0; }
            if (!out) out = (int32_t *) calloc((size_t) W * H, sizeof(int32_t));
            uint32_t isot = be16(p), psot = be32(p + 2);
            const uint8_t *sot = cs + pos;
            pos += 2 + len;
            if (pos + 2 > n || be16(cs + pos) != 0xFF93) { free(out); return 0; }   /* SOD */
            const uint8_t *tile_end = psot ? sot + psot : cs + n - 2;
            int ty0 = (int) isot * TH, th = ty0 + TH <= H ? TH : H - ty0;
            if (ty0 >= H || tile_end > cs + n) { free(out); return 0; }
            decode_tile(cs + pos + 2, tile_end, W, th, ty0, expn, mant, prec, guard, out + (size_t) ty0 * W);
            tiles_seen++;
            pos = (size_t) (tile_end - cs);
            continue;
        }
        pos += 2 + len;
    }
    if (!tiles_seen) { free(out); return 0; }
    *samples = out;
    if (height) *height = (size_t) H;
    if (width) *width = (size_t) W;
    return (size_t) W * H;
}

/* ================================================================================================
 * encoder
 * ============================================================================================== */
static const double NORMS_REAL[4][10] = {   /* synthesis-basis L2 norms OpenJPEG uses for the 9/7 (its 2/K convention) */
    {1.000, 1.965, 4.177, 8.403, 16.90, 33.84, 67.69, 135.3, 270.6, 540.9},
    {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0},
    {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0},
    {2.080, 3.865, 8.307, 17.18, 34.71, 69.59, 139.3, 278.6, 557.2}};

#define NMSEDEC_BITS 7
#define NMSEDEC_FRACBITS 6
static int16_t LUT_SIG[128], LUT_SIG0[128], LUT_REF[128], LUT_REF0[128];
static int luts_ready = 0;
static void init_luts(void)
{
    if (luts_ready) return;
    /* squared-error reduction estimates (T.800 J.14.4), fixed point with 13 fractional bits */
    for (int i = 0; i < 128; i++) {
        double t = i / pow(2, NMSEDEC_FRACBITS), u, v;
        int x;
        u = t; v = t - 1.5;
        x = (int) (floor((u * u - v * v) * pow(2, NMSEDEC_FRACBITS) + 0.5) / pow(2, NMSEDEC_FRACBITS) * 8192.0);
        LUT_SIG[i] = (int16_t) (x > 0 ? x : 0);
        x = (int) (floor((u * u) * pow(2, NMSEDEC_FRACBITS) + 0.5) / pow(2, NMSEDEC_FRACBITS) * 8192.0);
        LUT_SIG0[i] = (int16_t) (x > 0 ? x : 0);
        u = t - 1.0;
        v = (i & (1 << (NMSEDEC_BITS - 1))) ? t - 1.5 : t - 0.5;
        x = (int) (floor((u * u - v * v) * pow(2, NMSEDEC_FRACBITS) + 0.5) / pow(2, NMSEDEC_FRACBITS) * 8192.0);
        LUT_REF[i] = (int16_t) (x > 0 ? x : 0);
        x = (int) (floor((u * u) * pow(2, NMSEDEC_FRACBITS) + 0.5) / pow(2, NMSEDEC_FRACBITS) * 8192.0);
        LUT_REF0[i] = (int16_t) (x > 0 ? x : 0);
    }
    luts_ready = 1;
}
const int16_t *orc_j2k_lut(int which) { init_luts(); return which == 0 ? LUT_SIG : which == 1 ? LUT_SIG0 : which == 2 ? LUT_REF : LUT_REF0; }

static int nmsedec_sig(uint32_t x, int bpno) { return bpno > 0 ? LUT_SIG[(x >> bpno) & 127] : LUT_SIG0[x & 127]; }
static int nmsedec_ref(uint32_t x, int bpno) { return bpno > 0 ? LUT_REF[(x >> bpno) & 127] : LUT_REF0[x & 127]; }

/* ---- forward 9/7 on one interleaved line (even = low), OpenJPEG order and scaling (low/K, high*K) */
static void fdwt_step(float *w, int a, int b, int end, int m, float c)
{
    /* w[a + 2i ... ] is updated from its two neighbours of the other parity */
    float *fl = w + a, *fw = w + b + 1;
    int imax = end < m ? end : m;
    if (imax > 0) {
        fw[-1] += (fl[0] + fw[0]) * c;
        fw += 2;
        for (int i = 1; i < imax; i++) { fw[-1] += (fw[-2] + fw[0]) * c; fw += 2; }
    }
    if (m < end) fw[-1] += (2 * fw[-2]) * c;
}
static void fdwt97_line(float *w, int sn, int dn, int cas)
{
    const float invK = (float) (1.0 / 1.230174105);
    int a, b;
    if (cas == 0) { if (sn + dn <= 1) return; a = 0; b = 1; }
    else { if (!(sn > 0 || dn > 1)) return; a = 1; b = 0; }         /* a lone high-pass sample: see fdwt97_tile */
    fdwt_step(w, a, b, dn, dn < sn - b ? dn : sn - b, DWT_ALPHA);
    fdwt_step(w, b, a, sn, sn < dn - a ? sn : dn - a, DWT_BETA);
    fdwt_step(w, a, b, dn, dn < sn - b ? dn : sn - b, DWT_GAMMA);
    fdwt_step(w, b, a, sn, sn < dn - a ? sn : dn - a, DWT_DELTA);
    for (int i = 0; i < sn; i++) w[a + 2 * i] *= invK;
    for (int i = 0; i < dn; i++) w[b + 2 * i] *= DWT_K;
}

static void fdwt97_tile(float *buf, const tile_t *t)
{
    int W = t->W;
    float *line = (float *) malloc((size_t) (t->W > t->H ? t->W : t->H) * sizeof(float) + 64);
    for (int r = J2K_NRES - 1; r >= 1; r--) {
        int rw = t->res[r].x1 - t->res[r].x0, rh = t->res[r].y1 - t->res[r].y0;
        int sn = t->res[r - 1].y1 - t->res[r - 1].y0, dn = rh - sn, cas = t->res[r].y0 & 1;
        /* vertical pass first (opj_dwt_encode_procedure) */
        for (int x = 0; x < rw; x++) {
            if (rh == 1) break;
            for (int i = 0; i < rh; i++) line[i] = buf[(size_t) i * W + x];
            fdwt97_line(line, sn, dn, cas);
            for (int i = 0; i < sn; i++) buf[(size_t) i * W + x] = line[cas + 2 * i];
            for (int i = 0; i < dn; i++) buf[(size_t) (sn + i) * W + x] = line[1 - cas + 2 * i];
        }
        sn = t->res[r - 1].x1 - t->res[r - 1].x0; dn = rw - sn; cas = t->res[r].x0 & 1;
        for (int y = 0; y < rh; y++) {
            if (rw == 1) break;
            float *row = buf + (size_t) y * W;
            memcpy(line, row, (size_t) rw * sizeof(float));
            fdwt97_line(line, sn, dn, cas);
            for (int i = 0; i < sn; i++) row[i] = line[cas + 2 * i];
            for (int i = 0; i < dn; i++) row[sn + i] = line[1 - cas + 2 * i];
        }
    }
    free(line);
}

/* ---- T1 encoder (D.3 coding passes) with OpenJPEG's distortion bookkeeping */
typedef struct {
    t1_t t;
    uint32_t *mag;      /* |quantised| with 6 fractional bits */
} t1e_t;

static void t1e_sign(t1_t *t, mq_t *mq, int x, int y, int neg)
{
    int xb, cx = ctx_sc(t, x, y, &xb);
    mqe_encode(mq, cx, neg ^ xb);
    FL(t, x, y) |= F_SIG | (neg ? F_NEG : 0);
}

static void t1_encode_cblk(t1e_t *e, cblk_t *cb, int orient, int level, float stepsize)
{
    t1_t *t = &e->t;
    int w = t->w, h = t->h;
    uint32_t max = 0;
    for (int i = 0; i < w * h; i++) if (e->mag[i] > max) max = e->mag[i];
    cb->numbps = max ? (floorlog2((int) max) + 1) - NMSEDEC_FRACBITS : 0;
    cb->totalpasses = 0;
    cb->len = 0;
    if (cb->numbps <= 0) return;            /* numbps < 0 leaves the loop below empty in OpenJPEG; same outcome */

    size_t cap = (size_t) w * h * 4 + 64;
    uint8_t *store = (uint8_t *) calloc(cap + 2, 1);
    mq_t mq;
    mqe_init(&mq, store + 1);
    int bpno = cb->numbps - 1, passtype = 2, passno = 0;
    double cum = 0;
    for (; bpno >= 0; passno++) {
        int nmsedec = 0;
        uint32_t one = 1u << (bpno + NMSEDEC_FRACBITS);
        for (int y0 = 0; y0 < h; y0 += 4) {
            if (passtype == 0) {
                for (int x = 0; x < w; x++)
                    for (int y = y0; y < y0 + 4 && y < h; y++) {
                        if ((FL(t, x, y) & (F_SIG | F_VISIT)) || !has_sig_neighbour(t, x, y)) continue;
                        uint32_t m = e->mag[y * w + x];
                        int v = (m & one) ? 1 : 0;
                        mqe_encode(&mq, ctx_zc(t, x, y, orient), v);
                        if (v) {
                            nmsedec += nmsedec_sig(m, bpno);
                            t1e_sign(t, &mq, x, y, t->data[y * w + x] < 0);
                        }
                        FL(t, x, y) |= F_VISIT;
                    }
            } else if (passtype == 1) {
                for (int x = 0; x < w; x++)
                    for (int y = y0; y < y0 + 4 && y < h; y++) {
                        if ((FL(t, x, y) & (F_SIG | F_VISIT)) != F_SIG) continue;
                        uint32_t m = e->mag[y * w + x];
                        nmsedec += nmsedec_ref(m, bpno);
                        mqe_encode(&mq, ctx_mag(t, x, y), (m & one) ? 1 : 0);
                        FL(t, x, y) |= F_REFINED;
                    }
            } else {
                for (int x = 0; x < w; x++) {
                    int y = y0, agg = 0, partial = 0;
                    if (y0 + 3 < h) {
                        agg = 1;
                        for (int k = 0; k < 4; k++)
                            if ((FL(t, x, y0 + k) & (F_SIG | F_VISIT)) || has_sig_neighbour(t, x, y0 + k)) { agg = 0; break; }
                    }
                    if (agg) {
                        int runlen = 0;
                        for (; runlen < 4; runlen++)
                            if (e->mag[(y0 + runlen) * w + x] & one) break;
                        mqe_encode(&mq, CTX_AGG, runlen != 4);
                        if (runlen == 4) continue;
                        mqe_encode(&mq, CTX_UNI, runlen >> 1);
                        mqe_encode(&mq, CTX_UNI, runlen & 1);
                        y = y0 + runlen;
                        partial = 1;
                    }
                    for (; y < y0 + 4 && y < h; y++) {
                        if (!partial && (FL(t, x, y) & (F_SIG | F_VISIT))) continue;
                        uint32_t m = e->mag[y * w + x];
                        int v = (m & one) ? 1 : 0;
                        if (!partial) mqe_encode(&mq, ctx_zc(t, x, y, orient), v);
                        partial = 0;
                        if (v) {
                            nmsedec += nmsedec_sig(m, bpno);
                            t1e_sign(t, &mq, x, y, t->data[y * w + x] < 0);
                        }
                    }
                }
            }
        }
        if (passtype == 2)
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) FL(t, x, y) &= (uint8_t) ~F_VISIT;

        /* opj_t1_getwmsedec */
        /* the distortion weight uses the step WITHOUT the sub-band gain (OpenJPEG keeps its pre-2.4 behaviour) */
        double st = (double) stepsize / (double) (1 << (orient == 0 ? 0 : (orient == 3 ? 2 : 1)));
        double wm = 1.0 * NORMS_REAL[orient][level] * st * (double) (1 << bpno);
        wm *= wm * nmsedec / 8192.0;
        cum += wm;
        cb->disto[passno] = cum;
        if (passtype == 2 && bpno == 0) {                  /* the only terminated pass with cblksty 0 */
            mqe_flush(&mq);
            cb->rate[passno] = mqe_numbytes(&mq);
        } else {
            cb->rate[passno] = (int) ((uint32_t) mqe_numbytes(&mq) + 3u);
        }
        if (++passtype == 3) { passtype = 0; bpno--; }
    }
    cb->totalpasses = passno;
    /* pass rates must be non-decreasing and never end on 0xFF */
    int last = mqe_numbytes(&mq);
    for (int p = passno; p > 0;) {
        --p;
        if (cb->rate[p] > last) cb->rate[p] = last; else last = cb->rate[p];
    }
    uint8_t *bytes = store + 1;
    for (int p = 0; p < passno; p++)
        if (cb->rate[p] > 0 && bytes[cb->rate[p] - 1] == 0xFF) cb->rate[p]--;
    cb->len = mqe_numbytes(&mq);
    cb->data = (uint8_t *) malloc((size_t) cb->len + 1);
    memcpy(cb->data, bytes, (size_t) cb->len);
    free(store);
}

/* ---- bit writer (B.10.1) and tag-tree encoder (B.10.2) */
typedef struct { uint8_t *start, *p, *end; uint32_t buf; int ct; int ok; } biow_t;
static void biow_init(biow_t *b, uint8_t *p, size_t len) { b->start = b->p = p; b->end = p + len; b->buf = 0; b->ct = 8; b->ok = 1; }
static int biow_byteout(biow_t *b)
{
    b->buf = (b->buf << 8) & 0xFFFF;
    b->ct = b->buf == 0xFF00 ? 7 : 8;
    if (b->p >= b->end) { b->ok = 0; return 0; }
    *b->p++ = (uint8_t) (b->buf >> 8);
    return 1;
}
static void biow_bit(biow_t *b, int v)
{
    if (b->ct == 0) biow_byteout(b);
    b->ct--;
    b->buf |= (uint32_t) (v & 1) << b->ct;
}
static void biow_bits(biow_t *b, uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) biow_bit(b, (int) ((v >> i) & 1)); }
static int biow_flush(biow_t *b)
{
    if (!biow_byteout(b)) return 0;
    if (b->ct == 7 && !biow_byteout(b)) return 0;
    return b->ok;
}
static void tgt_reset(tgt_t *t) { for (int i = 0; i < t->nnodes; i++) { t->nodes[i].value = 999; t->nodes[i].low = 0; t->nodes[i].known = 0; } }
static void tgt_setvalue(tgt_t *t, int leaf, int value)
{
    tgnode_t *n = &t->nodes[leaf];
    while (n && n->value > value) { n->value = value; n = n->parent; }
}
static void tgt_encode(biow_t *b, tgt_t *t, int leaf, int threshold)
{
    tgnode_t *stk[32], **sp = stk, *node = &t->nodes[leaf];
    while (node->parent) { *sp++ = node; node = node->parent; }
    int low = 0;
    for (;;) {
        if (low > node->low) node->low = low; else low = node->low;
        while (low < threshold) {
            if (low >= node->value) {
                if (!node->known) { biow_bit(b, 1); node->known = 1; }
                break;
            }
            biow_bit(b, 0);
            ++low;
        }
        node->low = low;
        if (sp == stk) break;
        node = *--sp;
    }
}
static void put_numpasses(biow_t *b, int n)
{
    if (n == 1) biow_bits(b, 0, 1);
    else if (n == 2) biow_bits(b, 2, 2);
    else if (n <= 5) biow_bits(b, 0xC | (uint32_t) (n - 3), 4);
    else if (n <= 36) biow_bits(b, 0x1E0 | (uint32_t) (n - 6), 9);
    else biow_bits(b, 0xFF80 | (uint32_t) (n - 37), 16);
}

/* One LRCP pass over the tile for a single layer: writes all packets, returns bytes or -1 if they do not fit. */
static long t2_encode(tile_t *t, uint8_t *dest, size_t maxlen, int final)
{
    uint8_t *c = dest;
    size_t length = maxlen;
    for (int r = 0; r < J2K_NRES; r++) {
        res_t *rs = &t->res[r];
        tgt_t *incl[3] = {0}, *imsb[3] = {0};
        int empty = 1;
        for (int b = 0; b < rs->nbands; b++) {
            band_t *bd = &rs->bands[b];
            int nc = bd->ncw * bd->nch;
            if (!nc) continue;
            incl[b] = tgt_create(bd->ncw, bd->nch);
            imsb[b] = tgt_create(bd->ncw, bd->nch);
            tgt_reset(incl[b]); tgt_reset(imsb[b]);
            for (int ci = 0; ci < nc; ci++) {
                cblk_t *cb = &bd->cblks[ci];
                tgt_setvalue(imsb[b], ci, bd->numbps - cb->numbps);
                if (cb->npasses) empty = 0;
            }
        }
        biow_t bio;
        biow_init(&bio, c, length);
        (void) empty;
        biow_bit(&bio, 1);                       /* OpenJPEG 2.4.0 never signals an empty packet */
        {
            for (int b = 0; b < rs->nbands; b++) {
                band_t *bd = &rs->bands[b];
                int nc = bd->ncw * bd->nch;
                if (!nc) continue;
                for (int ci = 0; ci < nc; ci++)
                    if (bd->cblks[ci].npasses) tgt_setvalue(incl[b], ci, 0);
                for (int ci = 0; ci < nc; ci++) {
                    cblk_t *cb = &bd->cblks[ci];
                    tgt_encode(&bio, incl[b], ci, 1);
                    if (!cb->npasses) continue;
                    cb->numlenbits = 3;
                    tgt_encode(&bio, imsb[b], ci, 999);
                    put_numpasses(&bio, cb->npasses);
                    int seglen = cb->rate[cb->npasses - 1];
                    int inc = floorlog2(seglen) + 1 - (cb->numlenbits + floorlog2(cb->npasses));
                    if (inc < 0) inc = 0;
                    for (int k = 0; k < inc; k++) biow_bit(&bio, 1);       /* comma code */
                    biow_bit(&bio, 0);
                    cb->numlenbits += inc;
                    biow_bits(&bio, (uint32_t) seglen, cb->numlenbits + floorlog2(cb->npasses));
                }
            }
        }
        int ok = biow_flush(&bio);
        for (int b = 0; b < 3; b++) { tgt_free(incl[b]); tgt_free(imsb[b]); }
        if (!ok) return -1;
        size_t nb = (size_t) (bio.p - bio.start);
        c += nb; length -= nb;
        for (int b = 0; b < rs->nbands; b++) {
            band_t *bd = &rs->bands[b];
            for (int ci = 0; ci < bd->ncw * bd->nch; ci++) {
                cblk_t *cb = &bd->cblks[ci];
                if (!cb->npasses) continue;
                size_t sl = (size_t) cb->rate[cb->npasses - 1];
                if (sl > length) return -1;
                if (final) memcpy(c, cb->data, sl);
                c += sl; length -= sl;
            }
        }
    }
    return (long) (c - dest);
}

/* opj_tcd_makelayer for the single layer: choose the number of passes per code-block for a slope threshold */
static void make_layer(tile_t *t, double thresh)
{
    for (int r = 0; r < J2K_NRES; r++)
        for (int b = 0; b < t->res[r].nbands; b++) {
            band_t *bd = &t->res[r].bands[b];
            for (int ci = 0; ci < bd->ncw * bd->nch; ci++) {
                cblk_t *cb = &bd->cblks[ci];
                int n = 0;
                if (thresh < 0) n = cb->totalpasses;             /* "use all passes" */
                else for (int p = 0; p < cb->totalpasses; p++) {
                    uint32_t dr; double dd;
                    if (n == 0) { dr = (uint32_t) cb->rate[p]; dd = cb->disto[p]; }
                    else { dr = (uint32_t) (cb->rate[p] - cb->rate[n - 1]); dd = cb->disto[p] - cb->disto[n - 1]; }
                    if (!dr) { if (dd != 0) n = p + 1; continue; }
                    if (thresh - (dd / dr) < DBL_EPSILON) n = p + 1;
                }
                cb->npasses = n;
            }
        }
}

/* everything of the encoder that does not depend on the rate: transform, quantisation, tier-1 */
/* test hook: hands every code-block's quantised coefficients (6 fractional bits) to the caller, in packet order */
static void (*g_block_sink)(const int32_t *q, int w, int h, int orient, void *user) = NULL;
static void *g_block_sink_user = NULL;
void orc_j2k_set_block_sink(void (*fn)(const int32_t *, int, int, int, void *), void *user) { g_block_sink = fn; g_block_sink_user = user; }

static void j2k_analyse_at(const uint16_t *img, int H, int W, int ty0, tile_t *tp, int *expn, int *mant)
{
    init_luts();
    const int prec = 16, guard = 2;
    tile_init_at(tp, W, H, ty0);

    /* QCD step sizes: opj_dwt_calc_explicit_stepsizes + encode_stepsize */
    for (int bi = 0; bi < 3 * J2K_NRES - 2; bi++) {
        int resno = bi == 0 ? 0 : (bi - 1) / 3 + 1, orient = bi == 0 ? 0 : (bi - 1) % 3 + 1;
        int level = J2K_NRES - 1 - resno;
        double stepsize = 1.0 / NORMS_REAL[orient][level];
        int v = (int) floor(stepsize * 8192.0);
        int p = floorlog2(v) - 13, n = 11 - floorlog2(v);
        mant[bi] = (n < 0 ? v >> -n : v << n) & 0x7FF;
        expn[bi] = prec - p;
    }
    for (int r = 0; r < J2K_NRES; r++)
        for (int b = 0; b < tp->res[r].nbands; b++) {
            band_t *bd = &tp->res[r].bands[b];
            int bi = r == 0 ? 0 : 3 * (r - 1) + b + 1;
            band_set_quant(bd, expn[bi], mant[bi], prec, guard);
            /* encoder-side step carries the sub-band gain (OpenJPEG pairs this with its 2/K synthesis scaling) */
            int log2_gain = bd->orient == 0 ? 0 : (bd->orient == 3 ? 2 : 1);
            bd->stepsize = (float) ((1.0 + mant[bi] / 2048.0) * pow(2.0, (double) (prec + log2_gain - expn[bi])));
        }

    /* DC level shift to float, forward transform */
    float *buf = (float *) malloc((size_t) W * H * sizeof(float));
    for (size_t i = 0; i < (size_t) W * H; i++) buf[i] = (float) ((int) img[i] - (1 << (prec - 1)));
    fdwt97_tile(buf, tp);

    /* tier-1 */
    for (int r = 0; r < J2K_NRES; r++)
        for (int b = 0; b < tp->res[r].nbands; b++) {
            band_t *bd = &tp->res[r].bands[b];
            for (int ci = 0; ci < bd->ncw * bd->nch; ci++) {
                cblk_t *cb = &bd->cblks[ci];
                t1e_t e;
                int w = cb->x1 - cb->x0, h = cb->y1 - cb->y0;
                t1_alloc(&e.t, w, h);
                e.mag = (uint32_t *) malloc((size_t) w * h * sizeof(uint32_t));
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) {
                        float cf = buf[(size_t) (bd->offy + cb->y0 - bd->y0 + y) * W + bd->offx + cb->x0 - bd->x0 + x];
                        int q = (int) lrintf((cf / bd->stepsize) * (float) (1 << NMSEDEC_FRACBITS));
                        e.t.data[y * w + x] = q;
                        e.mag[y * w + x] = (uint32_t) (q < 0 ? -q : q);
                    }
                if (g_block_sink) g_block_sink(e.t.data, w, h, bd->orient, g_block_sink_user);
                t1_encode_cblk(&e, cb, bd->orient, bd->level, bd->stepsize);
                free(e.mag);
                t1_free(&e.t);
            }
        }
    free(buf);

}

static void j2k_analyse(const uint16_t *img, int H, int W, tile_t *tp, int *expn, int *mant) { j2k_analyse_at(img, H, W, 0, tp, expn, mant); }

#define PUT16(v) do { *p++ = (uint8_t) ((v) >> 8); *p++ = (uint8_t) (v); } while (0)
#define PUT32(v) do { PUT16((uint32_t) (v) >> 16); PUT16((uint32_t) (v) & 0xFFFF); } while (0)

/* main header (A.5.1, A.6.1, A.6.4, A.9.2) of an image of `tiles` tiles of H rows stacked along y */
static uint8_t *write_main_header(uint8_t *p, int W, int H, int tiles, const int *expn, const int *mant)
{
    static const char comment[] = "Created by OpenJPEG version 2.4.0";
    const int prec = 16, guard = 2;
    PUT16(0xFF4F);
    PUT16(0xFF51); PUT16(41); PUT16(0); PUT32(W); PUT32(H * tiles); PUT32(0); PUT32(0); PUT32(W); PUT32(H); PUT32(0); PUT32(0);
    PUT16(1); *p++ = (uint8_t) (prec - 1); *p++ = 1; *p++ = 1;
    PUT16(0xFF52); PUT16(12); *p++ = 0; *p++ = 0; PUT16(1); *p++ = 0; *p++ = J2K_NRES - 1; *p++ = 4; *p++ = 4; *p++ = 0; *p++ = 0;
    PUT16(0xFF5C); PUT16(3 + 2 * (3 * J2K_NRES - 2)); *p++ = (uint8_t) (2 + (guard << 5));
    for (int bi = 0; bi < 3 * J2K_NRES - 2; bi++) PUT16((uint32_t) ((expn[bi] << 11) | mant[bi]));
    PUT16(0xFF64); PUT16(4 + (int) strlen(comment)); PUT16(1);
    memcpy(p, comment, strlen(comment)); p += strlen(comment);
    return p;
}

/* one tile-part (SOT, SOD, packets) of an analysed tile at `p`; returns its end, or NULL when the layer does not fit.
 * The byte budget is opj_j2k_setup_encoder + opj_j2k_update_rates for one layer: every tile carries an equal share of
 * the main header. */
static uint8_t *write_tile_part(tile_t *t, uint8_t *p, uint8_t *end, float base_cr, size_t main_hdr, int tiles, int isot)
{
    const int prec = 16, W = t->W, H = t->H;
    float rate = base_cr / 2;
    if (rate <= 1.0f) rate = 0.0f;                               /* lossless: keep every pass */
    if (rate > 0.0f) {
        rate = (float) (((double) prec * (double) W * (double) H) / ((double) rate * 8.0)) - 0.0f;
        rate -= (float) main_hdr / (float) tiles;
        if (rate < 30.0f) rate = 30.0f;
    }

    /* PCRD: opj_tcd_rateallocate */
    double smin = DBL_MAX, smax = 0;
    for (int r = 0; r < J2K_NRES; r++)
        for (int b = 0; b < t->res[r].nbands; b++) {
            band_t *bd = &t->res[r].bands[b];
            for (int ci = 0; ci < bd->ncw * bd->nch; ci++) {
                cblk_t *cb = &bd->cblks[ci];
                for (int ps = 0; ps < cb->totalpasses; ps++) {
                    int dr; double dd;
                    if (ps == 0) { dr = cb->rate[0]; dd = cb->disto[0]; }
                    else { dr = cb->rate[ps] - cb->rate[ps - 1]; dd = cb->disto[ps] - cb->disto[ps - 1]; }
                    if (dr == 0) continue;
                    double slope = dd / dr;
                    if (slope < smin) smin = slope;
                    if (slope > smax) smax = slope;
                }
            }
        }
    uint8_t *sot = p;
    p += 12 + 2;
    size_t room = (size_t) (end - p) - 2;
    double good = -1;                                            /* rate 0: every pass */
    if (rate > 0.0f) {
        size_t maxlen = (size_t) ceil(rate);
        if (maxlen > room) maxlen = room;
        double lo = smin, hi = smax, thresh = 0, stable = 0;
        for (int i = 0; i < 128; i++) {
            thresh = (lo + hi) / 2;
            make_layer(t, thresh);
            if (t2_encode(t, p, maxlen, 0) < 0) { lo = thresh; continue; }
            hi = thresh;
            stable = thresh;
        }
        good = stable == 0 ? thresh : stable;
    }
    make_layer(t, good);
    long body = t2_encode(t, p, room, 1);
    if (body < 0) return NULL;
    uint8_t *after = p + body;
    /* SOT (A.4.2) / SOD */
    p = sot;
    PUT16(0xFF90); PUT16(10); PUT16(isot); PUT32(12 + 2 + body); *p++ = 0; *p++ = 1;
    PUT16(0xFF93);
    return after;
}

/* `tiles` frames of H x W stacked along y, one tile each (ebcc_codec.c:105-180 with n_tiles > 1: cp_tdy = frame
 * height, one opj_write_tile per frame); tiles == 1 is the single-tile image */
size_t orc_j2k_encode_tiled(const uint16_t *img, size_t tiles, size_t height, size_t width, float base_cr, uint8_t **out)
{
    const int W = (int) width, H = (int) height, F = (int) tiles;
    int expn[3 * J2K_NRES - 2], mant[3 * J2K_NRES - 2];
    size_t cap = 1024 + (size_t) W * H * 4 * F;
    uint8_t *o = (uint8_t *) calloc(cap, 1), *p = o;
    size_t main_hdr = 0;
    for (int k = 0; k < F; k++) {
        tile_t t;
        j2k_analyse_at(img + (size_t) k * W * H, H, W, k * H, &t, expn, mant);
        if (k == 0) { p = write_main_header(p, W, H, F, expn, mant); main_hdr = (size_t) (p - o); }
        p = write_tile_part(&t, p, o + cap, base_cr, main_hdr, F, k);
        tile_free(&t);
        if (!p) { free(o); return 0; }
    }
    PUT16(0xFFD9);
    *out = o;
    return (size_t) (p - o);
}

size_t orc_j2k_encode(const uint16_t *img, size_t height, size_t width, float base_cr, uint8_t **out)
{
    return orc_j2k_encode_tiled(img, 1, height, width, base_cr, out);
}

/* ================================================================================================
 * unit-level entry points used by the tests to pin the product's tier-1 coder block by block
 * ============================================================================================== */
/* q: w*h quantised coefficients WITH 6 fractional bits (what lrintf((c/step)*64) yields) */
int orc_j2k_t1_encode(const int32_t *q, int w, int h, int orient, int level, float stepsize, uint8_t *out, int out_cap,
                      int *numbps, int *rates, double *disto)
{
    init_luts();
    t1e_t e;
    cblk_t cb;
    memset(&cb, 0, sizeof cb);
    t1_alloc(&e.t, w, h);
    e.mag = (uint32_t *) malloc((size_t) w * h * sizeof(uint32_t));
    for (int i = 0; i < w * h; i++) { e.t.data[i] = q[i]; e.mag[i] = (uint32_t) (q[i] < 0 ? -q[i] : q[i]); }
    t1_encode_cblk(&e, &cb, orient, level, stepsize);
    *numbps = cb.numbps;
    for (int p = 0; p < cb.totalpasses; p++) { rates[p] = cb.rate[p]; disto[p] = cb.disto[p]; }
    int n = cb.len < out_cap ? cb.len : out_cap;
    if (cb.data) memcpy(out, cb.data, (size_t) n);
    free(cb.data); free(e.mag); t1_free(&e.t);
    return cb.totalpasses;
}

/* decode `npasses` passes of a code-block; out: w*h values in OpenJPEG's half-unit convention */
void orc_j2k_t1_decode(const uint8_t *data, int len, int numbps, int npasses, int w, int h, int orient, int32_t *out)
{
    t1_t t;
    cblk_t cb;
    memset(&cb, 0, sizeof cb);
    cb.data = (uint8_t *) data; cb.len = len; cb.numbps = numbps; cb.npasses = npasses;
    t1_alloc(&t, w, h);
    t1_decode_cblk(&t, &cb, orient);
    memcpy(out, t.data, (size_t) w * h * sizeof(int32_t));
    t1_free(&t);
}

/* per-code-block tier-1 results in packet order (resolution, band, raster): for pinning the device analysis */
int orc_j2k_analysis(const uint16_t *img, size_t height, size_t width, int *numbps, int *totalpasses, int *lens,
                     int *rates /* [nb][100] */, double *disto /* [nb][100] */, unsigned *hashes)
{
    tile_t t;
    int expn[3 * J2K_NRES - 2], mant[3 * J2K_NRES - 2];
    j2k_analyse(img, (int) height, (int) width, &t, expn, mant);
    int nb = 0;
    for (int r = 0; r < J2K_NRES; r++)
        for (int b = 0; b < t.res[r].nbands; b++) {
            band_t *bd = &t.res[r].bands[b];
            for (int ci = 0; ci < bd->ncw * bd->nch; ci++, nb++) {
                cblk_t *cb = &bd->cblks[ci];
                numbps[nb] = cb->numbps; totalpasses[nb] = cb->totalpasses; lens[nb] = cb->len;
                for (int p = 0; p < cb->totalpasses; p++) { rates[nb * 100 + p] = cb->rate[p]; disto[nb * 100 + p] = cb->disto[p]; }
                unsigned hsh = 0;
                for (int k = 0; k < cb->len; k++) hsh = hsh * 131 + cb->data[k];
                hashes[nb] = hsh;
            }
        }
    tile_free(&t);
    return nb;
}
