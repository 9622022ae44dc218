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
 * library through opj_backend.c (tests/golden/j2k_*.npz) - see tests/test_oracle_j2k.py.
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
    res_t res[J2K_NRES];
} tile_t;

static int ceildivpow2(int a, int b) { return (int) (((int64_t) a + ((int64_t) 1 << b) - 1) >> b); }
static int floorlog2(int a) { int l = 0; while (a > 1) { a >>= 1; l++; } return l; }

static void tile_init(tile_t *t, int W, int H)
{
    memset(t, 0, sizeof *t);
    t->W = W; t->H = H;
    for (int r = 0; r < J2K_NRES; r++) {
        int lv = J2K_NRES - 1 - r;
        res_t *rs = &t->res[r];
        rs->x0 = 0; rs->y0 = 0;
        rs->x1 = ceildivpow2(W, lv); rs->y1 = ceildivpow2(H, lv);
        rs->nbands = r == 0 ? 1 : 3;
        for (int b = 0; b < rs->nbands; b++) {
            band_t *bd = &rs->bands[b];
            bd->level = lv;
            if (r == 0) {
                bd->orient = 0;
                bd->x0 = 0; bd->y0 = 0; bd->x1 = rs->x1; bd->y1 = rs->y1;
                bd->offx = 0; bd->offy = 0;
            } else {
                bd->orient = b + 1;
                int xb = bd->orient & 1, yb = bd->orient >> 1;
                /* T.800 B-15: band bounds from the tile-component bounds */
                bd->x0 = ceildivpow2(0 - (xb << lv), lv + 1) < 0 ? 0 : ceildivpow2(0 - (xb << lv), lv + 1);
                bd->y0 = ceildivpow2(0 - (yb << lv), lv + 1) < 0 ? 0 : ceildivpow2(0 - (yb << lv), lv + 1);
                bd->x1 = ceildivpow2(W - (xb << lv), lv + 1);
                bd->y1 = ceildivpow2(H - (yb << lv), lv + 1);
                bd->offx = xb ? t->res[r - 1].x1 : 0;
                bd->offy = yb ? t->res[r - 1].y1 : 0;
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
static int mqe_numbytes(const mq_t *m) { return (int) (m->wp - m->start) + 1 - 1 + 0 + (m->wp >= m->start ? 0 : 0) + 0 + 1 - 1; }
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
static void t1_decode_cblk(t1_t *t, const cblk_t *cb, int orient)
{
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

/* x: interleaved samples (even = low); sn lows, dn highs (cas 0: the first sample is a low) */
static void idwt97_line(float *x, int sn, int dn)
{
    int n = sn + dn;
    if (!(dn > 0 || sn > 1)) return;
    for (int i = 0; i < sn; i++) x[2 * i] = x[2 * i] * DWT_K;
    for (int i = 0; i < dn; i++) x[2 * i + 1] = x[2 * i + 1] * DWT_TWO_INVK;
    (void) n;
    /* l[i] += c*(h[i-1] + h[i]) with symmetric extension; c = -delta */
    {
        float c = -DWT_DELTA;
        int m = sn < dn ? sn : dn;                       /* min(sn, dn - a), a = 0 */
        for (int i = 0; i < m; i++) {
            float hl = i == 0 ? x[1] : x[2 * i - 1];
            x[2 * i] = x[2 * i] + ((hl + x[2 * i + 1]) * c);
        }
        if (m < sn) { float c2 = c + c; x[2 * m] = x[2 * m] + x[2 * m - 1] * c2; }
    }
    /* h[i] += c*(l[i] + l[i+1]); c = -gamma */
    {
        float c = -DWT_GAMMA;
        int m = dn < sn - 1 ? dn : sn - 1;               /* min(dn, sn - b), b = 1 */
        for (int i = 0; i < m; i++) x[2 * i + 1] = x[2 * i + 1] + ((x[2 * i] + x[2 * i + 2]) * c);
        if (m < dn) { float c2 = c + c; x[2 * m + 1] = x[2 * m + 1] + x[2 * m] * c2; }
    }
    {
        float c = -DWT_BETA;
        int m = sn < dn ? sn : dn;
        for (int i = 0; i < m; i++) {
            float hl = i == 0 ? x[1] : x[2 * i - 1];
            x[2 * i] = x[2 * i] + ((hl + x[2 * i + 1]) * c);
        }
        if (m < sn) { float c2 = c + c; x[2 * m] = x[2 * m] + x[2 * m - 1] * c2; }
    }
    {
        float c = -DWT_ALPHA;
        int m = dn < sn - 1 ? dn : sn - 1;
        for (int i = 0; i < m; i++) x[2 * i + 1] = x[2 * i + 1] + ((x[2 * i] + x[2 * i + 2]) * c);
        if (m < dn) { float c2 = c + c; x[2 * m + 1] = x[2 * m + 1] + x[2 * m] * c2; }
    }
}

static void idwt97_tile(float *buf, const tile_t *t)
{
    int W = t->W;
    float *line = (float *) malloc((size_t) (t->W > t->H ? t->W : t->H) * sizeof(float) + 64);
    for (int r = 1; r < J2K_NRES; r++) {
        int rw = t->res[r].x1, rh = t->res[r].y1;
        int sn = t->res[r - 1].x1, dn = rw - sn;
        /* horizontal pass first (opj_dwt_decode_tile_97) */
        for (int y = 0; y < rh; y++) {
            float *row = buf + (size_t) y * W;
            for (int i = 0; i < sn; i++) line[2 * i] = row[i];
            for (int i = 0; i < dn; i++) line[2 * i + 1] = row[sn + i];
            idwt97_line(line, sn, dn);
            memcpy(row, line, (size_t) rw * sizeof(float));
        }
        sn = t->res[r - 1].y1; dn = rh - sn;
        for (int x = 0; x < rw; x++) {
            for (int i = 0; i < sn; i++) line[2 * i] = buf[(size_t) i * W + x];
            for (int i = 0; i < dn; i++) line[2 * i + 1] = buf[(size_t) (sn + i) * W + x];
            idwt97_line(line, sn, dn);
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

size_t orc_j2k_decode(const uint8_t *cs, size_t n, int32_t **samples, size_t *height, size_t *width)
{
    if (n < 4 || be16(cs) != 0xFF4F) return 0;
    size_t pos = 2;
    int W = 0, H = 0, prec = 16, guard = 2, nres = 0, qsty = 0;
    int expn[3 * J2K_NRES], mant[3 * J2K_NRES];
    memset(expn, 0, sizeof expn); memset(mant, 0, sizeof mant);
    const uint8_t *tile_data = NULL, *tile_end = NULL;
    while (pos + 4 <= n) {
        uint32_t mk = be16(cs + pos), len = be16(cs + pos + 2);
        const uint8_t *p = cs + pos + 4;
        if (mk == 0xFF51) {                         /* SIZ, A.5.1 */
            W = (int) (be32(p + 2) - be32(p + 10));
            H = (int) (be32(p + 6) - be32(p + 14));
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
        } else if (mk == 0xFF90) {                  /* SOT, A.4.2 */
            uint32_t psot = be32(p + 2);
            const uint8_t *sot = cs + pos;
            pos += 2 + len;
            if (be16(cs + pos) != 0xFF93) return 0; /* SOD */
            tile_data = cs + pos + 2;
            tile_end = psot ? sot + psot : cs + n - 2;
            break;
        }
        pos += 2 + len;
    }
    if (!tile_data || nres != J2K_NRES || qsty != 2 || W <= 0 || H <= 0) return 0;

    tile_t t;
    tile_init(&t, W, H);
    for (int r = 0; r < J2K_NRES; r++)
        for (int b = 0; b < t.res[r].nbands; b++) {
            int bi = r == 0 ? 0 : 3 * (r - 1) + b + 1;
            band_set_quant(&t.res[r].bands[b], expn[bi], mant[bi], prec, guard);
        }

    float *buf = (float *) calloc((size_t) W * H, sizeof(float));
    const uint8_t *p = tile_data;
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
    int32_t *out = (int32_t *) malloc((size_t) W * H * sizeof(int32_t));
    const int64_t shift = (int64_t) 1 << (prec - 1), vmax = ((int64_t) 1 << prec) - 1;
    for (size_t i = 0; i < (size_t) W * H; i++) {
        int64_t v = (int64_t) lrintf(buf[i]) + shift;
        out[i] = (int32_t) (v < 0 ? 0 : (v > vmax ? vmax : v));
    }
    free(buf);
    tile_free(&t);
    *samples = out;
    if (height) *height = (size_t) H;
    if (width) *width = (size_t) W;
    return (size_t) W * H;
}

size_t orc_j2k_encode(const uint16_t *img, size_t height, size_t width, float base_cr, uint8_t **out)
{
    (void) img; (void) height; (void) width; (void) base_cr; (void) out;
    (void) mqe_init; (void) mqe_encode; (void) mqe_flush; (void) mqe_numbytes;
    return 0;
}
