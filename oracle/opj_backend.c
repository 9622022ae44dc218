/*
 * opj_backend.c - drives the image's OpenJPEG 2.4.0 exactly the way the reference does
 * (/root/reference/src/ebcc_codec.c:105-180 encode, :1092-1136 decode).  TEST INFRASTRUCTURE ONLY.
 * OpenJPEG is the reference's own (un-vendored) dependency; this is how golden J2K vectors are
 * produced and how the frame-codec restatement is pinned.  The library is dlopen'd so that
 * libebcc_oracle.so still loads on a machine without it (backend 1 then reports failure).
 */
#ifdef ORC_HAVE_OPENJPEG
#define _GNU_SOURCE
#include "oracle.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <openjpeg.h>

typedef struct { uint8_t *buf; size_t cap, len, off; } mem_t;

static OPJ_SIZE_T mem_write(void *src, OPJ_SIZE_T n, void *ud)
{
    mem_t *m = (mem_t *) ud;
    while (m->off + n > m->cap) { m->cap = m->cap ? m->cap * 2 : 4096; m->buf = (uint8_t *) realloc(m->buf, m->cap); }
    memcpy(m->buf + m->off, src, n);
    m->off += n;
    if (m->off > m->len) m->len = m->off;
    return n;
}
static OPJ_SIZE_T mem_read(void *dst, OPJ_SIZE_T n, void *ud)
{
    mem_t *m = (mem_t *) ud;
    if (m->off >= m->len) return (OPJ_SIZE_T) -1;
    size_t k = n < m->len - m->off ? n : m->len - m->off;
    memcpy(dst, m->buf + m->off, k);
    m->off += k;
    return k;
}

static struct {
    void *h;
    void (*set_default_encoder_parameters)(opj_cparameters_t *);
    opj_image_t *(*image_create)(OPJ_UINT32, opj_image_cmptparm_t *, OPJ_COLOR_SPACE);
    opj_image_t *(*image_tile_create)(OPJ_UINT32, opj_image_cmptparm_t *, OPJ_COLOR_SPACE);
    OPJ_BOOL (*write_tile)(opj_codec_t *, OPJ_UINT32, OPJ_BYTE *, OPJ_UINT32, opj_stream_t *);
    opj_codec_t *(*create_compress)(OPJ_CODEC_FORMAT);
    OPJ_BOOL (*setup_encoder)(opj_codec_t *, opj_cparameters_t *, opj_image_t *);
    opj_stream_t *(*stream_default_create)(OPJ_BOOL);
    void (*stream_set_user_data)(opj_stream_t *, void *, opj_stream_free_user_data_fn);
    void (*stream_set_user_data_length)(opj_stream_t *, OPJ_UINT64);
    void (*stream_set_write_function)(opj_stream_t *, opj_stream_write_fn);
    void (*stream_set_read_function)(opj_stream_t *, opj_stream_read_fn);
    OPJ_BOOL (*start_compress)(opj_codec_t *, opj_image_t *, opj_stream_t *);
    OPJ_BOOL (*encode)(opj_codec_t *, opj_stream_t *);
    OPJ_BOOL (*end_compress)(opj_codec_t *, opj_stream_t *);
    void (*stream_destroy)(opj_stream_t *);
    void (*image_destroy)(opj_image_t *);
    void (*destroy_codec)(opj_codec_t *);
    void (*set_default_decoder_parameters)(opj_dparameters_t *);
    opj_codec_t *(*create_decompress)(OPJ_CODEC_FORMAT);
    OPJ_BOOL (*setup_decoder)(opj_codec_t *, opj_dparameters_t *);
    OPJ_BOOL (*read_header)(opj_stream_t *, opj_codec_t *, opj_image_t **);
    OPJ_BOOL (*decode)(opj_codec_t *, opj_stream_t *, opj_image_t *);
    OPJ_BOOL (*end_decompress)(opj_codec_t *, opj_stream_t *);
    const char *(*version)(void);
} J;

static int opj_load(void)
{
    if (J.h) return 1;
    const char *names[] = { "/opt/conda/lib/libopenjp2.so.7", "libopenjp2.so.7", NULL };
    for (int i = 0; names[i] && !J.h; i++) J.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!J.h) { fprintf(stderr, "oracle: libopenjp2.so.7 not found\n"); return 0; }
#define L(f) *(void **) &J.f = dlsym(J.h, "opj_" #f)
    L(set_default_encoder_parameters); L(image_create); L(image_tile_create); L(write_tile); L(create_compress); L(setup_encoder);
    L(stream_default_create); L(stream_set_user_data); L(stream_set_user_data_length);
    L(stream_set_write_function); L(stream_set_read_function); L(start_compress); L(encode);
    L(end_compress); L(stream_destroy); L(image_destroy); L(destroy_codec);
    L(set_default_decoder_parameters); L(create_decompress); L(setup_decoder); L(read_header);
    L(decode); L(end_decompress); L(version);
#undef L
    return 1;
}

const char *orc_opj_version(void) { return opj_load() ? J.version() : ""; }

/* `tiles` frames of height x width stacked along y; more than one = a tiled image written tile by tile (:121-124,
 * :145-147, :167-171) */
size_t orc_opj_encode_tiled(const uint16_t *img, size_t tiles, size_t height, size_t width, float base_cr, uint8_t **out)
{
    if (!opj_load()) return 0;
    opj_cparameters_t p;
    J.set_default_encoder_parameters(&p);
    p.tcp_numlayers = 1;
    p.cp_disto_alloc = 1;
    p.tcp_rates[0] = base_cr / 2;                    /* ebcc_codec.c:116 */
    p.irreversible = 1;
    p.cp_tx0 = 0; p.cp_ty0 = 0;
    if (tiles > 1) { p.tile_size_on = OPJ_TRUE; p.cp_tdx = (int) width; p.cp_tdy = (int) height; }
    const size_t rows = tiles * height;
    opj_image_cmptparm_t c; memset(&c, 0, sizeof c);
    c.dx = 1; c.dy = 1; c.w = (OPJ_UINT32) width; c.h = (OPJ_UINT32) rows; c.prec = 16; c.sgnd = 0;
    opj_image_t *im;
    if (tiles == 1) {
        im = J.image_create(1, &c, OPJ_CLRSPC_GRAY);
        for (size_t i = 0; i < rows * width; i++) im->comps[0].data[i] = img[i];
    } else {
        im = J.image_tile_create(1, &c, OPJ_CLRSPC_GRAY);
    }
    im->x0 = 0; im->y0 = 0; im->x1 = (OPJ_UINT32) width; im->y1 = (OPJ_UINT32) rows;
    opj_codec_t *cd = J.create_compress(OPJ_CODEC_J2K);
    if (!J.setup_encoder(cd, &p, im)) { J.image_destroy(im); J.destroy_codec(cd); return 0; }
    opj_stream_t *st = J.stream_default_create(OPJ_FALSE);
    mem_t m = { 0 };
    J.stream_set_user_data(st, &m, NULL);
    J.stream_set_user_data_length(st, 0);
    J.stream_set_write_function(st, mem_write);
    J.start_compress(cd, im, st);
    if (tiles > 1) {
        for (size_t k = 0; k < tiles; k++)
            J.write_tile(cd, (OPJ_UINT32) k, (OPJ_BYTE *) (img + k * height * width), (OPJ_UINT32) (height * width * sizeof(uint16_t)), st);
    } else {
        J.encode(cd, st);
    }
    J.end_compress(cd, st);
    J.stream_destroy(st); J.image_destroy(im); J.destroy_codec(cd);
    *out = m.buf;
    return m.len;
}

size_t orc_opj_encode(const uint16_t *img, size_t height, size_t width, float base_cr, uint8_t **out)
{
    return orc_opj_encode_tiled(img, 1, height, width, base_cr, out);
}

size_t orc_opj_decode(const uint8_t *cs, size_t n, int32_t **samples, size_t *h, size_t *w)
{
    if (!opj_load()) return 0;
    mem_t m = { (uint8_t *) cs, n, n, 0 };
    opj_stream_t *st = J.stream_default_create(OPJ_TRUE);
    J.stream_set_user_data(st, &m, NULL);
    J.stream_set_user_data_length(st, n);
    J.stream_set_read_function(st, mem_read);
    opj_dparameters_t dp;
    J.set_default_decoder_parameters(&dp);
    dp.decod_format = 0; dp.cp_layer = 0; dp.cp_reduce = 0;
    opj_codec_t *cd = J.create_decompress(OPJ_CODEC_J2K);
    J.setup_decoder(cd, &dp);
    opj_image_t *im = NULL;
    if (!J.read_header(st, cd, &im)) { J.stream_destroy(st); J.destroy_codec(cd); return 0; }
    J.decode(cd, st, im);
    J.end_decompress(cd, st);
    size_t ww = im->x1 - im->x0, hh = im->y1 - im->y0, npx = ww * hh;
    *samples = (int32_t *) malloc(npx * sizeof(int32_t));
    memcpy(*samples, im->comps[0].data, npx * sizeof(int32_t));
    if (h) *h = hh;
    if (w) *w = ww;
    J.stream_destroy(st); J.destroy_codec(cd); J.image_destroy(im);
    return npx;
}
#endif
