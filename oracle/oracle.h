/*
 * oracle.h - CPU restatement of the EBCC per-frame codec path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load libebcc_oracle.so.  The shipped library
 * (ebcc_amd/libh5z_ebcc.so) never links, includes or dlopens anything from this directory.
 *
 * Parity status
 *   - residual layer (padding, DC removal, CDF 9/7 lifting, truncation to integers, SPIHT
 *     list coder, MSB-first bit I/O, "IMS" header): PINNED bit-exactly against the reference's own
 *     C sources compiled unmodified (oracle/_ref/libspiht_ref.so, recipe in oracle/Makefile) and
 *     against the committed fixtures in tests/golden/ that were generated from that build.
 *   - frame codec orchestration (scaling, residual, both rate searches, truncation search,
 *     header, EBCK container): PINNED against oracle/_ref/libh5z_ebcc_ref.so (reference sources +
 *     the image's OpenJPEG 2.4.0 / zstd 1.4.9) and committed stream fixtures.
 *   - JPEG 2000 base layer: OpenJPEG is an un-vendored submodule of the reference (no pin visible);
 *     parity is defined against OpenJPEG 2.4.0 (/opt/conda/lib/libopenjp2.so.2.4.0 of this
 *     image).  See j2k_oracle.c for what is restated and what is pinned.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef EBCC_ORACLE_H
#define EBCC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- residual layer (src/spiht/{dwt.h,spiht_re.c,ml.h,bitio.h}) ---- */

/* geometry of the padded transform grid, src/spiht/dwt.h:41-59 */
typedef struct {
    uint32_t size_x, size_y;     /* image width / height                */
    uint32_t extra_x, extra_y;   /* padding to a multiple of 2^(stages+1) */
    uint32_t stride;             /* size_x + extra_x                    */
    uint32_t stages;
} orc_grid_t;

void orc_grid_init(orc_grid_t *g, size_t height, size_t width, size_t stages);

/* load_image + sub_dc + dwt2full + normalize  (src/spiht/spiht_re.c:435,461,466,467).
 * coeffs must hold (size_x+extra_x)*(size_y+extra_y) floats; returns the DC byte. */
int orc_spiht_analysis(const float *image, size_t height, size_t width, size_t stages, float *coeffs);

/* idwt2full + add_dc + crop + /255  (src/spiht/spiht_re.c:508-516); coeffs is destroyed. */
void orc_spiht_synthesis(float *coeffs, size_t height, size_t width, size_t stages, int dc, float *image_out);

/* spiht_encode, src/spiht/spiht_re.c:432-475.  *out is malloc'd. */
void orc_spiht_encode(const float *image, size_t height, size_t width, uint8_t **out, size_t *out_size,
                      size_t trunc_bits, size_t stages);

/* spiht_decode, src/spiht/spiht_re.c:477-520 */
void orc_spiht_decode(const uint8_t *in, size_t in_size, float *image_out, size_t height, size_t width,
                      size_t num_bits);

/* decode only the coefficient grid (no synthesis); returns dc, fills geometry. coeffs malloc'd. */
int orc_spiht_decode_coeffs(const uint8_t *in, size_t in_size, size_t num_bits, orc_grid_t *g, float **coeffs);

/* ---- frame codec (src/ebcc_codec.c) ---- */

#define ORC_NDIMS 3
typedef enum { ORC_NONE = 0, ORC_MAX_ERROR = 1, ORC_RELATIVE_ERROR = 2 } orc_residual_t;

/* same layout as codec_config_t, src/ebcc_codec.h:32-39 */
typedef struct {
    size_t dims[ORC_NDIMS];
    float base_cr;
    int residual_compression_type;
    float residual_cr;
    float error;
    size_t chunk_dims[ORC_NDIMS];
} orc_config_t;

/* J2K back-end used by the frame codec restatement.
 *   0 = j2k_oracle.c restatement (default)
 *   1 = OpenJPEG 2.4.0 loaded with dlopen("libopenjp2.so.7") (the reference's own dependency) */
void orc_set_j2k_backend(int backend);
int  orc_get_j2k_backend(void);

size_t orc_ebcc_encode(const float *data, const orc_config_t *config, uint8_t **out);
size_t orc_ebcc_decode(const uint8_t *data, size_t data_size, float **out);
size_t orc_ebcc_encode_chunking(const float *data, const orc_config_t *config, uint8_t **out);
size_t orc_ebcc_encode_chunking_compat(const float *data, const orc_config_t *config, uint8_t **out);
size_t orc_ebcc_decode_chunking(const uint8_t *data, size_t data_size, float **out);
void   orc_free(void *p);

/* trace of the last orc_ebcc_encode call (probe counts etc.) for test diagnostics */
typedef struct {
    int n_j2k_encodes, n_j2k_decodes, n_spiht_decodes;
    float final_cr;
    size_t coeffs_size, compressed_size, tail_size;
} orc_trace_t;
void orc_last_trace(orc_trace_t *t);

/* ---- JPEG 2000 base layer (j2k_oracle.c) ---- */

/* Encode a single-component 16-bit unsigned image exactly the way the reference configures
 * OpenJPEG (src/ebcc_codec.c:105-180): irreversible 9/7, 1 layer, rate = base_cr/2, 6 resolutions,
 * 64x64 code-blocks, LRCP, one tile.  *out malloc'd.  Returns bytes or 0. */
size_t orc_j2k_encode(const uint16_t *img, size_t height, size_t width, float base_cr, uint8_t **out);
/* `tiles` frames of height x width stacked along y, one tile each (ebcc_codec.c:105-180 with n_tiles > 1) */
size_t orc_j2k_encode_tiled(const uint16_t *img, size_t tiles, size_t height, size_t width, float base_cr, uint8_t **out);

/* Decode a codestream to int32 samples as OpenJPEG's opj_decode does (src/ebcc_codec.c:1092-1136
 * reads image->comps[0].data).  samples malloc'd. Returns number of pixels or 0. */
/* test hook: called with every code-block's quantised coefficients during orc_j2k_encode / orc_j2k_analysis */
void orc_j2k_set_block_sink(void (*fn)(const int32_t *q, int w, int h, int orient, void *user), void *user);
size_t orc_j2k_decode(const uint8_t *cs, size_t cs_size, int32_t **samples, size_t *height, size_t *width);

#ifdef __cplusplus
}
#endif
#endif
