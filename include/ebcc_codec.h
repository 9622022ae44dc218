/*
 * ebcc_codec.h - public C API of libh5z_ebcc.so (MI355X build).
 *
 * Drop-in for /root/reference/src/ebcc_codec.h:15-49: same symbol names, argument meaning, struct
 * layout (codec_config_t is 64 bytes on LP64), return conventions (encoders return bytes, decoders
 * return the number of floats, 0 = error) and ownership (outputs are system-malloc'd, release with
 * free_buffer; HDF5 itself frees filter output with free()).
 */
#ifndef EBCC_CODEC_H
#define EBCC_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifndef EBCC_API
#define EBCC_API __attribute__((visibility("default")))
#endif

#define NDIMS 3
#define EBCC_MIN_INTERNAL_IMAGE_DIM 32
#define EBCC_MAX_INTERNAL_IMAGE_DIM 2047
#define EBCC_VERSION_MAJOR 0
#define EBCC_VERSION_MINOR 1
#define EBCC_HEADER_VERSION 1
#define EBCC_HEADER_FLAG_CONST_FIELD 0x01
#define EBCC_HEADER_MAGIC "EBCC"
#define EBCC_CHUNKING_HEADER_VERSION 1
#define EBCC_CHUNKING_HEADER_MAGIC "EBCK"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { NONE, MAX_ERROR, RELATIVE_ERROR } residual_t;

typedef struct {
    size_t dims[NDIMS];                     /* (frames, height, width); frames*height is the J2K image height */
    float base_cr;                          /* initial JPEG 2000 compression ratio */
    residual_t residual_compression_type;
    float residual_cr;                      /* unused, kept for layout */
    float error;                            /* max abs error, or fraction of the data range */
    size_t chunk_dims[NDIMS];               /* all zero = one chunk */
} codec_config_t;

EBCC_API size_t ebcc_encode(float *data, codec_config_t *config, uint8_t **out_buffer);
EBCC_API size_t ebcc_decode(uint8_t *data, size_t data_size, float **out_buffer);
EBCC_API size_t ebcc_encode_chunking(float *data, codec_config_t *config, uint8_t **out_buffer);
EBCC_API size_t ebcc_encode_chunking_compat(float *data, codec_config_t *config, uint8_t **out_buffer);
EBCC_API size_t ebcc_decode_chunking(uint8_t *data, size_t data_size, float **out_buffer);
EBCC_API void free_buffer(void *buffer);

EBCC_API void print_config(codec_config_t *config);
EBCC_API void log_set_level_from_env(void);

/* HDF5 filter glue, /root/reference/src/h5z_ebcc.c:27-28,38 */
EBCC_API void populate_config(codec_config_t *config, size_t cd_nelmts, const unsigned int cd_values[], size_t buf_size);
EBCC_API int H5PLget_plugin_type(void);
EBCC_API const void *H5PLget_plugin_info(void);

#ifdef __cplusplus
}
#endif
#endif /* EBCC_CODEC_H */
