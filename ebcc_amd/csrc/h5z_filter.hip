// h5z_filter.hip - the HDF5 dynamically loaded filter of /root/reference/src/h5z_ebcc.c (id 308) on top of the C API:
// cd_values -> codec_config_t (populate_config, exported: the Zarr codec calls it), the filter callback, the plugin entry
// points.  Same argument meaning, return conventions and exit(1) contracts as the reference.
#include "host.hpp"

using namespace ebcc;

extern "C" {

// ---- HDF5 filter plugin, /root/reference/src/h5z_ebcc.c ---------------------------------------------
#define H5Z_FLAG_REVERSE 0x0100
typedef size_t (*H5Z_func_t)(unsigned int, size_t, const unsigned int[], size_t, size_t *, void **);
struct H5Z_class2_t {
    int version; int id; unsigned encoder_present; unsigned decoder_present; const char *name;
    void *can_apply; void *set_local; H5Z_func_t filter;
};

void populate_config(codec_config_t *config, size_t cd_nelmts, const unsigned int cd_values[], size_t buf_size)
{
    // h5z_ebcc.c:38-93, including exit(1) on invalid parameters
    if (cd_nelmts < 4) { log_fatal("EBCC filter requires at least 4 configuration values, got %lu", cd_nelmts); exit(1); }
    for (int i = 0; i < NDIMS; i++) config->chunk_dims[i] = 0;
    size_t th = cd_values[0], tw = cd_values[1];
    if (th < EBCC_MIN_INTERNAL_IMAGE_DIM || tw < EBCC_MIN_INTERNAL_IMAGE_DIM || th > EBCC_MAX_INTERNAL_IMAGE_DIM ||
        tw > EBCC_MAX_INTERNAL_IMAGE_DIM) {
        log_fatal("Tile size %lu x %lu is invalid, each dimension must be between %d and %d", th, tw,
                  EBCC_MIN_INTERNAL_IMAGE_DIM, EBCC_MAX_INTERNAL_IMAGE_DIM);
        exit(1);
    }
    size_t tile = th * tw;
    config->dims[0] = buf_size / sizeof(float);
    if (config->dims[0] < tile) { log_fatal("Buffer size %lu is smaller than the tile size %lu x %lu = %lu", config->dims[0], th, tw, tile); exit(1); }
    if (config->dims[0] % tile != 0) { log_fatal("Buffer size %lu is not divisible by the tile size %lu x %lu = %lu", config->dims[0], th, tw, tile); exit(1); }
    for (size_t i = 0; i < 2; i++) {
        size_t cur = cd_values[i];
        config->dims[0] /= cur;
        config->dims[i + 1] = cur;
    }
    if (config->dims[1] != 0 && config->dims[0] > EBCC_MAX_INTERNAL_IMAGE_DIM / config->dims[1]) {
        log_fatal("Flattened EBCC image height %lu x %lu exceeds the limit of %d", config->dims[0], config->dims[1],
                  EBCC_MAX_INTERNAL_IMAGE_DIM);
        exit(1);
    }
    config->base_cr = u2f(cd_values[2]);
    config->residual_compression_type = (residual_t) cd_values[3];
    if (config->residual_compression_type == MAX_ERROR || config->residual_compression_type == RELATIVE_ERROR) {
        if (cd_nelmts != 5) { log_fatal("EBCC filter: modes 1 and 2 need 5 configuration values"); exit(1); }
        config->error = u2f(cd_values[4]);
    }
}

static size_t H5Z_filter_ebcc(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes,
                              size_t *buf_size, void **buf)
{
    if (flags & H5Z_FLAG_REVERSE) {
        float *out = nullptr;
        *buf_size = ebcc_decode((uint8_t *) *buf, nbytes, &out);                               // element count (quirk Q1)
        free_buffer(*buf);
        *buf = out;
        return *buf_size;
    }
    codec_config_t config;
    memset(&config, 0, sizeof config);
    populate_config(&config, cd_nelmts, cd_values, *buf_size);
    uint8_t *out = nullptr;
    *buf_size = ebcc_encode((float *) *buf, &config, &out);
    free_buffer(*buf);
    *buf = out;
    return *buf_size;
}

static const H5Z_class2_t H5Z_EBCC[1] = {{1, 308, 1, 1, "HDF5 EBCC filter L&L", nullptr, nullptr, H5Z_filter_ebcc}};

int H5PLget_plugin_type(void) { return 0; }            // H5PL_TYPE_FILTER
const void *H5PLget_plugin_info(void) { return H5Z_EBCC; }

}  // extern "C"

