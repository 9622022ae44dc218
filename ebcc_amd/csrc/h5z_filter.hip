// h5z_filter.hip - the HDF5 dynamically loaded filter of /root/reference/src/h5z_ebcc.c (id 308) on top of the C API:
// cd_values -> codec_config_t (populate_config, exported: the Zarr codec calls it), the filter callback, the plugin entry
// points.  Same argument meaning, return conventions and exit(1) contracts as the reference.
#include <link.h>

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

// ---- Direct-chunk batch helpers for C callers (the Python form is ebcc_amd/h5_batch.py).
// HDF5 calls the filter above once per chunk, under its global lock: a frame coded alone pays the latency of the serial
// tier-1 kernels.  A dataset with one frame per chunk can instead be written and read in DEVICE BATCHES: the frames are
// coded together and stored as pre-filtered chunks (H5Dwrite_chunk), or the raw chunks are fetched (H5Dread_chunk) and
// decoded together.  The file is the same ordinary EBCC-filtered dataset either way (chunk bytes identical to what the
// callback writes).  HDF5 is not linked: its functions are taken from the libhdf5 the calling process has already loaded.
extern "C++" {
namespace {
typedef long long hid_like;                 // hid_t of HDF5 >= 1.10 (int64_t)
typedef unsigned long long hsize_like;      // hsize_t
struct H5Api {
    hid_like (*Dget_space)(hid_like) = nullptr;
    int (*Sget_simple_extent_ndims)(hid_like) = nullptr;
    int (*Sget_simple_extent_dims)(hid_like, hsize_like *, hsize_like *) = nullptr;
    int (*Sclose)(hid_like) = nullptr;
    hid_like (*Dget_create_plist)(hid_like) = nullptr;
    int (*Pget_chunk)(hid_like, int, hsize_like *) = nullptr;
    int (*Pget_filter_by_id2)(hid_like, int, unsigned int *, size_t *, unsigned int *, size_t, char *, unsigned int *) = nullptr;
    int (*Pclose)(hid_like) = nullptr;
    int (*Dwrite_chunk)(hid_like, hid_like, unsigned int, const hsize_like *, size_t, const void *) = nullptr;
    int (*Dread_chunk)(hid_like, hid_like, const hsize_like *, unsigned int *, void *) = nullptr;
    int (*Dget_chunk_storage_size)(hid_like, const hsize_like *, hsize_like *) = nullptr;
    bool ok = false;
    H5Api()
    {
        // the HDF5 library of this process: a Python extension loads it privately (RTLD_LOCAL), so the global scope may not
        // show it - look through the loaded objects for it and take a handle to that very instance
        struct Find { std::string path; } found;
        dl_iterate_phdr([](struct dl_phdr_info *info, size_t, void *data) -> int {
            const char *name = info->dlpi_name ? info->dlpi_name : "";
            const char *base = strrchr(name, '/');
            base = base ? base + 1 : name;
            if (strncmp(base, "libhdf5", 7) == 0 && !strstr(base, "_hl") && !strstr(base, "_cpp") && !strstr(base, "_fortran")) {
                static_cast<Find *>(data)->path = name;
                return 1;
            }
            return 0;
        }, &found);
        void *h = found.path.empty() ? nullptr : dlopen(found.path.c_str(), RTLD_NOW | RTLD_NOLOAD);
        auto sym = [&](const char *n) { void *p = h ? dlsym(h, n) : nullptr; return p ? p : dlsym(RTLD_DEFAULT, n); };
        Dget_space = (decltype(Dget_space)) sym("H5Dget_space");
        Sget_simple_extent_ndims = (decltype(Sget_simple_extent_ndims)) sym("H5Sget_simple_extent_ndims");
        Sget_simple_extent_dims = (decltype(Sget_simple_extent_dims)) sym("H5Sget_simple_extent_dims");
        Sclose = (decltype(Sclose)) sym("H5Sclose");
        Dget_create_plist = (decltype(Dget_create_plist)) sym("H5Dget_create_plist");
        Pget_chunk = (decltype(Pget_chunk)) sym("H5Pget_chunk");
        Pget_filter_by_id2 = (decltype(Pget_filter_by_id2)) sym("H5Pget_filter_by_id2");
        Pclose = (decltype(Pclose)) sym("H5Pclose");
        Dwrite_chunk = (decltype(Dwrite_chunk)) sym("H5Dwrite_chunk");
        Dread_chunk = (decltype(Dread_chunk)) sym("H5Dread_chunk");
        Dget_chunk_storage_size = (decltype(Dget_chunk_storage_size)) sym("H5Dget_chunk_storage_size");
        ok = Dget_space && Sget_simple_extent_ndims && Sget_simple_extent_dims && Sclose && Dget_create_plist && Pget_chunk &&
             Pget_filter_by_id2 && Pclose && Dwrite_chunk && Dread_chunk && Dget_chunk_storage_size;
    }
};
H5Api &h5() { static H5Api a; return a; }

// the dataset's geometry and EBCC filter parameters: rank, dims, frames (product of the leading dims), cd_values
struct FrameDataset {
    int rank = 0;
    hsize_like dims[32] = {0};
    size_t frames = 0, H = 0, W = 0;
    unsigned int cd[8] = {0};
    size_t cd_n = 0;
};
bool open_frame_dataset(hid_like dset, FrameDataset &d, const char *who)
{
    H5Api &a = h5();
    if (!a.ok) { log_fatal("%s: no HDF5 library with H5Dwrite_chunk / H5Dread_chunk is loaded in this process", who); return false; }
    const hid_like sp = a.Dget_space(dset);
    if (sp < 0) { log_fatal("%s: not a dataset", who); return false; }
    d.rank = a.Sget_simple_extent_ndims(sp);
    const bool dims_ok = d.rank >= 2 && d.rank <= 32 && a.Sget_simple_extent_dims(sp, d.dims, nullptr) == d.rank;
    a.Sclose(sp);
    if (!dims_ok) { log_fatal("%s: the dataset needs at least two dimensions", who); return false; }
    d.H = (size_t) d.dims[d.rank - 2]; d.W = (size_t) d.dims[d.rank - 1];
    d.frames = 1;
    for (int i = 0; i + 2 < d.rank; i++) d.frames *= (size_t) d.dims[i];
    const hid_like pl = a.Dget_create_plist(dset);
    if (pl < 0) { log_fatal("%s: no creation property list", who); return false; }
    hsize_like chunk[32];
    bool ok = a.Pget_chunk(pl, d.rank, chunk) == d.rank;
    for (int i = 0; ok && i < d.rank; i++) ok = chunk[i] == (i + 2 < d.rank ? 1ull : d.dims[i]);
    unsigned int flags = 0, fcfg = 0;
    d.cd_n = 8;
    char name[8];
    const bool filt = ok && a.Pget_filter_by_id2(pl, 308, &flags, &d.cd_n, d.cd, sizeof name, name, &fcfg) >= 0;
    a.Pclose(pl);
    if (!ok) { log_fatal("%s: the dataset's chunks must be single frames (1, ..., 1, %zu, %zu)", who, d.H, d.W); return false; }
    if (!filt || d.cd_n < 4 || d.cd_n > 5 || d.cd[0] != d.H || d.cd[1] != d.W) { log_fatal("%s: the dataset does not carry filter 308 for %zu x %zu frames", who, d.H, d.W); return false; }
    return true;
}
void chunk_offset(const FrameDataset &d, size_t frame, hsize_like *off)
{
    for (int i = d.rank - 3; i >= 0; i--) { off[i] = frame % d.dims[i]; frame /= d.dims[i]; }
    off[d.rank - 2] = 0; off[d.rank - 1] = 0;
}
}  // namespace
}  // extern "C++"

// frames [first_frame, first_frame + n_frames) of the dataset (counted in C order over its leading dimensions) from
// `frames` (host, n_frames x H x W fp32), coded with the dataset's own filter-308 parameters.  0 = ok.
int ebcc_h5_write_frames(long long dset_id, size_t first_frame, size_t n_frames, const float *frames)
{
    log_set_level_from_env();
    FrameDataset d;
    if (!frames || n_frames < 1 || !open_frame_dataset(dset_id, d, "ebcc_h5_write_frames")) return 1;
    if (first_frame + n_frames > d.frames) { log_fatal("ebcc_h5_write_frames: frames %zu .. %zu of %zu", first_frame, first_frame + n_frames, d.frames); return 1; }
    codec_config_t cfg;
    populate_config(&cfg, d.cd_n, d.cd, d.H * d.W * sizeof(float));                 // (one frame per chunk)
    std::vector<uint8_t *> outs(n_frames, nullptr);
    std::vector<size_t> sizes(n_frames, 0);
    const int rc = cached_encode_host_frames(frames, n_frames, (int) d.H, (int) d.W, &cfg, outs.data(), sizes.data());
    if (rc == 2) exit(1);                                                           // NaN / Inf: as the filter callback (check_nan_inf)
    int bad = rc;
    hsize_like off[32];
    for (size_t i = 0; i < n_frames && !bad; i++) {
        chunk_offset(d, first_frame + i, off);
        if (h5().Dwrite_chunk(dset_id, 0 /* H5P_DEFAULT */, 0, off, sizes[i], outs[i]) < 0) { log_fatal("ebcc_h5_write_frames: H5Dwrite_chunk failed for frame %zu", first_frame + i); bad = 1; }
    }
    for (uint8_t *p : outs) free(p);
    return bad ? 1 : 0;
}

// the same frames read back: raw chunks fetched with H5Dread_chunk, decoded as device batches into `frames_out` (host).  0 = ok.
int ebcc_h5_read_frames(long long dset_id, size_t first_frame, size_t n_frames, float *frames_out)
{
    log_set_level_from_env();
    FrameDataset d;
    if (!frames_out || n_frames < 1 || !open_frame_dataset(dset_id, d, "ebcc_h5_read_frames")) return 1;
    if (first_frame + n_frames > d.frames) { log_fatal("ebcc_h5_read_frames: frames %zu .. %zu of %zu", first_frame, first_frame + n_frames, d.frames); return 1; }
    std::vector<std::vector<uint8_t>> raw(n_frames);
    std::vector<const uint8_t *> ptrs(n_frames);
    std::vector<size_t> sizes(n_frames);
    hsize_like off[32];
    for (size_t i = 0; i < n_frames; i++) {
        chunk_offset(d, first_frame + i, off);
        hsize_like bytes = 0;
        unsigned int mask = 0;
        if (h5().Dget_chunk_storage_size(dset_id, off, &bytes) < 0 || bytes == 0) { log_fatal("ebcc_h5_read_frames: frame %zu has no chunk", first_frame + i); return 1; }
        raw[i].resize((size_t) bytes);
        if (h5().Dread_chunk(dset_id, 0, off, &mask, raw[i].data()) < 0 || mask != 0) { log_fatal("ebcc_h5_read_frames: H5Dread_chunk failed for frame %zu (filter mask %u)", first_frame + i, mask); return 1; }
        ptrs[i] = raw[i].data(); sizes[i] = raw[i].size();
    }
    return cached_decode_host_frames(ptrs.data(), sizes.data(), n_frames, (int) d.H, (int) d.W, frames_out) ? 1 : 0;
}

int H5PLget_plugin_type(void) { return 0; }            // H5PL_TYPE_FILTER
const void *H5PLget_plugin_info(void) { return H5Z_EBCC; }

}  // extern "C"

