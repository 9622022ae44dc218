/*
 * ebcc_hip.h - C-ABI of the MI355X (gfx950) EBCC engine: plain pointers and sizes only.
 *
 * The drop-in surface of the reference is include/ebcc_codec.h (same symbols as
 * /root/reference/src/ebcc_codec.h:41-49 plus the HDF5 plugin symbols of src/h5z_ebcc.c:14-28,38).
 * This header is ADDITIVE: device-resident batch entry points that the reference cannot offer
 * (its API is one host frame per call), plus unit-level entry points used by the parity tests.
 * Every function cites the reference interface it replaces.
 *
 * Conventions: return 0 on success, non-zero on error (message on stderr); "d_" pointers are HIP
 * device pointers on the context's device; all other pointers are host memory; work is enqueued on
 * the context's stream and the call returns after the stream has been synchronised unless stated.
 */
#ifndef EBCC_HIP_H
#define EBCC_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "ebcc_codec.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

typedef struct ebcc_hip_ctx ebcc_hip_ctx;

/* ---- context -------------------------------------------------------------------------------- */
int ebcc_hip_device_count(void);
/* One context per (device, frame geometry); owns a HIP stream and all workspaces for up to
 * max_frames frames of height x width.  Returns NULL on failure. */
ebcc_hip_ctx *ebcc_hip_create(int device, size_t max_frames, size_t height, size_t width);
void ebcc_hip_destroy(ebcc_hip_ctx *ctx);
/* the context's hipStream_t (so callers can order their own work against it) */
void *ebcc_hip_stream(ebcc_hip_ctx *ctx);
size_t ebcc_hip_workspace_bytes(const ebcc_hip_ctx *ctx);

/* raw device-memory helpers so that C / ctypes callers need no other HIP binding */
void *ebcc_hip_malloc(size_t bytes);
void ebcc_hip_free(void *d_ptr);
int ebcc_hip_memcpy_h2d(void *d_dst, const void *src, size_t bytes);
int ebcc_hip_memcpy_d2h(void *dst, const void *d_src, size_t bytes);

/* ---- residual layer --------------------------------------------------------------------------
 * Batch forms of spiht_encode / spiht_decode, /root/reference/src/spiht/spiht_re.h:20-21
 * (num_stages is fixed to WAVELET_LEVELS = 3, src/ebcc_codec.c:28,748). */

/* d_images: [n_frames][height][width] fp32 in [0,1].  trunc_bits[f] as in spiht_encode.
 * out_streams[f] receives a malloc()'d byte stream of out_sizes[f] bytes (free with free_buffer). */
int ebcc_hip_spiht_encode(ebcc_hip_ctx *ctx, const float *d_images, size_t n_frames, const size_t *trunc_bits,
                          uint8_t **out_streams, size_t *out_sizes);

/* streams[f]/sizes[f]/num_bits[f] as in spiht_decode(buffer_in, input_size, ..., num_bits);
 * d_images_out: [n_frames][height][width] fp32. */
int ebcc_hip_spiht_decode(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes,
                          const size_t *num_bits, size_t n_frames, float *d_images_out);

/* After ebcc_hip_spiht_encode on this context: the image spiht_decode(buf, trunc_bits/8, ..., trunc_bits)
 * would return for a prefix of each stream, rebuilt from the encoder's bookkeeping without parsing
 * (the device form of one probe of the truncation search, src/ebcc_codec.c:778-779). */
int ebcc_hip_spiht_decode_prefix(ebcc_hip_ctx *ctx, size_t n_frames, const size_t *trunc_bits, float *d_images_out);

/* Parity diagnostics: integer wavelet coefficients of the padded grid after load_image + sub_dc +
 * dwt2full + normalize (src/spiht/spiht_re.c:435,461,466,467).  coeffs: host [n_frames][padded pixels],
 * dc: host [n_frames]. */
int ebcc_hip_spiht_coeffs(ebcc_hip_ctx *ctx, const float *d_images, size_t n_frames, int32_t *coeffs, int *dc);
size_t ebcc_hip_padded_pixels(const ebcc_hip_ctx *ctx);

/* ---- JPEG 2000 base layer (unit level) ----------------------------------------------------------
 * Batch forms of j2k_encode_internal / j2k_decode_internal, /root/reference/src/ebcc_codec.c:105-180,
 * :1092-1136 (the reference reaches OpenJPEG there).  Frames are scaled to u16 with their own min/max as
 * ebcc_encode does (:675-689) and coded at rate cr[f]; minmax (host, [n][2], may be NULL) returns them. */
int ebcc_hip_j2k_encode(ebcc_hip_ctx *ctx, const float *d_frames, size_t n_frames, const float *cr, uint8_t **out_streams,
                        size_t *out_sizes, float *minmax);
/* After ebcc_hip_j2k_encode: the field j2k_decode_internal would return for those codestreams, decoded in
 * place from the encoder's code-block slots; nbad[f] = count(|x - d| > target[f]), err_sum[f] = sum(x - d). */
int ebcc_hip_j2k_emulated_decode(ebcc_hip_ctx *ctx, const float *d_frames, size_t n_frames, const float *target,
                                 float *d_out, unsigned long long *nbad, double *err_sum);
/* j2k_decode_internal for a batch of codestreams with (minval, maxval) pairs in minmax [n][2] */
int ebcc_hip_j2k_decode(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n_frames,
                        const float *minmax, float *d_out);

/* Host-only check of the codestream parser behind the decode path (no device needed): 0 = `cs` is accepted as a one-tile
 * codestream of height x width and every code-block entry lies inside it, 1 = rejected, 2 = accepted with an entry out of
 * bounds (never expected).  The reference leaves this to OpenJPEG (/root/reference/src/ebcc_codec.c:1096-1135). */
int ebcc_hip_j2k_parse_check(const uint8_t *cs, size_t n, size_t height, size_t width);

/* ---- frame codec -----------------------------------------------------------------------------
 * Batch forms of ebcc_encode / ebcc_decode (src/ebcc_codec.h:41-42) for frames resident in HBM.
 * config->dims must be {1, height, width} of the context (one frame per stream, as HDF5 chunks of
 * one frame / ebcc_encode_chunking with chunk_dims {1,H,W} produce).  Streams are malloc'd (free_buffer).
 * Return 0 = ok, 1 = error, 2 = NaN/Inf in the input; on failure entries of out_streams that are not NULL
 * still have to be freed.  A batch is coded as EBCC_HIP_SLICES concurrent slices (default: ebcc_hip_default_encode_slices() = 3 from 96 frames on, one slice below),
 * each on its own engine, stream and host thread; results do not depend on the slicing.  The slice engines are created on first use; ebcc_hip_prepare creates them ahead of time for batches of
 * n_frames (part of setting a context up, like ebcc_hip_create).  Returns 0. */
int ebcc_hip_prepare(ebcc_hip_ctx *ctx, size_t n_frames);
/* ebcc_encode / ebcc_decode / the chunking entry points and the HDF5 filter (/root/reference/src/ebcc_codec.h:41-49) keep
 * their engines between calls, one per device and frame geometry - tens of GB of device memory for batches of 256 frames of
 * 721 x 1440.  This gives that memory back; the next call makes the engines again.  Contexts of ebcc_hip_create are not
 * touched. */
void ebcc_hip_release_engines(void);
/* The same for a caller's context: its second engine set (ebcc_hip_encode_shard / ebcc_hip_decode_shard / the host-frames
 * entry points make it on their first call of more than one batch; it is as large as the context) is destroyed, the next
 * such call makes it again (also after a call that found no memory for it). */
void ebcc_hip_release_second_set(ebcc_hip_ctx *ctx);
/* Pageable host memory <-> device memory through the engine's pinned bounce buffers with several copying host threads
 * (what the chunking entry points use for their own arrays): ~3x hipMemcpy on a fresh pageable array.  0 = ok. */
int ebcc_hip_upload(ebcc_hip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int ebcc_hip_download(ebcc_hip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* Frames in pageable host memory <-> streams, any number of frames: what ebcc_encode_chunking / ebcc_decode_chunking
 * (/root/reference/src/ebcc_codec.c:1007-1046, :1322-1449) do between the array and the EBCK container, for callers that
 * keep the chunks themselves (HDF5 direct chunk writes / reads, ebcc_amd/h5_batch.py): staged uploads / downloads, batches
 * of the context's capacity on the two alternating engine sets, the output's pages mapped while the GPU decodes.
 * 0 = ok; on an encode error every stream made so far has been freed. */
int ebcc_hip_encode_host_frames(ebcc_hip_ctx *ctx, const float *h_frames, size_t n_frames, const codec_config_t *config,
                                uint8_t **out_streams, size_t *out_sizes);
int ebcc_hip_decode_host_frames(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n_frames,
                                float *h_frames_out);
/* Maps the pages of a host array that is about to receive a download (a fresh allocation of hundreds of MB is unmapped:
 * the download would fault it in page by page): asks for huge pages and touches every page from several threads, returns
 * when done.  DESTINATION arrays only - a zero is written to every page.  Meant to run on a second thread of the caller
 * beside ebcc_hip_decode_frames, as ebcc_decode_chunking (/root/reference/src/ebcc_codec.c:1322-1449) does for the
 * array it returns.  0 = ok. */
int ebcc_hip_prefault(void *h_dst, size_t bytes);
int ebcc_hip_encode_frames(ebcc_hip_ctx *ctx, const float *d_frames, size_t n_frames, const codec_config_t *config,
                           uint8_t **out_streams, size_t *out_sizes);
int ebcc_hip_decode_frames(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n_frames,
                           float *d_frames_out);
/* One GPU's share of a large array (BASELINE configs[3]: 4096 frames per GPU): any number of device-resident frames, coded
 * in batches of the context's capacity - the loop a caller of ebcc_hip_encode_frames would write, which is also what
 * ebcc_encode_chunking (/root/reference/src/ebcc_codec.c:1007-1046) does chunk by chunk - but on two alternating engine
 * sets, so that the host part of batch k (the level-22 zstd of the kept residual prefixes, about a quarter of a batch's
 * time, during which its kernels have nothing to do) runs beside the kernels of batch k + 1.  The second engine set is
 * created on first use and lives as long as the context; without memory for it the batches run one after the other.
 * Streams are identical to those of ebcc_hip_encode_frames.  0 = ok; on error every stream made so far has been freed. */
int ebcc_hip_encode_shard(ebcc_hip_ctx *ctx, const float *d_frames, size_t n_frames, const codec_config_t *config,
                          uint8_t **out_streams, size_t *out_sizes);
/* The decode counterpart: any number of streams to consecutive frames on the device, batches on the two engine sets side
 * by side (a batch of long residual streams is one wave per frame and leaves most of the chip idle; for host arrays,
 * ebcc_decode_chunking - /root/reference/src/ebcc_codec.c:1322-1449 - downloads one batch beside the next one's kernels the
 * same way).  Same frames as ebcc_hip_decode_frames batch by batch.  0 = ok. */
int ebcc_hip_decode_shard(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n_frames,
                          float *d_frames_out);

/* Direct-chunk batch path for C callers (netCDF-C / CDO-style pipelines; ebcc_amd/h5_batch.py is the Python form): a dataset
 * whose chunks are single frames - chunk dims (1, ..., 1, H, W), filter 308 as /root/reference/src/h5z_ebcc.c:38-93 reads it -
 * is written / read in device batches instead of one filter callback per chunk (/root/reference/src/h5z_ebcc.c:124-148 is
 * what HDF5 would call per chunk): frames [first_frame, first_frame + n_frames), counted in C order over the leading
 * dimensions, are coded with the dataset's own filter parameters and stored pre-filtered with H5Dwrite_chunk, or fetched with
 * H5Dread_chunk and decoded together.  dset_id: the hid_t (HDF5 >= 1.10) of the open dataset.  HDF5 is not linked: its
 * functions are taken from the libhdf5 the calling process has loaded.  Chunk bytes are identical to the callback's.
 * 0 = ok, 1 = error (logged); NaN / Inf in the frames exits with status 1 like the filter callback. */
int ebcc_h5_write_frames(long long dset_id, size_t first_frame, size_t n_frames, const float *frames);
int ebcc_h5_read_frames(long long dset_id, size_t first_frame, size_t n_frames, float *frames_out);

/* Worker threads of the process-wide host pool that runs the entropy stage (level-22 zstd of the residual prefixes) of
 * every slice of every call: EBCC_HOST_THREADS, else min(affinity mask, TWICE the container's CPU quota (cgroup cpu.max): the
 * stage comes in bursts, a burst may run wider than the quota as long as a period's total stays below it) divided by
 * LOCAL_WORLD_SIZE when the process is one rank of a multi-process job, minus min(slices, 2) for the threads that steer the
 * GPU (none subtracted from a share of 4 or fewer); at most 64. */
int ebcc_hip_host_threads(int slices);
/* Slices an encode batch of 96 frames or more runs as when EBCC_HIP_SLICES is not set (smaller batches: one). */
int ebcc_hip_default_encode_slices(void);
/* Slices an encode batch of n_frames runs as (EBCC_HIP_SLICES and the batch size taken into account). */
int ebcc_hip_encode_slices_for(size_t n_frames);
/* Host-side accounting since the last reset: out[0] usable CPUs (affinity and quota), out[1] CPU quota of the container in
 * CPUs (0: none), out[2] core-seconds spent in zstd, out[3] seconds the slices waited for the zstd workers, out[4] bytes
 * compressed, out[5] entropy batches, out[6] prefix bytes whose compression was proved unnecessary (ebcc_hip_zstd_floor).  bench.py prints them per rank (a run bound by the host's CPUs shows here). */
void ebcc_hip_host_stats(double *out, int reset);
/* Lower bound (bytes) of the zstd frame ZSTD_compress writes for [src, src + n) at any level, from the format alone (the
 * literals no match can cover cost at least their entropy; host_pool.hip: zstd_size_lower_bound); 0 = no bound (n above
 * 4 MB, or a libzstd that may split blocks).  The encoder uses it to decide the reference's "pure base layer beats base +
 * residual" comparison (src/ebcc_codec.c:838) without compressing prefixes that provably lose it. */
size_t ebcc_hip_zstd_floor(const uint8_t *src, size_t n);
/* Host-side check of the arithmetic identities the kernels rely on (the division-free s / 65535.0f of the fused inverse
 * wavelet level, for every s in [0, 65535]); returns the number of violations: 0. */
int ebcc_hip_selfcheck(void);
/* The tier-1 decoder's launch shape for a batch (host logic, no device work): table = the decode table the host parses out of
 * the packet headers, four ints per code-block (offset, segment bytes, bit-planes, passes); out[0..2] = code-blocks (by rank in
 * the longest-first order) that go 1, 2, 4 to a wave, out[3] = lanes per wave of all the others. */
void ebcc_hip_plan_decode_lanes(const int *table, int n_code_blocks, int out[4]);

/* Per-kernel timing with HIP events on the engine's stream (bench.py roofline leg).  Names: "t1_encode",
 * "t1_probe_decode", "t1_decode", "rate_alloc", "j2k_dwt_fwd", "spiht_encode".  Process-wide switch. */
void ebcc_hip_timing_enable(ebcc_hip_ctx *ctx, int on);
int ebcc_hip_timing_read(ebcc_hip_ctx *ctx, const char *name, double *total_ms, long *launches);

/* last error text of the calling thread ("" if none) */
const char *ebcc_hip_last_error(void);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* EBCC_HIP_H */
