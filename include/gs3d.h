/*
 * gs3d.h — C ABI of the MI355X-native 3D Gaussian Splatting core (libgs3d_hip.so).
 *
 * This is the drop-in boundary for LioQing/wgpu-3dgs-core's render-side API: every entry point
 * names the reference interface it replaces (paths relative to the reference repository root).
 * A Rust `extern "C"` block binds these 1:1; there are no C++/torch types in any
 * signature — only opaque handles, plain pointers and sizes.
 *
 * Conventions
 *   - Every function that can fail returns gs_status (0 = GS_OK, negative = error).  The variant
 *     fields of the matching Rust error enum (src/error.rs) are available, per thread, from
 *     gs_last_error().
 *   - Handles are created/destroyed explicitly.  A handle must not be used from two threads at
 *     once; distinct handles are independent.
 *   - Every launch takes a gs_stream (a hipStream_t): the analogue of the reference's
 *     CommandEncoder + queue.submit (src/compute_bundle.rs:114-132).  Work is asynchronous
 *     unless the function is documented as blocking.
 *   - There is no CPU fallback: without a HIP device gs_device_create fails with
 *     GS_ERR_NO_DEVICE and nothing else can be constructed.
 */
#ifndef GS3D_H
#define GS3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS3D_ABI_VERSION 1

typedef int32_t gs_status;

enum {
    GS_OK = 0,
    GS_ERR_INVALID_ARGUMENT = -1,
    GS_ERR_NO_DEVICE = -2,
    GS_ERR_HIP = -3,
    GS_ERR_OUT_OF_MEMORY = -4,
    /* GaussiansBufferUpdateError::CountMismatch{count, expected_count}      src/error.rs:66-70  */
    GS_ERR_COUNT_MISMATCH = -10,
    /* GaussiansBufferUpdateRangeError::CountMismatch{count,start,expected}  src/error.rs:73-81  */
    GS_ERR_RANGE_COUNT_MISMATCH = -11,
    /* GaussiansBufferTryFromBufferError::BufferSizeNotMultiple              src/error.rs:85-94  */
    GS_ERR_BUFFER_SIZE_NOT_MULTIPLE = -12,
    /* FixedSizeBufferWrapperError::BufferSizeMismatched                     src/error.rs:97-104 */
    GS_ERR_BUFFER_SIZE_MISMATCHED = -13,
    /* ComputeBundleCreateError::ResourceCountMismatch                       src/error.rs:109-117 */
    GS_ERR_RESOURCE_COUNT_MISMATCH = -14,
    /* ComputeBundleCreateError::WorkgroupSizeExceedsDeviceLimit             src/error.rs:118-125 */
    GS_ERR_WORKGROUP_SIZE_EXCEEDS_LIMIT = -15,
    /* ComputeBundleBuildError::{MissingBindGroupLayout, MissingResolver, MissingEntryPoint,
     * MissingMainShader, Wesl}                                              src/error.rs:129-143 */
    GS_ERR_MISSING_BIND_GROUP_LAYOUT = -16,
    GS_ERR_MISSING_RESOLVER = -17,
    GS_ERR_MISSING_ENTRY_POINT = -18,
    GS_ERR_MISSING_MAIN_SHADER = -19,
    GS_ERR_KERNEL_COMPILE = -20,
    /* the reference panics: gaussian_config.rs:131-133, 211-213, 230-232 */
    GS_ERR_LOSSY_CONFIG = -21,
    /* DownloadBufferError                                                   src/error.rs:55-63  */
    GS_ERR_DOWNLOAD = -22,
    /* the frame needs more than 2^32 (tile, Gaussian) pairs: pair indices are 32-bit */
    GS_ERR_PAIR_OVERFLOW = -23,
    /* std::io::Error (InvalidData / UnexpectedEof) of the PLY reader; message = the Rust message */
    GS_ERR_PLY = -24,
    /* std::io::Error of the SPZ reader / header validation; message = the Rust message */
    GS_ERR_SPZ = -25,
    /* gs_renderer_wait_frame: the frame produced more (tile, Gaussian) pairs than the renderer's pair
     * buffers hold (a = pairs, b = capacity).  The frame was SKIPPED: the image was not written (it
     * keeps what it held) rather than blended without its farthest pairs.  The next frame grows the
     * buffers: render again. */
    GS_ERR_PAIR_CAPACITY = -26,
    /* gs_renderer_wait_frame: the watchdog of the radix sort's LDS-atomic rank fired in this frame (an order
     * assumption about returning LDS atomics that the hardware documents do not promise; probed at
     * gs_device_create and checked by block 0 of every pass).  The frame's blend order may be wrong; the device
     * has been switched to the ballot-based rank for good: render again. */
    GS_ERR_RANK_ORDER = -27
};

/* Thread-local details of the last failing call on this thread.
 *   COUNT_MISMATCH:            a = count, b = expected_count
 *   RANGE_COUNT_MISMATCH:      a = count, b = start, c = expected_count
 *   BUFFER_SIZE_NOT_MULTIPLE:  a = buffer_size, b = expected_multiple_size
 *   BUFFER_SIZE_MISMATCHED:    a = buffer_size, b = expected_size
 *   RESOURCE_COUNT_MISMATCH:   a = resource_count, b = bind_group_layout_count
 *   WORKGROUP_SIZE_EXCEEDS_LIMIT: a = workgroup_size, b = device_limit
 *   HIP:                       a = hipError_t                                                   */
typedef struct gs_error_info {
    int32_t code;
    uint64_t a, b, c;
    char message[256];
} gs_error_info;

void gs_last_error(gs_error_info *out);
const char *gs_status_string(gs_status s);
uint32_t gs_abi_version(void);
/* wgpu::Adapter::get_info() analogue (AdapterInfo::driver_info): the HIP version this library was
 * COMPILED against (HIP_VERSION of the headers) and the versions of the runtime and driver it is
 * RUNNING on (hipRuntimeGetVersion / hipDriverGetVersion; 0 when the call fails).  A Python host that
 * also imports PyTorch runs the library on the runtime bundled with the torch wheel, which may be older
 * than the headers: every benchmark record prints both. */
void gs_hip_versions(int32_t *compiled, int32_t *runtime, int32_t *driver);

/* ------------------------------------------------------------------------------------------ */
/* Data model                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* GaussianShConfig / GaussianCov3dConfig families — src/gaussian_config.rs:15-233 */
typedef enum { GS_SH_SINGLE = 0, GS_SH_HALF = 1, GS_SH_NORM8 = 2, GS_SH_NONE = 3 } gs_sh_config;
typedef enum { GS_COV3D_ROT_SCALE = 0, GS_COV3D_SINGLE = 1, GS_COV3D_HALF = 2 } gs_cov3d_config;

/* struct Gaussian — src/gaussian.rs:53-60 (rot = xyzw; 224 bytes) */
typedef struct gs_gaussian {
    float rot[4];
    float pos[3];
    uint8_t color[4];
    float sh[45];
    float scale[3];
} gs_gaussian;

/* GaussianDisplayMode — src/buffer/gaussian_transform.rs:7-14 */
typedef enum { GS_DISPLAY_SPLAT = 0, GS_DISPLAY_ELLIPSE = 1, GS_DISPLAY_POINT = 2 } gs_display_mode;

/* GaussianTransformPod — src/buffer/gaussian_transform.rs:166-174 (8 bytes) */
typedef struct gs_gaussian_transform_pod {
    float size;
    uint8_t flags[4]; /* display_mode, sh_deg, no_sh0, max_std_dev (u8) */
} gs_gaussian_transform_pod;

/* ModelTransformPod — src/buffer/model_transform.rs:61-66 (48 bytes) */
typedef struct gs_model_transform_pod {
    float pos[3];
    float _pad0;
    float rot[4];
    float scale[3];
    float _pad1;
} gs_model_transform_pod;

/* std::mem::size_of::<GaussianPodWithSh{S}Cov3d{C}Configs>() — src/buffer/gaussian.rs:373-384 */
size_t gs_pod_size(gs_sh_config sh, gs_cov3d_config cov);
/* GaussianPod::features() — src/buffer/gaussian.rs:270-286; order: sh_single, sh_half, sh_norm8,
 * sh_none, cov3d_rot_scale, cov3d_single, cov3d_half */
gs_status gs_pod_features(gs_sh_config sh, gs_cov3d_config cov, uint8_t out[7]);
const char *gs_feature_name(uint32_t index);
/* G::from_gaussian — src/buffer/gaussian.rs:314-339 (host, multi-threaded) */
gs_status gs_pack(gs_sh_config sh, gs_cov3d_config cov, const gs_gaussian *in, size_t n, void *out);
/* Into<Gaussian> — src/buffer/gaussian.rs:341-363; GS_ERR_LOSSY_CONFIG where Rust panics */
gs_status gs_unpack_to_gaussian(gs_sh_config sh, gs_cov3d_config cov, const void *pods, size_t n,
                                gs_gaussian *out);

/* GaussianShDegree::new / GaussianMaxStdDev::new / GaussianTransformPod::new —
 * src/buffer/gaussian_transform.rs:25-30, 63-68, 178-194.  GS_ERR_INVALID_ARGUMENT where the
 * Rust constructors return None. */
gs_status gs_gaussian_transform_pod_new(float size, gs_display_mode mode, uint8_t sh_deg,
                                        uint8_t no_sh0, float max_std_dev,
                                        gs_gaussian_transform_pod *out);
void gs_gaussian_transform_pod_default(gs_gaussian_transform_pod *out);
gs_status gs_max_std_dev_encode(float max_std_dev, uint8_t *out);
float gs_max_std_dev_decode(uint8_t v);
/* ModelTransformPod::new / default — src/buffer/model_transform.rs:68-84 */
void gs_model_transform_pod_new(const float pos[3], const float rot_xyzw[4], const float scale[3],
                                gs_model_transform_pod *out);
void gs_model_transform_pod_default(gs_model_transform_pod *out);

/* ------------------------------------------------------------------------------------------ */
/* Inria PLY source format — src/source_format/ply.rs, src/gaussian.rs:70-125 (host side)      */
/* ------------------------------------------------------------------------------------------ */

/* PlyGaussianPod — src/source_format/ply.rs:11-21 (62 little-endian f32 = 248 bytes) */
typedef struct gs_ply_gaussian_pod {
    float pos[3];
    float normal[3];
    float color[3];   /* f_dc_0..2 */
    float sh[45];     /* f_rest_0..44, channel-planar */
    float alpha;      /* opacity (logit) */
    float scale[3];   /* log scale */
    float rot[4];     /* wxyz */
} gs_ply_gaussian_pod;

/* PlyGaussians::PLY_PROPERTIES — ply.rs:204-267 */
const char *gs_ply_property_name(uint32_t index);
/* Gaussian::from_ply / Gaussian::to_ply — src/gaussian.rs:70-125 */
void gs_gaussian_from_ply(const gs_ply_gaussian_pod *in, size_t n, gs_gaussian *out);
/* The exp of Gaussian::from_ply (f32::exp in the reference, src/gaussian.rs:76,81): the double-based
 * table algorithm glibc's expf uses, restated so that the host path above and the device path
 * (gs_pack_device_from_ply) give the same bits (csrc/gs_convert.h). */
float gs_expf(float x);
void gs_gaussian_to_ply(const gs_gaussian *in, size_t n, gs_ply_gaussian_pod *out);
/* PlyGaussians::read_from — ply.rs:292-408.  Call with out == NULL to get the vertex count, then
 * with a buffer.  Handles the Inria fast path (one memcpy) and custom property orders in ascii /
 * binary little / big endian.  Errors: GS_ERR_PLY with the reference's message, e.g.
 * "Gaussian vertex element not found in PLY header",
 * "Gaussian element property invalid or missing in PLY". */
gs_status gs_ply_read(const void *bytes, size_t len, gs_ply_gaussian_pod *out, size_t capacity,
                      size_t *count_out, int32_t *is_inria_out);
/* PlyGaussians::write_to — ply.rs:410-431 (binary_little_endian Inria layout).  Call with
 * out == NULL to get the size. */
gs_status gs_ply_write(const gs_ply_gaussian_pod *pods, size_t n, void *out, size_t capacity,
                       size_t *bytes_out);

/* ------------------------------------------------------------------------------------------ */
/* SPZ source format — src/source_format/spz.rs, src/gaussian.rs:126-352 (host side)           */
/* ------------------------------------------------------------------------------------------ */

/* SpzGaussiansHeaderPod — spz.rs:436-446 (16 bytes; magic 0x5053474e, versions 1..=3) */
typedef struct gs_spz_header {
    uint32_t magic;
    uint32_t version;
    uint32_t num_points;
    uint8_t sh_degree;
    uint8_t fractional_bits;
    uint8_t flags;      /* bit 0 = antialiased */
    uint8_t reserved;
} gs_spz_header;

/* SpzGaussiansFromGaussianSliceOptions — spz.rs:962-999 (default: version 3, sh 3, 12 fractional
 * bits, not antialiased, sh_quantize_bits [5,4,4]) */
typedef struct gs_spz_options {
    uint32_t version;
    uint8_t sh_degree;
    uint8_t fractional_bits;
    uint8_t antialiased;
    uint8_t _pad;
    uint32_t sh_quantize_bits[3];
} gs_spz_options;

void gs_spz_options_default(gs_spz_options *out);
/* SpzGaussians::read_from + iter_gaussian (gzip -> columns -> Gaussian::from_spz).  Call with
 * out == NULL for the count / header.  Errors: GS_ERR_SPZ with the reference's message, e.g.
 * "Invalid SPZ magic number: 0, expected 5053474E", "Unsupported SPZ version: 999, expected one of 1..=3" */
gs_status gs_spz_decode(const void *bytes, size_t len, gs_spz_header *header_out, gs_gaussian *out,
                        size_t capacity, size_t *count_out);
gs_status gs_spz_decode_decompressed(const void *bytes, size_t len, gs_spz_header *header_out,
                                     gs_gaussian *out, size_t capacity, size_t *count_out);
/* SpzGaussians::from_gaussians_with_options + write_to (Gaussian::to_spz -> columns -> gzip).
 * Call with out == NULL for the size. */
gs_status gs_spz_encode(const gs_gaussian *in, size_t n, const gs_spz_options *options, void *out,
                        size_t capacity, size_t *bytes_out);
gs_status gs_spz_encode_decompressed(const gs_gaussian *in, size_t n, const gs_spz_options *options,
                                     void *out, size_t capacity, size_t *bytes_out);
/* the gzip member around the payload on its own (flate2 GzDecoder / GzEncoder, spz.rs:945-959), so a
 * host mirror can keep the payload it read and write the identical columns back.  out == NULL: size */
gs_status gs_spz_decompress(const void *bytes, size_t len, void *out, size_t capacity, size_t *bytes_out);
gs_status gs_spz_compress(const void *bytes, size_t len, void *out, size_t capacity, size_t *bytes_out);

/* GaussiansSource / Gaussians — src/gaussian.rs:394-548: the unified representation, reduced to
 * what crosses an FFI: read a source format into Gaussians, write Gaussians as a source format.
 *   gs_gaussians_read  = Gaussians::read_from(reader, source)?.iter_gaussian()   (:478-497)
 *   gs_gaussians_write = Gaussians::from_gaussians_iter(iter, source).write_to(writer)   (:426-436, 513-524;
 *                        SPZ with the default options, as `iter.collect::<SpzGaussians>()` does)
 * GS_SOURCE_INTERNAL fails with the reference's InvalidInput messages ("cannot read Internal
 * Gaussians from buffer" / "cannot write Internal Gaussians to buffer") as GS_ERR_INVALID_ARGUMENT.
 * Call with out == NULL for the count / size. */
typedef enum { GS_SOURCE_INTERNAL = 0, GS_SOURCE_PLY = 1, GS_SOURCE_SPZ = 2 } gs_gaussians_source;
gs_status gs_gaussians_read(const void *bytes, size_t len, gs_gaussians_source source, gs_gaussian *out,
                            size_t capacity, size_t *count_out);
gs_status gs_gaussians_write(const gs_gaussian *in, size_t n, gs_gaussians_source source, void *out,
                             size_t capacity, size_t *bytes_out);

/* ------------------------------------------------------------------------------------------ */
/* Device, streams                                                                             */
/* ------------------------------------------------------------------------------------------ */

typedef struct gs_device gs_device;   /* wgpu::Device + wgpu::Queue */
typedef struct gs_stream gs_stream;   /* wgpu::CommandEncoder + queue.submit */

/* wgpu::Limits fields read by src/compute_bundle.rs:269-272 */
typedef struct gs_limits {
    uint32_t max_compute_workgroup_size_x;
    uint32_t max_compute_invocations_per_workgroup;
    uint32_t compute_units;
    uint32_t wavefront_size;
    uint64_t total_memory_bytes;
    char arch_name[64];
} gs_limits;

gs_status gs_device_create(int32_t hip_ordinal, gs_device **out);
void gs_device_destroy(gs_device *dev);
gs_status gs_device_limits(const gs_device *dev, gs_limits *out);
gs_status gs_device_synchronize(gs_device *dev);
/* 1 while the radix sorts of this device rank with returning LDS atomics (the order probe of gs_device_create passed
 * and the per-frame watchdog has not fired: GS_ERR_RANK_ORDER), 0 once they use the ballot-based rank (no reference
 * item: wgpu has no such primitive) */
int32_t gs_device_fast_rank(const gs_device *dev);
gs_status gs_stream_create(gs_device *dev, gs_stream **out);
/* Streams of one process that should run CONCURRENTLY on the device (two frames in flight: the latency-bound sort
 * chain of frame i + 1 under the VALU-bound blend of frame i) must sit on different hardware queues; HIP gives every
 * priority level its own queues, and streams of equal priority may share one (measured under torch: they do).
 * priority: hipStreamCreateWithPriority's number, between *least and *greatest of gs_device_stream_priority_range
 * (numerically lower = higher priority).  No reference item: a wgpu::Queue has no priority. */
gs_status gs_stream_create_with_priority(gs_device *dev, int32_t priority, gs_stream **out);
gs_status gs_device_stream_priority_range(gs_device *dev, int32_t *least, int32_t *greatest);
/* borrow an existing hipStream_t (e.g. the stream a caller's framework is using) */
gs_status gs_stream_wrap(gs_device *dev, void *hip_stream, gs_stream **out);
void *gs_stream_native(const gs_stream *s);
gs_status gs_stream_synchronize(gs_stream *s);
/* Lifetime: a renderer whose newest frame was enqueued on `s` stays usable — gs_stream_destroy records that frame's end
 * event on the stream before it goes, and later frames / waits of the renderer are ordered behind the event, not the
 * stream.  A WRAPPED stream (gs_stream_wrap) must therefore be handed to gs_stream_destroy BEFORE the caller destroys
 * the hipStream_t it wraps. */
void gs_stream_destroy(gs_stream *s);

/* ------------------------------------------------------------------------------------------ */
/* Buffers — trait BufferWrapper / FixedSizeBufferWrapper, src/buffer/mod.rs:17-150            */
/* ------------------------------------------------------------------------------------------ */

typedef struct gs_buffer gs_buffer; /* wgpu::Buffer: ref-counted device allocation */

gs_status gs_buffer_create(gs_device *dev, size_t bytes, const void *init_or_null, gs_buffer **out);
/* adopt device memory the caller owns (non-owning; the TryFrom<wgpu::Buffer> direction) */
gs_status gs_buffer_from_raw(gs_device *dev, void *device_ptr, size_t bytes, gs_buffer **out);
gs_buffer *gs_buffer_retain(gs_buffer *b); /* Clone = handle copy (buffer/gaussian.rs:16) */
void gs_buffer_release(gs_buffer *b);
size_t gs_buffer_size(const gs_buffer *b);
void *gs_buffer_device_ptr(const gs_buffer *b);
/* queue.write_buffer(buffer, offset, data) */
gs_status gs_buffer_write(gs_buffer *b, gs_stream *s, size_t offset, const void *src, size_t bytes);
/* BufferWrapper::download — src/buffer/mod.rs:27-42 (blocking: prepare_download + map + poll) */
gs_status gs_buffer_download(gs_buffer *b, gs_stream *s, void *dst, size_t bytes);
/* BufferWrapper::prepare_download — src/buffer/mod.rs:48-78: enqueues the copy of the whole buffer
 * into a host-visible staging buffer on `s` and returns at once (the wgpu original records
 * copy_buffer_to_buffer into the caller's encoder).
 * BufferWrapper::map_download — :80-101: gs_download_map blocks until the copy has landed (the
 * map_async + device.poll(wait) of the original) and exposes the bytes, valid until
 * gs_download_release; gs_download_ready polls without blocking.
 * Errors: GS_ERR_DOWNLOAD (DownloadBufferError). */
typedef struct gs_download gs_download;
gs_status gs_buffer_prepare_download(gs_buffer *b, gs_stream *s, gs_download **out);
int32_t gs_download_ready(gs_download *d);
gs_status gs_download_map(gs_download *d, const void **data_out, size_t *bytes_out);
void gs_download_release(gs_download *d);

/* GaussiansBuffer<G> — src/buffer/gaussian.rs:17-229 */
typedef struct gs_gaussians_buffer gs_gaussians_buffer;

/* new_with_pods (pods != NULL) / new_empty (pods == NULL) — :50-90 */
gs_status gs_gaussians_buffer_create(gs_device *dev, gs_sh_config sh, gs_cov3d_config cov,
                                     const void *pods_or_null, size_t len,
                                     gs_gaussians_buffer **out);
/* new(device, gaussians): upload the source records, pack on the device (gs_pack_device) — :21-30 */
gs_status gs_gaussians_buffer_create_from_gaussians(gs_device *dev, gs_sh_config sh,
                                                    gs_cov3d_config cov, const gs_gaussian *gaussians,
                                                    size_t len, gs_gaussians_buffer **out);
/* PlyGaussians -> GaussiansBuffer in one step: the 248-byte vertex records cross PCIe as they are (one
 * copy per slice) and ONE kernel does Gaussian::from_ply (src/gaussian.rs:70-92) fused with
 * G::from_gaussian (src/buffer/gaussian.rs:314-339) — what the reference does per vertex on the host
 * (src/source_format/ply.rs:386-390 iter_gaussian + src/buffer/gaussian.rs:21-30 new).  The PODs are
 * bit-equal to gs_gaussian_from_ply followed by gs_pack. */
gs_status gs_gaussians_buffer_create_from_ply(gs_device *dev, gs_sh_config sh, gs_cov3d_config cov,
                                              const gs_ply_gaussian_pod *ply, size_t len,
                                              gs_gaussians_buffer **out);
/* the same for [start, start + count) of an existing buffer (update_range semantics and errors) */
gs_status gs_gaussians_buffer_update_range_ply(gs_gaussians_buffer *g, gs_stream *s, size_t start,
                                               const gs_ply_gaussian_pod *ply, size_t count);
/* SpzGaussians -> GaussiansBuffer: inflate on the host (a gzip member is sequential), then the
 * decompressed columns cross PCIe once and ONE kernel does Gaussian::from_spz
 * (src/gaussian.rs:134-229 over spz.rs:739-771's columns) fused with G::from_gaussian.  The PODs are
 * bit-equal to gs_spz_decode followed by gs_pack; errors as gs_spz_decode (GS_ERR_SPZ). */
gs_status gs_gaussians_buffer_create_from_spz(gs_device *dev, gs_sh_config sh, gs_cov3d_config cov,
                                              const void *bytes, size_t len, gs_spz_header *header_out,
                                              gs_gaussians_buffer **out);
gs_status gs_gaussians_buffer_create_from_spz_decompressed(gs_device *dev, gs_sh_config sh, gs_cov3d_config cov,
                                                           const void *bytes, size_t len,
                                                           gs_spz_header *header_out,
                                                           gs_gaussians_buffer **out);
/* the kernel itself: `n` PlyGaussianPod records in DEVICE memory -> PODs in device memory */
gs_status gs_pack_device_from_ply(gs_device *dev, gs_stream *s, gs_sh_config sh, gs_cov3d_config cov,
                                  const gs_ply_gaussian_pod *ply_device, size_t n, void *pods_device);
/* TryFrom<wgpu::Buffer> — :213-229; the wrapper retains `buffer` */
/* G::from_gaussian on the device (src/buffer/gaussian.rs:314-339 for all 12 PODs): `n` source
 * records (struct Gaussian, 224 bytes each) already in device memory -> PODs in device memory, bit-equal
 * to gs_pack.  GaussiansBuffer::new / update / update_range with Gaussians go through this: the
 * source records cross PCIe once and are packed on the device. */
gs_status gs_pack_device(gs_device *dev, gs_stream *s, gs_sh_config sh, gs_cov3d_config cov,
                         const gs_gaussian *gaussians_device, size_t n, void *pods_device);
gs_status gs_gaussians_buffer_from_buffer(gs_buffer *buffer, gs_sh_config sh, gs_cov3d_config cov,
                                          gs_gaussians_buffer **out);
void gs_gaussians_buffer_destroy(gs_gaussians_buffer *g);
size_t gs_gaussians_buffer_len(const gs_gaussians_buffer *g); /* :93-95 */
gs_buffer *gs_gaussians_buffer_buffer(const gs_gaussians_buffer *g); /* BufferWrapper::buffer(), borrowed */
gs_sh_config gs_gaussians_buffer_sh(const gs_gaussians_buffer *g);
gs_cov3d_config gs_gaussians_buffer_cov3d(const gs_gaussians_buffer *g);
/* update_with_pod — :122-138 */
gs_status gs_gaussians_buffer_update(gs_gaussians_buffer *g, gs_stream *s, const void *pods,
                                     size_t count);
/* update_range_with_pod — :161-183 */
gs_status gs_gaussians_buffer_update_range(gs_gaussians_buffer *g, gs_stream *s, size_t start,
                                           const void *pods, size_t count);
/* update / update_range with Gaussians (host pack first) — :103-118, :143-158 */
gs_status gs_gaussians_buffer_update_gaussians(gs_gaussians_buffer *g, gs_stream *s,
                                               const gs_gaussian *gaussians, size_t count);
gs_status gs_gaussians_buffer_update_range_gaussians(gs_gaussians_buffer *g, gs_stream *s,
                                                     size_t start, const gs_gaussian *gaussians,
                                                     size_t count);
/* download::<G> (pods) and download_gaussians — :186-195 (blocking).  download_gaussians converts
 * POD -> Gaussian (GaussianPod::into_gaussian) on the DEVICE and copies struct Gaussian records back;
 * GS_ERR_LOSSY_CONFIG for ShNone / Cov3dSingle / Cov3dHalf, where the reference panics. */
gs_status gs_gaussians_buffer_download(gs_gaussians_buffer *g, gs_stream *s, void *pods_out,
                                       size_t count);
gs_status gs_gaussians_buffer_download_gaussians(gs_gaussians_buffer *g, gs_stream *s,
                                                 gs_gaussian *out, size_t count);
/* tell the wrapper that device code wrote the underlying buffer (e.g. a compute bundle bound it
 * read-write), so the renderer's block-planar mirror must be rebuilt on the next frame */
void gs_gaussians_buffer_mark_dirty(gs_gaussians_buffer *g);
/* The renderer reads a mirror of the buffer whose slots are, by default, in SPATIAL order: ids
 * sorted by the 30-bit Morton code of the position (10 bits per axis over the bounding box of all
 * positions; ties by index) — DESIGN.md §3.4a.  The order is computed whenever the whole buffer
 * is (re)mirrored (creation, gs_gaussians_buffer_update, mark_dirty); update_range keeps it.  It is
 * observable in exactly one way: (tile, Gaussian) pairs with bit-identical depth in one tile are
 * blended in mirror order.  Disable (index order) per buffer here or globally with
 * GS3D_SPATIAL_ORDER=0.  download_order writes order[slot] = Gaussian index (the identity when
 * disabled); count must equal the buffer length. */
gs_status gs_gaussians_buffer_set_spatial_order(gs_gaussians_buffer *g, int32_t enabled);
int32_t gs_gaussians_buffer_spatial_order(const gs_gaussians_buffer *g);
gs_status gs_gaussians_buffer_download_order(gs_gaussians_buffer *g, gs_stream *s, uint32_t *order_out,
                                             size_t count);

/* GaussianTransformBuffer — src/buffer/gaussian_transform.rs:104-163 (8-byte uniform) */
gs_status gs_gaussian_transform_buffer_create(gs_device *dev, gs_buffer **out);
gs_status gs_gaussian_transform_buffer_update(gs_buffer *b, gs_stream *s,
                                              const gs_gaussian_transform_pod *pod);
gs_status gs_gaussian_transform_buffer_from_buffer(gs_buffer *b); /* size check only */
/* ModelTransformBuffer — src/buffer/model_transform.rs:10-58 (48-byte uniform) */
gs_status gs_model_transform_buffer_create(gs_device *dev, gs_buffer **out);
gs_status gs_model_transform_buffer_update(gs_buffer *b, gs_stream *s,
                                           const gs_model_transform_pod *pod);
gs_status gs_model_transform_buffer_from_buffer(gs_buffer *b);

/* ------------------------------------------------------------------------------------------ */
/* ComputeBundle — src/compute_bundle.rs:49-351                                                */
/* ------------------------------------------------------------------------------------------ */

/* The kernel registry replaces shader::PACKAGE + wesl feature flags (src/shader.rs:8-56):
 * "features" select a template instantiation keyed by (sh, cov). */
typedef enum {
    /* tests/common/shader/array_map_add.wesl: group0 = {data rw u32[]}, group1 = {uniform u32};
     * data[i] += uniform + constant("add_constant") */
    GS_KERNEL_ARRAY_MAP_ADD = 0,
    /* tests/shader/gaussian.rs:26-59: group0 = {gaussians, output(56 f32)} */
    GS_KERNEL_TEST_GAUSSIAN = 1,
    /* tests/shader/gaussian_transform.rs:13-48: group0 = {transform uniform, output(4 u32)} */
    GS_KERNEL_TEST_GAUSSIAN_TRANSFORM = 2,
    /* tests/shader/model_transform.rs:14-51: group0 = {model uniform, output(38 f32)} */
    GS_KERNEL_TEST_MODEL_TRANSFORM = 3,
    /* AoS pods -> SoA f32 planes (color4, sh45, cov6) for `count` Gaussians: group0 = {gaussians, out} */
    GS_KERNEL_UNPACK_SOA = 4,
    GS_KERNEL_COUNT_ = 5
} gs_kernel_id;

typedef struct gs_bundle gs_bundle;

typedef struct gs_bundle_desc {
    const char *label;          /* may be NULL ("Compute Bundle") */
    gs_kernel_id kernel;        /* main_shader + entry_point */
    gs_sh_config sh;            /* wesl features */
    gs_cov3d_config cov;
    uint32_t bind_group_count;  /* bind_group_layouts.len() */
    const uint32_t *bindings_per_group;
    uint32_t workgroup_size;    /* 0 = None = device limit (compute_bundle.rs:274) */
    const char *const *constant_names; /* PipelineCompilationOptions.constants */
    const double *constant_values;
    uint32_t constant_count;
} gs_bundle_desc;

/* ComputeBundle::new_without_bind_groups — :260-341 */
gs_status gs_bundle_create(gs_device *dev, const gs_bundle_desc *desc, gs_bundle **out);
/* ComputeBundle::new — :141-188: `resources` = bind_group_count_given arrays of buffers;
 * GS_ERR_RESOURCE_COUNT_MISMATCH when group counts differ */
gs_status gs_bundle_create_with_bind_groups(gs_device *dev, const gs_bundle_desc *desc,
                                            gs_buffer *const *const *resources,
                                            const uint32_t *resource_counts,
                                            uint32_t resource_group_count, gs_bundle **out);
/* ComputeBundleBuilder::build with a caller-supplied shader — compute_bundle.rs:500-586.
 * `source` is HIP C++ (the MI355X replacement for the WESL main module).  It is compiled at run
 * time with hiprtc for the device's gfx arch; `#include <wgpu_3dgs_core.h>` imports the device
 * library (namespace gs: gaussian_unpack_color / _sh<SH> / _cov3d<SH,COV>, gaussian_transform_*,
 * model_*; the WESL package of src/shader.rs).  The entry point must be
 *     extern "C" __global__ void <entry_point>(gs::BundleArgs a, uint32_t count)
 * Feature flags arrive as macros: GS_SH, GS_COV (config indices) plus one macro per enabled
 * feature name (sh_single, cov3d_half, ... and every string of `defines`); the reference's
 * `override workgroup_size` is the macro `workgroup_size`; pipeline constants are macros of their
 * names.  Compilation errors -> GS_ERR_KERNEL_COMPILE (the ComputeBundleBuildError::Wesl analogue)
 * with the compiler log in gs_last_error().message. */
typedef struct gs_bundle_source_desc {
    const char *label;
    const char *source;
    const char *entry_point;
    gs_sh_config sh;
    gs_cov3d_config cov;
    uint32_t bind_group_count;
    const uint32_t *bindings_per_group;
    uint32_t workgroup_size;
    const char *const *constant_names;
    const double *constant_values;
    uint32_t constant_count;
    const char *const *defines;
    uint32_t define_count;
} gs_bundle_source_desc;
gs_status gs_bundle_create_from_source(gs_device *dev, const gs_bundle_source_desc *desc, gs_bundle **out);
/* create bind groups on a bundle built without them (`build` = `build_without_bind_groups` + this);
 * GS_ERR_RESOURCE_COUNT_MISMATCH as in compute_bundle.rs:161-168 */
gs_status gs_bundle_attach_bind_groups(gs_bundle *b, gs_buffer *const *const *resources,
                                       const uint32_t *resource_counts, uint32_t resource_group_count);
void gs_bundle_destroy(gs_bundle *b);
uint32_t gs_bundle_workgroup_size(const gs_bundle *b);
const char *gs_bundle_label(const gs_bundle *b);
uint32_t gs_bundle_bind_group_layout_count(const gs_bundle *b);
uint32_t gs_bundle_bind_group_count(const gs_bundle *b);
/* update_bind_group_with_binding_resources — :222-231; GS_ERR_INVALID_ARGUMENT when the index is
 * out of bounds (Rust returns None) */
gs_status gs_bundle_set_bind_group(gs_bundle *b, uint32_t index, gs_buffer *const *buffers,
                                   uint32_t count);
/* dispatch — :196-198: ceil(count / workgroup_size) workgroups on `s` */
gs_status gs_bundle_dispatch(gs_bundle *b, gs_stream *s, uint32_t count);
/* dispatch_with_bind_groups — :114-132 (ComputeBundle<()>::dispatch :344-351) */
gs_status gs_bundle_dispatch_with_bind_groups(gs_bundle *b, gs_stream *s, uint32_t count,
                                              gs_buffer *const *const *groups,
                                              const uint32_t *group_counts, uint32_t group_count);
/* number of workgroups the last dispatch launched (launch arithmetic check) */
uint32_t gs_bundle_last_workgroup_count(const gs_bundle *b);

/* ------------------------------------------------------------------------------------------ */
/* Render path (absent from the reference crate; see DESIGN.md §3 for its definition)          */
/* ------------------------------------------------------------------------------------------ */

/* view: column-major world->view, right-handed, camera looks down -Z, +Y up */
typedef struct gs_camera {
    float view[16];
    float pos[3];
    float fx, fy, cx, cy;
    float near_plane, far_plane;
    uint32_t width, height;
    float background[3];
} gs_camera;

/* 48-byte projected splat record; conic pre-scaled to (-A/2, -B, -C/2) */
typedef struct gs_projected {
    float mx, my;
    float ca, cb, cc;
    float opacity;
    float r, g, b;
    float depth;
    uint16_t tx0, ty0, tx1, ty1;
} gs_projected;

typedef struct gs_frame_stats {
    uint64_t gaussians;       /* N */
    uint64_t visible;         /* Gaussians with >= 1 tile */
    uint64_t pairs;           /* D = (tile, Gaussian) pairs */
    uint32_t tiles_x, tiles_y;
    uint32_t sort_passes;
    uint32_t timed_frames;    /* frames accumulated below since the last reset */
    /* accumulated stage time in ms (only while timing is enabled):
     * 0 repack, 1 preprocess + compaction of the visible Gaussians, 2 sizing pass (0 in steady
     * state), 3 depth sort, 4 pair expansion, 5 tile sort, 6 ranges, 7 blend, 8 whole frame */
    double stage_ms[12];
} gs_frame_stats;

typedef struct gs_renderer gs_renderer;

void gs_camera_look_at(const float eye[3], const float target[3], const float up[3],
                       float vfov_radians, uint32_t width, uint32_t height, float near_plane,
                       float far_plane, gs_camera *out);

gs_status gs_renderer_create(gs_device *dev, gs_renderer **out);
void gs_renderer_destroy(gs_renderer *r);
/* stage timing with HIP events on the launch stream (off by default) */
gs_status gs_renderer_set_timing(gs_renderer *r, int32_t enabled);
gs_status gs_renderer_reset_stats(gs_renderer *r);
/* Sharded frames (no reference item: the viewer owns the frame): a DEVICE word that
 * receives the flags of every following frame of this renderer (gs_frame_result.flags: bit 0 capacity exceeded, bit 1 skipped; 0 = the band was
 * rendered), written in stream order by the kernel that publishes the frame result — no extra
 * launch.  A rank points it at a word inside its chunk of the gather buffer, so the one all-gather
 * that exchanges the bands also tells every rank whether ANY band was skipped (pair capacity), and
 * the frame can be dropped as a whole instead of being shown torn.  NULL (the default) disables.
 * The word must stay valid until the frames that were enqueued with it have completed. */
gs_status gs_renderer_set_frame_flags_target(gs_renderer *r, uint32_t *device_word);
/* blocking: synchronises the stream of the last frame first */
gs_status gs_renderer_stats(gs_renderer *r, gs_frame_stats *out);

/* One frame: repack (if the Gaussians changed) -> preprocess + ordered compaction of the visible
 * Gaussians -> depth sort -> pair expansion in depth order -> stable tile sort -> tile ranges ->
 * blend.  Renders tile rows [band_ty0, band_ty1) (16-pixel rows; pass 0 and UINT32_MAX for the whole
 * image) into rgba_out_device, a device pointer to the FULL height x width x 4 f32 image (16-byte
 * aligned); only the band's rows are written.  All three GaussianDisplayModes of the transform are
 * rendered (Splat: Gaussian falloff; Ellipse: flat alpha inside the max_std_dev ellipse; Point:
 * flat alpha inside a 1.5-pixel dot — DESIGN.md §3.5a).
 *
 * (A)synchrony — the analogue of ComputeBundle::dispatch recording into a CommandEncoder that
 * executes at queue.submit (src/compute_bundle.rs:114-132): the call only ENQUEUES work on `s` and
 * returns; the image is complete when the stream has been synchronised (or gs_renderer_wait_frame
 * has returned).  The visible count V and the pair count D never come back to the host inside a
 * frame: grids are sized from host-side bounds and the kernels read V and D on the device.
 * The one exception is a SIZING frame — the first frame of a renderer, or the first after the
 * Gaussian count, image size or band changed — which blocks once in the middle to measure D and
 * size the pair buffers (GS_ERR_PAIR_OVERFLOW if D would exceed 2^32).  Later frames take the
 * capacity from the measured D of earlier frames (25 % head room over the last D, or over the last
 * step extrapolated three frames ahead while D keeps growing; grown lazily: buffer growth calls
 * hipFree, which synchronises the device).  If a frame nevertheless produces more pairs than fit, the
 * device notices before the blend and the frame is SKIPPED — the RGBA target is left untouched, no
 * image with missing splats is ever written — gs_renderer_wait_frame reports GS_ERR_PAIR_CAPACITY
 * (flags bits 0 and 1) and the next gs_render_frame has the larger buffers: a viewer that presents a
 * frame only after wait_frame / flags == 0 shows the previous frame once more, never a wrong one.
 *
 * Frames in flight: a renderer owns the scratch buffers of ONE frame, so consecutive frames on one
 * renderer run one after the other: in stream order on one stream, and when a frame is submitted on a
 * different stream than the renderer's previous frame it is ordered behind that frame's end-of-frame
 * event (a device-side wait, the call does not block).  To overlap frames (the sort chain of frame i + 1 fills the gaps
 * of frame i's blend: +20 % frames per second at 1 M Gaussians), use one gs_renderer per frame slot,
 * each on its own gs_stream, and hand them the frames in turn; the Gaussian buffer is shared (the
 * renderer-internal mirror of the buffer is built on the stream of the first frame that needs it and
 * frames on other streams wait for that build; UPDATING a buffer that frames on other streams may
 * still be reading is the caller's to order, as with any buffer used across streams).
 *
 * Limits (GS_ERR_INVALID_ARGUMENT beyond them): at most 2^22 tiles of 16 x 16 pixels and 65535 tiles
 * along either axis; at most 2^32 - 16 Gaussians. */
gs_status gs_render_frame(gs_renderer *r, gs_stream *s, gs_gaussians_buffer *gaussians,
                          const gs_gaussian_transform_pod *gaussian_transform,
                          const gs_model_transform_pod *model_transform, const gs_camera *camera,
                          uint32_t band_ty0, uint32_t band_ty1, float *rgba_out_device);

typedef struct gs_frame_result {
    uint64_t gaussians;       /* N */
    uint64_t visible;         /* V */
    uint64_t pairs;           /* D (the true count, also when it exceeded the capacity) */
    uint64_t pair_capacity;   /* pairs the renderer's buffers hold */
    uint32_t flags;           /* bit 0: pair capacity exceeded; bit 1: the frame was skipped (image not written);
                               * bit 2: the rank watchdog fired (GS_ERR_RANK_ORDER) */
    uint32_t launches;        /* kernel launches the frame enqueued (after the repack) */
} gs_frame_result;

/* Blocks until the last frame of `r` has completed and reports how it went: GS_OK,
 * GS_ERR_PAIR_CAPACITY (render again), GS_ERR_PAIR_OVERFLOW (> 2^32 pairs) or GS_ERR_HIP.
 * `out` may be NULL. */
gs_status gs_renderer_wait_frame(gs_renderer *r, gs_frame_result *out);

/* How the last frame sorted (blocking like gs_renderer_stats; no reference item: the sorts are the viewer's).  The depth
 * sort runs MSD-first — one scatter on the top 10 bits of the depth key, then one workgroup per bucket finishes the low
 * bits on its CU — while the buckets fit (`bucket_capacity` elements), and as LSD passes while they do not; the renderer
 * chooses per frame from the bucket sizes the previous frames reported.  Both produce the same order. */
typedef struct gs_sort_info {
    uint32_t depth_msd;          /* 1: MSD-first depth sort in the last frame, 0: LSD passes */
    uint32_t depth_bucket_max;   /* largest top-digit bucket the last frame saw (its own report) */
    uint32_t bucket_capacity;    /* elements a bucket may hold for the on-CU path (larger ones take a slow in-kernel fallback) */
    uint32_t tile_msd;           /* the same for the tile sort */
    uint32_t tile_bucket_max;
    uint32_t tile_masks;         /* 1: the last frame dropped unreachable tiles from small rects (tile rect version 4), 0: version 3 */
    uint32_t rounds;             /* rounds the last frame took: 1, or 2 (gs_renderer_set_rounds) */
    uint32_t round1;             /* two rounds: the nearest visible Gaussians its first round covered */
    uint32_t tiles_done;         /* two rounds: tiles the first round finished (every pixel at its final colour) */
    uint32_t partitioned;        /* two rounds: 1 = each round sorted only its own side of a depth threshold */
} gs_sort_info;
gs_status gs_renderer_sort_info(gs_renderer *r, gs_sort_info *out);
/* Pins the choice for the following frames: 1 MSD-first, 0 LSD passes, -1 the renderer chooses (default; the
 * environment variables GS3D_DEPTH_MSD / GS3D_TILE_MSD = 0 / 1 pin it for every renderer of the process).  A pinned
 * MSD-first sort stays correct whatever the bucket sizes (oversized buckets take the in-kernel fallback). */
gs_status gs_renderer_set_sort_mode(gs_renderer *r, int32_t depth_msd, int32_t tile_msd);
/* Tile rect version 4 (no reference item): rects of at most 3 x 3 tiles lose the tiles their splat cannot
 * reach — 5 % fewer (tile, Gaussian) pairs, the same image.  The test costs the preprocess kernel ~120 instructions per
 * Gaussian: hidden where that kernel waits for HBM, a net loss where it does not (50 M x 144 B: +3 %), so by default (-1)
 * it is on for records of 200 bytes or more (f32 SH) in scenes beyond the 256 MiB Infinity Cache; 1 / 0 pin it
 * (GS3D_TILE_MASKS=0/1 pins it for every renderer of the process). */
gs_status gs_renderer_set_tile_masks(gs_renderer *r, int32_t mode);
/* Two-round frames (DESIGN.md 4.2 "rounds"; no reference item: the stages are the viewer's).  A deep scene finishes most
 * of its tiles on the nearest fraction of its Gaussians; the rest is emitted, sorted and staged for nothing.  With two
 * rounds the frame of the nearest `first_round` visible Gaussians (0: a quarter of what the previous frame saw) is
 * rendered first; its blend marks the finished tiles and leaves the pixel state of the others in the image; the second
 * round drops every Gaussian whose rect (at most 3 x 3 tiles) lies in finished tiles and resumes the blend.  The image is
 * the single round's bit for bit; `pairs` of the frame result counts what was emitted (fewer), and the sorted / range
 * taps (gs_renderer_download_sorted, _ranges) refuse such a frame (GS_ERR_INVALID_ARGUMENT).  A frame flagged SKIPPED by its
 * second round has part of the first round's state in the image.  mode: 1 / 0 pin two rounds / one, -1 the renderer chooses
 * (GS3D_ROUNDS=0/1 and GS3D_ROUND1=<count> pin it for every renderer of the process). */
gs_status gs_renderer_set_rounds(gs_renderer *r, int32_t mode, uint32_t first_round);

/* Parity taps on the last frame (blocking).  Sizes: N records / N counts; D keys / D indices;
 * tiles_x*tiles_y*2 ranges. */
gs_status gs_renderer_download_projected(gs_renderer *r, gs_projected *proj_out,
                                         uint32_t *tiles_touched_out, size_t n);
gs_status gs_renderer_download_sorted(gs_renderer *r, uint64_t *keys_out, uint32_t *idx_out,
                                      uint64_t capacity, uint64_t *pairs_out);
gs_status gs_renderer_download_ranges(gs_renderer *r, uint32_t *ranges_out, size_t num_tiles);

/* Stand-alone device primitives used by the frame (also exported for tests and callers):
 * stable LSD radix sort of (u64 key, u32 value) pairs on bits [0, end_bit) — host buffers in/out,
 * blocking; and exclusive prefix sum of u32. */
gs_status gs_sort_pairs_u64(gs_device *dev, gs_stream *s, uint64_t *keys, uint32_t *values,
                            uint64_t count, uint32_t end_bit);
gs_status gs_exclusive_scan_u32(gs_device *dev, gs_stream *s, const uint32_t *in, uint32_t *out,
                                uint64_t count, uint64_t *total_out);

#ifdef __cplusplus
}
#endif
#endif /* GS3D_H */
