// gs3d.hpp — header-only C++17 mirror of wgpu-3dgs-core's Rust API over the C ABI (gs3d.h).
//
// Rust is not available in the build image, so this is the compiled-language host mirror the
// reference's users would recognise: same type names, same methods, Result<_, E> -> exceptions
// carrying the variant fields of src/error.rs.  Citations are relative to the reference.
#pragma once

#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gs3d.h"

namespace gs3d {

// ---- errors (src/error.rs:55-143) --------------------------------------------------------------
struct Error : std::runtime_error {
    gs_status status;
    uint64_t a, b, c;
    Error(const gs_error_info &i) : std::runtime_error(i.message), status(i.code), a(i.a), b(i.b), c(i.c) {}
};
struct GaussiansBufferUpdateError : Error { using Error::Error; uint64_t count() const { return a; } uint64_t expected_count() const { return b; } };
struct GaussiansBufferUpdateRangeError : Error { using Error::Error; uint64_t count() const { return a; } uint64_t start() const { return b; } uint64_t expected_count() const { return c; } };
struct GaussiansBufferTryFromBufferError : Error { using Error::Error; uint64_t buffer_size() const { return a; } uint64_t expected_multiple_size() const { return b; } };
struct FixedSizeBufferWrapperError : Error { using Error::Error; uint64_t buffer_size() const { return a; } uint64_t expected_size() const { return b; } };
struct ComputeBundleCreateError : Error { using Error::Error; };
struct ComputeBundleBuildError : std::runtime_error { using std::runtime_error::runtime_error; };
struct DownloadBufferError : Error { using Error::Error; };
struct LossyConfigError : Error { using Error::Error; };   // where the reference panics
struct NoDeviceError : Error { using Error::Error; };
struct PlyError : Error { using Error::Error; };   // std::io::Error of the PLY reader
struct SpzError : Error { using Error::Error; };   // std::io::Error of the SPZ reader
// gs_renderer_wait_frame: the frame exceeded the pair capacity sized from earlier frames (render again)
struct PairCapacityError : Error { using Error::Error; uint64_t pairs() const { return a; } uint64_t capacity() const { return b; } };
struct RankOrderError : Error { using Error::Error; };   // the radix rank watchdog fired: render again

inline void check(gs_status s) {
    if (s == GS_OK) return;
    gs_error_info i;
    gs_last_error(&i);
    if (i.code != s) { i.code = s; std::strncpy(i.message, gs_status_string(s), sizeof(i.message) - 1); }
    switch (s) {
    case GS_ERR_COUNT_MISMATCH: throw GaussiansBufferUpdateError(i);
    case GS_ERR_RANGE_COUNT_MISMATCH: throw GaussiansBufferUpdateRangeError(i);
    case GS_ERR_BUFFER_SIZE_NOT_MULTIPLE: throw GaussiansBufferTryFromBufferError(i);
    case GS_ERR_BUFFER_SIZE_MISMATCHED: throw FixedSizeBufferWrapperError(i);
    case GS_ERR_RESOURCE_COUNT_MISMATCH:
    case GS_ERR_WORKGROUP_SIZE_EXCEEDS_LIMIT: throw ComputeBundleCreateError(i);
    case GS_ERR_DOWNLOAD: throw DownloadBufferError(i);
    case GS_ERR_LOSSY_CONFIG: throw LossyConfigError(i);
    case GS_ERR_NO_DEVICE: throw NoDeviceError(i);
    case GS_ERR_PLY: throw PlyError(i);
    case GS_ERR_SPZ: throw SpzError(i);
    case GS_ERR_PAIR_CAPACITY: throw PairCapacityError(i);
    case GS_ERR_RANK_ORDER: throw RankOrderError(i);
    default: throw Error(i);
    }
}

using Gaussian = gs_gaussian;   // src/gaussian.rs:53-60

// ---- GaussianPod family (src/buffer/gaussian.rs:239-384) ---------------------------------------
template <gs_sh_config SH, gs_cov3d_config COV>
struct GaussianPod {
    static constexpr gs_sh_config sh = SH;
    static constexpr gs_cov3d_config cov3d = COV;
    static size_t size() { return gs_pod_size(SH, COV); }
    static std::vector<std::pair<std::string, bool>> features() {
        uint8_t f[7];
        check(gs_pod_features(SH, COV, f));
        std::vector<std::pair<std::string, bool>> out;
        for (uint32_t i = 0; i < 7; i++) out.emplace_back(gs_feature_name(i), f[i] != 0);
        return out;
    }
    // G::from_gaussian over a slice -> packed bytes
    static std::vector<uint8_t> from_gaussians(const std::vector<Gaussian> &g) {
        std::vector<uint8_t> pods(g.size() * size());
        check(gs_pack(SH, COV, g.data(), g.size(), pods.data()));
        return pods;
    }
    static std::vector<Gaussian> into_gaussians(const std::vector<uint8_t> &pods) {
        std::vector<Gaussian> g(pods.size() / size());
        check(gs_unpack_to_gaussian(SH, COV, pods.data(), g.size(), g.data()));
        return g;
    }
};
#define GS3D_POD(S, C, NAME) using NAME = GaussianPod<S, C>;
GS3D_POD(GS_SH_SINGLE, GS_COV3D_ROT_SCALE, GaussianPodWithShSingleCov3dRotScaleConfigs)
GS3D_POD(GS_SH_SINGLE, GS_COV3D_SINGLE, GaussianPodWithShSingleCov3dSingleConfigs)
GS3D_POD(GS_SH_SINGLE, GS_COV3D_HALF, GaussianPodWithShSingleCov3dHalfConfigs)
GS3D_POD(GS_SH_HALF, GS_COV3D_ROT_SCALE, GaussianPodWithShHalfCov3dRotScaleConfigs)
GS3D_POD(GS_SH_HALF, GS_COV3D_SINGLE, GaussianPodWithShHalfCov3dSingleConfigs)
GS3D_POD(GS_SH_HALF, GS_COV3D_HALF, GaussianPodWithShHalfCov3dHalfConfigs)
GS3D_POD(GS_SH_NORM8, GS_COV3D_ROT_SCALE, GaussianPodWithShNorm8Cov3dRotScaleConfigs)
GS3D_POD(GS_SH_NORM8, GS_COV3D_SINGLE, GaussianPodWithShNorm8Cov3dSingleConfigs)
GS3D_POD(GS_SH_NORM8, GS_COV3D_HALF, GaussianPodWithShNorm8Cov3dHalfConfigs)
GS3D_POD(GS_SH_NONE, GS_COV3D_ROT_SCALE, GaussianPodWithShNoneCov3dRotScaleConfigs)
GS3D_POD(GS_SH_NONE, GS_COV3D_SINGLE, GaussianPodWithShNoneCov3dSingleConfigs)
GS3D_POD(GS_SH_NONE, GS_COV3D_HALF, GaussianPodWithShNoneCov3dHalfConfigs)
#undef GS3D_POD

// ---- source formats (src/source_format/{ply,spz}.rs) --------------------------------------------
using PlyGaussianPod = gs_ply_gaussian_pod;   // ply.rs:11-21

class PlyGaussians {   // ply.rs:204-431 (both the Inria and Custom variants decode to pods here)
public:
    std::vector<PlyGaussianPod> pods;
    bool inria = true;
    static PlyGaussians read_from(const void *bytes, size_t len) {
        PlyGaussians p;
        size_t n = 0;
        int32_t is_inria = 0;
        check(gs_ply_read(bytes, len, nullptr, 0, &n, &is_inria));
        p.pods.resize(n);
        check(gs_ply_read(bytes, len, p.pods.data(), n, &n, &is_inria));
        p.inria = is_inria != 0;
        return p;
    }
    static PlyGaussians from_gaussians(const std::vector<Gaussian> &g) {   // FromIterator<Gaussian>
        PlyGaussians p;
        p.pods.resize(g.size());
        gs_gaussian_to_ply(g.data(), g.size(), p.pods.data());
        return p;
    }
    std::vector<uint8_t> write_to() const {
        size_t n = 0;
        check(gs_ply_write(pods.data(), pods.size(), nullptr, 0, &n));
        std::vector<uint8_t> out(n);
        check(gs_ply_write(pods.data(), pods.size(), out.data(), out.size(), &n));
        return out;
    }
    std::vector<Gaussian> iter_gaussian() const {   // IterGaussian
        std::vector<Gaussian> g(pods.size());
        gs_gaussian_from_ply(pods.data(), pods.size(), g.data());
        return g;
    }
    size_t len() const { return pods.size(); }
    bool is_empty() const { return pods.empty(); }
};

struct SpzGaussiansFromGaussianSliceOptions : gs_spz_options {   // spz.rs:962-999
    SpzGaussiansFromGaussianSliceOptions() { gs_spz_options_default(this); }
};

class SpzGaussians {   // spz.rs:514-959: keeps the decompressed payload (header + columns) as read / encoded
public:
    gs_spz_header header{};
    std::vector<uint8_t> payload;       // what write_decompressed emits: reading then writing keeps the columns
    std::vector<Gaussian> gaussians;    // Gaussian::from_spz of every point (what iter_gaussian yields)

    explicit SpzGaussians(std::vector<uint8_t> decompressed) : payload(std::move(decompressed)) {
        size_t n = 0;
        check(gs_spz_decode_decompressed(payload.data(), payload.size(), &header, nullptr, 0, &n));
        gaussians.resize(n);
        check(gs_spz_decode_decompressed(payload.data(), payload.size(), &header, gaussians.data(), n, &n));
    }
    static SpzGaussians read_from(const void *bytes, size_t len) { return SpzGaussians(sized(gs_spz_decompress, bytes, len)); }
    static SpzGaussians read_decompressed(const void *bytes, size_t len) {
        return SpzGaussians(std::vector<uint8_t>((const uint8_t *)bytes, (const uint8_t *)bytes + len));
    }
    static SpzGaussians from_gaussians_with_options(const std::vector<Gaussian> &g, const gs_spz_options &o = SpzGaussiansFromGaussianSliceOptions()) {
        return SpzGaussians(write_gaussians_decompressed(g, o));
    }
    static SpzGaussians from_gaussians(const std::vector<Gaussian> &g) { return from_gaussians_with_options(g); }
    std::vector<uint8_t> write_to() const { return sized(gs_spz_compress, payload.data(), payload.size()); }
    const std::vector<uint8_t> &write_decompressed() const { return payload; }
    // one-shot helpers over the fused entry points
    static std::vector<uint8_t> write_gaussians(const std::vector<Gaussian> &g, const gs_spz_options &o = SpzGaussiansFromGaussianSliceOptions()) { return sized(gs_spz_encode, g.data(), g.size(), &o); }
    static std::vector<uint8_t> write_gaussians_decompressed(const std::vector<Gaussian> &g, const gs_spz_options &o = SpzGaussiansFromGaussianSliceOptions()) { return sized(gs_spz_encode_decompressed, g.data(), g.size(), &o); }
    const std::vector<Gaussian> &iter_gaussian() const { return gaussians; }
    size_t len() const { return gaussians.size(); }
    bool is_empty() const { return gaussians.empty(); }
    bool operator==(const SpzGaussians &o) const { return payload == o.payload; }

private:
    template <class F, class... A>
    static std::vector<uint8_t> sized(F fn, A... head) {   // the ABI's two-call pattern: size query, then fill
        size_t n = 0;
        check(fn(head..., nullptr, 0, &n));
        std::vector<uint8_t> out(n);
        check(fn(head..., out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
};

// Gaussians / GaussiansSource (src/gaussian.rs:394-548): the format-agnostic read / write over the
// fused entry points gs_gaussians_read / gs_gaussians_write.  Source::Internal cannot be read from or
// written to a byte buffer (the reference's io::Error messages come back in Error::what()).
enum class GaussiansSource : uint32_t { Internal = GS_SOURCE_INTERNAL, Ply = GS_SOURCE_PLY, Spz = GS_SOURCE_SPZ };

class Gaussians {
public:
    GaussiansSource source = GaussiansSource::Internal;
    std::vector<Gaussian> gaussians;

    static Gaussians from_gaussians(std::vector<Gaussian> g) {
        Gaussians out;
        out.gaussians = std::move(g);
        return out;
    }
    static Gaussians read_from(const void *bytes, size_t len, GaussiansSource source) {
        Gaussians out;
        out.source = source;
        size_t n = 0;
        check(gs_gaussians_read(bytes, len, (gs_gaussians_source)source, nullptr, 0, &n));
        out.gaussians.resize(n);
        check(gs_gaussians_read(bytes, len, (gs_gaussians_source)source, out.gaussians.data(), n, &n));
        return out;
    }
    // write in `as` (default: the format the Gaussians were read from)
    std::vector<uint8_t> write_to(GaussiansSource as) const {
        size_t n = 0;
        check(gs_gaussians_write(gaussians.data(), gaussians.size(), (gs_gaussians_source)as, nullptr, 0, &n));
        std::vector<uint8_t> out(n);
        check(gs_gaussians_write(gaussians.data(), gaussians.size(), (gs_gaussians_source)as, out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
    std::vector<uint8_t> write_to() const { return write_to(source); }
    const std::vector<Gaussian> &iter_gaussian() const { return gaussians; }
    size_t len() const { return gaussians.size(); }
    bool is_empty() const { return gaussians.empty(); }
};

// ---- device / stream / buffer --------------------------------------------------------------------
class Device {
  public:
    explicit Device(int ordinal = 0) { check(gs_device_create(ordinal, &h_)); }
    ~Device() { gs_device_destroy(h_); }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    gs_limits limits() const { gs_limits l; check(gs_device_limits(h_, &l)); return l; }
    bool fast_rank() const { return gs_device_fast_rank(h_) != 0; }
    gs_device *raw() const { return h_; }
  private:
    gs_device *h_ = nullptr;
};

class Stream {   // CommandEncoder + queue.submit
  public:
    explicit Stream(Device &d) { check(gs_stream_create(d.raw(), &h_)); }
    Stream(Device &d, int priority) { check(gs_stream_create_with_priority(d.raw(), priority, &h_)); }   // frames in flight: one priority each
    ~Stream() { gs_stream_destroy(h_); }
    Stream(const Stream &) = delete;
    void synchronize() { check(gs_stream_synchronize(h_)); }
    gs_stream *raw() const { return h_; }
  private:
    gs_stream *h_ = nullptr;
};

class Buffer {   // wgpu::Buffer + BufferWrapper (src/buffer/mod.rs:17-102); Clone = handle copy
  public:
    Buffer(Device &d, size_t bytes, const void *init = nullptr) { check(gs_buffer_create(d.raw(), bytes, init, &h_)); }
    explicit Buffer(gs_buffer *retained) : h_(retained) {}
    Buffer(const Buffer &o) : h_(gs_buffer_retain(o.h_)) {}
    Buffer(Buffer &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Buffer &operator=(Buffer o) { std::swap(h_, o.h_); return *this; }
    ~Buffer() { gs_buffer_release(h_); }
    size_t size() const { return gs_buffer_size(h_); }
    void *device_ptr() const { return gs_buffer_device_ptr(h_); }
    void write(Stream &s, size_t offset, const void *src, size_t bytes) { check(gs_buffer_write(h_, s.raw(), offset, src, bytes)); }
    template <class T> std::vector<T> download(Stream &s) const {   // BufferWrapper::download::<T>
        std::vector<T> out(size() / sizeof(T));
        check(gs_buffer_download(h_, s.raw(), out.data(), out.size() * sizeof(T)));
        return out;
    }
    // BufferWrapper::prepare_download / map_download (src/buffer/mod.rs:48-101): the copy is
    // enqueued now and mapped (waited for) later
    class Download {
      public:
        explicit Download(gs_download *d) : d_(d) {}
        Download(Download &&o) noexcept : d_(o.d_) { o.d_ = nullptr; }
        Download(const Download &) = delete;
        ~Download() { gs_download_release(d_); }
        bool ready() const { return gs_download_ready(d_) != 0; }
        template <class T> std::vector<T> map() {
            const void *p = nullptr;
            size_t n = 0;
            check(gs_download_map(d_, &p, &n));
            const T *t = static_cast<const T *>(p);
            return std::vector<T>(t, t + n / sizeof(T));
        }
      private:
        gs_download *d_;
    };
    Download prepare_download(Stream &s) const { gs_download *d = nullptr; check(gs_buffer_prepare_download(h_, s.raw(), &d)); return Download(d); }
    gs_buffer *raw() const { return h_; }
  private:
    gs_buffer *h_ = nullptr;
};

template <class G>
class GaussiansBuffer {   // src/buffer/gaussian.rs:17-229
  public:
    GaussiansBuffer(Device &d, const std::vector<Gaussian> &g) { check(gs_gaussians_buffer_create_from_gaussians(d.raw(), G::sh, G::cov3d, g.data(), g.size(), &h_)); }
    // PlyGaussians -> buffer with Gaussian::from_ply + G::from_gaussian fused in one device kernel
    static GaussiansBuffer new_from_ply(Device &d, const std::vector<PlyGaussianPod> &ply) { GaussiansBuffer b; check(gs_gaussians_buffer_create_from_ply(d.raw(), G::sh, G::cov3d, ply.data(), ply.size(), &b.h_)); return b; }
    // SPZ file bytes -> buffer: host inflate, then Gaussian::from_spz + G::from_gaussian in one device kernel
    static GaussiansBuffer new_from_spz(Device &d, const void *bytes, size_t len) { GaussiansBuffer b; check(gs_gaussians_buffer_create_from_spz(d.raw(), G::sh, G::cov3d, bytes, len, nullptr, &b.h_)); return b; }
    void update_range_from_ply(Stream &s, size_t start, const std::vector<PlyGaussianPod> &ply) { check(gs_gaussians_buffer_update_range_ply(h_, s.raw(), start, ply.data(), ply.size())); }
    static GaussiansBuffer new_with_pods(Device &d, const std::vector<uint8_t> &pods) { GaussiansBuffer b; check(gs_gaussians_buffer_create(d.raw(), G::sh, G::cov3d, pods.data(), pods.size() / G::size(), &b.h_)); return b; }
    static GaussiansBuffer new_empty(Device &d, size_t len) { GaussiansBuffer b; check(gs_gaussians_buffer_create(d.raw(), G::sh, G::cov3d, nullptr, len, &b.h_)); return b; }
    static GaussiansBuffer try_from(const Buffer &buf) { GaussiansBuffer b; check(gs_gaussians_buffer_from_buffer(buf.raw(), G::sh, G::cov3d, &b.h_)); return b; }
    GaussiansBuffer(GaussiansBuffer &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    GaussiansBuffer(const GaussiansBuffer &) = delete;
    ~GaussiansBuffer() { gs_gaussians_buffer_destroy(h_); }
    size_t len() const { return gs_gaussians_buffer_len(h_); }
    bool is_empty() const { return len() == 0; }
    Buffer buffer() const { return Buffer(gs_buffer_retain(gs_gaussians_buffer_buffer(h_))); }
    void update(Stream &s, const std::vector<Gaussian> &g) { check(gs_gaussians_buffer_update_gaussians(h_, s.raw(), g.data(), g.size())); }
    void update_with_pod(Stream &s, const std::vector<uint8_t> &pods) { check(gs_gaussians_buffer_update(h_, s.raw(), pods.data(), pods.size() / G::size())); }
    void update_range(Stream &s, size_t start, const std::vector<Gaussian> &g) { check(gs_gaussians_buffer_update_range_gaussians(h_, s.raw(), start, g.data(), g.size())); }
    void update_range_with_pod(Stream &s, size_t start, const std::vector<uint8_t> &pods) { check(gs_gaussians_buffer_update_range(h_, s.raw(), start, pods.data(), pods.size() / G::size())); }
    std::vector<uint8_t> download(Stream &s) const { std::vector<uint8_t> out(len() * G::size()); check(gs_gaussians_buffer_download(h_, s.raw(), out.data(), len())); return out; }
    std::vector<Gaussian> download_gaussians(Stream &s) const { std::vector<Gaussian> out(len()); check(gs_gaussians_buffer_download_gaussians(h_, s.raw(), out.data(), len())); return out; }
    // renderer-side mirror order (gs3d.h: spatial by default); order[slot] = Gaussian index
    void set_spatial_order(bool enabled) { check(gs_gaussians_buffer_set_spatial_order(h_, enabled ? 1 : 0)); }
    bool spatial_order() const { return gs_gaussians_buffer_spatial_order(h_) != 0; }
    std::vector<uint32_t> download_order(Stream &s) const { std::vector<uint32_t> out(len()); check(gs_gaussians_buffer_download_order(h_, s.raw(), out.data(), out.size())); return out; }
    void mark_dirty() { gs_gaussians_buffer_mark_dirty(h_); }
    gs_gaussians_buffer *raw() const { return h_; }
  private:
    GaussiansBuffer() = default;
    gs_gaussians_buffer *h_ = nullptr;
};

// GaussianTransformPod helpers (src/buffer/gaussian_transform.rs)
inline std::optional<gs_gaussian_transform_pod> gaussian_transform_pod(float size, gs_display_mode mode, uint8_t sh_deg, bool no_sh0, float max_std_dev) {
    gs_gaussian_transform_pod p;
    if (gs_gaussian_transform_pod_new(size, mode, sh_deg, no_sh0, max_std_dev, &p) != GS_OK) return std::nullopt;
    return p;
}
struct GaussianTransformBuffer : Buffer {   // :104-163
    explicit GaussianTransformBuffer(Device &d) : Buffer(make(d)) {}
    void update_with_pod(Stream &s, const gs_gaussian_transform_pod &p) { check(gs_gaussian_transform_buffer_update(raw(), s.raw(), &p)); }
    static GaussianTransformBuffer try_from(const Buffer &b) { check(gs_gaussian_transform_buffer_from_buffer(b.raw())); return GaussianTransformBuffer(b); }
  private:
    explicit GaussianTransformBuffer(const Buffer &b) : Buffer(b) {}
    static gs_buffer *make(Device &d) { gs_buffer *b; check(gs_gaussian_transform_buffer_create(d.raw(), &b)); return b; }
};
struct ModelTransformBuffer : Buffer {      // src/buffer/model_transform.rs:10-58
    explicit ModelTransformBuffer(Device &d) : Buffer(make(d)) {}
    void update(Stream &s, const float pos[3], const float rot[4], const float scale[3]) { gs_model_transform_pod p; gs_model_transform_pod_new(pos, rot, scale, &p); check(gs_model_transform_buffer_update(raw(), s.raw(), &p)); }
    static ModelTransformBuffer try_from(const Buffer &b) { check(gs_model_transform_buffer_from_buffer(b.raw())); return ModelTransformBuffer(b); }
  private:
    explicit ModelTransformBuffer(const Buffer &b) : Buffer(b) {}
    static gs_buffer *make(Device &d) { gs_buffer *b; check(gs_model_transform_buffer_create(d.raw(), &b)); return b; }
};

// ---- ComputeBundle (src/compute_bundle.rs) -------------------------------------------------------
class ComputeBundle {
  public:
    ComputeBundle(gs_bundle *h, bool managed) : h_(h), managed_(managed) {}
    ComputeBundle(ComputeBundle &&o) noexcept : h_(o.h_), managed_(o.managed_) { o.h_ = nullptr; }
    ~ComputeBundle() { gs_bundle_destroy(h_); }
    uint32_t workgroup_size() const { return gs_bundle_workgroup_size(h_); }
    std::optional<std::string> label() const { const char *l = gs_bundle_label(h_); return l ? std::optional<std::string>(l) : std::nullopt; }
    void dispatch(Stream &s, uint32_t count) { check(gs_bundle_dispatch(h_, s.raw(), count)); }
    void dispatch(Stream &s, uint32_t count, const std::vector<std::vector<const Buffer *>> &groups) {
        std::vector<std::vector<gs_buffer *>> raw(groups.size());
        std::vector<gs_buffer *const *> ptrs;
        std::vector<uint32_t> counts;
        for (size_t i = 0; i < groups.size(); i++) {
            for (auto *b : groups[i]) raw[i].push_back(b->raw());
            ptrs.push_back(raw[i].data());
            counts.push_back((uint32_t)raw[i].size());
        }
        check(gs_bundle_dispatch_with_bind_groups(h_, s.raw(), count, ptrs.data(), counts.data(), (uint32_t)groups.size()));
    }
    uint32_t last_workgroup_count() const { return gs_bundle_last_workgroup_count(h_); }
  private:
    gs_bundle *h_;
    bool managed_;
};

class ComputeBundleBuilder {   // :364-593; required fields checked in the reference's order (:505-519)
  public:
    ComputeBundleBuilder &label(std::string l) { label_ = std::move(l); return *this; }
    ComputeBundleBuilder &bind_group_layout(uint32_t bindings) { layouts_.push_back(bindings); return *this; }
    ComputeBundleBuilder &resolver() { resolver_ = true; return *this; }   // the built-in kernel registry
    ComputeBundleBuilder &entry_point(std::string e) { entry_ = std::move(e); return *this; }
    ComputeBundleBuilder &main_shader(gs_kernel_id k) { kernel_ = k; return *this; }
    template <class G> ComputeBundleBuilder &features() { sh_ = G::sh; cov_ = G::cov3d; return *this; }
    ComputeBundleBuilder &workgroup_size(uint32_t w) { wg_ = w; return *this; }
    ComputeBundleBuilder &constant(std::string name, double v) { cnames_.push_back(std::move(name)); cvals_.push_back(v); return *this; }
    ComputeBundle build(Device &d, const std::vector<std::vector<const Buffer *>> &resources) {
        validate();
        gs_bundle_desc desc = make_desc();
        std::vector<std::vector<gs_buffer *>> raw(resources.size());
        std::vector<gs_buffer *const *> ptrs;
        std::vector<uint32_t> counts;
        for (size_t i = 0; i < resources.size(); i++) {
            for (auto *b : resources[i]) raw[i].push_back(b->raw());
            ptrs.push_back(raw[i].data());
            counts.push_back((uint32_t)raw[i].size());
        }
        gs_bundle *h;
        check(gs_bundle_create_with_bind_groups(d.raw(), &desc, ptrs.data(), counts.data(), (uint32_t)resources.size(), &h));
        return ComputeBundle(h, true);
    }
    ComputeBundle build_without_bind_groups(Device &d) {
        validate();
        gs_bundle_desc desc = make_desc();
        gs_bundle *h;
        check(gs_bundle_create(d.raw(), &desc, &h));
        return ComputeBundle(h, false);
    }
  private:
    void validate() const {
        if (layouts_.empty()) throw ComputeBundleBuildError("missing bind group layout for compute bundle");
        if (!resolver_) throw ComputeBundleBuildError("missing resolver for compute bundle");
        if (!entry_) throw ComputeBundleBuildError("missing entry point for compute bundle");
        if (!kernel_) throw ComputeBundleBuildError("missing main shader for compute bundle");
    }
    gs_bundle_desc make_desc() {
        cptrs_.clear();
        for (auto &n : cnames_) cptrs_.push_back(n.c_str());
        gs_bundle_desc d{};
        d.label = label_ ? label_->c_str() : nullptr;
        d.kernel = *kernel_;
        d.sh = sh_;
        d.cov = cov_;
        d.bind_group_count = (uint32_t)layouts_.size();
        d.bindings_per_group = layouts_.data();
        d.workgroup_size = wg_;
        d.constant_names = cptrs_.data();
        d.constant_values = cvals_.data();
        d.constant_count = (uint32_t)cvals_.size();
        return d;
    }
    std::optional<std::string> label_, entry_;
    std::vector<uint32_t> layouts_;
    bool resolver_ = false;
    std::optional<gs_kernel_id> kernel_;
    gs_sh_config sh_ = GS_SH_SINGLE;
    gs_cov3d_config cov_ = GS_COV3D_ROT_SCALE;
    uint32_t wg_ = 0;
    std::vector<std::string> cnames_;
    std::vector<const char *> cptrs_;
    std::vector<double> cvals_;
};

// ---- renderer ------------------------------------------------------------------------------------
class Renderer {
  public:
    explicit Renderer(Device &d) { check(gs_renderer_create(d.raw(), &h_)); }
    ~Renderer() { gs_renderer_destroy(h_); }
    Renderer(const Renderer &) = delete;
    template <class G>
    void render(Stream &s, GaussiansBuffer<G> &g, const gs_gaussian_transform_pod &gt, const gs_model_transform_pod &mt,
                const gs_camera &cam, float *rgba_device, uint32_t band_ty0 = 0, uint32_t band_ty1 = 0xffffffffu) {
        check(gs_render_frame(h_, s.raw(), g.raw(), &gt, &mt, &cam, band_ty0, band_ty1, rgba_device));
    }
    // render() only enqueues; wait_frame() blocks until the frame is complete and throws
    // PairCapacityError when it exceeded the pair capacity (the next frame has larger buffers)
    gs_frame_result wait_frame() { gs_frame_result fr; check(gs_renderer_wait_frame(h_, &fr)); return fr; }
    gs_frame_stats stats() { gs_frame_stats st; check(gs_renderer_stats(h_, &st)); return st; }
    // how the last frame sorted / how many rounds it took, and the pins of those per-frame choices (-1: the renderer decides;
    // every setting renders the same image; gs3d.h has the details)
    gs_sort_info sort_info() { gs_sort_info si; check(gs_renderer_sort_info(h_, &si)); return si; }
    void set_sort_mode(int32_t depth_msd, int32_t tile_msd = -1) { check(gs_renderer_set_sort_mode(h_, depth_msd, tile_msd)); }
    void set_tile_masks(int32_t mode) { check(gs_renderer_set_tile_masks(h_, mode)); }
    void set_rounds(int32_t mode, uint32_t first_round = 0) { check(gs_renderer_set_rounds(h_, mode, first_round)); }
    // sharded frames: a DEVICE word that receives every following frame's flags in stream order (nullptr: off)
    void set_frame_flags_target(uint32_t *device_word) { check(gs_renderer_set_frame_flags_target(h_, device_word)); }
  private:
    gs_renderer *h_ = nullptr;
};

// Frames in flight (no reference item: the viewer owns the frame loop; the seam is src/compute_bundle.rs:196-198 —
// dispatch only records, the caller owns the submission order).  `frames` renderers, each on a stream of its own
// PRIORITY, take the frames in turn: what a viewer with double / triple buffering does.  A frame is a chain of
// dependent kernels, latency-bound in its sorts and VALU-bound in its blend, so frames of different renderers overlap
// on the device when their streams sit on different hardware queues.  HIP keeps separate queues per priority LEVEL and
// may put streams of one level on one queue — observed, not promised: where the lanes share a queue they run one after
// the other and the ring costs what one stream costs.  Every frame is the frame its renderer would
// have rendered alone.  Presentation in order: render() returns the lane, wait(lane) blocks until THAT lane's newest
// frame is complete (lanes complete out of order when frames differ in cost).
class FrameRing {
  public:
    explicit FrameRing(Device &d, size_t frames = 3) {
        int32_t least = 0, greatest = 0;
        check(gs_device_stream_priority_range(d.raw(), &least, &greatest));
        const int32_t cycle3[3] = {greatest, least, 0};
        if (frames == 0) frames = 1;
        for (size_t k = 0; k < frames; k++) {
            const int32_t prio = least != greatest ? cycle3[k % 3] : 0;
            priorities_.push_back(prio);
            streams_.push_back(std::make_unique<Stream>(d, (int)prio));
            renderers_.push_back(std::make_unique<Renderer>(d));
        }
    }
    size_t size() const { return renderers_.size(); }
    // enqueues one frame on the next lane and returns the lane (its stream: stream(lane))
    template <class G>
    size_t render(GaussiansBuffer<G> &g, const gs_gaussian_transform_pod &gt, const gs_model_transform_pod &mt, const gs_camera &cam,
                  float *rgba_device, uint32_t band_ty0 = 0, uint32_t band_ty1 = 0xffffffffu) {
        const size_t lane = next_++ % renderers_.size();
        renderers_[lane]->render(*streams_[lane], g, gt, mt, cam, rgba_device, band_ty0, band_ty1);
        return lane;
    }
    // blocks until the newest frame of `lane` is complete; throws as Renderer::wait_frame does
    gs_frame_result wait(size_t lane) { return renderers_.at(lane)->wait_frame(); }
    void synchronize() { for (auto &s : streams_) s->synchronize(); }
    Stream &stream(size_t lane) { return *streams_.at(lane); }
    Renderer &renderer(size_t lane) { return *renderers_.at(lane); }
    int32_t priority(size_t lane) const { return priorities_.at(lane); }
  private:
    // (renderers are destroyed before their streams: gs_renderer_destroy synchronises the stream of its last frame)
    std::vector<std::unique_ptr<Stream>> streams_;
    std::vector<std::unique_ptr<Renderer>> renderers_;
    std::vector<int32_t> priorities_;
    size_t next_ = 0;
};

// HIP version of the headers the library was compiled with / of the runtime / of the driver it runs on
struct HipVersions { int32_t compiled, runtime, driver; };
inline HipVersions hip_versions() { HipVersions v{0, 0, 0}; gs_hip_versions(&v.compiled, &v.runtime, &v.driver); return v; }

// G::from_gaussian on the device: `count` struct Gaussian records in `gaussians` -> PODs in `pods`
template <class G>
inline void pack_device(Device &d, Stream &s, const Buffer &gaussians, size_t count, Buffer &pods) {
    check(gs_pack_device(d.raw(), s.raw(), G::sh, G::cov3d, static_cast<const gs_gaussian *>(gaussians.device_ptr()), count,
                         pods.device_ptr()));
}

}  // namespace gs3d
