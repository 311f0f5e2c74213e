// gs3d.hip — host side of libgs3d_hip.so: the C ABI of include/gs3d.h over the gfx950 kernels.
// Built by hipcc --offload-arch=gfx950 -ffp-contract=off (see build.py).  No CPU fallback: every
// compute entry point needs a HIP device.
#include "gs_internal.h"

#include <dlfcn.h>

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gs_bundle_kernels.h"
#include "gs_render_kernels.h"
#include "gs_pack_kernels.h"
#include "_gen_kernel_lib_src.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------

static thread_local gs_error_info t_err = {0, 0, 0, 0, {0}};

gs_status gs_fail(gs_status code, uint64_t a, uint64_t b, uint64_t c, const char *fmt, ...) {
    t_err.code = code;
    t_err.a = a;
    t_err.b = b;
    t_err.c = c;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err.message, sizeof(t_err.message), fmt, ap);
    va_end(ap);
    return code;
}

#define GS_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? GS_ERR_OUT_OF_MEMORY : GS_ERR_HIP,            \
                        (uint64_t)e_, 0, 0, "%s failed: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

#define GS_TRY(expr)                  \
    do {                              \
        gs_status s_ = (expr);        \
        if (s_ != GS_OK) return s_;   \
    } while (0)

#define fail gs_fail

extern "C" void gs_last_error(gs_error_info *out) {
    if (out) *out = t_err;
}

extern "C" uint32_t gs_abi_version(void) { return GS3D_ABI_VERSION; }

extern "C" void gs_hip_versions(int32_t *compiled, int32_t *runtime, int32_t *driver) {
    if (compiled) *compiled = HIP_VERSION;
    int v = 0;
    if (runtime) *runtime = hipRuntimeGetVersion(&v) == hipSuccess ? v : 0;
    v = 0;
    if (driver) *driver = hipDriverGetVersion(&v) == hipSuccess ? v : 0;
}

extern "C" const char *gs_status_string(gs_status s) {
    switch (s) {
    case GS_OK: return "ok";
    case GS_ERR_INVALID_ARGUMENT: return "invalid argument";
    case GS_ERR_NO_DEVICE: return "no HIP device";
    case GS_ERR_HIP: return "HIP runtime error";
    case GS_ERR_OUT_OF_MEMORY: return "out of device memory";
    case GS_ERR_COUNT_MISMATCH: return "Gaussians count mismatch";
    case GS_ERR_RANGE_COUNT_MISMATCH: return "Gaussians range count mismatch";
    case GS_ERR_BUFFER_SIZE_NOT_MULTIPLE: return "buffer size and expected multiple size mismatch";
    case GS_ERR_BUFFER_SIZE_MISMATCHED: return "buffer size and expected size mismatch";
    case GS_ERR_RESOURCE_COUNT_MISMATCH: return "resource count and bind group layout count mismatch";
    case GS_ERR_WORKGROUP_SIZE_EXCEEDS_LIMIT: return "workgroup size exceeds device limit";
    case GS_ERR_MISSING_BIND_GROUP_LAYOUT: return "missing bind group layout for compute bundle";
    case GS_ERR_MISSING_RESOLVER: return "missing resolver for compute bundle";
    case GS_ERR_MISSING_ENTRY_POINT: return "missing entry point for compute bundle";
    case GS_ERR_MISSING_MAIN_SHADER: return "missing main shader for compute bundle";
    case GS_ERR_KERNEL_COMPILE: return "kernel compilation failed";
    case GS_ERR_LOSSY_CONFIG: return "configuration cannot be converted back to a Gaussian";
    case GS_ERR_DOWNLOAD: return "buffer download failed";
    case GS_ERR_PAIR_OVERFLOW: return "more than 2^32 (tile, Gaussian) pairs";
    case GS_ERR_PAIR_CAPACITY: return "pair capacity exceeded; render again";
    case GS_ERR_RANK_ORDER: return "radix rank watchdog fired; device switched to the ballot-based rank; render again";
    case GS_ERR_PLY: return "PLY read error";
    case GS_ERR_SPZ: return "SPZ read error";
    default: return "unknown";
    }
}

// ------------------------------------------------------------------------------------------------
// host-side data model: packing (G::from_gaussian) — runs on the host in the reference too
// ------------------------------------------------------------------------------------------------

static bool valid_cfg(int sh, int cov) { return sh >= 0 && sh <= 3 && cov >= 0 && cov <= 2; }

extern "C" size_t gs_pod_size(gs_sh_config sh, gs_cov3d_config cov) {
    if (!valid_cfg(sh, cov)) return 0;
    return (size_t)gs::pod_bytes(sh, cov);
}

static const char *k_feature_names[7] = {"sh_single", "sh_half", "sh_norm8", "sh_none",
                                         "cov3d_rot_scale", "cov3d_single", "cov3d_half"};

extern "C" const char *gs_feature_name(uint32_t index) {
    return index < 7 ? k_feature_names[index] : nullptr;
}

extern "C" gs_status gs_pod_features(gs_sh_config sh, gs_cov3d_config cov, uint8_t out[7]) {
    if (!valid_cfg(sh, cov) || !out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "bad config");
    for (int i = 0; i < 7; i++) out[i] = (i == (int)sh) || (i == 4 + (int)cov);
    return GS_OK;
}

static float f16_to_f32_host(uint16_t h) {
    union { uint32_t u; float f; } o;
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        o.f = std::ldexp((float)m, -24);
        o.u |= sign;
    } else if (e == 31) {
        o.u = sign | 0x7f800000u | (m << 13);
    } else {
        o.u = sign | ((e + 112u) << 23) | (m << 13);
    }
    return o.f;
}

template <class F>
static void parallel_for(size_t n, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    size_t threads = n < 65536 ? 1 : (hw ? (hw > 32 ? 32 : hw) : 4);
    if (const char *e = std::getenv("GS3D_HOST_THREADS")) threads = std::atoi(e) > 0 ? (size_t)std::atoi(e) : threads;
    if (threads <= 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    size_t per = (n + threads - 1) / threads;
    for (size_t t = 0; t < threads; t++) {
        size_t a = t * per, b = a + per < n ? a + per : n;
        if (a >= b) break;
        pool.emplace_back([=] { fn(a, b); });
    }
    for (auto &th : pool) th.join();
}

// G::from_gaussian on the host: the same encoders the device kernel uses (gs_pack_kernels.h)
static void pack_one(int sh, int cov, const gs_gaussian &g, uint8_t *p, size_t stride) {
    static_assert(sizeof(gs_gaussian) == gs::GAUSSIAN_WORDS * 4, "gs_gaussian layout");
    uint32_t gw[gs::GAUSSIAN_WORDS], pw[56];
    std::memcpy(gw, &g, sizeof(gw));
    gs::pack_words(sh, cov, gw, pw);
    std::memcpy(p, pw, stride);
}

extern "C" gs_status gs_pack(gs_sh_config sh, gs_cov3d_config cov, const gs_gaussian *in, size_t n,
                             void *out) {
    if (!valid_cfg(sh, cov) || (n && (!in || !out)))
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "gs_pack: bad argument");
    size_t stride = gs_pod_size(sh, cov);
    uint8_t *o = (uint8_t *)out;
    parallel_for(n, [=](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) pack_one(sh, cov, in[i], o + i * stride, stride);
    });
    return GS_OK;
}

extern "C" gs_status gs_unpack_to_gaussian(gs_sh_config sh, gs_cov3d_config cov, const void *pods,
                                           size_t n, gs_gaussian *out) {
    if (!valid_cfg(sh, cov) || (n && (!pods || !out)))
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "gs_unpack_to_gaussian: bad argument");
    if (sh == GS_SH_NONE)
        return fail(GS_ERR_LOSSY_CONFIG, 0, 0, 0, "Cannot convert from SH None configuration");
    if (cov == GS_COV3D_SINGLE)
        return fail(GS_ERR_LOSSY_CONFIG, 0, 0, 0, "Cannot convert from Cov3d Single configuration");
    if (cov == GS_COV3D_HALF)
        return fail(GS_ERR_LOSSY_CONFIG, 0, 0, 0, "Cannot convert from Cov3d Half configuration");
    size_t stride = gs_pod_size(sh, cov);
    const uint8_t *in = (const uint8_t *)pods;
    parallel_for(n, [=](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            uint32_t pw[56], gw[gs::GAUSSIAN_WORDS];
            std::memcpy(pw, in + i * stride, stride);
            gs::unpack_words(sh, pw, gw);       // the same code the device kernel runs (k_unpack_pods)
            std::memcpy(&out[i], gw, sizeof(gw));
        }
    });
    return GS_OK;
}

extern "C" gs_status gs_max_std_dev_encode(float v, uint8_t *out) {
    if (!out || !(v >= 0.0f && v <= 3.0f))
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "max_std_dev must be in [0, 3]");
    *out = (uint8_t)(v / 3.0f * 255.0f);
    return GS_OK;
}

extern "C" float gs_max_std_dev_decode(uint8_t v) { return (float)v / 255.0f * 3.0f; }

extern "C" gs_status gs_gaussian_transform_pod_new(float size, gs_display_mode mode, uint8_t sh_deg,
                                                   uint8_t no_sh0, float max_std_dev,
                                                   gs_gaussian_transform_pod *out) {
    if (!out || (int)mode < 0 || (int)mode > 2)
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "bad display mode");
    if (sh_deg > 3) return fail(GS_ERR_INVALID_ARGUMENT, sh_deg, 0, 0, "SH degree must be in [0, 3]");
    uint8_t sd;
    GS_TRY(gs_max_std_dev_encode(max_std_dev, &sd));
    out->size = size;
    out->flags[0] = (uint8_t)mode;
    out->flags[1] = sh_deg;
    out->flags[2] = no_sh0 ? 1 : 0;
    out->flags[3] = sd;
    return GS_OK;
}

extern "C" void gs_gaussian_transform_pod_default(gs_gaussian_transform_pod *out) {
    gs_gaussian_transform_pod_new(1.0f, GS_DISPLAY_SPLAT, 3, 0, 3.0f, out);
}

extern "C" void gs_model_transform_pod_new(const float pos[3], const float rot[4],
                                           const float scale[3], gs_model_transform_pod *out) {
    std::memset(out, 0, sizeof(*out));
    std::memcpy(out->pos, pos, 12);
    std::memcpy(out->rot, rot, 16);
    std::memcpy(out->scale, scale, 12);
}

extern "C" void gs_model_transform_pod_default(gs_model_transform_pod *out) {
    const float p[3] = {0, 0, 0}, r[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    gs_model_transform_pod_new(p, r, s, out);
}

// ------------------------------------------------------------------------------------------------
// device / stream / buffer
// ------------------------------------------------------------------------------------------------

struct gs_device {
    int ordinal;
    gs_limits limits;
    hipStream_t internal;  // for blocking helper work
    // probe result: returning LDS atomics hand out lane-ordered values.  Written by any renderer of the device whose rank
    // watchdog fired (gs_render_frame / gs_renderer_wait_frame, possibly from different host threads) and read once per
    // frame (t_rank_fault) and by the stand-alone sorts: atomic, relaxed — it only ever goes from true to false.
    std::atomic<bool> lds_atomic_ordered{false};
    // renderers of this device: gs_stream_destroy records the end-of-frame event of those whose last frame sits on the
    // stream that is going away (the event is otherwise recorded lazily, when the renderer moves to another stream)
    std::mutex renderers_mu;
    std::vector<struct gs_renderer *> renderers;
};

struct gs_stream {
    gs_device *dev;
    hipStream_t s;
    bool owned;
};

struct gs_buffer {
    gs_device *dev;
    void *ptr;
    size_t bytes;
    bool owned;
    std::atomic<int> refs;
};

static gs_status use_device(const gs_device *dev) {
    if (!dev) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null device");
    GS_HIP(hipSetDevice(dev->ordinal));
    // HIP's "last error" is sticky per thread: drop whatever an earlier, already reported failure
    // left behind so that the hipGetLastError() checks after our launches only see our launches
    (void)hipGetLastError();
    return GS_OK;
}

extern "C" gs_status gs_device_create(int32_t ordinal, gs_device **out) {
    if (!out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null out");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(GS_ERR_NO_DEVICE, (uint64_t)e, 0, 0, "no HIP device available (%s)",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (ordinal < 0 || ordinal >= count)
        return fail(GS_ERR_NO_DEVICE, 0, 0, 0, "device ordinal %d out of range [0, %d)", ordinal, count);
    GS_HIP(hipSetDevice(ordinal));
    hipDeviceProp_t props;
    GS_HIP(hipGetDeviceProperties(&props, ordinal));
    gs_device *d = new gs_device();
    d->ordinal = ordinal;
    std::memset(&d->limits, 0, sizeof(d->limits));
    d->limits.max_compute_workgroup_size_x = (uint32_t)props.maxThreadsDim[0];
    d->limits.max_compute_invocations_per_workgroup = (uint32_t)props.maxThreadsPerBlock;
    d->limits.compute_units = (uint32_t)props.multiProcessorCount;
    d->limits.wavefront_size = (uint32_t)props.warpSize;
    d->limits.total_memory_bytes = (uint64_t)props.totalGlobalMem;
    std::snprintf(d->limits.arch_name, sizeof(d->limits.arch_name), "%s", props.gcnArchName);
    hipError_t se = hipStreamCreateWithFlags(&d->internal, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete d;
        return fail(GS_ERR_HIP, (uint64_t)se, 0, 0, "hipStreamCreate failed: %s", hipGetErrorString(se));
    }
    // probe the LDS-atomic ordering the fast radix ranking relies on (gs_render_kernels.h)
    d->lds_atomic_ordered = false;
    if (!std::getenv("GS3D_DISABLE_FAST_RANK")) {
        uint32_t *bad = nullptr;
        if (hipMalloc((void **)&bad, 4) == hipSuccess) {
            uint32_t h = 1;
            if (hipMemset(bad, 0, 4) == hipSuccess) {
                // both digit widths of the sorts (256 and 512 counters per wave), the scatter's own access pattern
                hipLaunchKernelGGL(gs::k_probe_lds_atomic_order<8>, dim3(512), dim3(gs::SORT_THREADS), 0, d->internal,
                                   8u, 0x3D650001u, bad);
                hipLaunchKernelGGL(gs::k_probe_lds_atomic_order<9>, dim3(512), dim3(gs::SORT_THREADS), 0, d->internal,
                                   8u, 0x3D650002u, bad);
                if (hipStreamSynchronize(d->internal) == hipSuccess &&
                    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost) == hipSuccess)
                    d->lds_atomic_ordered = (h == 0);
            }
            (void)hipFree(bad);
        }
        (void)hipGetLastError();
    }
    *out = d;
    return GS_OK;
}

extern "C" int32_t gs_device_fast_rank(const gs_device *dev) { return dev && dev->lds_atomic_ordered.load() ? 1 : 0; }

extern "C" void gs_device_destroy(gs_device *dev) {
    if (!dev) return;
    (void)hipSetDevice(dev->ordinal);
    (void)hipStreamDestroy(dev->internal);
    delete dev;
}

extern "C" gs_status gs_device_limits(const gs_device *dev, gs_limits *out) {
    if (!dev || !out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    *out = dev->limits;
    return GS_OK;
}

extern "C" gs_status gs_device_synchronize(gs_device *dev) {
    GS_TRY(use_device(dev));
    GS_HIP(hipDeviceSynchronize());
    return GS_OK;
}

extern "C" gs_status gs_stream_create(gs_device *dev, gs_stream **out) {
    if (!out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null out");
    GS_TRY(use_device(dev));
    hipStream_t s;
    GS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = new gs_stream{dev, s, true};
    return GS_OK;
}

extern "C" gs_status gs_device_stream_priority_range(gs_device *dev, int32_t *least, int32_t *greatest) {
    GS_TRY(use_device(dev));
    int lo = 0, hi = 0;
    GS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (least) *least = lo;
    if (greatest) *greatest = hi;
    return GS_OK;
}

extern "C" gs_status gs_stream_create_with_priority(gs_device *dev, int32_t priority, gs_stream **out) {
    if (!out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null out");
    GS_TRY(use_device(dev));
    int lo = 0, hi = 0;
    GS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (priority > lo || priority < hi)
        return fail(GS_ERR_INVALID_ARGUMENT, (uint64_t)(int64_t)priority, 0, 0, "stream priority %d outside [%d (least), %d (greatest)]",
                    priority, lo, hi);
    hipStream_t s;
    GS_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
    *out = new gs_stream{dev, s, true};
    return GS_OK;
}

extern "C" gs_status gs_stream_wrap(gs_device *dev, void *hip_stream, gs_stream **out) {
    if (!dev || !out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    *out = new gs_stream{dev, (hipStream_t)hip_stream, false};
    return GS_OK;
}

extern "C" void *gs_stream_native(const gs_stream *s) { return s ? (void *)s->s : nullptr; }

extern "C" gs_status gs_stream_synchronize(gs_stream *s) {
    if (!s) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null stream");
    GS_TRY(use_device(s->dev));
    GS_HIP(hipStreamSynchronize(s->s));
    return GS_OK;
}

static void renderers_leave_stream(gs_device *dev, hipStream_t st);

extern "C" void gs_stream_destroy(gs_stream *s) {
    if (!s) return;
    // A renderer whose last frame was enqueued on this stream records its end-of-frame event lazily, on the stream, when
    // it moves elsewhere (gs_render_frame): do that now, while the handle is alive.  This holds for wrapped streams too
    // (gs_stream_wrap): destroy the gs_stream BEFORE the hipStream_t it wraps.
    (void)hipSetDevice(s->dev->ordinal);
    renderers_leave_stream(s->dev, s->s);
    if (s->owned) {
        (void)hipSetDevice(s->dev->ordinal);
        (void)hipStreamDestroy(s->s);
    }
    delete s;
}

static hipStream_t stream_of(gs_device *dev, gs_stream *s) { return s ? s->s : dev->internal; }

extern "C" gs_status gs_buffer_create(gs_device *dev, size_t bytes, const void *init,
                                      gs_buffer **out) {
    if (!out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null out");
    GS_TRY(use_device(dev));
    void *p = nullptr;
    // a zero-sized wgpu buffer is legal; keep a 16-byte allocation so the pointer is valid
    GS_HIP(hipMalloc(&p, bytes ? bytes : 16));
    if (init && bytes) {
        hipError_t e = hipMemcpy(p, init, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return fail(GS_ERR_HIP, (uint64_t)e, 0, 0, "hipMemcpy failed: %s", hipGetErrorString(e));
        }
    } else if (bytes) {
        // wgpu zero-initialises new buffers
        hipError_t e = hipMemset(p, 0, bytes);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return fail(GS_ERR_HIP, (uint64_t)e, 0, 0, "hipMemset failed: %s", hipGetErrorString(e));
        }
    }
    gs_buffer *b = new gs_buffer();
    b->dev = dev;
    b->ptr = p;
    b->bytes = bytes;
    b->owned = true;
    b->refs.store(1);
    *out = b;
    return GS_OK;
}

extern "C" gs_status gs_buffer_from_raw(gs_device *dev, void *device_ptr, size_t bytes,
                                        gs_buffer **out) {
    if (!dev || !out || (!device_ptr && bytes))
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    gs_buffer *b = new gs_buffer();
    b->dev = dev;
    b->ptr = device_ptr;
    b->bytes = bytes;
    b->owned = false;
    b->refs.store(1);
    *out = b;
    return GS_OK;
}

extern "C" gs_buffer *gs_buffer_retain(gs_buffer *b) {
    if (b) b->refs.fetch_add(1);
    return b;
}

extern "C" void gs_buffer_release(gs_buffer *b) {
    if (!b) return;
    if (b->refs.fetch_sub(1) == 1) {
        if (b->owned && b->ptr) {
            (void)hipSetDevice(b->dev->ordinal);
            (void)hipFree(b->ptr);
        }
        delete b;
    }
}

extern "C" size_t gs_buffer_size(const gs_buffer *b) { return b ? b->bytes : 0; }
extern "C" void *gs_buffer_device_ptr(const gs_buffer *b) { return b ? b->ptr : nullptr; }

extern "C" gs_status gs_buffer_write(gs_buffer *b, gs_stream *s, size_t offset, const void *src,
                                     size_t bytes) {
    if (!b || (bytes && !src)) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    if (offset > b->bytes || bytes > b->bytes - offset)
        return fail(GS_ERR_INVALID_ARGUMENT, offset, bytes, b->bytes,
                    "write of %zu bytes at offset %zu overruns buffer of %zu bytes", bytes, offset,
                    b->bytes);
    if (!bytes) return GS_OK;
    GS_TRY(use_device(b->dev));
    hipStream_t st = stream_of(b->dev, s);
    GS_HIP(hipMemcpyAsync((uint8_t *)b->ptr + offset, src, bytes, hipMemcpyHostToDevice, st));
    // queue.write_buffer captures `src` at call time: do not return while the host memory may
    // still be read by an in-flight staged copy.
    GS_HIP(hipStreamSynchronize(st));
    return GS_OK;
}

extern "C" gs_status gs_buffer_download(gs_buffer *b, gs_stream *s, void *dst, size_t bytes) {
    if (!b || (bytes && !dst)) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    if (bytes > b->bytes)
        return fail(GS_ERR_INVALID_ARGUMENT, bytes, b->bytes, 0,
                    "download of %zu bytes from buffer of %zu bytes", bytes, b->bytes);
    if (!bytes) return GS_OK;
    GS_TRY(use_device(b->dev));
    hipStream_t st = stream_of(b->dev, s);
    hipError_t e = hipMemcpyAsync(dst, b->ptr, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
        return fail(GS_ERR_DOWNLOAD, (uint64_t)e, 0, 0, "download failed: %s", hipGetErrorString(e));
    return GS_OK;
}

// BufferWrapper::prepare_download / map_download (src/buffer/mod.rs:48-101): the copy into a
// host-visible staging buffer is only ENQUEUED; the caller maps (waits for) it later.
struct gs_download {
    gs_device *dev;
    void *pinned;
    size_t bytes;
    hipEvent_t done;
};

extern "C" gs_status gs_buffer_prepare_download(gs_buffer *b, gs_stream *s, gs_download **out) {
    if (!b || !out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    *out = nullptr;
    GS_TRY(use_device(b->dev));
    gs_download *d = new gs_download();
    d->dev = b->dev;
    d->bytes = b->bytes;
    d->pinned = nullptr;
    hipError_t e = hipHostMalloc(&d->pinned, b->bytes ? b->bytes : 1, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->done, hipEventDisableTiming);
    else d->done = nullptr;
    hipStream_t st = stream_of(b->dev, s);
    if (e == hipSuccess && b->bytes) e = hipMemcpyAsync(d->pinned, b->ptr, b->bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipEventRecord(d->done, st);
    if (e != hipSuccess) {
        if (d->pinned) (void)hipHostFree(d->pinned);
        if (d->done) (void)hipEventDestroy(d->done);
        delete d;
        return fail(GS_ERR_DOWNLOAD, (uint64_t)e, 0, 0, "download failed: %s", hipGetErrorString(e));
    }
    *out = d;
    return GS_OK;
}

extern "C" int32_t gs_download_ready(gs_download *d) {
    if (!d) return 0;
    (void)hipSetDevice(d->dev->ordinal);
    const hipError_t e = hipEventQuery(d->done);
    (void)hipGetLastError();
    return e == hipSuccess ? 1 : 0;
}

extern "C" gs_status gs_download_map(gs_download *d, const void **data_out, size_t *bytes_out) {
    if (!d || !data_out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    GS_TRY(use_device(d->dev));
    const hipError_t e = hipEventSynchronize(d->done);
    if (e != hipSuccess)
        return fail(GS_ERR_DOWNLOAD, (uint64_t)e, 0, 0, "download failed: %s", hipGetErrorString(e));
    *data_out = d->pinned;
    if (bytes_out) *bytes_out = d->bytes;
    return GS_OK;
}

extern "C" void gs_download_release(gs_download *d) {
    if (!d) return;
    (void)hipSetDevice(d->dev->ordinal);
    (void)hipEventSynchronize(d->done);     // the copy may still be writing the staging memory
    (void)hipHostFree(d->pinned);
    (void)hipEventDestroy(d->done);
    delete d;
}

// ------------------------------------------------------------------------------------------------
// GaussiansBuffer<G>
// ------------------------------------------------------------------------------------------------

struct gs_gaussians_buffer {
    gs_buffer *buf;
    int sh, cov;
    // block-planar mirror read by the preprocess kernel; rebuilt lazily
    void *planar;
    size_t planar_stride;  // capacity in Gaussians = len rounded up to whole 1024-blocks
    // Gaussians [dirty_lo, dirty_hi) of the AoS buffer are newer than the mirror (empty when lo >= hi).
    // update_range dirties only its own range, so an editor's per-edit update re-mirrors a few
    // Gaussians instead of the whole scene.
    size_t dirty_lo, dirty_hi;
    // spatial mirror order: order[slot] = Gaussian index, inv[index] = slot; both
    // null while the mirror is in index order.  `order` is a ref-counted gs_buffer so that the
    // renderer's parity taps can keep the order of the frame they describe.
    bool spatial;
    gs_buffer *order;
    void *inv;
    void *block_bounds;      // 8 floats per 1024-slot block (k_block_bounds); null = not available
    size_t partial_since_order;   // Gaussians rewritten by partial updates since the order was built
    // The mirror is (re)built on the stream of whichever frame first needs it; frames on OTHER streams
    // (several renderers keeping frames in flight on one buffer) wait for this event before reading it.
    hipEvent_t mirror_ready = nullptr;
    hipStream_t mirror_stream = nullptr;
    void mark(size_t lo, size_t hi) {
        if (lo >= hi) return;
        partial_since_order += hi - lo;
        if (dirty_lo >= dirty_hi) { dirty_lo = lo; dirty_hi = hi; return; }
        if (lo < dirty_lo) dirty_lo = lo;
        if (hi > dirty_hi) dirty_hi = hi;
    }
    void mark_all() { dirty_lo = 0; dirty_hi = (size_t)-1; }
};

static size_t pod_stride(const gs_gaussians_buffer *g) { return (size_t)gs::pod_bytes(g->sh, g->cov); }

extern "C" size_t gs_gaussians_buffer_len(const gs_gaussians_buffer *g) {
    return g ? g->buf->bytes / pod_stride(g) : 0;
}

extern "C" gs_status gs_gaussians_buffer_from_buffer(gs_buffer *buffer, gs_sh_config sh,
                                                     gs_cov3d_config cov,
                                                     gs_gaussians_buffer **out) {
    if (!buffer || !out || !valid_cfg(sh, cov))
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "bad argument");
    size_t stride = gs_pod_size(sh, cov);
    // the kernels move records as 16-byte vectors: an adopted raw pointer (gs_buffer_from_raw) must
    // be 16-byte aligned, as every hipMalloc allocation is
    if ((uintptr_t)buffer->ptr & 15u)
        return fail(GS_ERR_INVALID_ARGUMENT, (uint64_t)(uintptr_t)buffer->ptr, 16, 0,
                    "Gaussian buffer device pointer must be 16-byte aligned");
    if (buffer->bytes % stride != 0)
        return fail(GS_ERR_BUFFER_SIZE_NOT_MULTIPLE, buffer->bytes, stride, 0,
                    "buffer size and expected multiple size mismatch: %zu %% %zu != 0",
                    buffer->bytes, stride);
    gs_gaussians_buffer *g = new gs_gaussians_buffer();
    g->buf = gs_buffer_retain(buffer);
    g->sh = sh;
    g->cov = cov;
    g->planar = nullptr;
    g->planar_stride = 0;
    {   // default: spatial order on; GS3D_SPATIAL_ORDER=0 keeps the mirror in index order
        const char *e = std::getenv("GS3D_SPATIAL_ORDER");
        g->spatial = !(e && e[0] == '0');
    }
    g->order = nullptr;
    g->inv = nullptr;
    g->block_bounds = nullptr;
    g->partial_since_order = 0;
    g->mark_all();
    *out = g;
    return GS_OK;
}

extern "C" gs_status gs_gaussians_buffer_create(gs_device *dev, gs_sh_config sh, gs_cov3d_config cov,
                                                const void *pods, size_t len,
                                                gs_gaussians_buffer **out) {
    if (!out || !valid_cfg(sh, cov)) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "bad argument");
    gs_buffer *b = nullptr;
    GS_TRY(gs_buffer_create(dev, len * gs_pod_size(sh, cov), pods, &b));
    gs_status st = gs_gaussians_buffer_from_buffer(b, sh, cov, out);
    gs_buffer_release(b);
    return st;
}

#define GS_CFG_TABLE(kernel)                                                                     \
    {                                                                                            \
        {kernel<0, 0>, kernel<0, 1>, kernel<0, 2>}, {kernel<1, 0>, kernel<1, 1>, kernel<1, 2>},  \
        {kernel<2, 0>, kernel<2, 1>, kernel<2, 2>}, {kernel<3, 0>, kernel<3, 1>, kernel<3, 2>},  \
    }

typedef void (*pack_fn)(const uint32_t *, uint64_t, uint32_t *);
static pack_fn k_tbl_pack[4][3] = GS_CFG_TABLE(gs::k_pack_pods);

extern "C" gs_status gs_pack_device(gs_device *dev, gs_stream *s, gs_sh_config sh, gs_cov3d_config cov,
                                    const gs_gaussian *gaussians_device, size_t n, void *pods_device) {
    if (!dev || !valid_cfg(sh, cov) || (n && (!gaussians_device || !pods_device)))
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "gs_pack_device: bad argument");
    if (((uintptr_t)gaussians_device | (uintptr_t)pods_device) & 3u)
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "gs_pack_device: pointers must be 4-byte aligned");
    GS_TRY(use_device(dev));
    if (!n) return GS_OK;
    const uint64_t groups = ((uint64_t)n + gs::PACK_GROUP - 1) / gs::PACK_GROUP;
    if (groups > 0x7fffffffull) return fail(GS_ERR_INVALID_ARGUMENT, n, 0, 0, "too many Gaussians");
    hipLaunchKernelGGL(k_tbl_pack[sh][cov], dim3((uint32_t)groups), dim3(256), 0, stream_of(dev, s),
                       (const uint32_t *)gaussians_device, (uint64_t)n, (uint32_t *)pods_device);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

static pack_fn k_tbl_from_ply[4][3] = GS_CFG_TABLE(gs::k_from_ply_pods);

extern "C" gs_status gs_pack_device_from_ply(gs_device *dev, gs_stream *s, gs_sh_config sh, gs_cov3d_config cov,
                                             const gs_ply_gaussian_pod *ply_device, size_t n, void *pods_device) {
    if (!dev || !valid_cfg(sh, cov) || (n && (!ply_device || !pods_device)))
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "gs_pack_device_from_ply: bad argument");
    if (((uintptr_t)ply_device | (uintptr_t)pods_device) & 3u)
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "gs_pack_device_from_ply: pointers must be 4-byte aligned");
    GS_TRY(use_device(dev));
    if (!n) return GS_OK;
    const uint64_t groups = ((uint64_t)n + gs::PACK_GROUP - 1) / gs::PACK_GROUP;
    if (groups > 0x7fffffffull) return fail(GS_ERR_INVALID_ARGUMENT, n, 0, 0, "too many Gaussians");
    hipLaunchKernelGGL(k_tbl_from_ply[sh][cov], dim3((uint32_t)groups), dim3(256), 0, stream_of(dev, s),
                       (const uint32_t *)ply_device, (uint64_t)n, (uint32_t *)pods_device);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// Source records on the host (struct Gaussian, or PlyGaussianPod when `from_ply`) -> PODs in `g` at
// [start, start + count): the records cross PCIe as they are, slice by slice through TWO staging
// buffers (the host stages and submits slice i + 1 while the kernel of slice i runs; copies and kernels
// share one stream, so the transfers themselves run one after the other — the link is the bound:
// 49.7 GB/s of PLY bytes at 50 M vertices), and are converted on the device.
// The caller's memory may be reused when this returns.
static gs_status upload_records(gs_gaussians_buffer *g, gs_stream *s, size_t start, const void *records, size_t count,
                                bool from_ply) {
    constexpr size_t PACK_SLICE = 2u << 20;     // 2 Mi records: 2 x 496 MiB of staging at most
    gs_device *dev = g->buf->dev;
    GS_TRY(use_device(dev));
    if (!count) return GS_OK;
    hipStream_t st = stream_of(dev, s);
    const size_t rec = from_ply ? sizeof(gs_ply_gaussian_pod) : sizeof(gs_gaussian);
    const size_t slice = count < PACK_SLICE ? count : PACK_SLICE;
    const int nbuf = count > slice ? 2 : 1;
    void *staging[2] = {nullptr, nullptr};
    hipEvent_t used[2] = {nullptr, nullptr};
    gs_status rc = GS_OK;
    for (int i = 0; i < nbuf && rc == GS_OK; i++) {
        hipError_t e = hipMalloc(&staging[i], slice * rec);
        if (e != hipSuccess) rc = fail(GS_ERR_OUT_OF_MEMORY, slice * rec, 0, 0, "hipMalloc failed: %s", hipGetErrorString(e));
        else if ((e = hipEventCreateWithFlags(&used[i], hipEventDisableTiming)) != hipSuccess)
            rc = fail(GS_ERR_HIP, (uint64_t)e, 0, 0, "hipEventCreate failed: %s", hipGetErrorString(e));
    }
    const size_t stride = pod_stride(g);
    int k = 0;
    for (size_t first = 0; first < count && rc == GS_OK; first += slice, k ^= (nbuf - 1)) {
        const size_t cnt = count - first < slice ? count - first : slice;
        // the staging buffer's previous kernel must have read it before the next copy overwrites it
        if (first >= (size_t)nbuf * slice && hipEventSynchronize(used[k]) != hipSuccess) {
            rc = fail(GS_ERR_HIP, 0, 0, 0, "upload failed");
            break;
        }
        hipError_t e = hipMemcpyAsync(staging[k], (const uint8_t *)records + first * rec, cnt * rec, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            rc = fail(GS_ERR_HIP, (uint64_t)e, 0, 0, "upload failed: %s", hipGetErrorString(e));
            break;
        }
        void *dst = (uint8_t *)g->buf->ptr + (start + first) * stride;
        rc = from_ply ? gs_pack_device_from_ply(dev, s, (gs_sh_config)g->sh, (gs_cov3d_config)g->cov,
                                                (const gs_ply_gaussian_pod *)staging[k], cnt, dst)
                      : gs_pack_device(dev, s, (gs_sh_config)g->sh, (gs_cov3d_config)g->cov, (const gs_gaussian *)staging[k],
                                       cnt, dst);
        if (rc == GS_OK && hipEventRecord(used[k], st) != hipSuccess) rc = fail(GS_ERR_HIP, 0, 0, 0, "upload failed");
    }
    // the staging buffers are freed below and the caller's memory may be reused: wait for everything
    if (hipStreamSynchronize(st) != hipSuccess && rc == GS_OK) rc = fail(GS_ERR_HIP, 0, 0, 0, "pack failed");
    for (int i = 0; i < 2; i++) {
        if (used[i]) (void)hipEventDestroy(used[i]);
        if (staging[i]) (void)hipFree(staging[i]);
    }
    return rc;
}

static gs_status upload_gaussians(gs_gaussians_buffer *g, gs_stream *s, size_t start, const gs_gaussian *gaussians,
                                  size_t count) {
    return upload_records(g, s, start, gaussians, count, false);
}

extern "C" gs_status gs_gaussians_buffer_create_from_ply(gs_device *dev, gs_sh_config sh, gs_cov3d_config cov,
                                                         const gs_ply_gaussian_pod *ply, size_t len,
                                                         gs_gaussians_buffer **out) {
    if (!valid_cfg(sh, cov) || (len && !ply) || !out)
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "bad argument");
    GS_TRY(gs_gaussians_buffer_create(dev, sh, cov, nullptr, len, out));
    gs_status rc = upload_records(*out, nullptr, 0, ply, len, true);
    if (rc != GS_OK) {
        gs_gaussians_buffer_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

// SPZ: the decompressed payload crosses PCIe once; one kernel decodes the columns (Gaussian::from_spz,
// src/gaussian.rs:134-229) and packs (G::from_gaussian)
typedef void (*spz_fn)(gs::SpzView, uint64_t, uint32_t *);
static spz_fn k_tbl_from_spz[4][3] = GS_CFG_TABLE(gs::k_from_spz_pods);

extern "C" gs_status gs_gaussians_buffer_create_from_spz_decompressed(gs_device *dev, gs_sh_config sh,
                                                                      gs_cov3d_config cov, const void *bytes,
                                                                      size_t len, gs_spz_header *header_out,
                                                                      gs_gaussians_buffer **out) {
    if (!valid_cfg(sh, cov) || !bytes || !out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "bad argument");
    *out = nullptr;
    gs_spz_header h;
    size_t off[6];
    uint32_t ncoef = 0;
    GS_TRY(gs_spz_payload_layout(bytes, len, &h, off, &ncoef));
    if (header_out) *header_out = h;
    const size_t n = h.num_points;
    GS_TRY(gs_gaussians_buffer_create(dev, sh, cov, nullptr, n, out));
    if (!n) return GS_OK;
    gs_status rc = GS_OK;
    void *staging = nullptr;
    const size_t used = off[5] + n * 3u * ncoef;             // the payload the header declares (len may be larger)
    hipError_t e = hipMalloc(&staging, used);
    if (e != hipSuccess) rc = fail(GS_ERR_OUT_OF_MEMORY, used, 0, 0, "hipMalloc failed: %s", hipGetErrorString(e));
    if (rc == GS_OK && (e = hipMemcpyAsync(staging, bytes, used, hipMemcpyHostToDevice, dev->internal)) != hipSuccess)
        rc = fail(GS_ERR_HIP, (uint64_t)e, 0, 0, "upload failed: %s", hipGetErrorString(e));
    if (rc == GS_OK) {
        const uint8_t *b = (const uint8_t *)staging;
        gs::SpzView v{b + off[0], b + off[1], b + off[2], b + off[3], b + off[4], b + off[5], h.version, h.fractional_bits, ncoef};
        const uint64_t groups = ((uint64_t)n + gs::PACK_GROUP - 1) / gs::PACK_GROUP;
        hipLaunchKernelGGL(k_tbl_from_spz[sh][cov], dim3((uint32_t)groups), dim3(256), 0, dev->internal, v, (uint64_t)n,
                           (uint32_t *)(*out)->buf->ptr);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(dev->internal);
        if (e != hipSuccess) rc = fail(GS_ERR_HIP, (uint64_t)e, 0, 0, "SPZ decode kernel failed: %s", hipGetErrorString(e));
    }
    if (staging) (void)hipFree(staging);
    if (rc != GS_OK) {
        gs_gaussians_buffer_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

extern "C" gs_status gs_gaussians_buffer_create_from_spz(gs_device *dev, gs_sh_config sh, gs_cov3d_config cov,
                                                         const void *bytes, size_t len, gs_spz_header *header_out,
                                                         gs_gaussians_buffer **out) {
    if (!bytes || !out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "bad argument");
    std::vector<uint8_t> raw;
    GS_TRY(gs_spz_gunzip(bytes, len, raw));
    return gs_gaussians_buffer_create_from_spz_decompressed(dev, sh, cov, raw.data(), raw.size(), header_out, out);
}

extern "C" gs_status gs_gaussians_buffer_update_range_ply(gs_gaussians_buffer *g, gs_stream *s, size_t start,
                                                          const gs_ply_gaussian_pod *ply, size_t count) {
    if (!g) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null buffer");
    size_t len = gs_gaussians_buffer_len(g);
    if (count > len || start > len - count)
        return fail(GS_ERR_RANGE_COUNT_MISMATCH, count, start, len,
                    "Gaussians count mismatch: %zu + %zu > %zu", count, start, len);
    if (count && !ply) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null ply records");
    g->mark(start, start + count);
    return upload_records(g, s, start, ply, count, true);
}

extern "C" gs_status gs_gaussians_buffer_create_from_gaussians(gs_device *dev, gs_sh_config sh,
                                                               gs_cov3d_config cov,
                                                               const gs_gaussian *gaussians,
                                                               size_t len,
                                                               gs_gaussians_buffer **out) {
    if (!valid_cfg(sh, cov) || (len && !gaussians) || !out)
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "bad argument");
    GS_TRY(gs_gaussians_buffer_create(dev, sh, cov, nullptr, len, out));
    gs_status rc = upload_gaussians(*out, nullptr, 0, gaussians, len);
    if (rc != GS_OK) {
        gs_gaussians_buffer_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

extern "C" void gs_gaussians_buffer_destroy(gs_gaussians_buffer *g) {
    if (!g) return;
    (void)hipSetDevice(g->buf->dev->ordinal);
    if (g->planar) (void)hipFree(g->planar);
    if (g->inv) (void)hipFree(g->inv);
    if (g->block_bounds) (void)hipFree(g->block_bounds);
    if (g->mirror_ready) (void)hipEventDestroy(g->mirror_ready);
    if (g->order) gs_buffer_release(g->order);
    gs_buffer_release(g->buf);
    delete g;
}

extern "C" gs_buffer *gs_gaussians_buffer_buffer(const gs_gaussians_buffer *g) {
    return g ? g->buf : nullptr;
}
extern "C" gs_sh_config gs_gaussians_buffer_sh(const gs_gaussians_buffer *g) {
    return (gs_sh_config)(g ? g->sh : 0);
}
extern "C" gs_cov3d_config gs_gaussians_buffer_cov3d(const gs_gaussians_buffer *g) {
    return (gs_cov3d_config)(g ? g->cov : 0);
}

extern "C" gs_status gs_gaussians_buffer_update(gs_gaussians_buffer *g, gs_stream *s,
                                                const void *pods, size_t count) {
    if (!g) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null buffer");
    size_t len = gs_gaussians_buffer_len(g);
    if (count != len)
        return fail(GS_ERR_COUNT_MISMATCH, count, len, 0, "Gaussians count mismatch: %zu != %zu",
                    count, len);
    g->mark_all();
    return gs_buffer_write(g->buf, s, 0, pods, count * pod_stride(g));
}

extern "C" gs_status gs_gaussians_buffer_update_range(gs_gaussians_buffer *g, gs_stream *s,
                                                      size_t start, const void *pods,
                                                      size_t count) {
    if (!g) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null buffer");
    size_t len = gs_gaussians_buffer_len(g);
    if (count > len || start > len - count)      // (start + count could wrap)
        return fail(GS_ERR_RANGE_COUNT_MISMATCH, count, start, len,
                    "Gaussians count mismatch: %zu + %zu > %zu", count, start, len);
    g->mark(start, start + count);
    return gs_buffer_write(g->buf, s, start * pod_stride(g), pods, count * pod_stride(g));
}

extern "C" gs_status gs_gaussians_buffer_update_gaussians(gs_gaussians_buffer *g, gs_stream *s,
                                                          const gs_gaussian *gaussians,
                                                          size_t count) {
    if (!g) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null buffer");
    size_t len = gs_gaussians_buffer_len(g);
    if (count != len)
        return fail(GS_ERR_COUNT_MISMATCH, count, len, 0, "Gaussians count mismatch: %zu != %zu",
                    count, len);
    if (count && !gaussians) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null gaussians");
    g->mark_all();
    return upload_gaussians(g, s, 0, gaussians, count);
}

extern "C" gs_status gs_gaussians_buffer_update_range_gaussians(gs_gaussians_buffer *g, gs_stream *s,
                                                                size_t start,
                                                                const gs_gaussian *gaussians,
                                                                size_t count) {
    if (!g) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null buffer");
    size_t len = gs_gaussians_buffer_len(g);
    if (count > len || start > len - count)
        return fail(GS_ERR_RANGE_COUNT_MISMATCH, count, start, len,
                    "Gaussians count mismatch: %zu + %zu > %zu", count, start, len);
    if (count && !gaussians) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null gaussians");
    g->mark(start, start + count);
    return upload_gaussians(g, s, start, gaussians, count);
}

extern "C" gs_status gs_gaussians_buffer_download(gs_gaussians_buffer *g, gs_stream *s,
                                                  void *pods_out, size_t count) {
    if (!g) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null buffer");
    if (count > gs_gaussians_buffer_len(g))
        return fail(GS_ERR_INVALID_ARGUMENT, count, gs_gaussians_buffer_len(g), 0, "count too large");
    return gs_buffer_download(g->buf, s, pods_out, count * pod_stride(g));
}

// GaussiansBuffer::download::<Gaussian> (src/buffer/gaussian.rs:186-196): the PODs are converted back
// on the DEVICE (k_unpack_pods, slice by slice through a staging buffer) and the struct Gaussian
// records come over PCIe ready to use; bit-equal to downloading the PODs and gs_unpack_to_gaussian.
typedef void (*unpack_fn)(const uint32_t *, uint64_t, uint32_t *);
static unpack_fn k_tbl_unpack[3] = {gs::k_unpack_pods<0>, gs::k_unpack_pods<1>, gs::k_unpack_pods<2>};

extern "C" gs_status gs_gaussians_buffer_download_gaussians(gs_gaussians_buffer *g, gs_stream *s,
                                                            gs_gaussian *out, size_t count) {
    if (!g) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null buffer");
    if (g->sh == GS_SH_NONE || g->cov != GS_COV3D_ROT_SCALE)
        return gs_unpack_to_gaussian((gs_sh_config)g->sh, (gs_cov3d_config)g->cov, out, 0, out);   // the reference's error
    if (count > gs_gaussians_buffer_len(g))
        return fail(GS_ERR_INVALID_ARGUMENT, count, gs_gaussians_buffer_len(g), 0, "count too large");
    if (!count) return GS_OK;
    if (!out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null out");
    gs_device *dev = g->buf->dev;
    GS_TRY(use_device(dev));
    hipStream_t st = stream_of(dev, s);
    constexpr size_t SLICE = 2u << 20;
    const size_t slice = count < SLICE ? count : SLICE;
    void *staging = nullptr;
    hipError_t e = hipMalloc(&staging, slice * sizeof(gs_gaussian));
    if (e != hipSuccess)
        return fail(GS_ERR_OUT_OF_MEMORY, slice * sizeof(gs_gaussian), 0, 0, "hipMalloc failed: %s", hipGetErrorString(e));
    gs_status rc = GS_OK;
    const size_t stride = pod_stride(g);
    for (size_t first = 0; first < count && rc == GS_OK; first += slice) {
        const size_t cnt = count - first < slice ? count - first : slice;
        const uint64_t groups = ((uint64_t)cnt + gs::PACK_GROUP - 1) / gs::PACK_GROUP;
        hipLaunchKernelGGL(k_tbl_unpack[g->sh], dim3((uint32_t)groups), dim3(256), 0, st,
                           (const uint32_t *)((const uint8_t *)g->buf->ptr + first * stride), (uint64_t)cnt, (uint32_t *)staging);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out + first, staging, cnt * sizeof(gs_gaussian), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(GS_ERR_DOWNLOAD, (uint64_t)e, 0, 0, "download failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(staging);
    return rc;
}

extern "C" void gs_gaussians_buffer_mark_dirty(gs_gaussians_buffer *g) {
    if (g) g->mark_all();
}

// ------------------------------------------------------------------------------------------------
// fixed-size uniform buffers
// ------------------------------------------------------------------------------------------------

static gs_status fixed_from_buffer(gs_buffer *b, size_t expected) {
    if (!b) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null buffer");
    if (b->bytes != expected)
        return fail(GS_ERR_BUFFER_SIZE_MISMATCHED, b->bytes, expected, 0,
                    "buffer size and expected size mismatch: %zu != %zu", b->bytes, expected);
    return GS_OK;
}

extern "C" gs_status gs_gaussian_transform_buffer_create(gs_device *dev, gs_buffer **out) {
    gs_gaussian_transform_pod pod;
    gs_gaussian_transform_pod_default(&pod);
    return gs_buffer_create(dev, sizeof(pod), &pod, out);
}
extern "C" gs_status gs_gaussian_transform_buffer_update(gs_buffer *b, gs_stream *s,
                                                         const gs_gaussian_transform_pod *pod) {
    GS_TRY(fixed_from_buffer(b, sizeof(*pod)));
    return gs_buffer_write(b, s, 0, pod, sizeof(*pod));
}
extern "C" gs_status gs_gaussian_transform_buffer_from_buffer(gs_buffer *b) {
    return fixed_from_buffer(b, sizeof(gs_gaussian_transform_pod));
}
extern "C" gs_status gs_model_transform_buffer_create(gs_device *dev, gs_buffer **out) {
    gs_model_transform_pod pod;
    gs_model_transform_pod_default(&pod);
    return gs_buffer_create(dev, sizeof(pod), &pod, out);
}
extern "C" gs_status gs_model_transform_buffer_update(gs_buffer *b, gs_stream *s,
                                                      const gs_model_transform_pod *pod) {
    GS_TRY(fixed_from_buffer(b, sizeof(*pod)));
    return gs_buffer_write(b, s, 0, pod, sizeof(*pod));
}
extern "C" gs_status gs_model_transform_buffer_from_buffer(gs_buffer *b) {
    return fixed_from_buffer(b, sizeof(gs_model_transform_pod));
}

// ------------------------------------------------------------------------------------------------
// ComputeBundle
// ------------------------------------------------------------------------------------------------

typedef void (*bundle_kernel_fn)(gs::BundleArgs, uint32_t);

static bundle_kernel_fn k_tbl_test_gaussian[4][3] = GS_CFG_TABLE(gs::k_test_gaussian);
static bundle_kernel_fn k_tbl_unpack_soa[4][3] = GS_CFG_TABLE(gs::k_unpack_soa);

struct gs_bundle {
    gs_device *dev;
    std::string label;
    gs_kernel_id kernel;
    int sh, cov;
    uint32_t workgroup_size;
    std::vector<uint32_t> layout;                     // bindings per group
    std::vector<std::vector<gs_buffer *>> groups;     // managed bind groups (retained)
    bool managed;
    bool has_additional_constant;
    uint32_t additional_constant;
    uint32_t last_workgroups;
    // bundles compiled from source (gs_bundle_create_from_source)
    bool from_source = false;
    hipModule_t module = nullptr;
    hipFunction_t func = nullptr;
};

extern "C" gs_status gs_bundle_create(gs_device *dev, const gs_bundle_desc *desc, gs_bundle **out) {
    if (!dev || !desc || !out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    if ((int)desc->kernel < 0 || desc->kernel >= GS_KERNEL_COUNT_ || !valid_cfg(desc->sh, desc->cov))
        return fail(GS_ERR_MISSING_MAIN_SHADER, 0, 0, 0, "unknown kernel id %d", (int)desc->kernel);
    if (desc->bind_group_count == 0 || !desc->bindings_per_group)
        return fail(GS_ERR_MISSING_BIND_GROUP_LAYOUT, 0, 0, 0,
                    "missing bind group layout for compute bundle");
    // compute_bundle.rs:269-281
    uint32_t limit = dev->limits.max_compute_workgroup_size_x <
                             dev->limits.max_compute_invocations_per_workgroup
                         ? dev->limits.max_compute_workgroup_size_x
                         : dev->limits.max_compute_invocations_per_workgroup;
    uint32_t wg = desc->workgroup_size ? desc->workgroup_size : limit;
    if (wg > limit)
        return fail(GS_ERR_WORKGROUP_SIZE_EXCEEDS_LIMIT, wg, limit, 0,
                    "workgroup size exceeds device limit: %u > %u", wg, limit);
    uint32_t total = 0;
    for (uint32_t i = 0; i < desc->bind_group_count; i++) total += desc->bindings_per_group[i];
    if (total > (uint32_t)gs::MAX_BINDINGS)
        return fail(GS_ERR_INVALID_ARGUMENT, total, gs::MAX_BINDINGS, 0, "too many bindings");
    gs_bundle *b = new gs_bundle();
    b->dev = dev;
    b->label = desc->label ? desc->label : "";
    b->kernel = desc->kernel;
    b->sh = desc->sh;
    b->cov = desc->cov;
    b->workgroup_size = wg;
    b->layout.assign(desc->bindings_per_group, desc->bindings_per_group + desc->bind_group_count);
    b->managed = false;
    b->has_additional_constant = false;
    b->additional_constant = 0;
    b->last_workgroups = 0;
    for (uint32_t i = 0; i < desc->constant_count; i++) {
        if (desc->constant_names && desc->constant_names[i] &&
            !std::strcmp(desc->constant_names[i], "additional_constant")) {
            b->has_additional_constant = true;
            b->additional_constant = (uint32_t)desc->constant_values[i];
        }
    }
    *out = b;
    return GS_OK;
}

static void release_group(std::vector<gs_buffer *> &g) {
    for (gs_buffer *x : g) gs_buffer_release(x);
    g.clear();
}

extern "C" gs_status gs_bundle_set_bind_group(gs_bundle *b, uint32_t index, gs_buffer *const *buffers,
                                              uint32_t count) {
    if (!b) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null bundle");
    if (index >= b->groups.size())
        return fail(GS_ERR_INVALID_ARGUMENT, index, b->groups.size(), 0, "bind group index out of bounds");
    if (count != b->layout[index])
        return fail(GS_ERR_INVALID_ARGUMENT, count, b->layout[index], 0,
                    "bind group %u expects %u bindings, got %u", index, b->layout[index], count);
    std::vector<gs_buffer *> g;
    for (uint32_t i = 0; i < count; i++) {
        if (!buffers[i]) {
            release_group(g);
            return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null binding");
        }
        g.push_back(gs_buffer_retain(buffers[i]));
    }
    release_group(b->groups[index]);
    b->groups[index] = g;
    return GS_OK;
}

extern "C" gs_status gs_bundle_create_with_bind_groups(gs_device *dev, const gs_bundle_desc *desc,
                                                       gs_buffer *const *const *resources,
                                                       const uint32_t *resource_counts,
                                                       uint32_t resource_group_count,
                                                       gs_bundle **out) {
    gs_bundle *b = nullptr;
    GS_TRY(gs_bundle_create(dev, desc, &b));
    // compute_bundle.rs:161-168
    if (resource_group_count != b->layout.size()) {
        size_t layouts = b->layout.size();
        gs_bundle_destroy(b);
        return fail(GS_ERR_RESOURCE_COUNT_MISMATCH, resource_group_count, layouts, 0,
                    "resource count and bind group layout count mismatch: %u != %zu",
                    resource_group_count, layouts);
    }
    b->managed = true;
    b->groups.resize(b->layout.size());
    for (uint32_t i = 0; i < resource_group_count; i++) {
        gs_status st = gs_bundle_set_bind_group(b, i, resources[i], resource_counts[i]);
        if (st != GS_OK) {
            gs_bundle_destroy(b);
            return st;
        }
    }
    *out = b;
    return GS_OK;
}

extern "C" void gs_bundle_destroy(gs_bundle *b) {
    if (!b) return;
    for (auto &g : b->groups) release_group(g);
    if (b->module) {
        (void)hipSetDevice(b->dev->ordinal);
        (void)hipModuleUnload(b->module);
    }
    delete b;
}

extern "C" gs_status gs_bundle_create_from_source(gs_device *dev, const gs_bundle_source_desc *desc,
                                                  gs_bundle **out) {
    if (!dev || !desc || !out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    // same order of checks as ComputeBundleBuilder::build (compute_bundle.rs:505-519)
    if (desc->bind_group_count == 0 || !desc->bindings_per_group)
        return fail(GS_ERR_MISSING_BIND_GROUP_LAYOUT, 0, 0, 0, "missing bind group layout for compute bundle");
    if (!desc->entry_point) return fail(GS_ERR_MISSING_ENTRY_POINT, 0, 0, 0, "missing entry point for compute bundle");
    if (!desc->source) return fail(GS_ERR_MISSING_MAIN_SHADER, 0, 0, 0, "missing main shader for compute bundle");
    if (!valid_cfg(desc->sh, desc->cov)) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "bad config");
    uint32_t limit = dev->limits.max_compute_workgroup_size_x <
                             dev->limits.max_compute_invocations_per_workgroup
                         ? dev->limits.max_compute_workgroup_size_x
                         : dev->limits.max_compute_invocations_per_workgroup;
    uint32_t wg = desc->workgroup_size ? desc->workgroup_size : limit;
    if (wg > limit)
        return fail(GS_ERR_WORKGROUP_SIZE_EXCEEDS_LIMIT, wg, limit, 0,
                    "workgroup size exceeds device limit: %u > %u", wg, limit);
    uint32_t total = 0;
    for (uint32_t i = 0; i < desc->bind_group_count; i++) total += desc->bindings_per_group[i];
    if (total > (uint32_t)gs::MAX_BINDINGS)
        return fail(GS_ERR_INVALID_ARGUMENT, total, gs::MAX_BINDINGS, 0, "too many bindings");
    GS_TRY(use_device(dev));

    std::vector<std::string> opts;
    opts.push_back(std::string("--offload-arch=") + dev->limits.arch_name);
    opts.push_back("-std=c++17");
    opts.push_back("-ffp-contract=off");
    opts.push_back("-DGS_SH=" + std::to_string((int)desc->sh));
    opts.push_back("-DGS_COV=" + std::to_string((int)desc->cov));
    opts.push_back(std::string("-D") + k_feature_names[desc->sh] + "=1");
    opts.push_back(std::string("-D") + k_feature_names[4 + desc->cov] + "=1");
    opts.push_back("-Dworkgroup_size=" + std::to_string(wg));
    for (uint32_t i = 0; i < desc->define_count; i++)
        if (desc->defines && desc->defines[i]) opts.push_back(std::string("-D") + desc->defines[i] + "=1");
    for (uint32_t i = 0; i < desc->constant_count; i++) {
        if (!desc->constant_names || !desc->constant_names[i]) continue;
        double v = desc->constant_values[i];
        char buf[64];
        if (v == (double)(long long)v) std::snprintf(buf, sizeof(buf), "%lld", (long long)v);
        else std::snprintf(buf, sizeof(buf), "%.17g", v);
        opts.push_back(std::string("-D") + desc->constant_names[i] + "=" + buf);
    }
    std::string entry = desc->entry_point;
    opts.push_back("-Dmain=gs_entry_main");   // `main` cannot name a kernel in C++: always renamed
    if (entry == "main") entry = "gs_entry_main";
    std::vector<const char *> copts;
    for (auto &o : opts) copts.push_back(o.c_str());

    hiprtcProgram prog;
    const char *hdr_src[1] = {k_kernel_lib_src};
    const char *hdr_names[1] = {"wgpu_3dgs_core.h"};
    hiprtcResult rr = hiprtcCreateProgram(&prog, desc->source, "main_shader.hip", 1, hdr_src, hdr_names);
    if (rr != HIPRTC_SUCCESS)
        return fail(GS_ERR_KERNEL_COMPILE, (uint64_t)rr, 0, 0, "hiprtcCreateProgram: %s", hiprtcGetErrorString(rr));
    rr = hiprtcCompileProgram(prog, (int)copts.size(), copts.data());
    if (rr != HIPRTC_SUCCESS) {
        size_t n = 0;
        (void)hiprtcGetProgramLogSize(prog, &n);
        std::string log(n ? n : 1, '\0');
        if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
        (void)hiprtcDestroyProgram(&prog);
        // keep the first error lines
        size_t e = log.find("error");
        std::string shown = e == std::string::npos ? log : log.substr(e > 80 ? e - 80 : 0);
        return fail(GS_ERR_KERNEL_COMPILE, (uint64_t)rr, 0, 0, "kernel compilation failed: %.200s", shown.c_str());
    }
    size_t code_size = 0;
    (void)hiprtcGetCodeSize(prog, &code_size);
    std::vector<char> code(code_size);
    (void)hiprtcGetCode(prog, code.data());
    (void)hiprtcDestroyProgram(&prog);

    gs_bundle *b = new gs_bundle();
    b->dev = dev;
    b->label = desc->label ? desc->label : "";
    b->kernel = GS_KERNEL_COUNT_;
    b->sh = desc->sh;
    b->cov = desc->cov;
    b->workgroup_size = wg;
    b->layout.assign(desc->bindings_per_group, desc->bindings_per_group + desc->bind_group_count);
    b->managed = false;
    b->has_additional_constant = false;
    b->additional_constant = 0;
    b->last_workgroups = 0;
    b->from_source = true;
    hipError_t he = hipModuleLoadData(&b->module, code.data());
    if (he == hipSuccess) he = hipModuleGetFunction(&b->func, b->module, entry.c_str());
    if (he != hipSuccess) {
        gs_bundle_destroy(b);
        return fail(GS_ERR_MISSING_ENTRY_POINT, (uint64_t)he, 0, 0, "entry point '%s' not found in the compiled module: %s",
                    desc->entry_point, hipGetErrorString(he));
    }
    *out = b;
    return GS_OK;
}

// bind groups for a bundle created without them (gs_bundle_create / _from_source): makes it managed
extern "C" gs_status gs_bundle_attach_bind_groups(gs_bundle *b, gs_buffer *const *const *resources,
                                                  const uint32_t *resource_counts,
                                                  uint32_t resource_group_count) {
    if (!b) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null bundle");
    if (resource_group_count != b->layout.size())
        return fail(GS_ERR_RESOURCE_COUNT_MISMATCH, resource_group_count, b->layout.size(), 0,
                    "resource count and bind group layout count mismatch: %u != %zu",
                    resource_group_count, b->layout.size());
    b->managed = true;
    b->groups.resize(b->layout.size());
    for (uint32_t i = 0; i < resource_group_count; i++)
        GS_TRY(gs_bundle_set_bind_group(b, i, resources[i], resource_counts[i]));
    return GS_OK;
}

extern "C" uint32_t gs_bundle_workgroup_size(const gs_bundle *b) { return b ? b->workgroup_size : 0; }
extern "C" const char *gs_bundle_label(const gs_bundle *b) {
    return (b && !b->label.empty()) ? b->label.c_str() : nullptr;
}
extern "C" uint32_t gs_bundle_bind_group_layout_count(const gs_bundle *b) {
    return b ? (uint32_t)b->layout.size() : 0;
}
extern "C" uint32_t gs_bundle_bind_group_count(const gs_bundle *b) {
    return b ? (uint32_t)b->groups.size() : 0;
}
extern "C" uint32_t gs_bundle_last_workgroup_count(const gs_bundle *b) {
    return b ? b->last_workgroups : 0;
}

// minimum byte size each binding must have so that the kernel cannot run out of bounds
static gs_status validate_bindings(const gs_bundle *b, const gs::BundleArgs &a, uint32_t nbind) {
    auto need = [&](uint32_t i, uint64_t bytes) -> gs_status {
        if (i >= nbind || a.size[i] < bytes)
            return fail(GS_ERR_INVALID_ARGUMENT, i, i < nbind ? a.size[i] : 0, bytes,
                        "binding %u is smaller than the %llu bytes the kernel accesses", i,
                        (unsigned long long)bytes);
        return GS_OK;
    };
    uint64_t pod = (uint64_t)gs::pod_bytes(b->sh, b->cov);
    switch (b->kernel) {
    case GS_KERNEL_ARRAY_MAP_ADD:
        GS_TRY(need(0, 0));
        if (b->layout.size() > 1) GS_TRY(need(1, 4));
        break;
    case GS_KERNEL_TEST_GAUSSIAN:
        GS_TRY(need(0, pod));
        GS_TRY(need(1, 56 * 4));
        break;
    case GS_KERNEL_TEST_GAUSSIAN_TRANSFORM:
        GS_TRY(need(0, 8));
        GS_TRY(need(1, 16));
        break;
    case GS_KERNEL_TEST_MODEL_TRANSFORM:
        GS_TRY(need(0, 48));
        GS_TRY(need(1, 12));
        GS_TRY(need(2, 44 * 4));
        break;
    case GS_KERNEL_UNPACK_SOA:
        GS_TRY(need(0, 0));
        GS_TRY(need(1, (a.size[0] / pod) * 55 * 4));
        break;
    default: break;
    }
    return GS_OK;
}

static gs_status dispatch_groups(gs_bundle *b, gs_stream *s, uint32_t count,
                                 const std::vector<std::vector<gs_buffer *>> &groups) {
    if (!s) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null stream");
    if (groups.size() != b->layout.size())
        return fail(GS_ERR_INVALID_ARGUMENT, groups.size(), b->layout.size(), 0,
                    "dispatch needs %zu bind groups, got %zu", b->layout.size(), groups.size());
    gs::BundleArgs a;
    std::memset(&a, 0, sizeof(a));
    uint32_t n = 0;
    for (size_t gi = 0; gi < groups.size(); gi++) {
        if (groups[gi].size() != b->layout[gi])
            return fail(GS_ERR_INVALID_ARGUMENT, groups[gi].size(), b->layout[gi], 0,
                        "bind group %zu is not set or has the wrong binding count", gi);
        for (gs_buffer *buf : groups[gi]) {
            a.ptr[n] = buf->ptr;
            a.size[n] = buf->bytes;
            n++;
        }
    }
    a.reg_second_group = b->layout.size() > 1 ? 1u : 0u;
    a.reg_has_constant = b->has_additional_constant ? 1u : 0u;
    a.reg_constant = b->additional_constant;
    if (!b->from_source) GS_TRY(validate_bindings(b, a, n));
    GS_TRY(use_device(b->dev));
    // compute_bundle.rs:131 — dispatch_workgroups(count.div_ceil(workgroup_size), 1, 1)
    uint32_t wgs = count / b->workgroup_size + (count % b->workgroup_size != 0);
    b->last_workgroups = wgs;
    if (wgs == 0) return GS_OK;
    if (b->from_source) {
        struct { gs::BundleArgs a; uint32_t count; } params{a, count};
        size_t psize = sizeof(params);
        void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &params, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize,
                          HIP_LAUNCH_PARAM_END};
        GS_HIP(hipModuleLaunchKernel(b->func, wgs, 1, 1, b->workgroup_size, 1, 1, 0, s->s, nullptr, config));
        return GS_OK;
    }
    bundle_kernel_fn fn = nullptr;
    switch (b->kernel) {
    case GS_KERNEL_ARRAY_MAP_ADD: fn = gs::k_array_map_add; break;
    case GS_KERNEL_TEST_GAUSSIAN: fn = k_tbl_test_gaussian[b->sh][b->cov]; break;
    case GS_KERNEL_TEST_GAUSSIAN_TRANSFORM: fn = gs::k_test_gaussian_transform; break;
    case GS_KERNEL_TEST_MODEL_TRANSFORM: fn = gs::k_test_model_transform; break;
    case GS_KERNEL_UNPACK_SOA: fn = k_tbl_unpack_soa[b->sh][b->cov]; break;
    default: return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "bad kernel");
    }
    hipLaunchKernelGGL(fn, dim3(wgs), dim3(b->workgroup_size), 0, s->s, a, count);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

extern "C" gs_status gs_bundle_dispatch(gs_bundle *b, gs_stream *s, uint32_t count) {
    if (!b) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null bundle");
    if (!b->managed)
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0,
                    "bundle was created without bind groups: use gs_bundle_dispatch_with_bind_groups");
    return dispatch_groups(b, s, count, b->groups);
}

extern "C" gs_status gs_bundle_dispatch_with_bind_groups(gs_bundle *b, gs_stream *s, uint32_t count,
                                                         gs_buffer *const *const *groups,
                                                         const uint32_t *group_counts,
                                                         uint32_t group_count) {
    if (!b || (group_count && (!groups || !group_counts)))
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    std::vector<std::vector<gs_buffer *>> gv(group_count);
    for (uint32_t i = 0; i < group_count; i++) {
        for (uint32_t k = 0; k < group_counts[i]; k++) {
            if (!groups[i][k]) return fail(GS_ERR_INVALID_ARGUMENT, i, k, 0, "null binding");
            gv[i].push_back(groups[i][k]);
        }
    }
    return dispatch_groups(b, s, count, gv);
}

// ------------------------------------------------------------------------------------------------
// renderer
// ------------------------------------------------------------------------------------------------

extern "C" void gs_camera_look_at(const float eye[3], const float target[3], const float up[3],
                                  float vfov, uint32_t width, uint32_t height, float near_plane,
                                  float far_plane, gs_camera *out) {
    // glam Mat4::look_at_rh; evaluated in double and rounded once
    double f[3], s[3], u[3], fl = 0, sl = 0;
    for (int i = 0; i < 3; i++) {
        f[i] = (double)target[i] - (double)eye[i];
        fl += f[i] * f[i];
    }
    fl = std::sqrt(fl);
    for (int i = 0; i < 3; i++) f[i] /= fl;
    s[0] = f[1] * up[2] - f[2] * up[1];
    s[1] = f[2] * up[0] - f[0] * up[2];
    s[2] = f[0] * up[1] - f[1] * up[0];
    for (int i = 0; i < 3; i++) sl += s[i] * s[i];
    sl = std::sqrt(sl);
    for (int i = 0; i < 3; i++) s[i] /= sl;
    u[0] = s[1] * f[2] - s[2] * f[1];
    u[1] = s[2] * f[0] - s[0] * f[2];
    u[2] = s[0] * f[1] - s[1] * f[0];
    double e[3] = {eye[0], eye[1], eye[2]};
    double ds = s[0] * e[0] + s[1] * e[1] + s[2] * e[2];
    double du = u[0] * e[0] + u[1] * e[1] + u[2] * e[2];
    double df = f[0] * e[0] + f[1] * e[1] + f[2] * e[2];
    float *v = out->view;
    v[0] = (float)s[0]; v[1] = (float)u[0]; v[2] = (float)-f[0]; v[3] = 0.0f;
    v[4] = (float)s[1]; v[5] = (float)u[1]; v[6] = (float)-f[1]; v[7] = 0.0f;
    v[8] = (float)s[2]; v[9] = (float)u[2]; v[10] = (float)-f[2]; v[11] = 0.0f;
    v[12] = (float)-ds; v[13] = (float)-du; v[14] = (float)df; v[15] = 1.0f;
    std::memcpy(out->pos, eye, 12);
    double focal = 0.5 * (double)height / std::tan(0.5 * (double)vfov);
    out->fx = (float)focal;
    out->fy = (float)focal;
    out->cx = 0.5f * (float)width;
    out->cy = 0.5f * (float)height;
    out->near_plane = near_plane;
    out->far_plane = far_plane;
    out->width = width;
    out->height = height;
    out->background[0] = out->background[1] = out->background[2] = 0.0f;
}

struct DevArray {
    void *ptr = nullptr;
    size_t bytes = 0;
};

static gs_status dev_reserve(DevArray &a, size_t bytes) {
    if (a.bytes >= bytes && a.ptr) return GS_OK;
    if (a.ptr) GS_HIP(hipFree(a.ptr));
    a.ptr = nullptr;
    a.bytes = 0;
    size_t want = bytes + bytes / 8 + 256;
    GS_HIP(hipMalloc(&a.ptr, want));
    a.bytes = want;
    return GS_OK;
}

static void dev_free(DevArray &a) {
    if (a.ptr) (void)hipFree(a.ptr);
    a.ptr = nullptr;
    a.bytes = 0;
}

// stage indices of gs_frame_stats.stage_ms
enum { ST_REPACK = 0, ST_PRE, ST_SCAN, ST_DSORT, ST_EXPAND, ST_TSORT, ST_RANGES, ST_BLEND, ST_FRAME, ST_COUNT };

// what sized the scratch of the last frame: a change means the pair count may jump, so the next
// frame measures it first (one blocking "sizing" frame) instead of trusting the history
struct FrameShape {
    uint64_t n = 0;
    uint32_t width = 0, height = 0, band0 = 0, band1 = 0;
    bool operator==(const FrameShape &o) const {
        return n == o.n && width == o.width && height == o.height && band0 == o.band0 && band1 == o.band1;
    }
};

struct gs_renderer {
    gs_device *dev;
    DevArray order_r2, keep_bits, r2_scan, box_table;         // two-round frames: mirror slots of the Gaussians round 2 keeps, in depth order; one bit per slot
    DevArray recs, depth, rect, sorted_rect, exp_sums, cursors, chunk_tiles, chunk_vis, state, zero_region, scan_tmp, block_list;
    DevArray cull_status;                 // k_block_cull: one (tag << 10 | count) word per group of 256 blocks
    DevArray chunk_hist;                  // [chunks][256] first-digit histogram of every chunk's depth keys (PreOut::chunk_hist)
    bool list_mode = false;               // the last frame's per-slot arrays are in LIST space (k_block_cull ran)
    bool rank_inject_set = false;         // GS3D_TEST_RANK_FAULT: the watchdog's test hook has been armed
    // Depth sort of the frame: MSD-first (one scatter on the top digit + k_bucket_sort) or the LSD passes.  The choice
    // follows the largest top-digit bucket the last frames reported (FrameResult::depth_bucket_max): MSD-first while
    // the buckets fit a workgroup's registers, LSD while they do not; a shape's first frame guesses from N.
    bool depth_msd = false;               // mode of the last frame
    uint32_t depth_bucket_seen = 0;       // newest reported bucket size the mode was chosen from (diagnostic)
    bool tile_msd = false;                // the tile sort of the last frame was MSD-first
    int depth_msd_req = -1, tile_msd_req = -1;   // gs_renderer_set_sort_mode: -1 = the renderer chooses
    int tile_masks_req = -1;              // gs_renderer_set_tile_masks
    bool tile_masks = false;              // the last frame ran tile rect version 4
    bool two_round = false;               // the last frame took two rounds (its taps hold round 2 only)
    bool partitioned = false;             // ... and sorted each round on its own (gs_sort_info)
    int rounds_req = -1;                  // gs_renderer_set_rounds: -1 the renderer decides, 0 one round, 1 two
    uint32_t round1_req = 0;              // ... Gaussians of round 1 (0: a quarter of the visible ones)
    uint32_t round1 = 0;                  // Gaussians the last two-round frame's first round covered
    // what the renderer's own choice of the rounds rests on: the pair count of a single-round frame of this shape (the
    // sizing pass's, or the newest single-round report with the visible count it came with), the rounds of the frames
    // behind the two result blocks, and the feedback state (the length of round 1 is scaled up while round 1 finishes
    // too few tiles; past 3.4 x the renderer stays with one round until the shape changes)
    uint64_t full_pairs = 0;
    uint32_t full_pairs_v = 0, rounds_epoch = 0, rounds_fb_gen = 0;
    uint8_t done_rounds[2] = {1, 1};
    uint32_t done_round_k[2] = {0, 0};
    float round_scale = 1.0f;
    bool rounds_off = false;
    uint32_t rounds_off_gen = 0;          // the frame that switched the rounds off (another try 512 frames later)
    uint64_t round_cap = 0;               // the pair bound the last two-round frame used for its grids (0: the buffers' capacity)
    uint32_t round_cap_k = 0;             // ... and the length of round 1 it was measured with
    bool auto_deep = false;               // the renderer's last own choice (kept while no report is available)
    bool auto_all_done = false;           // the newest two-round report of this shape: round 1 finished every tile (round 2 was skipped)
    uint64_t auto_k = 0;
    bool wt_pairs = true;                 // k_pairs_emit stores write-through (gs::store16)
    uint64_t tile_msd_fail_d = 0;         // pair count at which the MSD-first tile sort last reported an oversized bucket (0: never)
    bool state_tile_bmax_dirty = false;   // FrameState::tile_bucket_max holds a value of an MSD-first frame
    uint32_t cull_last_gen = 0, cull_last_groups = 0;   // frame / group count of the last k_block_cull (status tags)
    DevArray dkeys[2], dvals[2];          // (depth bits - bias, mirror slot), capacity N
    DevArray tkeys[2], tvals[2];          // (tile id, mirror slot), capacity pair_capacity
    DevArray ghist, digit_totals, bucket_starts;
    gs::FrameResult *results;             // pinned, [2]: one per frame parity
    uint32_t *host_counters;              // pinned: sizing pass total
    uint64_t pair_capacity;
    FrameShape shape;
    uint32_t gen;                         // frame generation (tags the pinned results)
    hipEvent_t done[2];                   // end of the frame of each parity
    bool done_valid[2];
    uint32_t done_gen[2];
    uint32_t done_shape[2];               // shape_epoch of the frame behind each done event
    uint32_t shape_epoch = 0;             // counts the changes of `shape` (a new epoch starts with a sizing frame)
    // last frame (host-side knowledge; V and D live in results[gen & 1])
    uint64_t n;
    uint32_t tiles_x, tiles_y, sort_passes;
    uint32_t key_bias;
    int dsorted_side, tsorted_side;
    bool wide_tiles;  // tile keys are u32 (more than 65536 tiles) instead of u16
    bool rect32;      // the last frame's tile rects are packed (gs::rect_pack32)
    uint32_t launches;                    // kernel launches of the last frame (diagnostic)
    uint32_t *flags_target = nullptr;     // device word that receives every frame's flags (gs_renderer_set_frame_flags_target)
    hipStream_t last_stream;
    bool have_frame;                      // last_stream is meaningful (the null stream is a valid stream)
    bool last_stream_gone = false;        // ... but has been destroyed since (gs_stream_destroy recorded done[gen & 1] on it)
    gs_buffer *last_order;   // mirror order of the last frame's buffer (null = index order), for the taps
    // timing
    bool timing;
    hipEvent_t ev[ST_COUNT + 2];
    bool ev_valid;
    bool ev_pending;
    double stage_ms[ST_COUNT];
    uint32_t timed_frames;
};

// gs_stream_destroy: the end-of-frame event of every renderer whose last frame is on `st` is recorded now, and the
// renderer remembers that the stream is gone (it must not be touched again: waits go through the event).
static void renderers_leave_stream(gs_device *dev, hipStream_t st) {
    std::lock_guard<std::mutex> lock(dev->renderers_mu);
    for (gs_renderer *r : dev->renderers) {
        if (!r->have_frame || r->last_stream_gone || r->last_stream != st) continue;
        r->done_valid[r->gen & 1u] = hipEventRecord(r->done[r->gen & 1u], st) == hipSuccess;
        r->last_stream_gone = true;
    }
    (void)hipGetLastError();
}

// host wait for the renderer's last frame: its stream, or — the stream was destroyed — its end-of-frame event
static hipError_t sync_last_frame(gs_renderer *r) {
    if (!r->have_frame) return hipSuccess;
    if (!r->last_stream_gone) return hipStreamSynchronize(r->last_stream);
    return r->done_valid[r->gen & 1u] ? hipEventSynchronize(r->done[r->gen & 1u]) : hipSuccess;
}

extern "C" gs_status gs_renderer_create(gs_device *dev, gs_renderer **out) {
    if (!out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null out");
    GS_TRY(use_device(dev));
    gs_renderer *r = new gs_renderer();
    r->dev = dev;
    r->host_counters = nullptr;
    r->results = nullptr;
    hipError_t e = hipHostMalloc((void **)&r->host_counters, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&r->results, 2 * sizeof(gs::FrameResult), hipHostMallocMapped | hipHostMallocCoherent);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        // no system-scope fence at the event: what the host reads after it (the frame result) lives in
        // coherent pinned memory, and images are fetched with copies that synchronise by themselves
        static const bool fence = std::getenv("GS3D_EVENT_FENCE") && std::getenv("GS3D_EVENT_FENCE")[0] == '1';
        e = hipEventCreateWithFlags(&r->done[i], hipEventDisableTiming | (fence ? 0u : hipEventDisableSystemFence));
        r->done_valid[i] = false;
        r->done_gen[i] = 0;
        r->done_shape[i] = 0;
    }
    if (e != hipSuccess) {
        if (r->host_counters) (void)hipHostFree(r->host_counters);
        if (r->results) (void)hipHostFree(r->results);
        delete r;
        return fail(GS_ERR_HIP, (uint64_t)e, 0, 0, "renderer allocation failed: %s", hipGetErrorString(e));
    }
    std::memset(r->results, 0, 2 * sizeof(gs::FrameResult));
    r->pair_capacity = 0;
    r->gen = 0;
    r->n = 0;
    r->tiles_x = r->tiles_y = r->sort_passes = 0;
    r->key_bias = 0;
    r->dsorted_side = r->tsorted_side = 0;
    r->wide_tiles = false;
    r->rect32 = false;
    r->launches = 0;
    r->last_stream = nullptr;
    r->have_frame = false;
    r->last_order = nullptr;
    r->timing = false;
    r->ev_valid = false;
    r->ev_pending = false;
    r->timed_frames = 0;
    for (int i = 0; i < ST_COUNT; i++) r->stage_ms[i] = 0.0;
    {
        std::lock_guard<std::mutex> lock(dev->renderers_mu);
        dev->renderers.push_back(r);
    }
    *out = r;
    return GS_OK;
}

extern "C" void gs_renderer_destroy(gs_renderer *r) {
    if (!r) return;
    (void)hipSetDevice(r->dev->ordinal);
    {
        std::lock_guard<std::mutex> lock(r->dev->renderers_mu);
        auto &v = r->dev->renderers;
        for (size_t i = 0; i < v.size(); i++)
            if (v[i] == r) {
                v[i] = v.back();
                v.pop_back();
                break;
            }
    }
    (void)sync_last_frame(r);   // kernels of the last frame write pinned memory
    DevArray *arrs[] = {&r->recs, &r->depth, &r->rect, &r->sorted_rect, &r->exp_sums, &r->cursors, &r->chunk_tiles, &r->chunk_vis,
                        &r->state, &r->zero_region, &r->order_r2, &r->keep_bits, &r->r2_scan, &r->box_table, &r->scan_tmp, &r->block_list, &r->cull_status, &r->chunk_hist, &r->dkeys[0], &r->dkeys[1], &r->dvals[0],
                        &r->dvals[1], &r->tkeys[0], &r->tkeys[1], &r->tvals[0], &r->tvals[1], &r->ghist,
                        &r->digit_totals, &r->bucket_starts};
    for (DevArray *a : arrs) dev_free(*a);
    if (r->host_counters) (void)hipHostFree(r->host_counters);
    if (r->results) (void)hipHostFree(r->results);
    for (auto &e : r->done) (void)hipEventDestroy(e);
    if (r->last_order) gs_buffer_release(r->last_order);
    if (r->ev_valid)
        for (auto &e : r->ev) (void)hipEventDestroy(e);
    delete r;
}

extern "C" gs_status gs_renderer_set_timing(gs_renderer *r, int32_t enabled) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null renderer");
    GS_TRY(use_device(r->dev));
    if (enabled && !r->ev_valid) {
        for (auto &e : r->ev) GS_HIP(hipEventCreate(&e));
        r->ev_valid = true;
    }
    r->timing = enabled != 0;
    return GS_OK;
}

extern "C" gs_status gs_renderer_set_frame_flags_target(gs_renderer *r, uint32_t *device_word) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null renderer");
    if ((uintptr_t)device_word & 3u) return fail(GS_ERR_INVALID_ARGUMENT, (uint64_t)(uintptr_t)device_word, 4, 0, "unaligned flags word");
    r->flags_target = device_word;
    return GS_OK;
}

// fold the events of the previous timed frame into the accumulators
static gs_status collect_timing(gs_renderer *r) {
    if (!r->ev_pending) return GS_OK;
    GS_HIP(hipEventSynchronize(r->ev[ST_COUNT]));
    for (int i = 0; i < ST_FRAME; i++) {
        float ms = 0;
        GS_HIP(hipEventElapsedTime(&ms, r->ev[i], r->ev[i + 1]));
        r->stage_ms[i] += ms;
    }
    float ms = 0;
    GS_HIP(hipEventElapsedTime(&ms, r->ev[0], r->ev[ST_FRAME]));
    r->stage_ms[ST_FRAME] += ms;
    r->timed_frames++;
    r->ev_pending = false;
    return GS_OK;
}

extern "C" gs_status gs_renderer_reset_stats(gs_renderer *r) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null renderer");
    GS_TRY(use_device(r->dev));
    GS_TRY(collect_timing(r));
    for (int i = 0; i < ST_COUNT; i++) r->stage_ms[i] = 0.0;
    r->timed_frames = 0;
    return GS_OK;
}

// result block of the most recent frame (valid once its stream work has completed)
static const gs::FrameResult &last_result(const gs_renderer *r) { return r->results[r->gen & 1u]; }

extern "C" gs_status gs_renderer_wait_frame(gs_renderer *r, gs_frame_result *out) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null renderer");
    GS_TRY(use_device(r->dev));
    if (out) std::memset(out, 0, sizeof(*out));
    if (!r->have_frame) return GS_OK;       // no frame yet
    GS_HIP(sync_last_frame(r));
    const gs::FrameResult &fr = last_result(r);
    if (out) {
        out->gaussians = r->n;
        out->visible = fr.visible;
        out->pairs = fr.pairs_total;
        out->pair_capacity = r->pair_capacity;
        out->flags = fr.flags;
        out->launches = r->launches;
    }
    if (fr.gen != r->gen)
        return fail(GS_ERR_HIP, fr.gen, r->gen, 0, "the frame did not complete (result generation %u, expected %u)",
                    fr.gen, r->gen);
    if (fr.pairs_total > 0xfffffff0ull)
        return fail(GS_ERR_PAIR_OVERFLOW, r->n, 0, 0,
                    "the frame needs more than 2^32 (tile, Gaussian) pairs; pair indices are 32-bit");
    if (fr.flags & gs::FRAME_FLAG_RANK_FAULT) {
        // The device drops to the ballot-based rank, and THIS renderer's watchdog word is cleared whether or not it was
        // this renderer that flipped the switch: with several renderers on one device (FrameRing, parallel.lanes) the
        // second one to report used to find the switch already off, keep its word set, and flag every later frame.
        r->dev->lds_atomic_ordered.store(false);
        (void)hipMemset(&((gs::FrameState *)r->state.ptr)->rank_fault, 0, sizeof(uint32_t));   // the stream is idle here
        return fail(GS_ERR_RANK_ORDER, 0, 0, 0,
                    "the LDS-atomic rank of the radix sort returned an out-of-order value in this frame: its blend order "
                    "may be wrong; the device has been switched to the ballot-based rank: render again");
    }
    if (fr.flags & gs::FRAME_FLAG_PAIR_OVERFLOW)
        return fail(GS_ERR_PAIR_CAPACITY, fr.pairs_total, r->pair_capacity, 0,
                    "the frame produced %llu (tile, Gaussian) pairs but the pair buffers hold %llu: the frame was "
                    "skipped (the image was not written); render again (the next frame grows the buffers)",
                    (unsigned long long)fr.pairs_total, (unsigned long long)r->pair_capacity);
    return GS_OK;
}

extern "C" gs_status gs_renderer_stats(gs_renderer *r, gs_frame_stats *out) {
    if (!r || !out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    GS_TRY(use_device(r->dev));
    GS_HIP(sync_last_frame(r));
    GS_TRY(collect_timing(r));
    std::memset(out, 0, sizeof(*out));
    const gs::FrameResult &fr = last_result(r);
    out->gaussians = r->n;
    out->visible = r->have_frame ? fr.visible : 0;
    out->pairs = r->have_frame ? fr.pairs_total : 0;
    out->tiles_x = r->tiles_x;
    out->tiles_y = r->tiles_y;
    out->sort_passes = r->sort_passes;
    out->timed_frames = r->timed_frames;
    static_assert(ST_COUNT <= 12, "gs_frame_stats.stage_ms too small");
    for (int i = 0; i < ST_COUNT; i++) out->stage_ms[i] = r->stage_ms[i];
    return GS_OK;
}

extern "C" gs_status gs_renderer_sort_info(gs_renderer *r, gs_sort_info *out) {
    if (!r || !out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    GS_TRY(use_device(r->dev));
    std::memset(out, 0, sizeof(*out));
    out->bucket_capacity = gs::BKT_CAP;
    if (!r->have_frame) return GS_OK;
    GS_HIP(sync_last_frame(r));
    const gs::FrameResult &fr = last_result(r);
    out->depth_msd = r->depth_msd ? 1u : 0u;
    out->depth_bucket_max = fr.gen == r->gen ? fr.depth_bucket_max : 0u;
    out->tile_msd = r->tile_msd ? 1u : 0u;
    out->tile_masks = r->tile_masks ? 1u : 0u;
    out->rounds = r->two_round ? 2u : 1u;
    out->round1 = r->two_round ? r->round1 : 0u;
    out->tiles_done = r->two_round && fr.gen == r->gen ? fr.tiles_done : 0u;
    out->partitioned = r->two_round && r->partitioned ? 1u : 0u;
    if (r->tile_msd && r->state.ptr)      // (the result block carries the PREVIOUS frame's: read this frame's from the device)
        GS_HIP(hipMemcpy(&out->tile_bucket_max, &((gs::FrameState *)r->state.ptr)->tile_bucket_max, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return GS_OK;
}

extern "C" gs_status gs_renderer_set_tile_masks(gs_renderer *r, int32_t mode) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null renderer");
    if (mode < -1 || mode > 1) return fail(GS_ERR_INVALID_ARGUMENT, (uint64_t)(int64_t)mode, 0, 0, "tile mask modes are -1, 0 or 1");
    r->tile_masks_req = mode;
    return GS_OK;
}

extern "C" gs_status gs_renderer_set_rounds(gs_renderer *r, int32_t mode, uint32_t first_round) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null renderer");
    if (mode < -1 || mode > 1) return fail(GS_ERR_INVALID_ARGUMENT, (uint64_t)(int64_t)mode, 0, 0, "round modes are -1, 0 or 1");
    r->rounds_req = mode;
    r->round1_req = first_round;
    return GS_OK;
}

extern "C" gs_status gs_renderer_set_sort_mode(gs_renderer *r, int32_t depth_msd, int32_t tile_msd) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null renderer");
    if (depth_msd < -1 || depth_msd > 1 || tile_msd < -1 || tile_msd > 1)
        return fail(GS_ERR_INVALID_ARGUMENT, (uint64_t)(int64_t)depth_msd, (uint64_t)(int64_t)tile_msd, 0, "sort modes are -1, 0 or 1");
    r->depth_msd_req = depth_msd;
    r->tile_msd_req = tile_msd;
    return GS_OK;
}

typedef void (*preprocess_fn)(const uint4 *, uint32_t, gs::FrameConsts, gs::PreOut);
typedef void (*block_bounds_fn)(const uint4 *, uint32_t, float *);
#define GS_CFG_TABLE_X(kernel, ...)                                                                              \
    {                                                                                                             \
        {kernel<0, 0, __VA_ARGS__>, kernel<0, 1, __VA_ARGS__>, kernel<0, 2, __VA_ARGS__>},                        \
        {kernel<1, 0, __VA_ARGS__>, kernel<1, 1, __VA_ARGS__>, kernel<1, 2, __VA_ARGS__>},                        \
        {kernel<2, 0, __VA_ARGS__>, kernel<2, 1, __VA_ARGS__>, kernel<2, 2, __VA_ARGS__>},                        \
        {kernel<3, 0, __VA_ARGS__>, kernel<3, 1, __VA_ARGS__>, kernel<3, 2, __VA_ARGS__>},                        \
    }
// [non-temporal mirror loads][...]: the cache policy is a template parameter (a run-time flag put a
// branch and a wait behind every load)
static preprocess_fn k_tbl_preprocess[2][4][3] = {GS_CFG_TABLE_X(gs::k_preprocess, false), GS_CFG_TABLE_X(gs::k_preprocess, true)};
// [nt][pipelined]
static preprocess_fn k_tbl_preprocess_banded[2][2][4][3] = {
    {GS_CFG_TABLE_X(gs::k_preprocess_banded, false, false), GS_CFG_TABLE_X(gs::k_preprocess_banded, true, false)},
    {GS_CFG_TABLE_X(gs::k_preprocess_banded, false, true), GS_CFG_TABLE_X(gs::k_preprocess_banded, true, true)}};
static block_bounds_fn k_tbl_block_bounds[4][3] = GS_CFG_TABLE(gs::k_block_bounds);

// frame constants from the uniforms
static void make_frame_consts(const gs_gaussian_transform_pod *gt, const gs_model_transform_pod *mt,
                              const gs_camera *cam, uint32_t band_ty0, uint32_t band_ty1,
                              gs::FrameConsts &fc) {
    gs::ModelTransform m;
    std::memcpy(&m, mt, sizeof(m));
    gs::model_transform_mat(m, fc.M);
    gs::model_transform_inv_sr_mat(m, fc.ISR);
    float sr[9];
    gs::model_scale_rot_mat(m, sr);
    std::memcpy(fc.V, cam->view, 64);
    for (int r = 0; r < 3; r++) {
        float sg = r == 0 ? 1.0f : -1.0f;
        float w0 = sg * cam->view[0 + r], w1 = sg * cam->view[4 + r], w2 = sg * cam->view[8 + r];
        for (int c = 0; c < 3; c++)
            fc.WS[3 * r + c] = (w0 * sr[3 * c + 0] + w1 * sr[3 * c + 1]) + w2 * sr[3 * c + 2];
    }
    std::memcpy(fc.cam_pos, cam->pos, 12);
    fc.fx = cam->fx;
    fc.fy = cam->fy;
    fc.cx = cam->cx;
    fc.cy = cam->cy;
    fc.near_plane = cam->near_plane;
    fc.far_plane = cam->far_plane;
    uint32_t flags;
    std::memcpy(&flags, gt->flags, 4);
    fc.size2 = gt->size * gt->size;
    fc.limx = 1.3f * ((0.5f * (float)cam->width) / cam->fx);
    fc.limy = 1.3f * ((0.5f * (float)cam->height) / cam->fy);
    fc.max_std_dev = gs::gaussian_transform_max_std_dev(flags);
    std::memcpy(fc.bg, cam->background, 12);
    fc.sh_deg = gs::gaussian_transform_sh_deg(flags);
    fc.no_sh0 = gs::gaussian_transform_no_sh0(flags) ? 1u : 0u;
    fc.width = cam->width;
    fc.height = cam->height;
    fc.tiles_x = (cam->width + 15u) / 16u;
    fc.tiles_y = (cam->height + 15u) / 16u;
    fc.band_ty0 = band_ty0 < fc.tiles_y ? band_ty0 : fc.tiles_y;
    fc.band_ty1 = band_ty1 < fc.tiles_y ? band_ty1 : fc.tiles_y;
    if (fc.band_ty1 < fc.band_ty0) fc.band_ty1 = fc.band_ty0;
    fc.mask_culled_records = 0;
    fc.nt_loads = 0;
    // in display mode Splat the tile rect is clipped to the splat's visible box.
    // GS3D_RECT_V1=1 keeps the unclipped rect of spec version 1 (same images, more pairs) for A/B runs
    // and for the parity tests against the version-1 goldens.
    {
        static const bool rect_v1 = std::getenv("GS3D_RECT_V1") && std::getenv("GS3D_RECT_V1")[0] == '1';
        fc.clip_rect = gt->flags[0] == GS_DISPLAY_SPLAT && !rect_v1 ? 1u : 0u;
        // rect version 4: small rects lose the tiles their splat cannot reach.  GS3D_TILE_MASKS=0: version 3
        // (make_frame_consts only records that the display mode allows it; gs_render_frame decides: gs_renderer_set_tile_masks)
        fc.tile_masks = fc.clip_rect;
    }
    fc.ellipse_pmin = -0.5f * (fc.max_std_dev * fc.max_std_dev);
    {   // packed 4-byte tile rects while both tile counts fit 8 bits (images up to 4096 px); GS3D_RECT32=0: always uint2
        static const bool rect32_off = std::getenv("GS3D_RECT32") && std::getenv("GS3D_RECT32")[0] == '0';
        fc.rect32 = !rect32_off && fc.tiles_x <= 256u && fc.tiles_y <= 256u && fc.tiles_x * fc.tiles_y <= 32768u ? 1u : 0u;
        // write-through stores of the 16-byte-per-lane outputs (gs::store16).  GS3D_WT_STORES=0/1
        // GS3D_WT_STORES: bit 0 = the pairs of k_pairs_emit, bit 1 = the image
        static const int wt_env = std::getenv("GS3D_WT_STORES") ? std::atoi(std::getenv("GS3D_WT_STORES")) : 3;
        fc.wt_stores = (wt_env & 2) ? 1u : 0u;
        fc.wt_pairs = (wt_env & 1) ? 1u : 0u;
        static const int wtr_env = std::getenv("GS3D_WT_RECORDS") ? std::atoi(std::getenv("GS3D_WT_RECORDS")) : 0;
        fc.wt_records = (uint32_t)wtr_env & 3u;
    }
    // block culling gain (see block_is_culled): size^2 |W R_m S_m|_2^2, the squared SPECTRAL norm of the linear part
    // whatever the caller's view and model matrices are: the largest eigenvalue of A = (WS)^T (WS), in double by the
    // closed form for symmetric 3x3 matrices, never above the trace (= the squared Frobenius norm, the bound of rounds
    // 2-3); 0.1 % head room for the f32 arithmetic.  The Jacobian's norm is taken per block on the device.
    {
        double a[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) a[i][j] += (double)fc.WS[3 * k + i] * (double)fc.WS[3 * k + j];
        const double tr = a[0][0] + a[1][1] + a[2][2];
        double lmax = tr;
        const double p1 = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double q = tr / 3.0;
        const double p2 = (a[0][0] - q) * (a[0][0] - q) + (a[1][1] - q) * (a[1][1] - q) + (a[2][2] - q) * (a[2][2] - q) + 2.0 * p1;
        const double pp = std::sqrt(p2 / 6.0);
        if (pp == 0.0) {
            lmax = q * (1.0 + 1e-6);          // A = q I (a rotation times a uniform scale: the identity model transform)
        } else if (pp > 0.0 && std::isfinite(pp)) {
            double b[3][3];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) b[i][j] = (a[i][j] - (i == j ? q : 0.0)) / pp;
            double rdet = (b[0][0] * (b[1][1] * b[2][2] - b[1][2] * b[2][1]) - b[0][1] * (b[1][0] * b[2][2] - b[1][2] * b[2][0]) +
                           b[0][2] * (b[1][0] * b[2][1] - b[1][1] * b[2][0])) / 2.0;
            rdet = rdet < -1.0 ? -1.0 : (rdet > 1.0 ? 1.0 : rdet);
            const double l = q + 2.0 * pp * std::cos(std::acos(rdet) / 3.0);
            if (std::isfinite(l) && l > 0.0 && l <= tr) lmax = l * (1.0 + 1e-6) + 1e-30;
        }
        fc.cull_gain = (float)(1.001 * (double)fc.size2 * lmax);
    }
    if (!(fc.cull_gain > 0.0f) || !(fc.cull_gain < 1e30f)) fc.cull_gain = 0.0f;   // degenerate uniforms: no block culling
}

static uint32_t bit_length(uint32_t v) {
    uint32_t b = 0;
    while (v) {
        b++;
        v >>= 1;
    }
    return b;
}

// depth keys: 9-bit digits when that saves a pass (e.g. 27 significant bits: 3 passes instead of 4);
// otherwise 8-bit digits, whose 256-bin tiles write longer runs and fit more workgroups per CU
static uint32_t depth_radix_bits(uint32_t key_bits) {
    return (key_bits + 8) / 9 < (key_bits + 7) / 8 ? (uint32_t)gs::RADIX_BITS_MAX : (uint32_t)gs::RADIX_BITS;
}

// mask of the FIRST digit of a sort of `key_bits` bits in passes of at most `rb` bits (the balanced widths of
// run_sort_items: 27 bits -> 9 + 9 + 9, 13 -> 7 + 6); the preprocess kernel counts that digit per chunk
static uint32_t first_digit_mask(uint32_t key_bits, uint32_t rb) {
    const uint32_t passes = (key_bits + rb - 1) / rb;
    const uint32_t bits = passes ? (key_bits + passes - 1) / passes : 0u;
    return (1u << bits) - 1u;
}

// The frame's compacting first pass (see gs_render_kernels.h, COMPACT): dense keys in, the index is
// the value, chunks without visible Gaussians are skipped, V comes out in *visible_out.
struct SortCompact {
    const uint32_t *dense_keys = nullptr;   // [N] keys by slot, 0xffffffff = culled (read by pass 0 instead of keys[0])
    const uint32_t *chunk_vis = nullptr;
    uint32_t *visible_out = nullptr;
    uint32_t dense_count = 0;        // N: the first pass runs over all slots (host bound: sizes the grid)
    const uint32_t *dense_count_dev = nullptr;   // optional device word: the slots that really hold data (list frames)
    const uint32_t *chunk_hist = nullptr;        // per-chunk histogram of the first digit, counted by the preprocess kernel
};

// Stable LSD radix sort of (key, u32 value) pairs on key bits [0, end_bit), RB bits per pass at
// most (digit widths balanced over the passes), ping-ponging between side 0 and side 1; the side
// holding the result is returned.  `sc` = host bound of the count (sizes the grid) and, optionally,
// the device word holding the real count.  With `compact`, pass 0 reads keys[0] as the dense key
// array of preprocess and ignores vals[0].
template <int TILE>
static void launch_scan_rows(uint32_t rows, hipStream_t st, uint32_t *ghist, uint32_t stride, gs::SortCount sc,
                             uint32_t *totals) {
    // rows of up to SCAN_ROWS_SMALL_MAX blocks (stride = the host's bound of the block count): one wave per row
    static const bool small_off = std::getenv("GS3D_SCAN_ROWS_SMALL") && std::getenv("GS3D_SCAN_ROWS_SMALL")[0] == '0';
    if (stride <= gs::SCAN_ROWS_SMALL_MAX && !small_off)
        hipLaunchKernelGGL((gs::k_sort_scan_rows_small<TILE>), dim3((rows + 3u) / 4u), dim3(256), 0, st, ghist, stride, sc,
                           totals, rows);
    else
        hipLaunchKernelGGL((gs::k_sort_scan_rows<TILE>), dim3(rows), dim3(gs::SCAN_ROWS_THREADS), 0, st, ghist, stride, sc,
                           totals);
}

// Watchdog word of the LDS-atomic rank for the scatters launched by the current gs_render_frame call (FrameState::
// rank_fault; null outside a frame: the stand-alone sorts are not watched).  Thread-local instead of one more
// parameter through five levels of sort templates.
static thread_local uint32_t *t_rank_fault = nullptr;
// ... and what else the frame hands its scatters the same way: the watchdog's sample of this frame (gs::k_sort_scatter:
// tile = watch mod live tiles, round = (watch / live tiles) mod ITEMS; the frame generation, so that every position is
// visited over a few hundred frames), and the word that receives the largest top-digit bucket of the depth sort
// (only the LSD sort's LAST pass gets it: t_top_pass).
static thread_local uint32_t t_watch = 0;
static thread_local uint32_t *t_bucket_max = nullptr;
static thread_local bool t_top_pass = false;
static thread_local uint32_t *t_bucket_starts = nullptr;    // the MSD-first sorts' scatter pass writes every bucket's start here
// ... and which side of a partitioned two-round frame's depth threshold the COMPACT pass keeps (gs::CompactPred; default: all)
static thread_local gs::CompactPred t_compact_pred;

// one scatter launch (FAST_RANK chosen by the device probe); KO = type of the keys the pass writes
template <typename KI, typename KO, int RB, bool COMPACT, int ITEMS>
static void launch_scatter(const gs_device *dev, hipStream_t st, uint32_t sgrid, const KI *kin, const uint32_t *vin, KO *kout,
                           uint32_t ko_shift, uint32_t *vout, gs::SortCount psc, uint32_t shift, uint32_t digit_mask,
                           const uint32_t *ghist, const uint32_t *totals, const uint32_t *cv, uint32_t *vo, uint32_t pnb,
                           uint32_t xr) {
    // inputs that cannot stay in the L2s anyway are read non-temporally (top bit of the last argument; see
    // k_sort_scatter).  GS3D_NT_SCATTER=0/1 forces.
    static const int nt_env = std::getenv("GS3D_NT_SCATTER") ? std::atoi(std::getenv("GS3D_NT_SCATTER")) : -1;
    const bool nt = !COMPACT && (nt_env >= 0 ? nt_env != 0 : (uint64_t)psc.count * (sizeof(KI) + 4u) > (32ull << 20));
    const uint32_t xr_nt = xr | (nt ? 0x80000000u : 0u);
    uint32_t *bmax = t_top_pass ? t_bucket_max : nullptr;
    // FAST_RANK: chosen by the device probe and the frame's watchdog (t_rank_fault: null outside a frame and once the device
    // has dropped to the ballot-based rank); PRED: a partitioned two-round frame's compacting pass (t_compact_pred)
    auto go = [&](auto fast_tag, auto pred_tag, uint32_t *rf, uint32_t watch) {
        constexpr bool FAST = decltype(fast_tag)::value, PRED = decltype(pred_tag)::value;
        hipLaunchKernelGGL((gs::k_sort_scatter<KI, FAST, RB, COMPACT, ITEMS, KO, PRED>), dim3(sgrid), dim3(gs::SORT_THREADS), 0, st, kin, vin,
                           kout, ko_shift, vout, psc, shift, digit_mask, ghist, totals, cv, vo, pnb, xr_nt, rf, watch, bmax, t_bucket_starts,
                           PRED ? t_compact_pred : gs::CompactPred());
    };
    const bool fast = t_rank_fault || dev->lds_atomic_ordered.load(std::memory_order_relaxed);
    uint32_t *rf = t_rank_fault;
    const uint32_t watch = t_rank_fault ? t_watch : 0u;
    if constexpr (COMPACT) {
        if (t_compact_pred.tau_dev) {
            if (fast) go(std::true_type(), std::true_type(), rf, watch);
            else go(std::false_type(), std::true_type(), (uint32_t *)nullptr, 0u);
            return;
        }
    }
    if (fast) go(std::true_type(), std::false_type(), rf, watch);
    else go(std::false_type(), std::false_type(), (uint32_t *)nullptr, 0u);
}

// one radix pass: histogram -> row scan -> scatter
template <typename KI, typename KO, int RB, bool COMPACT, int ITEMS>
static void launch_pass(const gs_device *dev, hipStream_t st, uint32_t sgrid, const KI *kin, const uint32_t *vin, KO *kout,
                        uint32_t ko_shift, uint32_t *vout, gs::SortCount psc, uint32_t shift, uint32_t digit_mask, DevArray &ghist,
                        DevArray &digit_totals, const uint32_t *cv, uint32_t *vo, uint32_t pnb, uint32_t xr,
                        const uint32_t *chunk_hist = nullptr, uint32_t chunk_hist_words = (uint32_t)gs::PP_THREADS) {
    constexpr uint32_t R = 1u << RB;
    constexpr int TILE = gs::SORT_THREADS * ITEMS;
    // GS3D_CHUNK_HIST=0: the compacting pass counts its histogram from the keys again (A/B, tests)
    static const bool chunk_hist_off = std::getenv("GS3D_CHUNK_HIST") && std::getenv("GS3D_CHUNK_HIST")[0] == '0';
    bool summed = false;
    if constexpr (COMPACT && TILE % gs::PP_CHUNK == 0) {
        if (chunk_hist && !chunk_hist_off) {
            // the preprocess kernel counted this digit per chunk: sum the chunks' rows instead of re-reading the keys
            hipLaunchKernelGGL((gs::k_sort_hist_chunks<RB, ITEMS>), dim3(sgrid), dim3(gs::SORT_THREADS), 0, st, chunk_hist, psc,
                               digit_mask, (uint32_t *)ghist.ptr, cv, pnb, xr, chunk_hist_words);
            summed = true;
        }
    }
    if (!summed) {
        bool done = false;
        if constexpr (COMPACT) {
            if (t_compact_pred.tau_dev) {
                hipLaunchKernelGGL((gs::k_sort_hist<KI, RB, COMPACT, ITEMS, true>), dim3(sgrid), dim3(gs::SORT_THREADS), 0, st, kin, psc, shift,
                                   digit_mask, (uint32_t *)ghist.ptr, cv, pnb, xr, t_compact_pred);
                done = true;
            }
        }
        if (!done)
            hipLaunchKernelGGL((gs::k_sort_hist<KI, RB, COMPACT, ITEMS>), dim3(sgrid), dim3(gs::SORT_THREADS), 0, st, kin, psc, shift,
                               digit_mask, (uint32_t *)ghist.ptr, cv, pnb, xr, gs::CompactPred());
    }
    (void)R;
    launch_scan_rows<TILE>(digit_mask + 1u, st, (uint32_t *)ghist.ptr, pnb, psc, (uint32_t *)digit_totals.ptr);   // live rows only
    launch_scatter<KI, KO, RB, COMPACT, ITEMS>(dev, st, sgrid, kin, vin, kout, ko_shift, vout, psc, shift, digit_mask,
                                               (const uint32_t *)ghist.ptr, (const uint32_t *)digit_totals.ptr, cv, vo, pnb, xr);
}

static uint32_t xcd_span_for(uint32_t pnb, uint32_t &sgrid);

template <typename K, int RB, int ITEMS>
static gs_status run_sort_items(const gs_device *dev, void *const keys[2], void *const vals[2], DevArray &ghist,
                             DevArray &digit_totals, gs::SortCount sc, uint32_t end_bit, const SortCompact *compact,
                             hipStream_t st, int &result_side, uint32_t &passes_out, uint32_t &launches,
                             const gs::ExpandIO *source = nullptr) {
    constexpr uint32_t R = 1u << RB;
    constexpr uint32_t TILE = (uint32_t)(gs::SORT_THREADS * ITEMS);
    uint32_t passes = (end_bit + RB - 1) / RB;
    if ((compact || source) && passes == 0) passes = 1;   // the compaction (and V) / the generation (and D) must happen even for a 0-bit key range
    passes_out = passes;
    result_side = 0;
    if (sc.count == 0 || passes == 0) return GS_OK;
    const uint32_t nb = (uint32_t)(((uint64_t)sc.count + TILE - 1) / TILE);
    GS_TRY(dev_reserve(ghist, (size_t)nb * R * 4));
    GS_TRY(dev_reserve(digit_totals, R * 4));
    int side = 0;
    uint32_t shift = 0;
    for (uint32_t p = 0; p < passes; p++) {
        // balanced digit widths: e.g. 13 bits -> 7 + 6, 15 -> 8 + 7, 27 -> 9 + 9 + 9, 32 -> 8 + 8 + 8 + 8
        const uint32_t bits = end_bit > shift ? (end_bit - shift + (passes - p) - 1) / (passes - p) : 0u;
        const uint32_t digit_mask = (1u << bits) - 1u;
        const bool first = compact && p == 0;
        const K *kin = first ? (const K *)compact->dense_keys : (const K *)keys[side];
        const uint32_t *vin = (const uint32_t *)vals[side];
        // the sorted depth keys themselves are never read (compact = the frame's depth sort): its last pass
        // writes the order only
        K *kout = compact && p == passes - 1 ? (K *)nullptr : (K *)keys[side ^ 1];
        uint32_t *vout = (uint32_t *)vals[side ^ 1];
        const gs::SortCount psc = first ? gs::SortCount{compact->dense_count, compact->dense_count_dev} : sc;
        const uint32_t pnb = first ? (uint32_t)(((uint64_t)compact->dense_count + TILE - 1) / TILE) : nb;
        if (first) GS_TRY(dev_reserve(ghist, (size_t)pnb * R * 4));
        const uint32_t *cv = first ? compact->chunk_vis : nullptr;
        uint32_t *vo = first ? compact->visible_out : nullptr;
        // XCD-aware tile order in the scatter (see scatter_tile_of): XCD x takes `xr` consecutive tiles
        // of every group of 8 * xr.  xr grows with the number of tiles (a group must stay a small part
        // of the pass) between 4 and 64.  GS3D_XCD_REMAP=0 disables, GS3D_XCD_REMAP_C=<n> forces a size.
        uint32_t sgrid = 0;
        const uint32_t xr = xcd_span_for(pnb, sgrid);
        t_top_pass = compact != nullptr && p == passes - 1;   // the depth sort's pass on its top digit reports the largest bucket
        if (source && p == 0) {
            // the pairs come from the depth-ordered rects: k_pairs_emit writes this pass's input
            // (keys[side] / vals[side]) and its histogram at once
            if constexpr (sizeof(K) <= 4) {
                gs::ExpandIO src = *source;
                src.tvals = (uint32_t *)vals[side];
                if (sizeof(K) == 2 && src.rect32)
                    hipLaunchKernelGGL((gs::k_pairs_emit<K, RB, ITEMS, sizeof(K) == 2>), dim3(sgrid), dim3(gs::SORT_THREADS), 0, st,
                                       src, digit_mask, (uint32_t *)ghist.ptr, (K *)keys[side], pnb, xr, 0u);
                else
                    hipLaunchKernelGGL((gs::k_pairs_emit<K, RB, ITEMS, false>), dim3(sgrid), dim3(gs::SORT_THREADS), 0, st, src,
                                       digit_mask, (uint32_t *)ghist.ptr, (K *)keys[side], pnb, xr, 0u);
                launch_scan_rows<(int)TILE>(digit_mask + 1u, st, (uint32_t *)ghist.ptr, pnb, psc, (uint32_t *)digit_totals.ptr);
                launch_scatter<K, K, RB, false, ITEMS>(dev, st, sgrid, kin, vin, kout, 0u, vout, psc, shift, digit_mask,
                                                       (const uint32_t *)ghist.ptr, (const uint32_t *)digit_totals.ptr, cv, vo, pnb, xr);
            }
        } else if constexpr (sizeof(K) == 4) {
            // Narrow keys (the frame's depth sort only): the pass BEFORE the last stores key >> (shift of
            // the last pass) as u16 — all the last pass still needs — and the last pass runs on 16-bit keys:
            // 2 bytes less per element written, 2 x 2 bytes less read.  GS3D_NARROW_KEYS=0 switches it off.
            static const bool narrow_on = !(std::getenv("GS3D_NARROW_KEYS") && std::getenv("GS3D_NARROW_KEYS")[0] == '0');
            const bool narrow = narrow_on && compact && passes >= 2;
            const bool writes_narrow = narrow && p == passes - 2, reads_narrow = narrow && p == passes - 1;
            if (reads_narrow) {
                launch_pass<uint16_t, uint16_t, RB, false, ITEMS>(dev, st, sgrid, (const uint16_t *)keys[side], vin, (uint16_t *)nullptr, 0u,
                                                                  vout, psc, 0u, digit_mask, ghist, digit_totals, cv, vo, pnb, xr);
            } else if (writes_narrow) {
                const uint32_t next_shift = shift + bits;
                if (first)
                    launch_pass<uint32_t, uint16_t, RB, true, ITEMS>(dev, st, sgrid, (const uint32_t *)kin, vin, (uint16_t *)keys[side ^ 1],
                                                                     next_shift, vout, psc, shift, digit_mask, ghist, digit_totals, cv, vo, pnb, xr,
                                                                     compact->chunk_hist);
                else
                    launch_pass<uint32_t, uint16_t, RB, false, ITEMS>(dev, st, sgrid, (const uint32_t *)kin, vin, (uint16_t *)keys[side ^ 1],
                                                                      next_shift, vout, psc, shift, digit_mask, ghist, digit_totals, cv, vo, pnb, xr);
            } else if (first) {
                launch_pass<K, K, RB, true, ITEMS>(dev, st, sgrid, kin, vin, kout, 0u, vout, psc, shift, digit_mask, ghist, digit_totals, cv, vo, pnb, xr,
                                                   compact->chunk_hist);
            } else {
                launch_pass<K, K, RB, false, ITEMS>(dev, st, sgrid, kin, vin, kout, 0u, vout, psc, shift, digit_mask, ghist, digit_totals, cv, vo, pnb, xr);
            }
        } else {
            launch_pass<K, K, RB, false, ITEMS>(dev, st, sgrid, kin, vin, kout, 0u, vout, psc, shift, digit_mask, ghist, digit_totals, cv, vo, pnb, xr);
        }
        launches += 3;
        shift += bits;
        side ^= 1;
    }
    t_top_pass = false;
    GS_HIP(hipGetLastError());
    result_side = side;
    return GS_OK;
}

// XCD-aware tile order of a radix pass over `pnb` tiles (see run_sort_items): the span factor and the padded grid
static uint32_t xcd_span_for(uint32_t pnb, uint32_t &sgrid) {
    static const bool remap_on = !(std::getenv("GS3D_XCD_REMAP") && std::getenv("GS3D_XCD_REMAP")[0] == '0');
    static const int remap_c = std::getenv("GS3D_XCD_REMAP_C") ? std::atoi(std::getenv("GS3D_XCD_REMAP_C")) : 0;
    uint32_t xr = 0;
    if (remap_on && pnb >= 256u) {
        xr = 4u;
        while (xr < 64u && xr * 2u * 256u <= pnb) xr *= 2u;
        if (remap_c > 1) xr = (uint32_t)remap_c;
    }
    sgrid = xr ? 8u * xr * ((pnb + 8u * xr - 1u) / (8u * xr)) : pnb;
    return xr;
}

// The histogram half of that pass on its own (partitioned two-round frames: gs::k_round_threshold needs the totals of the top
// digit before any sort runs): the chunk rows of the preprocess kernel summed per tile, the rows scanned; digit_totals
// holds the 1024 totals afterwards.
template <int ITEMS>
static gs_status run_top_digit_totals(gs_renderer *r, hipStream_t st, uint32_t dense_count, const uint32_t *dense_count_dev) {
    constexpr int RB = gs::MSD_TOP_BITS;
    constexpr uint32_t R = 1u << RB, TILE = (uint32_t)(gs::SORT_THREADS * ITEMS);
    const uint32_t pnb = (uint32_t)(((uint64_t)dense_count + TILE - 1) / TILE);
    GS_TRY(dev_reserve(r->ghist, (size_t)pnb * R * 4));
    GS_TRY(dev_reserve(r->digit_totals, R * 4));
    uint32_t sgrid = 0;
    const uint32_t xr = xcd_span_for(pnb, sgrid);
    const gs::SortCount psc{dense_count, dense_count_dev};
    hipLaunchKernelGGL((gs::k_sort_hist_chunks<RB, ITEMS>), dim3(sgrid), dim3(gs::SORT_THREADS), 0, st, (const uint32_t *)r->chunk_hist.ptr, psc,
                       R - 1u, (uint32_t *)r->ghist.ptr, (const uint32_t *)r->chunk_vis.ptr, pnb, xr, R / 2u);
    launch_scan_rows<(int)TILE>(R, st, (uint32_t *)r->ghist.ptr, pnb, psc, (uint32_t *)r->digit_totals.ptr);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// The frame's depth sort, MSD-first (round 5; gs_render_kernels.h, "Bucket sort"): ONE compacting scatter pass on the TOP
// 10 bits of the depth key — its histogram summed from the rows the preprocess kernel counted — then one workgroup per
// bucket sorts the remaining low bits on its CU and writes the final order to vals[0].  4 launches instead of 9; right
// when the buckets fit a workgroup (up to ~1-2 M visible Gaussians spread in depth), which the caller decides from the
// bucket sizes the last frames reported.  `scratch_*`: two passes of the chunked fallback (oversized buckets).
template <int ITEMS>
static gs_status run_depth_msd_items(gs_renderer *r, hipStream_t st, const SortCompact &cp, uint32_t dbits, uint32_t top_range,
                                     void *scratch_keys, void *scratch_vals, uint32_t *bucket_max, uint32_t &passes_out) {
    constexpr int RB = gs::MSD_TOP_BITS, RBL = gs::RADIX_BITS_MAX;     // top digit 10 bits; two bucket passes of up to 9 below it
    constexpr uint32_t R = 1u << RB, TILE = (uint32_t)(gs::SORT_THREADS * ITEMS);
    const gs_device *dev = r->dev;
    const uint32_t low_bits = dbits - (uint32_t)RB;
    const uint32_t pnb = (uint32_t)(((uint64_t)cp.dense_count + TILE - 1) / TILE);
    GS_TRY(dev_reserve(r->ghist, (size_t)pnb * R * 4));
    GS_TRY(dev_reserve(r->digit_totals, R * 4));
    GS_TRY(dev_reserve(r->bucket_starts, R * 4));
    uint32_t sgrid = 0;
    const uint32_t xr = xcd_span_for(pnb, sgrid);
    const gs::SortCount psc{cp.dense_count, cp.dense_count_dev};
    t_bucket_starts = (uint32_t *)r->bucket_starts.ptr;
    launch_pass<uint32_t, uint32_t, RB, true, ITEMS>(dev, st, sgrid, cp.dense_keys, (const uint32_t *)nullptr, (uint32_t *)r->dkeys[1].ptr, 0u,
                                                     (uint32_t *)r->dvals[1].ptr, psc, low_bits, R - 1u, r->ghist, r->digit_totals,
                                                     cp.chunk_vis, cp.visible_out, pnb, xr, cp.chunk_hist, R / 2u);
    t_bucket_starts = nullptr;
    gs::BucketSortIO io;
    io.totals = (const uint32_t *)r->digit_totals.ptr;
    io.starts = (const uint32_t *)r->bucket_starts.ptr;
    io.nb = top_range < R ? top_range : R;          // digits past the far plane's cannot occur (their totals are zero)
    io.keys_in = r->dkeys[1].ptr;
    io.vals_in = (const uint32_t *)r->dvals[1].ptr;
    io.keys_tmp = scratch_keys;
    io.vals_tmp = (uint32_t *)scratch_vals;
    io.keys_out = nullptr;                          // nobody reads the sorted depth keys
    io.vals_out = (uint32_t *)r->dvals[0].ptr;
    io.low_bits = low_bits;
    io.bucket_max = bucket_max;
    io.ranges = nullptr;
    io.num_tiles = 0;
    io.rank_fault = t_rank_fault;
    io.watch = t_watch;
    if (t_rank_fault || dev->lds_atomic_ordered.load(std::memory_order_relaxed))
        hipLaunchKernelGGL((gs::k_bucket_sort<uint32_t, RBL, gs::BKT_THREADS, true>), dim3(io.nb), dim3(gs::BKT_THREADS), 0, st, io);
    else
        hipLaunchKernelGGL((gs::k_bucket_sort<uint32_t, RBL, gs::BKT_THREADS, false>), dim3(io.nb), dim3(gs::BKT_THREADS), 0, st, io);
    GS_HIP(hipGetLastError());
    r->launches += 4;
    passes_out = 1u + (low_bits + RBL - 1u) / RBL;
    return GS_OK;
}

// The frame's tile sort, MSD-first (u16 tile ids, more than 1024 tiles): k_pairs_emit counts the TOP 10 bits of the tile id
// while it writes the pairs, one scatter pass partitions them into buckets of 2^low_bits consecutive tiles (each in depth
// order), and k_bucket_sort finishes every bucket with one counting pass on the low bits — which also yields the tiles'
// [start, end) ranges: no second histogram / row scan / scatter and no range kernel, 4 launches instead of 7.  Result on
// side 0 (keys too: the parity tap rebuilds the 64-bit keys from them).
template <int ITEMS>
static gs_status run_tile_msd_items(gs_renderer *r, hipStream_t st, const gs::ExpandIO &eo, gs::SortCount tc, uint32_t tile_bits,
                                    uint32_t num_tiles, uint32_t *ranges, uint32_t *bucket_max, uint32_t &passes_out) {
    constexpr int RB = gs::MSD_TOP_BITS, RBL = 6;        // top digit 10 bits; up to 6 bits (65536 tiles) left for the buckets
    constexpr uint32_t R = 1u << RB, TILE = (uint32_t)(gs::SORT_THREADS * ITEMS);
    const gs_device *dev = r->dev;
    const uint32_t low_bits = tile_bits - (uint32_t)RB;
    const uint32_t pnb = (uint32_t)(((uint64_t)tc.count + TILE - 1) / TILE);
    GS_TRY(dev_reserve(r->ghist, (size_t)pnb * R * 4));
    GS_TRY(dev_reserve(r->digit_totals, R * 4));
    uint32_t sgrid = 0;
    const uint32_t xr = xcd_span_for(pnb, sgrid);
    gs::ExpandIO src = eo;
    src.tvals = (uint32_t *)r->tvals[0].ptr;
    if (src.rect32)
        hipLaunchKernelGGL((gs::k_pairs_emit<uint16_t, RB, ITEMS, true>), dim3(sgrid), dim3(gs::SORT_THREADS), 0, st, src, R - 1u,
                           (uint32_t *)r->ghist.ptr, (uint16_t *)r->tkeys[0].ptr, pnb, xr, low_bits);
    else
        hipLaunchKernelGGL((gs::k_pairs_emit<uint16_t, RB, ITEMS, false>), dim3(sgrid), dim3(gs::SORT_THREADS), 0, st, src, R - 1u,
                           (uint32_t *)r->ghist.ptr, (uint16_t *)r->tkeys[0].ptr, pnb, xr, low_bits);
    launch_scan_rows<(int)TILE>(R, st, (uint32_t *)r->ghist.ptr, pnb, tc, (uint32_t *)r->digit_totals.ptr);
    GS_TRY(dev_reserve(r->bucket_starts, R * 4));
    t_bucket_starts = (uint32_t *)r->bucket_starts.ptr;
    launch_scatter<uint16_t, uint16_t, RB, false, ITEMS>(dev, st, sgrid, (const uint16_t *)r->tkeys[0].ptr, (const uint32_t *)r->tvals[0].ptr,
                                                         (uint16_t *)r->tkeys[1].ptr, 0u, (uint32_t *)r->tvals[1].ptr, tc, low_bits, R - 1u,
                                                         (const uint32_t *)r->ghist.ptr, (const uint32_t *)r->digit_totals.ptr,
                                                         (const uint32_t *)nullptr, (uint32_t *)nullptr, pnb, xr);
    t_bucket_starts = nullptr;
    gs::BucketSortIO io;
    io.totals = (const uint32_t *)r->digit_totals.ptr;
    io.starts = (const uint32_t *)r->bucket_starts.ptr;
    io.nb = ((num_tiles - 1u) >> low_bits) + 1u;
    io.keys_in = r->tkeys[1].ptr;
    io.vals_in = (const uint32_t *)r->tvals[1].ptr;
    io.keys_tmp = nullptr;                          // one pass: the chunked path needs no scratch
    io.vals_tmp = nullptr;
    io.keys_out = r->tkeys[0].ptr;
    io.vals_out = (uint32_t *)r->tvals[0].ptr;
    io.low_bits = low_bits;
    io.bucket_max = bucket_max;
    io.ranges = ranges;
    io.num_tiles = num_tiles;
    io.rank_fault = t_rank_fault;
    io.watch = t_watch;
    if (t_rank_fault || dev->lds_atomic_ordered.load(std::memory_order_relaxed))
        hipLaunchKernelGGL((gs::k_bucket_sort<uint16_t, RBL, gs::BKT_THREADS_SMALL, true>), dim3(io.nb), dim3(gs::BKT_THREADS_SMALL), 0, st, io);
    else
        hipLaunchKernelGGL((gs::k_bucket_sort<uint16_t, RBL, gs::BKT_THREADS_SMALL, false>), dim3(io.nb), dim3(gs::BKT_THREADS_SMALL), 0, st, io);
    GS_HIP(hipGetLastError());
    r->launches += 4;
    passes_out = low_bits ? 2u : 1u;
    return GS_OK;
}

// tile size from the host-side bound of the element count (see SortCfg)
template <typename K, int RB>
static gs_status run_sort_rb(const gs_device *dev, void *const keys[2], void *const vals[2], DevArray &ghist,
                             DevArray &digit_totals, gs::SortCount sc, uint32_t end_bit, const SortCompact *compact,
                             hipStream_t st, int &result_side, uint32_t &passes_out, uint32_t &launches,
                             const gs::ExpandIO *source = nullptr) {
    const uint64_t bound = compact ? compact->dense_count : sc.count;
    if constexpr (gs::SortCfg<K>::ITEMS_LARGE != gs::SortCfg<K>::ITEMS) {
        // u32 keys: larger tiles for larger sorts.  u16 tile keys: a tile's digit runs should hold >= 32
        // elements — 4096-key tiles do for digits of up to 7 bits (1080p: 13 tile bits = 7 + 6) and fit
        // twice as many workgroups per CU (10 M: emit + tile sort 376 -> 358 us); 8-bit digits (4K: 15
        // bits = 8 + 7) keep 8192-key tiles (848 vs 875 us with the small ones).
        bool large = bound >= (4u << 20);
        if (sizeof(K) == 4) {
            // GS3D_DEPTH_SORT_LARGE=0/1 forces 4096- / 8192-key tiles for 32-bit keys (A/B runs)
            static const int force32 = std::getenv("GS3D_DEPTH_SORT_LARGE") ? std::atoi(std::getenv("GS3D_DEPTH_SORT_LARGE")) : -1;
            if (force32 >= 0) large = force32 != 0;
        }
        if (sizeof(K) == 2) {
            const uint32_t passes = (end_bit + RB - 1) / RB;
            large = passes ? (end_bit + passes - 1) / passes > 7u : false;
            // GS3D_TILE_SORT_LARGE=0/1 forces 4096- / 8192-key tiles for the 16-bit tile keys (A/B runs)
            static const int force = std::getenv("GS3D_TILE_SORT_LARGE") ? std::atoi(std::getenv("GS3D_TILE_SORT_LARGE")) : -1;
            if (force >= 0) large = force != 0;
        }
        if (large)
            return run_sort_items<K, RB, gs::SortCfg<K>::ITEMS_LARGE>(dev, keys, vals, ghist, digit_totals, sc, end_bit,
                                                                    compact, st, result_side, passes_out, launches, source);
    }
    return run_sort_items<K, RB, gs::SortCfg<K>::ITEMS>(dev, keys, vals, ghist, digit_totals, sc, end_bit, compact, st,
                                                       result_side, passes_out, launches, source);
}

// host-known count (spatial order build, stand-alone sort)
template <typename K>
static gs_status sort_pairs_device(const gs_device *dev, void *const keys[2], void *const vals[2],
                                   DevArray &ghist, DevArray &digit_totals, uint32_t count,
                                   uint32_t end_bit, hipStream_t st, int &result_side,
                                   uint32_t &passes_out) {
    uint32_t launches = 0;
    return run_sort_rb<K, gs::RADIX_BITS>(dev, keys, vals, ghist, digit_totals, gs::SortCount{count, nullptr}, end_bit,
                                          nullptr, st, result_side, passes_out, launches);
}

// (Re)build the block-planar mirror on `st`.  A whole-buffer rebuild in spatial mode first computes
// the order: bounding box of the positions -> 30-bit Morton keys -> stable radix sort -> order / inv.
static gs_status build_spatial_order(gs_gaussians_buffer *g, hipStream_t st, size_t len) {
    gs_device *dev = g->buf->dev;
    const uint32_t n = (uint32_t)len, pod_words = (uint32_t)(pod_stride(g) / 4);
    DevArray keys[2], vals[2], gh, dt, partial, bbox;
    gs_status rc = GS_OK;
    for (int i = 0; i < 2 && rc == GS_OK; i++) {
        rc = dev_reserve(keys[i], (size_t)n * 4);
        if (rc == GS_OK) rc = dev_reserve(vals[i], (size_t)n * 4);
    }
    const uint32_t pgrid = n / 256u + 1u < 1024u ? n / 256u + 1u : 1024u;
    if (rc == GS_OK) rc = dev_reserve(partial, (size_t)pgrid * 24);
    if (rc == GS_OK) rc = dev_reserve(bbox, 24);
    int side = 0;
    uint32_t passes = 0;
    if (rc == GS_OK) {
        hipLaunchKernelGGL(gs::k_bbox_partial, dim3(pgrid), dim3(256), 0, st, (const uint32_t *)g->buf->ptr,
                           pod_words, n, (float *)partial.ptr);
        hipLaunchKernelGGL(gs::k_bbox_final, dim3(1), dim3(256), 0, st, (const float *)partial.ptr, pgrid,
                           (float *)bbox.ptr);
        hipLaunchKernelGGL(gs::k_morton_keys, dim3((n + 255u) / 256u), dim3(256), 0, st,
                           (const uint32_t *)g->buf->ptr, pod_words, n, (const float *)bbox.ptr,
                           (uint32_t *)keys[0].ptr, (uint32_t *)vals[0].ptr);
        void *k2[2] = {keys[0].ptr, keys[1].ptr};
        void *v2[2] = {vals[0].ptr, vals[1].ptr};
        rc = sort_pairs_device<uint32_t>(dev, k2, v2, gh, dt, n, 30, st, side, passes);
    }
    if (rc == GS_OK) {
        // the sorted values ARE the order: keep that array as a ref-counted buffer, free the rest
        gs_buffer *ob = new gs_buffer();
        ob->dev = dev;
        ob->ptr = vals[side].ptr;
        ob->bytes = (size_t)n * 4;
        ob->owned = true;
        ob->refs.store(1);
        vals[side].ptr = nullptr;
        vals[side].bytes = 0;
        if (g->order) gs_buffer_release(g->order);
        g->order = ob;
        if (g->inv) (void)hipFree(g->inv);
        g->inv = nullptr;
        hipError_t e = hipMalloc(&g->inv, (size_t)n * 4);
        if (e != hipSuccess) rc = fail(GS_ERR_OUT_OF_MEMORY, (size_t)n * 4, 0, 0, "hipMalloc failed: %s", hipGetErrorString(e));
        else
            hipLaunchKernelGGL(gs::k_invert_order, dim3((n + 255u) / 256u), dim3(256), 0, st,
                               (const uint32_t *)ob->ptr, n, (uint32_t *)g->inv);
    }
    if (rc == GS_OK && hipGetLastError() != hipSuccess) rc = fail(GS_ERR_HIP, 0, 0, 0, "spatial order launch failed");
    // the scratch arrays may still be in use by the queued kernels: hipFree synchronises the device
    for (int i = 0; i < 2; i++) {
        dev_free(keys[i]);
        dev_free(vals[i]);
    }
    for (DevArray *a : {&gh, &dt, &partial, &bbox}) dev_free(*a);
    return rc;
}

static gs_status ensure_planar(gs_gaussians_buffer *g, hipStream_t st) {
    size_t len = gs_gaussians_buffer_len(g);
    size_t stride = (len + gs::PLANAR_BLOCK - 1) / gs::PLANAR_BLOCK * gs::PLANAR_BLOCK;   // whole blocks
    uint32_t chunks = (uint32_t)(pod_stride(g) / 16);
    if (!g->planar || g->planar_stride != stride) {
        if (g->planar) GS_HIP(hipFree(g->planar));
        g->planar = nullptr;
        if (g->block_bounds) GS_HIP(hipFree(g->block_bounds));
        g->block_bounds = nullptr;
        GS_HIP(hipMalloc(&g->planar, (stride ? stride : gs::PLANAR_BLOCK) * 16 * chunks));
        g->planar_stride = stride;
        g->mark_all();
    }
    size_t lo = g->dirty_lo, hi = g->dirty_hi < len ? g->dirty_hi : len;
    // Re-sort policy: a partial update keeps the slots of the Gaussians it rewrites, so the spatial
    // order decays under an editor's stream of update_range calls (moved Gaussians stay in the
    // blocks of their old neighbourhood: block bounds grow, culling gets weaker — results never
    // change).  Once the partial updates since the last ordering add up to a quarter of the buffer,
    // the next frame rebuilds the order (one Morton sort + ordered repack, ~2 ms per 10 M).
    if (g->spatial && g->order && lo < hi && g->partial_since_order > len / 4) {
        lo = 0;
        hi = len;
    }
    if (lo < hi) {
        const bool whole = lo == 0 && hi == len;
        if (whole) g->partial_since_order = 0;
        const bool want_order = g->spatial && len > 1;
        if (whole || want_order != (g->order != nullptr)) {
            // whole-buffer (re)mirror: this is where the spatial order is (re)computed or dropped
            if (want_order) {
                GS_TRY(build_spatial_order(g, st, len));
            } else {
                if (g->order) gs_buffer_release(g->order);
                g->order = nullptr;
                if (g->inv) GS_HIP(hipFree(g->inv));
                g->inv = nullptr;
            }
            lo = 0;
            hi = len;
        }
        uint64_t count = hi - lo;
        if (g->order && lo == 0 && hi == len) {
            uint32_t grid = (uint32_t)((count + gs::REPACK_GROUP - 1) / gs::REPACK_GROUP);
            hipLaunchKernelGGL(gs::k_repack_planar_ordered, dim3(grid), dim3(256), 0, st,
                               (const uint4 *)g->buf->ptr, (uint4 *)g->planar, (const uint32_t *)g->order->ptr,
                               count, chunks);
        } else if (g->order) {
            uint64_t total = count * chunks;
            uint32_t grid = (uint32_t)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
            hipLaunchKernelGGL(gs::k_repack_planar_scatter, dim3(grid), dim3(256), 0, st,
                               (const uint4 *)g->buf->ptr, (uint4 *)g->planar, (const uint32_t *)g->inv,
                               (uint64_t)lo, count, chunks);
        } else {
            uint32_t grid = (uint32_t)((count + gs::REPACK_GROUP - 1) / gs::REPACK_GROUP);
            hipLaunchKernelGGL(gs::k_repack_planar, dim3(grid), dim3(256), 0, st,
                               (const uint4 *)g->buf->ptr, (uint4 *)g->planar, (uint64_t)lo, count, chunks);
        }
        GS_HIP(hipGetLastError());
        // per-block bounds for the preprocess kernels' block culling (recomputed for every block:
        // 48 bytes per Gaussian, cheaper than tracking which blocks a partial update touched)
        const uint32_t nblocks = (uint32_t)((len + gs::PLANAR_BLOCK - 1) / gs::PLANAR_BLOCK);
        if (!g->block_bounds) GS_HIP(hipMalloc(&g->block_bounds, (stride / gs::PLANAR_BLOCK + 1) * 32));
        hipLaunchKernelGGL(k_tbl_block_bounds[g->sh][g->cov], dim3(nblocks), dim3(gs::PP_THREADS), 0, st,
                           (const uint4 *)g->planar, (uint32_t)len, (float *)g->block_bounds);
        GS_HIP(hipGetLastError());
        if (!g->mirror_ready) GS_HIP(hipEventCreateWithFlags(&g->mirror_ready, hipEventDisableTiming));
        GS_HIP(hipEventRecord(g->mirror_ready, st));
        g->mirror_stream = st;
    } else if (g->mirror_ready && st != g->mirror_stream) {
        // the mirror was built on another stream: order this stream behind that build
        GS_HIP(hipStreamWaitEvent(st, g->mirror_ready, 0));
    }
    g->dirty_lo = g->dirty_hi = 0;
    return GS_OK;
}

extern "C" gs_status gs_gaussians_buffer_set_spatial_order(gs_gaussians_buffer *g, int32_t enabled) {
    if (!g) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null buffer");
    if (g->spatial != (enabled != 0)) {
        g->spatial = enabled != 0;
        g->mark_all();
    }
    return GS_OK;
}

extern "C" int32_t gs_gaussians_buffer_spatial_order(const gs_gaussians_buffer *g) {
    return g && g->spatial ? 1 : 0;
}

extern "C" gs_status gs_gaussians_buffer_download_order(gs_gaussians_buffer *g, gs_stream *s,
                                                        uint32_t *order_out, size_t count) {
    if (!g || (count && !order_out)) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    size_t len = gs_gaussians_buffer_len(g);
    if (count != len)
        return fail(GS_ERR_COUNT_MISMATCH, count, len, 0, "Gaussians count mismatch: %zu != %zu", count, len);
    GS_TRY(use_device(g->buf->dev));
    hipStream_t st = stream_of(g->buf->dev, s);
    GS_TRY(ensure_planar(g, st));
    if (!g->order) {
        for (size_t i = 0; i < len; i++) order_out[i] = (uint32_t)i;
        return GS_OK;
    }
    GS_HIP(hipStreamSynchronize(st));
    hipError_t e = hipMemcpy(order_out, g->order->ptr, len * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(GS_ERR_DOWNLOAD, (uint64_t)e, 0, 0, "download failed: %s", hipGetErrorString(e));
    return GS_OK;
}

// roctx ranges around the stages of a frame (the tracing hook of this path).  The marker
// library is looked up at run time and only when GS3D_ROCTX=1, so the product has no link-time
// dependency on a profiler and pays nothing otherwise.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *e = std::getenv("GS3D_ROCTX");
        if (!e || e[0] != '1') return;
        for (const char *name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
static Roctx &roctx() {
    static Roctx r;
    return r;
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) : on(roctx().push != nullptr) {
        if (on) roctx().push(name);
    }
    void next(const char *name) {
        if (on) {
            roctx().pop();
            roctx().push(name);
        }
    }
    ~RoctxRange() {
        if (on) roctx().pop();
    }
};

static gs_status reserve_pairs(gs_renderer *r, uint64_t pairs, bool wide) {
    if (pairs <= r->pair_capacity && r->tkeys[0].ptr && wide == r->wide_tiles) return GS_OK;
    uint64_t cap = pairs;
    if (cap < r->pair_capacity) cap = r->pair_capacity;
    if (cap > 0xfffffff0ull) cap = 0xfffffff0ull;   // pair indices are 32-bit
    for (int i = 0; i < 2; i++) {
        GS_TRY(dev_reserve(r->tkeys[i], cap * (wide ? 4 : 2) + 64));
        GS_TRY(dev_reserve(r->tvals[i], cap * 4 + 64));
    }
    GS_TRY(dev_reserve(r->cursors, (size_t)(cap / gs::CURSOR_SLOTS + 2) * sizeof(gs::PairCursorRec)));
    r->pair_capacity = cap;
    r->wide_tiles = wide;
    return GS_OK;
}

// pair capacity for a frame expected to produce `d` pairs: 25 % head room for a moving camera
static uint64_t capacity_for(uint64_t d) { return d + d / 4 + 65536; }

static uint32_t float_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

// device memory that must start out as zeros (the frame state: V and D live there between kernels)
static gs_status reserve_zeroed(DevArray &a, size_t bytes, hipStream_t st) {
    if (a.bytes >= bytes && a.ptr) return GS_OK;
    GS_TRY(dev_reserve(a, bytes));
    GS_HIP(hipMemsetAsync(a.ptr, 0, a.bytes, st));
    return GS_OK;
}

extern "C" gs_status gs_render_frame(gs_renderer *r, gs_stream *s, gs_gaussians_buffer *g,
                                     const gs_gaussian_transform_pod *gt,
                                     const gs_model_transform_pod *mt, const gs_camera *cam,
                                     uint32_t band_ty0, uint32_t band_ty1, float *rgba) {
    if (!r || !s || !g || !gt || !mt || !cam || !rgba)
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    if (g->buf->dev != r->dev || s->dev != r->dev)
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "objects belong to different devices");
    if (cam->width == 0 || cam->height == 0 || cam->width > 65535u * 16u || cam->height > 65535u * 16u)
        return fail(GS_ERR_INVALID_ARGUMENT, cam->width, cam->height, 0, "bad image size");
    if ((uintptr_t)rgba & 15u)
        return fail(GS_ERR_INVALID_ARGUMENT, (uint64_t)(uintptr_t)rgba, 16, 0,
                    "the RGBA frame must be 16-byte aligned (pixels are stored as float4)");
    uint32_t mode = gt->flags[0];
    if (mode > GS_DISPLAY_POINT)
        return fail(GS_ERR_INVALID_ARGUMENT, mode, 0, 0, "unknown GaussianDisplayMode %u", mode);
    if (!(cam->near_plane >= 0.0f))
        return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "the near plane must be >= 0 (depth keys are the bits of a positive float)");
    GS_TRY(use_device(r->dev));
    hipStream_t st = s->s;
    GS_TRY(collect_timing(r));
    // Frames of one renderer share its scratch buffers and result blocks, so they must run one after
    // the other.  On one stream that is stream order; a caller that moves the renderer to ANOTHER
    // stream gets the same guarantee from the previous frame's end-of-frame event (a device-side wait,
    // the host does not block).
    // No event at the end of every frame (round 4: it cost ~4 us of a pipelined 1 M frame).  The event a stream change
    // needs is recorded on the PREVIOUS stream when the change happens (everything enqueued there is in front of it),
    // and the capacity history reads the self-validating result blocks (gen stored last, read first and last)
    // without asking an event first.  GS3D_FRAME_EVENT=1 restores the per-frame event (and the query in front of
    // every history read).
    static const bool frame_event = std::getenv("GS3D_FRAME_EVENT") && std::getenv("GS3D_FRAME_EVENT")[0] == '1';
    if (r->have_frame && (st != r->last_stream || r->last_stream_gone)) {
        // (a stream that has been destroyed since recorded the event on its way out: gs_stream_destroy; a new stream may
        // have received the old handle's value, hence the flag and not the comparison alone)
        if (!frame_event && !r->last_stream_gone) {
            GS_HIP(hipEventRecord(r->done[r->gen & 1u], r->last_stream));
            r->done_valid[r->gen & 1u] = true;
        }
        if (r->done_valid[r->gen & 1u]) GS_HIP(hipStreamWaitEvent(st, r->done[r->gen & 1u], 0));
    }
    const bool timing = r->timing && r->ev_valid;
    static const char *const k_stage_names[ST_COUNT] = {"gs3d:repack", "gs3d:preprocess", "gs3d:sizing", "gs3d:depth_sort",
                                                        "gs3d:expand", "gs3d:tile_sort", "gs3d:ranges", "gs3d:blend",
                                                        "gs3d:end"};
    RoctxRange range("gs3d:frame");
    RoctxRange stage("gs3d:setup");
    auto mark = [&](int i) {
        if (timing) (void)hipEventRecord(r->ev[i], st);
        stage.next(k_stage_names[i]);
    };

    gs::FrameConsts fc;
    make_frame_consts(gt, mt, cam, band_ty0, band_ty1, fc);
    size_t n64 = gs_gaussians_buffer_len(g);
    {
        // Tile rect version 4: pinned by the caller or the environment, otherwise on where the preprocess kernel waits for
        // HBM long enough to hide the test's ~120 instructions per Gaussian: records of 200 bytes or more (f32 SH) in a
        // scene beyond the Infinity Cache.  Same-box A/B, frame time with / without (gpurun_out/r05i/ab_masks3.txt): 10 M x
        // 224 B at 4K 1.521 / 1.558 ms, at 1080p 0.926-0.951 / 0.926-0.959 (tile sort -12 us); 50 M x 144 B 3.37-3.45 /
        // 3.31-3.37 (preprocess +65..130 us, tile sort -20..55); 1 M x 48 B 0.318 / 0.320 (preprocess +5 us).
        static const int masks_env = std::getenv("GS3D_TILE_MASKS") ? std::atoi(std::getenv("GS3D_TILE_MASKS")) : -1;
        const int pinned = r->tile_masks_req >= 0 ? r->tile_masks_req : masks_env;
        const uint64_t pod_bytes = (uint64_t)gs::pod_words(g->sh, g->cov) * 4u;
        const bool want = pinned >= 0 ? pinned != 0 : pod_bytes >= 200u && (uint64_t)n64 * pod_bytes > (512ull << 20);
        fc.tile_masks = fc.tile_masks && want ? 1u : 0u;
        r->tile_masks = fc.tile_masks != 0u;
        r->wt_pairs = fc.wt_pairs != 0u;
    }
    if (n64 > 0xfffffff0ull) return fail(GS_ERR_INVALID_ARGUMENT, n64, 0, 0, "too many Gaussians");
    uint32_t n = (uint32_t)n64;
    uint32_t nchunks = (n + gs::PP_CHUNK - 1) / gs::PP_CHUNK;
    uint32_t num_tiles = fc.tiles_x * fc.tiles_y;
    // one 256-Gaussian expansion chunk may touch at most 256 * num_tiles pairs: keep that inside 32 bits
    if ((uint64_t)fc.tiles_x * fc.tiles_y > (1ull << 22))
        return fail(GS_ERR_INVALID_ARGUMENT, cam->width, cam->height, 0, "more than 2^22 tiles");
    // tile rects are packed as 16-bit tile coordinates
    if (fc.tiles_x > 0xffffu || fc.tiles_y > 0xffffu)
        return fail(GS_ERR_INVALID_ARGUMENT, cam->width, cam->height, 0, "more than 65535 tiles along one axis");
    const bool wide = num_tiles > 65536u;
    const size_t nn = n ? n : 1, nc = nchunks ? nchunks : 1;

    // ---- what the previous frames told us (never blocks: an unfinished frame is simply not consulted) ----
    uint64_t want_capacity = r->pair_capacity;
    bool rank_fault_seen = false;
    uint64_t hist_d[2] = {0, 0};
    uint32_t hist_gen[2] = {0, 0};
    uint32_t hist_v[2] = {0, 0};
    uint32_t hist_bmax[2] = {0, 0}, hist_tmax[2] = {0, 0}, hist_tdone[2] = {0, 0}, hist_topen[2] = {0, 0}, hist_rmax[2] = {0, 0};
    for (int i = 0; i < 2; i++) {
        if (frame_event) {
            if (!r->done_valid[i] || hipEventQuery(r->done[i]) != hipSuccess) continue;
        } else if (!r->done_gen[i]) {
            continue;
        }
        const gs::FrameResult &fr = r->results[i];
        // gen is published last with a system-scope release (publish_result): read it first — and once more behind the
        // fields (a frame two generations on may be overwriting the block while it is read)
        if (__atomic_load_n(&fr.gen, __ATOMIC_ACQUIRE) != r->done_gen[i]) continue;
        const uint64_t f_pairs = fr.pairs_total;
        const uint32_t f_vis = fr.visible;
        const uint32_t f_flags = fr.flags;
        const uint32_t f_bmax = fr.depth_bucket_max, f_tmax = fr.tile_bucket_max, f_tdone = fr.tiles_done, f_topen = fr.tiles_open, f_rmax = fr.round_pairs_max;
        if (__atomic_load_n(&fr.gen, __ATOMIC_ACQUIRE) != r->done_gen[i] || f_pairs > 0xfffffff0ull) continue;
        hist_d[i] = f_pairs;
        hist_gen[i] = r->done_gen[i];
        hist_v[i] = f_vis;
        hist_bmax[i] = f_bmax;
        hist_tmax[i] = f_tmax;
        hist_tdone[i] = f_tdone;
        hist_topen[i] = f_topen;
        hist_rmax[i] = f_rmax;
        if (f_flags & gs::FRAME_FLAG_RANK_FAULT) rank_fault_seen = true;
        // grow when the last measured D leaves less than 1/8 of head room
        if (f_pairs + f_pairs / 8 > r->pair_capacity && capacity_for(f_pairs) > want_capacity)
            want_capacity = capacity_for(f_pairs);
    }
    // A camera that keeps closing in: D grows frame over frame, and this frame is two or three frames
    // ahead of the newest result (frames are pipelined).  Extrapolate the last step three frames ahead
    // and size for that, so that a steady zoom does not run into the skip path.
    // Only a TREND is extrapolated: both results must come from the current shape epoch (same N, image
    // size and band — a switch from a band to the full frame, or a resize, is a discontinuity, not a zoom),
    // and the extrapolation is capped at twice the newest D: one jump of the camera must not turn into
    // pair buffers of 4 x D that never shrink.
    if (hist_gen[0] && hist_gen[1] && (hist_gen[0] + 1u == hist_gen[1] || hist_gen[1] + 1u == hist_gen[0]) &&
        r->done_shape[0] == r->shape_epoch && r->done_shape[1] == r->shape_epoch) {
        const int newer = hist_gen[0] > hist_gen[1] ? 0 : 1;
        const uint64_t d_new = hist_d[newer], d_old = hist_d[newer ^ 1];
        if (d_new > d_old) {
            uint64_t ahead = d_new + 3u * (d_new - d_old);
            if (ahead > 2u * d_new) ahead = 2u * d_new;
            if (ahead + ahead / 8 > r->pair_capacity && capacity_for(ahead) > want_capacity) want_capacity = capacity_for(ahead);
        }
    }
    (void)hipGetLastError();   // hipEventQuery reports hipErrorNotReady through the sticky error too
    FrameShape shape;
    shape.n = n;
    shape.width = cam->width;
    shape.height = cam->height;
    shape.band0 = fc.band_ty0;
    shape.band1 = fc.band_ty1;
    const bool sizing = n != 0 && (r->pair_capacity == 0 || !(shape == r->shape));

    mark(ST_REPACK);
    GS_TRY(ensure_planar(g, st));
    if (r->last_order != g->order) {
        if (r->last_order) gs_buffer_release(r->last_order);
        r->last_order = g->order ? gs_buffer_retain(g->order) : nullptr;
    }
    mark(ST_PRE);

    // per-slot outputs of preprocess: whole chunks — a list frame addresses them in list space, where the
    // buffer's last, partial block may sit anywhere and its lanes past N are written too (as "culled")
    const size_t nslots = nc * (size_t)gs::PP_CHUNK;
    GS_TRY(dev_reserve(r->recs, nslots * 4 * gs::REC_WORDS + 16));
    GS_TRY(dev_reserve(r->depth, nslots * 4));
    GS_TRY(dev_reserve(r->rect, nslots * 8));
    GS_TRY(dev_reserve(r->sorted_rect, (nn + 1024) * 8));   // padded: k_pairs_emit reads whole batches
    GS_TRY(dev_reserve(r->chunk_tiles, nc * 4));
    GS_TRY(dev_reserve(r->chunk_vis, nc * 4));
    GS_TRY(dev_reserve(r->chunk_hist, nc * (size_t)(gs::PRE_HIST_BINS / 2) * 4));
    GS_TRY(dev_reserve(r->scan_tmp, nc * 4));
    for (int i = 0; i < 2; i++) {
        GS_TRY(dev_reserve(r->dkeys[i], nn * 4));
        GS_TRY(dev_reserve(r->dvals[i], (nn + 1024) * 4));
    }
    {
        // The sorts' histogram rows and digit totals at the largest size any depth pass of this scene may ask for (the top-digit
        // histogram of an MSD-first sort or of a partitioned two-round frame: 1024 rows): reserved HERE, in the frame that sizes
        // the scene, because growing them later means a hipFree under frames in flight — a device-wide wait at best (and under
        // rocprofv3 --pmc the 10 M bench hung in it: the first partitioned frame doubled `ghist` behind eight queued frames).
        const size_t depth_tile = (size_t)gs::SORT_THREADS * (n >= (4u << 20) ? gs::SortCfg<uint32_t>::ITEMS_LARGE : gs::SortCfg<uint32_t>::ITEMS);
        GS_TRY(dev_reserve(r->ghist, ((size_t)nn + depth_tile - 1) / depth_tile * ((size_t)4 << gs::MSD_TOP_BITS)));
        GS_TRY(dev_reserve(r->digit_totals, (size_t)4 << gs::MSD_TOP_BITS));
    }
    const uint32_t exp_grid = (n + gs::EXP_CHUNK - 1) / gs::EXP_CHUNK;   // V <= N
    GS_TRY(dev_reserve(r->exp_sums, (size_t)(exp_grid ? exp_grid : 1) * 4));
    GS_TRY(reserve_zeroed(r->state, sizeof(gs::FrameState), st));
    {
        // Watchdog of the LDS-atomic rank (scatter_ranked): a completed frame reported a rank that was not the
        // ballot-based one -> this device sorts with the ballot-based rank from now on, and the flag is cleared in
        // stream order (the kernels that could set it are no longer launched).
        gs::FrameState *fs = (gs::FrameState *)r->state.ptr;
        if (rank_fault_seen) {
            // (the word is this renderer's own and is cleared whoever switched the device: see gs_renderer_wait_frame)
            r->dev->lds_atomic_ordered.store(false);
            GS_HIP(hipMemsetAsync(&fs->rank_fault, 0, sizeof(uint32_t), st));
        }
        // GS3D_TEST_RANK_FAULT=1 (tests): the watchdog's expectation is off by one, so it fires in the first frame
        static const bool inject = std::getenv("GS3D_TEST_RANK_FAULT") && std::getenv("GS3D_TEST_RANK_FAULT")[0] == '1';
        if (inject && !r->rank_inject_set) {
            const uint32_t one = 1u;
            GS_HIP(hipMemcpyAsync(&fs->rank_inject, &one, sizeof(one), hipMemcpyHostToDevice, st));
            GS_HIP(hipStreamSynchronize(st));      // `one` lives on this stack frame
            r->rank_inject_set = true;
        }
        // read ONCE per frame: every scatter of the frame ranks the same way, whatever another renderer's thread does
        t_rank_fault = r->dev->lds_atomic_ordered.load() ? &fs->rank_fault : nullptr;
        t_bucket_max = &fs->depth_bucket_max;
        t_top_pass = false;
    }
    struct RankFaultScope {
        ~RankFaultScope() {
            t_rank_fault = nullptr;
            t_bucket_max = nullptr;
            t_top_pass = false;
        }
    } rank_fault_scope;

    r->gen++;
    const uint32_t gen = r->gen;
    {
        // the rank watchdog's sample of this frame (GS3D_TEST_RANK_WATCH=<n> pins it: tests)
        static const char *watch_env = std::getenv("GS3D_TEST_RANK_WATCH");
        t_watch = watch_env ? (uint32_t)std::strtoul(watch_env, nullptr, 10) : gen;
    }
    // From here on kernels of this frame may be in the stream.  Whatever way the function is left — also
    // through GS_TRY / GS_HIP after an allocation or launch failure — the end-of-frame event of this
    // generation is recorded behind them, so the next frame on ANOTHER stream waits for exactly these
    // kernels before it touches the shared scratch and state buffers (a frame that failed half-way used
    // to leave done[gen & 1] pointing at frame gen - 2).  Its result block carries no `gen`, so the
    // capacity history skips it.
    struct DoneGuard {
        gs_renderer *r;
        hipStream_t st;
        uint32_t gen;
        bool record;
        ~DoneGuard() {
            if (record) {
                (void)hipEventRecord(r->done[gen & 1u], st);
                r->done_valid[gen & 1u] = true;
            } else {
                r->done_valid[gen & 1u] = false;      // recorded when (if) the renderer moves to another stream
            }
            r->done_gen[gen & 1u] = gen;
            r->done_shape[gen & 1u] = r->shape_epoch;
            r->done_rounds[gen & 1u] = r->two_round ? 2 : 1;
            r->done_round_k[gen & 1u] = r->two_round ? r->round1 : 0u;
        }
    } done_guard{r, st, gen, frame_event};
    gs::FrameState *state = (gs::FrameState *)r->state.ptr;
    gs::FrameResult *result = &r->results[gen & 1u];
    r->n = n;
    r->tiles_x = fc.tiles_x;
    r->tiles_y = fc.tiles_y;
    r->rect32 = fc.rect32 != 0u;
    r->last_stream = st;
    r->last_stream_gone = false;
    r->have_frame = true;
    r->launches = 0;
    r->list_mode = false;

    // depth keys = bits of the (positive) view depth minus the bits of the near plane: every visible
    // depth lies in (near, far), so only bit_length(bits(far) - bits(near)) bits need sorting — a
    // bound the host knows without looking at the scene
    const uint32_t near_bits = float_bits(cam->near_plane > 0.0f ? cam->near_plane : 0.0f);
    const uint32_t far_bits = cam->far_plane > 0.0f ? float_bits(cam->far_plane) : 0u;
    const uint32_t dbits = far_bits > near_bits ? bit_length(far_bits - near_bits) : 0u;
    r->key_bias = near_bits;
    const uint32_t tile_bits = bit_length(num_tiles ? num_tiles - 1 : 0);

    // ---- which depth sort (gs_renderer::depth_msd) ----
    // MSD-first needs a top digit of 10 bits and at most two bucket passes (of 9) below it: 11..28 key bits (the bench's planes,
    // 0.1 / 100, give 27); with fewer or more bits the LSD passes stand.
    const uint32_t msd_low_bits = dbits > (uint32_t)gs::MSD_TOP_BITS ? dbits - (uint32_t)gs::MSD_TOP_BITS : 0u;
    const bool msd_possible = n != 0 && dbits > (uint32_t)gs::MSD_TOP_BITS && msd_low_bits <= 2u * (uint32_t)gs::RADIX_BITS_MAX;
    bool depth_msd = false;
    if (msd_possible) {
        static const int msd_env = std::getenv("GS3D_DEPTH_MSD") ? std::atoi(std::getenv("GS3D_DEPTH_MSD")) : -1;
        const int newer = hist_gen[0] > hist_gen[1] ? 0 : 1;
        const int pinned = r->depth_msd_req >= 0 ? r->depth_msd_req : msd_env;
        if (pinned >= 0) {
            depth_msd = pinned != 0;
        } else if (!sizing && hist_gen[newer] && r->done_shape[newer] == r->shape_epoch && hist_bmax[newer]) {
            // hysteresis: leave MSD-first when a bucket no longer fits the register path, come back below 7/8 of it
            const uint32_t b = hist_bmax[newer];
            r->depth_bucket_seen = b;
            depth_msd = r->depth_msd ? b <= gs::BKT_CAP : b <= gs::BKT_CAP - gs::BKT_CAP / 8u;
        } else if (sizing) {
            // no report yet: the buckets of a scene this small probably fit (and if not, the bucket kernel's chunked path
            // still sorts them correctly, and the report of this very frame corrects the choice)
            depth_msd = n <= (4u << 20);
        } else {
            depth_msd = r->depth_msd;      // frames in flight between the sizing frame and its report: keep the guess
        }
    }
    r->depth_msd = depth_msd;

    gs::TileKeys tile_keys{nullptr, nullptr, 0u, 0u, 0u, nullptr, nullptr};   // null keys: the blend reads its ranges from the range array
    // GS3D_BLEND_GROUPS = 1 (half-tile lists), 2 (8x8 blocks) or 4 (8x4 blocks, default)
    static const int groups = std::getenv("GS3D_BLEND_GROUPS") ? std::atoi(std::getenv("GS3D_BLEND_GROUPS")) : 4;
    uint32_t band_tiles = (fc.band_ty1 - fc.band_ty0) * fc.tiles_x;
    // The blend launch of one round (0: the frame's only one).
    auto launch_blend = [&](uint32_t round) -> gs_status {
        if (!band_tiles) return GS_OK;
        typedef void (*blend_fn)(uint32_t *, const uint32_t *, const uint32_t *, gs::FrameConsts, float4 *,
                                 const gs::FrameState *, gs::TileKeys);
        static const blend_fn tbl[3][3] = {
            {gs::k_blend<0>, gs::k_blend_grouped<0, 2>, gs::k_blend_grouped<0, 4>},
            {gs::k_blend<1>, gs::k_blend_grouped<1, 2>, gs::k_blend_grouped<1, 4>},
            {gs::k_blend<2>, gs::k_blend_grouped<2, 2>, gs::k_blend_grouped<2, 4>}};
        static const blend_fn tbl_rounds[3][2] = {{gs::k_blend_grouped<0, 2, true>, gs::k_blend_grouped<0, 4, true>},
                                                  {gs::k_blend_grouped<1, 2, true>, gs::k_blend_grouped<1, 4, true>},
                                                  {gs::k_blend_grouped<2, 2, true>, gs::k_blend_grouped<2, 4, true>}};
        if (round != 0u && groups == 1) return fail(GS_ERR_INVALID_ARGUMENT, round, 0, 0, "two-round frames need the grouped blend");
        const blend_fn blend = round != 0u ? tbl_rounds[mode][groups == 2 ? 0 : 1] : tbl[mode][groups == 1 ? 0 : groups == 2 ? 1 : 2];
        tile_keys.round = round;
        hipLaunchKernelGGL(blend, dim3(band_tiles), dim3(gs::BLEND_THREADS), 0, st,
                           (uint32_t *)r->zero_region.ptr, (const uint32_t *)r->tvals[r->tsorted_side].ptr,
                           (const uint32_t *)r->recs.ptr, fc, (float4 *)rgba, (const gs::FrameState *)r->state.ptr, tile_keys);
        GS_HIP(hipGetLastError());
        r->launches++;
        return GS_OK;
    };
    r->two_round = false;
    // ---- one round, or two (DESIGN.md §4.2 "rounds"): decided before anything is launched, because a partitioned frame
    //      changes what the preprocess kernel counts and what the depth sorts see ----
    // A deep scene finishes most of its tiles on the nearest fraction of its Gaussians; everything behind them is
    // emitted, sorted and staged for nothing.  Two rounds: the frame of the nearest K visible Gaussians first, whose
    // blend leaves a bit per finished tile and the pixel state of the others; then the rest, without the Gaussians whose
    // (small) rect lies in finished tiles, resumed by the same blend.  The image is the single round's, bit for bit: a
    // tile's list is the concatenation of its two lists, and a dropped Gaussian touches finished pixels only.
    static const int rounds_env = std::getenv("GS3D_ROUNDS") ? std::atoi(std::getenv("GS3D_ROUNDS")) : -1;
    static const long round1_env = std::getenv("GS3D_ROUND1") ? std::atol(std::getenv("GS3D_ROUND1")) : 0;
    uint32_t round_k = 0;
    bool two_round = false;
    if (r->rounds_epoch != r->shape_epoch) {      // a new shape: the feedback starts over
        r->rounds_epoch = r->shape_epoch;
        r->round_scale = 1.0f;
        r->rounds_off = false;
        r->auto_deep = false;
        r->auto_k = 0;
    }
    if (band_tiles && groups != 1 && n > 4096u) {
        const int newer = hist_gen[0] > hist_gen[1] ? 0 : 1;
        const bool have = !sizing && hist_gen[newer] && r->done_shape[newer] == r->shape_epoch;
        const uint32_t v_est = have ? hist_v[newer] : n;
        if (have && r->done_rounds[newer] == 1) {
            r->full_pairs = hist_d[newer];
            r->full_pairs_v = hist_v[newer];
        }
        // pairs a single round would emit now: the measured count, scaled with the visible Gaussians since
        const double d_full = r->full_pairs_v ? (double)r->full_pairs * (double)v_est / (double)r->full_pairs_v : (double)r->full_pairs;
        // The renderer's own choice.  Measured (same-box A/B, gpurun_out/r05r): 10 M at 1080p (2 970 pairs per tile) -9 %,
        // at 4K (1 716) -8 %, 50 M (14 800) -24 %; the 1 M scene (296 pairs per tile) finishes its tiles only at the end of
        // their lists.  Round 1 is given ~250 pairs per tile: the bench scenes finish EVERY tile from ~170 on (k_round2_gate
        // then skips round 2), and a shorter round 1 is a shorter tile sort (same-box sweep, gpurun_out/r05x/ab_k.txt: 10 M
        // 0.813 / 0.801 / 0.790 / 0.786 ms at 400 / 270 / 210 / 170 pairs per tile, 50 M 2.07 / 2.06 / 2.02 / 2.02) — and a frame
        // takes two rounds when that is at most a third of its Gaussians and the pairs to save outweigh the launches of a
        // second round.  The feedback below lengthens a round 1 that turns out too short.
        const double per_tile = d_full / (double)band_tiles;
        double k_auto = per_tile > 0.0 ? (double)v_est * 250.0 / per_tile * (double)r->round_scale : 0.0;
        if (have && r->done_rounds[newer] == 2 && hist_gen[newer] != r->rounds_fb_gen) {
            // feedback: a round 1 that finishes less than 60 % of the tiles it has pairs for was too short (or the scene
            // does not occlude)
            r->rounds_fb_gen = hist_gen[newer];
            if ((uint64_t)hist_tdone[newer] * 10u < ((uint64_t)hist_tdone[newer] + hist_topen[newer]) * 6u) {
                r->round_scale *= 1.5f;
                if (r->round_scale > 3.4f) {
                    r->rounds_off = true;
                    r->rounds_off_gen = r->gen;
                }
            } else if (hist_topen[newer] != 0u && (uint64_t)hist_topen[newer] * 10u <= (uint64_t)hist_tdone[newer] + hist_topen[newer] &&
                       r->round_scale < 2.7f) {
                // nearly there (at most a tenth of the tiles with pairs left open): a little longer and round 2 is skipped
                r->round_scale *= 1.25f;
            }
        }
        if (r->rounds_off && r->gen - r->rounds_off_gen > 512u) {
            // ... but not for ever: the camera may have moved into a view that does occlude; another try every 512 frames
            r->rounds_off = false;
            r->round_scale = 1.0f;
        }
        bool deep = have && !r->rounds_off && d_full >= 12.0e6 && per_tile >= 1200.0 && k_auto * 3.0 <= (double)v_est;
        if (!have && !sizing && r->rounds_epoch == r->shape_epoch) {
            // frames in flight: no finished report to consult (both result blocks belong to frames still running):
            // what the last frame with a report decided stands
            deep = r->auto_deep && !r->rounds_off;
            k_auto = (double)r->auto_k;
        }
        r->auto_deep = deep;
        r->auto_k = (uint64_t)k_auto;
        const int pinned = r->rounds_req >= 0 ? r->rounds_req : rounds_env;
        two_round = pinned >= 0 ? pinned != 0 : deep;
        const uint64_t k = r->round1_req ? r->round1_req : round1_env > 0 ? (uint64_t)round1_env : pinned > 0 && !deep ? v_est / 4u : (uint64_t)k_auto;
        round_k = (uint32_t)((k + 2047u) / 2048u * 2048u < n ? (k + 2047u) / 2048u * 2048u : 0u);
        if (round_k == 0u) two_round = false;
    }
    // A two-round frame is PARTITIONED when the depth keys have a top digit to cut at (and the frame is not the one that
    // sizes the pair buffers): the preprocess kernel counts the top 10 bits of every key, k_round_threshold picks the digit
    // boundary with at least round_k Gaussians in front of it, and each round's depth sort — LSD passes whose compacting
    // first pass takes only its side of the boundary (gs::CompactPred) — sorts what that round renders: the nearest ones,
    // then what k_round2_slot_bits keeps of the rest.  Otherwise round 2 is compacted out of the full depth order
    // (k_round2_count / _write).
    // Measured, same-box A/B.  With a round 2 that runs (gpurun_out/r05t/ab3.txt): 50 M 2.42 against 2.53 ms (depth-sort stage
    // 0.318 against 0.462: a threshold + two sorts whose first pass streams the 200 MB of dense keys for 1-2 M survivors,
    // against one full sort + the compaction), 10 M 0.908 against 0.865 (two first passes of ~50 us each at their launch-bound
    // floors cost more than the full sort of 7 M keys saves).  With a round 2 that k_round2_gate skips — round 1 finished every
    // tile — only round 1's sort remains (gpurun_out/r05x/ab_part.txt): 10 M 0.806 against 0.817, 4K 1.29 against 1.32, 50 M 2.00
    // against 2.15.  So: from 32 M Gaussians, or when the newest two-round report of this shape says that round 1 finished
    // every tile (kept while no report is available); GS3D_ROUND_PARTITION=0/1 forces.
    static const int partition_env = std::getenv("GS3D_ROUND_PARTITION") ? std::atoi(std::getenv("GS3D_ROUND_PARTITION")) : -1;
    bool partition_auto = n >= (32u << 20);
    {
        const int newer = hist_gen[0] > hist_gen[1] ? 0 : 1;
        if (!sizing && hist_gen[newer] && r->done_shape[newer] == r->shape_epoch) {
            if (r->done_rounds[newer] == 2) r->auto_all_done = hist_tdone[newer] == band_tiles && hist_topen[newer] == 0u;
        } else if (sizing) {
            r->auto_all_done = false;
        }
        partition_auto = partition_auto || r->auto_all_done;
    }
    const bool partition = two_round && !sizing && dbits > (uint32_t)gs::MSD_TOP_BITS && n < (1u << 30) &&
                           (partition_env >= 0 ? partition_env != 0 : partition_auto);
    r->partitioned = partition;
    if (partition) {
        depth_msd = false;
        r->depth_msd = false;
    }
    if (n == 0) {
        // nothing to project: clear the ranges, blend the background
        GS_TRY(dev_reserve(r->zero_region, (size_t)num_tiles * 8));
        GS_HIP(hipMemsetAsync(r->zero_region.ptr, 0, (size_t)num_tiles * 8, st));
        GS_TRY(reserve_pairs(r, 1, wide));
        hipLaunchKernelGGL(gs::k_publish_result, dim3(1), dim3(64), 0, st, result, state, gen, r->flags_target);
        GS_HIP(hipGetLastError());
        r->launches++;
        mark(ST_SCAN); mark(ST_DSORT); mark(ST_EXPAND); mark(ST_TSORT); mark(ST_RANGES);
        r->sort_passes = 0;
        r->dsorted_side = r->tsorted_side = 0;
    } else {
        // the clear job of this frame, spread over the preprocess grid: tile ranges + the
        // expansion's super-chunk sums
        const size_t ranges_words = (size_t)num_tiles * 2;                                   // even: keeps the u64 sums aligned
        const size_t esb_words = 2 * ((size_t)exp_grid / gs::EXP_SB + 1);                    // u64 super-chunk sums of the expansion
        const size_t done_words = ((size_t)num_tiles + 31) / 32 + 1;                        // two-round frames: finished tiles
        // (a two-round frame needs the sums of both rounds and the bits; cleared in every frame: ~1 KB)
        const size_t zero_words = ranges_words + 2 * esb_words + 2 * done_words;
        GS_TRY(dev_reserve(r->zero_region, zero_words * 4));
        uint32_t *zero = (uint32_t *)r->zero_region.ptr;
        uint32_t *esb = zero + ranges_words;
        uint32_t *esb2 = esb + esb_words;
        uint32_t *done_bits = esb2 + esb_words;
        uint32_t *open_bits = done_bits + done_words;

        if (!sizing && want_capacity > r->pair_capacity) GS_TRY(reserve_pairs(r, want_capacity, wide));
        if (!sizing) GS_TRY(reserve_pairs(r, r->pair_capacity, wide));   // key width may have changed

        // Records with SH take the two-phase kernel (geometry chunks first, SH chunks only for the
        // lanes that survive culling): with the mirror in spatial order whole 128-byte lines of
        // culled Gaussians are never fetched; with a random order it costs the same as the
        // single-phase kernel (measured).  GS3D_FORCE_BANDED=0/1 overrides for experiments.
        static const int force_banded = std::getenv("GS3D_FORCE_BANDED") ? std::atoi(std::getenv("GS3D_FORCE_BANDED")) : -1;
        const bool banded = g->sh != GS_SH_NONE && (force_banded >= 0 ? force_banded != 0 : true);
        static const int mask_env = std::getenv("GS3D_MASK_REC") ? std::atoi(std::getenv("GS3D_MASK_REC")) : -1;
        fc.mask_culled_records = mask_env >= 0 ? (uint32_t)mask_env : (g->order != nullptr ? 1u : 0u);
        // Non-temporal loads of the mirror once it no longer fits the 256 MiB Infinity Cache: nothing of
        // it survives until the next frame anyway (10 M x 224 B: preprocess 0.440 -> 0.421 ms); a mirror that
        // does fit is re-read from the caches frame after frame and loses that with nt (1 M x 48 B:
        // 22 -> 27 us).  GS3D_NT_LOADS=0/1 forces.
        static const int nt_env = std::getenv("GS3D_NT_LOADS") ? std::atoi(std::getenv("GS3D_NT_LOADS")) : -1;
        fc.nt_loads = nt_env >= 0 ? (uint32_t)nt_env : ((uint64_t)n * gs::pod_words(g->sh, g->cov) * 4u > (512ull << 20) ? 1u : 0u);
        static const bool block_cull_off = std::getenv("GS3D_BLOCK_CULL") && std::getenv("GS3D_BLOCK_CULL")[0] == '0';
        // SH-less records are 48 bytes: the block test (one more dependent load per workgroup)
        // costs more than skipping them saves (measured at 1 M: +4 us on a 20 us kernel)
        if (block_cull_off || !g->block_bounds || !banded) fc.cull_gain = 0.0f;

        gs::PreOut po;
        po.recs = (uint32_t *)r->recs.ptr;
        po.rect = (uint2 *)r->rect.ptr;
        po.depth = (uint32_t *)r->depth.ptr;
        po.chunk_tiles = (uint32_t *)r->chunk_tiles.ptr;
        po.chunk_vis = (uint32_t *)r->chunk_vis.ptr;
        po.zero_ptr = zero;
        po.zero_words = (uint32_t)zero_words;
        po.key_bias = near_bits;
        po.block_bounds = (const float *)g->block_bounds;
        po.chunk_hist = (uint32_t *)r->chunk_hist.ptr;
        const bool top_hist = depth_msd || partition;      // the chunk rows count the TOP 10 bits (else the LSD sort's first digit)
        po.digit_mask = top_hist ? (1u << gs::MSD_TOP_BITS) - 1u : first_digit_mask(dbits, depth_radix_bits(dbits));
        po.digit_shift = top_hist ? msd_low_bits : 0u;
        po.hist_words = top_hist ? (1u << gs::MSD_TOP_BITS) / 2u : (uint32_t)gs::PP_THREADS;
        po.block_list = nullptr;
        po.block_count = nullptr;
        // Block list (k_block_cull): one thread per block tests it, the survivors are handed to the first
        // workgroups of the preprocess grid.  GS3D_BLOCK_LIST=0 keeps the test inside the preprocess kernel.
        // It pays when most blocks are culled (a rank's band of 8 at 50 M: preprocess 0.57 -> 0.41 ms) and
        // costs its launch when few are (whole 1080p frame at 10 M: +4 us), so it is taken when the newest
        // finished frame of this shape saw less than half of the Gaussians, or — no such frame yet — when
        // the frame is a band.  GS3D_BLOCK_LIST=0/1 forces.
        static const int block_list_env = std::getenv("GS3D_BLOCK_LIST") ? std::atoi(std::getenv("GS3D_BLOCK_LIST")) : -1;
        bool use_list = fc.band_ty1 - fc.band_ty0 < fc.tiles_y;
        if (!sizing && (hist_gen[0] || hist_gen[1])) use_list = hist_v[hist_gen[0] > hist_gen[1] ? 0 : 1] < n / 2u;
        if (block_list_env >= 0) use_list = block_list_env != 0;
        // the list frame keeps its outputs in list space: list_slots = blocks * 1024 must fit 32 bits, and the
        // look-back of k_block_cull is written for at most 2^20 groups of 256 blocks
        if (n > 0xfffff000u) use_list = false;
        r->list_mode = false;
        if (fc.cull_gain > 0.0f && use_list) {
            const uint32_t groups = (nchunks + 255u) / 256u;
            GS_TRY(dev_reserve(r->block_list, (size_t)nchunks * 4));
            GS_TRY(dev_reserve(r->cull_status, (size_t)groups * 4));
            // Tag of this frame's status words: 22 bits of the generation.  A word must never hold this tag
            // before its group publishes: consecutive list frames over the same groups overwrite every word
            // with the previous tag; in every other case (first use, a gap of frames without the list, another
            // group count) the words are cleared first, and the tag 0 is skipped.
            const uint32_t tag = (gen & 0x3fffffu) ? (gen & 0x3fffffu) : 0x3fffffu;
            if (r->cull_last_gen + 1u != gen || r->cull_last_groups != groups || (gen & 0x3fffffu) <= 1u)
                GS_HIP(hipMemsetAsync(r->cull_status.ptr, 0, (size_t)groups * 4, st));
            r->cull_last_gen = gen;
            r->cull_last_groups = groups;
            hipLaunchKernelGGL(gs::k_block_cull, dim3(groups), dim3(256), 0, st, (const float *)g->block_bounds, nchunks, fc,
                               (uint32_t *)r->block_list.ptr, state, (uint32_t *)r->cull_status.ptr, tag, groups);
            GS_HIP(hipGetLastError());
            r->launches++;
            po.block_list = (const uint32_t *)r->block_list.ptr;
            po.block_count = &state->list_blocks;
            r->list_mode = true;
        }
        // GS3D_PRE_PIPELINE=0: the two-phase kernel without the prefetch of the next Gaussian's geometry chunks
        static const bool pre_serial = std::getenv("GS3D_PRE_PIPELINE") && std::getenv("GS3D_PRE_PIPELINE")[0] == '0';
        const int nt = fc.nt_loads ? 1 : 0;
        hipLaunchKernelGGL((banded ? k_tbl_preprocess_banded[nt][pre_serial ? 0 : 1] : k_tbl_preprocess[nt])[g->sh][g->cov], dim3(nchunks),
                           dim3(gs::PP_THREADS), 0, st, (const uint4 *)g->planar, n, fc, po);
        GS_HIP(hipGetLastError());
        r->launches++;
        mark(ST_SCAN);

        if (sizing) {
            // First frame of this shape: measure D before sizing the pair buffers (the only blocking
            // step; steady-state frames take the capacity from the history instead)
            gs::ScanJob jt{(const uint32_t *)r->chunk_tiles.ptr, (uint32_t *)r->scan_tmp.ptr, r->host_counters, nchunks,
                           r->list_mode ? &state->list_blocks : nullptr};
            hipLaunchKernelGGL(gs::k_scan_chunks, dim3(1), dim3(1024), 0, st, jt, jt);
            GS_HIP(hipGetLastError());
            GS_HIP(hipStreamSynchronize(st));
            const uint32_t d = r->host_counters[0];
            if (d == 0xffffffffu) {
                r->shape = FrameShape();
                return fail(GS_ERR_PAIR_OVERFLOW, n, 0, 0,
                            "the frame needs more than 2^32 (tile, Gaussian) pairs; pair indices are 32-bit");
            }
            r->full_pairs = d;              // (the visible count it belongs to comes with the frame's report)
            r->full_pairs_v = 0;
            const uint64_t cap = capacity_for(d) > want_capacity ? capacity_for(d) : want_capacity;
            GS_TRY(reserve_pairs(r, cap, wide));
        }
        if (!(r->shape == shape)) r->shape_epoch++;
        r->shape = shape;
        // A two-round frame sizes its grids — and bounds each round — by what a ROUND emitted last time, not by the single-round
        // pair count that sized the buffers (50 M: 3 M pairs per round in buffers for 150 M: the emission's and the tile sort's
        // 37 000 mostly empty workgroups cost 20-40 us per kernel): twice the larger round of the newest two-round report, with
        // the usual head room.  A round that still outgrows it skips the frame like any pair overflow (the next one has the
        // report); the first two-round frame of a shape, and any frame without a report, use the buffers' capacity.
        uint32_t capacity = (uint32_t)r->pair_capacity;
        if (two_round) {
            const int newer = hist_gen[0] > hist_gen[1] ? 0 : 1;
            uint64_t want = 0;
            // (only a report of a frame whose round 1 was as long as this one's says anything about this frame's rounds)
            if (!sizing && hist_gen[newer] && r->done_shape[newer] == r->shape_epoch && r->done_rounds[newer] == 2 && hist_rmax[newer] &&
                r->done_round_k[newer] == round_k)
                want = capacity_for(2ull * hist_rmax[newer]);
            else if (!sizing && !hist_gen[newer] && r->round_cap && r->round_cap_k == round_k)
                want = r->round_cap;          // frames in flight: the last bound stands
            if (want && want < capacity) capacity = (uint32_t)want;
            r->round_cap = want;
            r->round_cap_k = round_k;
        }
        mark(ST_DSORT);

        // ---- depth sort of the visible Gaussians; its first pass reads the dense per-slot keys and
        //      compacts (count V stays on the device, grids from N) ----
        int dside = 0;
        uint32_t dpasses = 0;
        // pred: which Gaussians the compacting first pass takes (all visible ones; or one side of a partitioned frame's
        // threshold); their count goes to *visible_out
        auto depth_sort = [&](const gs::CompactPred &pred, uint32_t *visible_out, const uint32_t *dense_dev = nullptr) -> gs_status {
            void *k2[2] = {r->dkeys[0].ptr, r->dkeys[1].ptr};
            void *v2[2] = {r->dvals[0].ptr, r->dvals[1].ptr};
            SortCompact cp;
            cp.dense_keys = (const uint32_t *)r->depth.ptr;
            cp.chunk_vis = (const uint32_t *)r->chunk_vis.ptr;
            cp.visible_out = visible_out;
            cp.dense_count = n;
            // list frame: only the surviving blocks' slots; round 2 of a partitioned frame: none when the gate found nothing left
            cp.dense_count_dev = dense_dev ? dense_dev : r->list_mode ? &state->list_slots : nullptr;
            // (a partitioned frame's chunk rows count the top digit of ALL visible keys: its passes count their own)
            cp.chunk_hist = pred.tau_dev ? nullptr : (const uint32_t *)r->chunk_hist.ptr;
            const gs::SortCount dc{n, visible_out};
            t_compact_pred = pred;
            struct PredScope {
                ~PredScope() { t_compact_pred = gs::CompactPred(); }
            } pred_scope;
            uint32_t passes = 0;
            if (depth_msd) {
                // scratch of the bucket kernel's chunked path: the LSD sort's side 0 keys and the (not yet written)
                // depth-ordered rects — the dense keys themselves stay intact for the parity taps
                const uint32_t top_range = ((far_bits - near_bits) >> msd_low_bits) + 1u;
                if (n >= (4u << 20))
                    GS_TRY((run_depth_msd_items<gs::SortCfg<uint32_t>::ITEMS_LARGE>(r, st, cp, dbits, top_range, r->dkeys[0].ptr,
                                                                                  r->sorted_rect.ptr, &state->depth_bucket_max, passes)));
                else
                    GS_TRY((run_depth_msd_items<gs::SortCfg<uint32_t>::ITEMS>(r, st, cp, dbits, top_range, r->dkeys[0].ptr,
                                                                            r->sorted_rect.ptr, &state->depth_bucket_max, passes)));
                dside = 0;
            } else if (depth_radix_bits(dbits) == (uint32_t)gs::RADIX_BITS_MAX)
                GS_TRY((run_sort_rb<uint32_t, gs::RADIX_BITS_MAX>(r->dev, k2, v2, r->ghist, r->digit_totals, dc, dbits, &cp,
                                                                  st, dside, passes, r->launches)));
            else
                GS_TRY((run_sort_rb<uint32_t, gs::RADIX_BITS>(r->dev, k2, v2, r->ghist, r->digit_totals, dc, dbits, &cp, st,
                                                              dside, passes, r->launches)));
            dpasses += passes;
            return GS_OK;
        };
        gs::CompactPred pred1;                  // round 1 of a partitioned frame: the keys in front of the threshold
        if (partition) {
            // the totals of the top digit (the histogram half of the MSD-first sort's pass), then the cut
            if (n >= (4u << 20))
                GS_TRY((run_top_digit_totals<gs::SortCfg<uint32_t>::ITEMS_LARGE>(r, st, n, r->list_mode ? &state->list_slots : nullptr)));
            else
                GS_TRY((run_top_digit_totals<gs::SortCfg<uint32_t>::ITEMS>(r, st, n, r->list_mode ? &state->list_slots : nullptr)));
            gs::ThresholdIO ti;
            ti.totals = (const uint32_t *)r->digit_totals.ptr;
            ti.state = state;
            ti.target = round_k;
            ti.low_bits = msd_low_bits;
            hipLaunchKernelGGL(gs::k_round_threshold, dim3(1), dim3(512), 0, st, ti);
            GS_HIP(hipGetLastError());
            r->launches += 3;
            GS_TRY(dev_reserve(r->keep_bits, nslots / 8 + 64));
            pred1.tau_dev = &state->depth_tau;
            pred1.keep_bits = (const uint32_t *)r->keep_bits.ptr;      // (side 0 loads and ignores them)
            pred1.side = 0u;
        }
        GS_TRY(depth_sort(pred1, partition ? &state->round1_visible : &state->visible));

        // ---- pairs in depth order, tile sort, tile ranges: once per round ----
        int tside = 0;
        uint32_t tpasses = 0;
        // round: 0 the frame's only one; 1: the nearest `limit` Gaussians of the depth order; 2: the survivors of
        // k_round2_write (order_r2)
        auto pairs_round = [&](uint32_t round, const uint32_t *order, const uint32_t *count_dev, uint32_t limit, uint32_t *esb_r) -> gs_status {
            if (round <= 1u) mark(ST_EXPAND);
            gs::ExpandIO eo;
            eo.order = order;
            eo.rect = (const uint2 *)r->rect.ptr;
            eo.sorted_rect = (uint2 *)r->sorted_rect.ptr;
            eo.count_dev = count_dev;
            eo.limit = limit;
            eo.round = round;
            eo.sums = (uint32_t *)r->exp_sums.ptr;
            eo.sb_sums = (unsigned long long *)esb_r;
            eo.tvals = (uint32_t *)r->tvals[0].ptr;
            eo.state = state;
            eo.result = result;
            eo.capacity = capacity;
            eo.tiles_x = fc.tiles_x;
            eo.gen = gen;
            eo.sb_bound = exp_grid / gs::EXP_SB + 1;
            eo.rect32 = fc.rect32;
            eo.flags_dev = r->flags_target;
            eo.wt_stores = r->wt_pairs ? 1u : 0u;
            // Where a wave of k_pairs_emit starts: found by the wave itself (a search over the super-chunk
            // sums: one step per 256 of them) or looked up in a table that k_pairs_cursors writes first.
            // The table costs a launch and wins once the search needs more than one step (A/B on one box:
            // 1 M 0.369 vs 0.366 ms, 10 M 1.227 vs 1.226, 50 M 4.60 vs 4.80).  GS3D_CURSOR_KERNEL=0/1 forces.
            static const int cursor_env = std::getenv("GS3D_CURSOR_KERNEL") ? std::atoi(std::getenv("GS3D_CURSOR_KERNEL")) : -1;
            const bool cursor_kernel = cursor_env >= 0 ? cursor_env != 0 : eo.sb_bound > 256u;
            eo.cursors = cursor_kernel ? (gs::PairCursorRec *)r->cursors.ptr : nullptr;
            // XCD-aware span order of the gather: XCD x takes C consecutive spans of every group of 8 C, so that its L2
            // serves part of the gather (neighbours in depth order are often neighbours in the mirror).  Same-box sweep
            // (gpurun_out/r04q/ab*.log): C = 0 / 16 / 64 / 256 / 1024 -> 58.3 / 53.3 / 49.1 / 54.3 / 90.6 us at 10 M, 261 / 258 /
            // 231 / 228 / 293 us at 50 M, 12.0 / - / 10.5 / 21 / 26 us at 1 M.  GS3D_EXPAND_XCD=<C> forces, 0 = dispatch order.
            static const int exp_xcd = std::getenv("GS3D_EXPAND_XCD") ? std::atoi(std::getenv("GS3D_EXPAND_XCD")) : 64;
            uint32_t count_grid = (exp_grid + gs::EXP_COUNT_CHUNKS - 1) / gs::EXP_COUNT_CHUNKS;
            eo.xcd_chunk = 0;
            if (exp_xcd > 0 && count_grid >= 256u) {
                eo.xcd_chunk = (uint32_t)exp_xcd;
                count_grid = 8u * eo.xcd_chunk * ((count_grid + 8u * eo.xcd_chunk - 1u) / (8u * eo.xcd_chunk));
            }
            if (eo.rect32)
                hipLaunchKernelGGL(gs::k_expand_count<true>, dim3(count_grid), dim3(gs::EXP_CHUNK), 0, st, eo);
            else
                hipLaunchKernelGGL(gs::k_expand_count<false>, dim3(count_grid), dim3(gs::EXP_CHUNK), 0, st, eo);
            r->launches++;
            if (cursor_kernel) {
                hipLaunchKernelGGL(gs::k_pairs_cursors, dim3(eo.sb_bound), dim3(gs::EXP_SB), 0, st, eo);
                r->launches++;
            }
            GS_HIP(hipGetLastError());
            if (round <= 1u) mark(ST_TSORT);

            // ---- stable sort on the tile id alone (pairs are generated in depth order by its first pass) ----
            tside = 0;
            const gs::SortCount tc{capacity, &state->pairs};
            // Which tile sort (gs_renderer::tile_msd).  MSD-first needs u16 tile ids with more than 10 bits; its buckets are
            // 2^(bits - 10) consecutive tiles, so what decides is the pair count: up to an average of a quarter of the register
            // path's capacity per bucket it is tried, and a frame that reports a bucket beyond the capacity (FrameResult::
            // tile_bucket_max, one frame late) sends the renderer back to the LSD passes until the pair count has dropped by
            // a quarter below the count that failed.
            bool tile_msd = false;
            if (!wide && tile_bits > (uint32_t)gs::MSD_TOP_BITS && capacity != 0u && round == 0u) {
                static const int tmsd_env = std::getenv("GS3D_TILE_MSD") ? std::atoi(std::getenv("GS3D_TILE_MSD")) : -1;
                const int pinned = r->tile_msd_req >= 0 ? r->tile_msd_req : tmsd_env;
                const int newer = hist_gen[0] > hist_gen[1] ? 0 : 1;
                // pairs this frame is expected to hold: the newest report of this shape, else what sized the buffers
                const uint64_t d_est = !sizing && hist_gen[newer] && r->done_shape[newer] == r->shape_epoch ? hist_d[newer]
                                                                                                             : (uint64_t)capacity * 4u / 5u;
                if (!sizing && hist_gen[newer] && r->done_shape[newer] == r->shape_epoch && hist_tmax[newer] > gs::BKT_CAP_SMALL)
                    r->tile_msd_fail_d = d_est ? d_est : 1u;
                if (sizing) r->tile_msd_fail_d = 0;
                // Measured at 1 M (gpurun_out/r05c/kt_1m.txt): the 1020 buckets of ~2 500 pairs cost the bucket kernel 22 us (one
                // 1024-thread workgroup with 157 KB of LDS per bucket: four rounds of workgroups whose fixed costs dominate) and
                // the 10-bit first pass 6 us more than the 7-bit one — 60 us against the LSD sort's 55.  So the renderer does
                // not choose it by itself (GS3D_TILE_MSD_AUTO=1 lets it); pinned, it is exact (tests/test_gpu_msd_sort.py).
                static const bool tile_auto = std::getenv("GS3D_TILE_MSD_AUTO") && std::getenv("GS3D_TILE_MSD_AUTO")[0] == '1';
                if (pinned >= 0)
                    tile_msd = pinned != 0;
                else
                    tile_msd = tile_auto && d_est <= (uint64_t)gs::BKT_CAP_SMALL * 256u &&
                               (r->tile_msd_fail_d == 0 || d_est < r->tile_msd_fail_d - r->tile_msd_fail_d / 4u);
            }
            r->tile_msd = tile_msd;
            if (tile_msd) {
                t_bucket_max = nullptr;
                if (tile_bits > (uint32_t)gs::MSD_TOP_BITS + 6u) return fail(GS_ERR_INVALID_ARGUMENT, tile_bits, 0, 0, "tile id bits");
                GS_TRY((run_tile_msd_items<gs::SortCfg<uint16_t>::ITEMS>(r, st, eo, tc, tile_bits, num_tiles, zero, &state->tile_bucket_max,
                                                                              tpasses)));
                tside = 0;
            } else {
                // (a frame whose tile sort is LSD reports no bucket size: the next result must not carry a stale one)
                if (r->state_tile_bmax_dirty) GS_HIP(hipMemsetAsync(&state->tile_bucket_max, 0, sizeof(uint32_t), st));
                void *k2[2] = {r->tkeys[0].ptr, r->tkeys[1].ptr};
                void *v2[2] = {r->tvals[0].ptr, r->tvals[1].ptr};
                const gs::ExpandIO *src = &eo;
                if (wide)
                    GS_TRY((run_sort_rb<uint32_t, gs::RADIX_BITS>(r->dev, k2, v2, r->ghist, r->digit_totals, tc, tile_bits, nullptr,
                                                                  st, tside, tpasses, r->launches, src)));
                else
                    GS_TRY((run_sort_rb<uint16_t, gs::RADIX_BITS>(r->dev, k2, v2, r->ghist, r->digit_totals, tc, tile_bits, nullptr,
                                                                  st, tside, tpasses, r->launches, src)));
            }
            r->state_tile_bmax_dirty = tile_msd;
            if (round <= 1u) mark(ST_RANGES);
            // Tile ranges, three ways (the MSD-first tile sort has written them already: k_bucket_sort).  (1) One pass over the sorted keys (k_tile_ranges).  (2) A 32-ary search per tile
            // (k_tile_ranges_search) once reading every key again costs more than a few dependent probes per tile: from
            // a pair capacity of 8 M (GS3D_RANGES_SEARCH=0/1 forces).  (3) The same search run by the blend workgroups
            // themselves (blend_tile_range_wg): no launch in front of the blend, but a workgroup that waits for its
            // probes is occupancy the VALU-bound blend misses — same-box A/B (gpurun_out/r04l/ab.log): blend +3.5 us at
            // 1 M and 10 M, +10 us at 4K, +12 us at 50 M against 6.8 / 10.5 / 35 / 19 us of range kernel saved: frames
            // +1.5 % at 1 M, +-0 at 10 M, -0.5 % at 50 M, -2.3 % at 4K.  It is taken where it pays: images of more than
            // 16384 tiles, where neither stand-alone kernel is cheap (GS3D_RANGES_IN_BLEND=0/1 forces).
            static const int in_blend_env = std::getenv("GS3D_RANGES_IN_BLEND") ? std::atoi(std::getenv("GS3D_RANGES_IN_BLEND")) : -1;
            // (a two-round frame: always — the range array is cleared once per frame, and a launch per round is saved)
            const bool ranges_in_blend = round != 0u || (in_blend_env >= 0 ? in_blend_env != 0 : num_tiles > 16384u);
            static const int ranges_env = std::getenv("GS3D_RANGES_SEARCH") ? std::atoi(std::getenv("GS3D_RANGES_SEARCH")) : -1;
            const bool ranges_search = ranges_env >= 0 ? ranges_env != 0 : capacity >= (8u << 20);
            if (tile_msd) {
                // nothing to do
            } else if (capacity && ranges_in_blend) {
                tile_keys.keys = r->tkeys[tside].ptr;
                tile_keys.count_dev = &state->pairs;
                tile_keys.count_bound = capacity;
                tile_keys.wide = wide ? 1u : 0u;
            } else if (capacity && ranges_search) {
                if (wide)
                    hipLaunchKernelGGL(gs::k_tile_ranges_search<uint32_t>, dim3((num_tiles + 3u) / 4u), dim3(256), 0, st,
                                       (const uint32_t *)r->tkeys[tside].ptr, tc, zero, num_tiles);
                else
                    hipLaunchKernelGGL(gs::k_tile_ranges_search<uint16_t>, dim3((num_tiles + 3u) / 4u), dim3(256), 0, st,
                                       (const uint16_t *)r->tkeys[tside].ptr, tc, zero, num_tiles);
                GS_HIP(hipGetLastError());
                r->launches++;
            } else if (capacity) {
                if (wide)
                    hipLaunchKernelGGL(gs::k_tile_ranges<uint32_t>, dim3((uint32_t)(((uint64_t)capacity + 1023) / 1024)), dim3(256), 0, st,
                                       (const uint32_t *)r->tkeys[tside].ptr, tc, zero);
                else
                    hipLaunchKernelGGL(gs::k_tile_ranges<uint16_t>, dim3((uint32_t)(((uint64_t)capacity + 2047) / 2048)), dim3(256), 0, st,
                                       (const uint16_t *)r->tkeys[tside].ptr, tc, zero);
                GS_HIP(hipGetLastError());
                r->launches++;
            }
            r->tsorted_side = tside;          // (launch_blend reads the pairs of THIS round: round 1's blend runs before the frame ends)
            return GS_OK;
        };
        if (!capacity) two_round = false;
        r->two_round = two_round;
        r->round1 = round_k;
        if (!two_round) {
            GS_TRY(pairs_round(0u, (const uint32_t *)r->dvals[dside].ptr, &state->visible, 0xffffffffu, esb));
        } else {
            tile_keys.done = done_bits;
            tile_keys.open = open_bits;
            if (partition)
                GS_TRY(pairs_round(1u, (const uint32_t *)r->dvals[dside].ptr, &state->round1_visible, 0xffffffffu, esb));
            else
                GS_TRY(pairs_round(1u, (const uint32_t *)r->dvals[dside].ptr, &state->visible, round_k, esb));
            mark(ST_BLEND);
            GS_TRY(launch_blend(1u));
            {
                if (!partition) GS_TRY(dev_reserve(r->order_r2, (nn + 1024) * 4));      // (its largest size at once; padded like the sorts' values)
                gs::Round2IO ro;
                ro.order = (const uint32_t *)r->dvals[dside].ptr + round_k;
                ro.rect = (const uint2 *)r->rect.ptr;
                ro.done = done_bits;
                ro.open = open_bits;
                ro.order_out = (uint32_t *)r->order_r2.ptr;
                ro.state = state;
                ro.first = round_k;
                ro.groups = (n - round_k + gs::R2_GROUP - 1u) / gs::R2_GROUP;
                ro.tiles_x = fc.tiles_x;
                ro.num_tiles = num_tiles;
                // (per-slot arrays cover whole chunks: nslots; a list frame's live in list space, bounded by the same)
                ro.slots = (uint32_t)nslots;
                GS_TRY(dev_reserve(r->keep_bits, nslots / 8 + 64));
                ro.keep_bits = (uint32_t *)r->keep_bits.ptr;
                const uint32_t bits_grid = (uint32_t)((nslots + 256u * gs::R2_SLOT_ITEMS - 1u) / (256u * gs::R2_SLOT_ITEMS));
                {
                    typedef void (*bits_fn)(gs::Round2IO);
                    static const bits_fn tbl[2][2] = {{gs::k_round2_slot_bits<false, false>, gs::k_round2_slot_bits<false, true>},
                                                      {gs::k_round2_slot_bits<true, false>, gs::k_round2_slot_bits<true, true>}};
                    GS_TRY(dev_reserve(r->box_table, ((size_t)num_tiles + 8) * 2));
                    ro.box_table = (const uint16_t *)r->box_table.ptr;
                    hipLaunchKernelGGL(gs::k_round2_gate, dim3(1), dim3(256), 0, st, ro, band_tiles, n,
                                       r->list_mode ? (const uint32_t *)&state->list_slots : (const uint32_t *)nullptr);
                    hipLaunchKernelGGL(gs::k_round2_box_table, dim3((num_tiles + 255u) / 256u), dim3(256), 0, st, (const uint32_t *)done_bits,
                                       (uint16_t *)r->box_table.ptr, fc.tiles_x, fc.tiles_y, (const gs::FrameState *)state);
                    r->launches += 2;
                    const bool lds = num_tiles <= gs::R2_LDS_TILES;
                    const size_t lds_bytes = lds ? ((size_t)num_tiles + 7) / 8 * 16 : 0;
                    const uint32_t persistent = bits_grid < 1024u ? bits_grid : 1024u;      // (4 / 2 workgroups per CU at 1080p / 4K)
                    hipLaunchKernelGGL(tbl[fc.rect32 ? 1 : 0][lds ? 1 : 0], dim3(persistent), dim3(256), lds_bytes, st, ro);
                }
                GS_HIP(hipGetLastError());
                r->launches++;
                if (partition) {
                    // the depth sort of what is left: keys behind the threshold whose slot bit is set
                    gs::CompactPred pred2;
                    pred2.tau_dev = &state->depth_tau;
                    pred2.keep_bits = (const uint32_t *)r->keep_bits.ptr;
                    pred2.side = 1u;
                    GS_TRY(depth_sort(pred2, &state->round2_visible, &state->round2_dense));
                } else {
                    GS_TRY(dev_reserve(r->r2_scan, ((size_t)nn / gs::R2_GROUP + 1) * (2 * 4 + 32 * 8) + 64));      // (its largest size: never regrown under frames in flight)
                    ro.masks = (unsigned long long *)r->r2_scan.ptr;
                    ro.counts = (uint32_t *)(ro.masks + (size_t)ro.groups * 32);
                    ro.offsets = ro.counts + ro.groups;
                    hipLaunchKernelGGL(gs::k_round2_count, dim3(ro.groups), dim3(256), 0, st, ro);
                    gs::ScanJob js{ro.counts, ro.offsets, &state->round2_visible, ro.groups, nullptr};
                    hipLaunchKernelGGL(gs::k_scan_chunks, dim3(1), dim3(1024), 0, st, js, js);
                    hipLaunchKernelGGL(gs::k_round2_write, dim3(ro.groups), dim3(256), 0, st, ro);
                    r->launches += 3;
                }
                GS_HIP(hipGetLastError());
            }
            GS_TRY(pairs_round(2u, partition ? (const uint32_t *)r->dvals[dside].ptr : (const uint32_t *)r->order_r2.ptr, &state->round2_visible,
                               0xffffffffu, esb2));
        }
        r->sort_passes = dpasses + tpasses;
        r->dsorted_side = dside;
        r->tsorted_side = tside;
    }
    if (!r->two_round) mark(ST_BLEND);
    GS_TRY(launch_blend(r->two_round ? 2u : 0u));
    mark(ST_FRAME);
    if (timing) {
        (void)hipEventRecord(r->ev[ST_COUNT], st);
        r->ev_pending = true;
    }
    return GS_OK;   // done_guard records the end-of-frame event
}

static gs_status download_sync(gs_renderer *r, void *dst, const void *src, size_t bytes) {
    if (!bytes) return GS_OK;
    GS_HIP(sync_last_frame(r));
    hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail(GS_ERR_DOWNLOAD, (uint64_t)e, 0, 0, "download failed: %s", hipGetErrorString(e));
    return GS_OK;
}

// Which Gaussian every OUTPUT slot of the last frame holds (0xffffffff: none — padding past N).  The
// per-slot arrays of a frame (records, depth keys, rects, the values of the sorts) are indexed by
// mirror slot, or — when k_block_cull handed the preprocess kernel its blocks — by LIST slot: list
// position * 1024 + lane.  Either way the taps translate back to Gaussian indices here: through the
// block list (if any) and the buffer's mirror order (if any).
static gs_status download_slot_map(gs_renderer *r, std::vector<uint32_t> &gaussian_of_slot) {
    gaussian_of_slot.clear();
    if (!r->have_frame || !r->n) return GS_OK;
    std::vector<uint32_t> order;
    if (r->last_order) {
        order.resize(r->n);
        GS_TRY(download_sync(r, order.data(), r->last_order->ptr, r->n * 4));
    }
    if (!r->list_mode) {
        gaussian_of_slot.resize(r->n);
        for (size_t slot = 0; slot < r->n; slot++) gaussian_of_slot[slot] = order.empty() ? (uint32_t)slot : order[slot];
        return GS_OK;
    }
    gs::FrameState fs;
    GS_TRY(download_sync(r, &fs, r->state.ptr, sizeof(fs)));
    std::vector<uint32_t> list(fs.list_blocks);
    GS_TRY(download_sync(r, list.data(), r->block_list.ptr, (size_t)fs.list_blocks * 4));
    gaussian_of_slot.assign((size_t)fs.list_blocks * gs::PP_CHUNK, 0xffffffffu);
    for (size_t b = 0; b < list.size(); b++)
        for (size_t l = 0; l < gs::PP_CHUNK; l++) {
            const size_t slot = (size_t)list[b] * gs::PP_CHUNK + l;
            if (slot < r->n) gaussian_of_slot[b * gs::PP_CHUNK + l] = order.empty() ? (uint32_t)slot : order[slot];
        }
    return GS_OK;
}

// depth bits of every output slot of the last frame (0xffffffff = culled): the dense keys of
// preprocess plus the key bias; chunks without visible Gaussians may be block-culled and stale
static gs_status download_slot_depths(gs_renderer *r, size_t slots, std::vector<uint32_t> &depth) {
    depth.assign(slots, 0xffffffffu);
    if (!r->have_frame || !slots) return GS_OK;
    const size_t nchunks = (slots + gs::PP_CHUNK - 1) / gs::PP_CHUNK;
    std::vector<uint32_t> chunk_vis(nchunks);
    GS_TRY(download_sync(r, depth.data(), r->depth.ptr, slots * 4));
    GS_TRY(download_sync(r, chunk_vis.data(), r->chunk_vis.ptr, nchunks * 4));
    for (size_t slot = 0; slot < slots; slot++) {
        if (chunk_vis[slot / gs::PP_CHUNK] == 0u) depth[slot] = 0xffffffffu;
        else if (depth[slot] != 0xffffffffu) depth[slot] += r->key_bias;
    }
    return GS_OK;
}

// The device keeps the projected data as dense per-slot arrays (36-byte blend record, tile rect)
// plus the compacted depth keys; the 48-byte gs_projected view of DESIGN.md §3.3 is assembled here.
extern "C" gs_status gs_renderer_download_projected(gs_renderer *r, gs_projected *proj_out,
                                                    uint32_t *tiles_out, size_t n) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null renderer");
    if (n > r->n) return fail(GS_ERR_INVALID_ARGUMENT, n, r->n, 0, "n exceeds last frame");
    GS_TRY(use_device(r->dev));
    static_assert(sizeof(gs_projected) == 48, "record size");
    if (!n) return GS_OK;
    // a Gaussian whose block the list dropped has no slot at all: it is culled
    if (tiles_out) std::memset(tiles_out, 0, n * sizeof(uint32_t));
    if (proj_out) std::memset(proj_out, 0, n * sizeof(gs_projected));
    // the device arrays are indexed by output slot; a partial request (n < N) still needs all slots
    std::vector<uint32_t> slot_map, depth;
    GS_TRY(download_slot_map(r, slot_map));
    const size_t total = slot_map.size();
    if (!total) return GS_OK;
    std::vector<uint32_t> recs(total * gs::REC_WORDS);
    std::vector<uint2> rect(total);
    std::vector<uint32_t> kept(total);      // tiles of the rect that the frame emits pairs for (all of them, or — rect version 4 — fewer)
    GS_TRY(download_sync(r, recs.data(), r->recs.ptr, total * 4 * gs::REC_WORDS));
    GS_TRY(download_slot_depths(r, total, depth));
    if (r->rect32) {
        std::vector<uint32_t> packed(total);
        GS_TRY(download_sync(r, packed.data(), r->rect.ptr, total * 4));
        for (size_t k = 0; k < total; k++) {      // (culled slots: never looked at)
            uint32_t rows;
            gs::rect_unpack32(packed[k], r->tiles_x ? r->tiles_x : 1u, rect[k].x, rect[k].y, rows);
            kept[k] = gs::rect_count32(packed[k]);
        }
    } else {
        std::vector<uint2> raw(total);
        GS_TRY(download_sync(r, raw.data(), r->rect.ptr, total * 8));
        for (size_t k = 0; k < total; k++) {
            uint32_t rows;
            gs::rect_unpack64(raw[k], rect[k].x, rect[k].y, rows);
            kept[k] = gs::rect_count64(raw[k]);
        }
    }
    for (size_t slot = 0; slot < total; slot++) {
        const size_t i = slot_map[slot];   // Gaussian index of this slot
        if (i >= n) continue;
        // a slot absent from the depth keys is culled (its chunk may even have been block-culled,
        // in which case its per-slot arrays are stale)
        const bool vis = depth[slot] != 0xffffffffu;
        if (tiles_out) tiles_out[i] = vis ? kept[slot] : 0u;
        if (proj_out) {
            gs_projected &p = proj_out[i];
            if (vis) {
                std::memcpy(&p, &recs[slot * gs::REC_WORDS], 36);
                std::memcpy(&p.depth, &depth[slot], 4);
                p.tx0 = (uint16_t)(rect[slot].x & 0xffffu);
                p.ty0 = (uint16_t)(rect[slot].x >> 16);
                p.tx1 = (uint16_t)(rect[slot].y & 0xffffu);
                p.ty1 = (uint16_t)(rect[slot].y >> 16);
            }
        }
    }
    return GS_OK;
}

// The frame sorts (depth) and (tile) separately; the canonical 64-bit key of DESIGN.md §3.4 is
// rebuilt here as tile << 32 | depth bits of the pair's Gaussian.
extern "C" gs_status gs_renderer_download_sorted(gs_renderer *r, uint64_t *keys_out, uint32_t *idx_out,
                                                 uint64_t capacity, uint64_t *pairs_out) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null renderer");
    if (r->have_frame && r->two_round)
        return fail(GS_ERR_INVALID_ARGUMENT, 2, 0, 0, "the last frame took two rounds: its pair arrays hold the second round only");
    GS_TRY(use_device(r->dev));
    GS_HIP(sync_last_frame(r));
    uint64_t d = r->have_frame ? last_result(r).pairs_total : 0;
    if (d > r->pair_capacity) d = r->pair_capacity;     // an overflowed frame only holds this many
    if (pairs_out) *pairs_out = d;
    uint64_t m = d < capacity ? d : capacity;
    if (!m) return GS_OK;
    std::vector<uint32_t> idx(m), slot_map;   // output slots of the pairs; slot -> Gaussian
    GS_TRY(download_sync(r, idx.data(), r->tvals[r->tsorted_side].ptr, m * 4));
    GS_TRY(download_slot_map(r, slot_map));
    if (idx_out)
        for (uint64_t j = 0; j < m; j++) idx_out[j] = slot_map[idx[j]];
    if (keys_out) {
        std::vector<uint32_t> depth;
        GS_TRY(download_slot_depths(r, slot_map.size(), depth));
        if (r->wide_tiles) {
            std::vector<uint32_t> t(m);
            GS_TRY(download_sync(r, t.data(), r->tkeys[r->tsorted_side].ptr, m * 4));
            for (uint64_t j = 0; j < m; j++) keys_out[j] = ((uint64_t)t[j] << 32) | depth[idx[j]];
        } else {
            std::vector<uint16_t> t(m);
            GS_TRY(download_sync(r, t.data(), r->tkeys[r->tsorted_side].ptr, m * 2));
            for (uint64_t j = 0; j < m; j++) keys_out[j] = ((uint64_t)t[j] << 32) | depth[idx[j]];
        }
    }
    return GS_OK;
}

extern "C" gs_status gs_renderer_download_ranges(gs_renderer *r, uint32_t *ranges_out,
                                                 size_t num_tiles) {
    if (!r || !ranges_out) return fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    if (num_tiles > (size_t)r->tiles_x * r->tiles_y)
        return fail(GS_ERR_INVALID_ARGUMENT, num_tiles, 0, 0, "too many tiles");
    if (r->have_frame && r->two_round)
        return fail(GS_ERR_INVALID_ARGUMENT, 2, 0, 0, "the last frame took two rounds: its tile ranges are the second round's");
    GS_TRY(use_device(r->dev));
    return download_sync(r, ranges_out, r->zero_region.ptr, num_tiles * 8);
}

// ------------------------------------------------------------------------------------------------
// stand-alone primitives
// ------------------------------------------------------------------------------------------------

extern "C" gs_status gs_sort_pairs_u64(gs_device *dev, gs_stream *s, uint64_t *keys, uint32_t *values,
                                       uint64_t count, uint32_t end_bit) {
    if (!dev || (count && (!keys || !values)) || end_bit > 64 || count > 0xfffffff0ull)
        return fail(GS_ERR_INVALID_ARGUMENT, count, end_bit, 0, "bad argument");
    GS_TRY(use_device(dev));
    if (!count) return GS_OK;
    hipStream_t st = stream_of(dev, s);
    DevArray k[2], v[2], gh, dt;
    gs_status rc = GS_OK;
    for (int i = 0; i < 2 && rc == GS_OK; i++) {
        rc = dev_reserve(k[i], count * 8);
        if (rc == GS_OK) rc = dev_reserve(v[i], count * 4);
    }
    if (rc == GS_OK) {
        hipError_t e = hipMemcpyAsync(k[0].ptr, keys, count * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(v[0].ptr, values, count * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(GS_ERR_HIP, (uint64_t)e, 0, 0, "upload failed: %s", hipGetErrorString(e));
    }
    int side = 0;
    uint32_t passes = 0;
    if (rc == GS_OK) {
        void *k2[2] = {k[0].ptr, k[1].ptr};
        void *v2[2] = {v[0].ptr, v[1].ptr};
        rc = sort_pairs_device<uint64_t>(dev, k2, v2, gh, dt, (uint32_t)count, end_bit, st, side, passes);
    }
    if (rc == GS_OK) {
        hipError_t e = hipMemcpyAsync(keys, k[side].ptr, count * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(values, v[side].ptr, count * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(GS_ERR_HIP, (uint64_t)e, 0, 0, "download failed: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < 2; i++) {
        dev_free(k[i]);
        dev_free(v[i]);
    }
    dev_free(gh);
    dev_free(dt);
    return rc;
}

namespace gs {
// chunk sums for the stand-alone scan (the frame fuses this into k_preprocess)
__global__ __launch_bounds__(PP_THREADS) void k_chunk_sums(const uint32_t *__restrict__ in, uint32_t n,
                                                           uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_red[4];
    uint32_t base = blockIdx.x * PP_CHUNK + threadIdx.x * PP_ITEMS;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < PP_ITEMS; k++) v += base + k < n ? in[base + k] : 0u;
    v = wave_reduce_add(v);
    if ((threadIdx.x & 63u) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// per-chunk exclusive scan written out
__global__ __launch_bounds__(PP_THREADS) void k_chunk_scan_write(const uint32_t *__restrict__ in,
                                                                 const uint32_t *__restrict__ chunk_offsets,
                                                                 uint32_t n, uint32_t *__restrict__ out) {
    __shared__ uint32_t s_scan[4];
    uint32_t base = blockIdx.x * PP_CHUNK + threadIdx.x * PP_ITEMS;
    uint32_t c[PP_ITEMS], sum = 0;
#pragma unroll
    for (int k = 0; k < PP_ITEMS; k++) {
        c[k] = base + k < n ? in[base + k] : 0u;
        sum += c[k];
    }
    uint32_t total;
    uint32_t off = chunk_offsets[blockIdx.x] + block_exclusive_scan_256(sum, s_scan, total);
#pragma unroll
    for (int k = 0; k < PP_ITEMS; k++) {
        if (base + k < n) out[base + k] = off;
        off += c[k];
    }
}
}  // namespace gs

extern "C" gs_status gs_exclusive_scan_u32(gs_device *dev, gs_stream *s, const uint32_t *in,
                                           uint32_t *out, uint64_t count, uint64_t *total_out) {
    if (!dev || (count && (!in || !out)) || count > 0xfffffff0ull)
        return fail(GS_ERR_INVALID_ARGUMENT, count, 0, 0, "bad argument");
    GS_TRY(use_device(dev));
    if (total_out) *total_out = 0;
    if (!count) return GS_OK;
    hipStream_t st = stream_of(dev, s);
    uint32_t n = (uint32_t)count;
    uint32_t nchunks = (n + gs::PP_CHUNK - 1) / gs::PP_CHUNK;
    DevArray din, dout, sums, offs, tot;
    gs_status rc = dev_reserve(din, count * 4);
    if (rc == GS_OK) rc = dev_reserve(dout, count * 4);
    if (rc == GS_OK) rc = dev_reserve(sums, (size_t)nchunks * 4);
    if (rc == GS_OK) rc = dev_reserve(offs, (size_t)nchunks * 4);
    if (rc == GS_OK) rc = dev_reserve(tot, 16);
    uint32_t total = 0;
    if (rc == GS_OK) {
        hipError_t e = hipMemcpyAsync(din.ptr, in, count * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gs::k_chunk_sums, dim3(nchunks), dim3(gs::PP_THREADS), 0, st,
                               (const uint32_t *)din.ptr, n, (uint32_t *)sums.ptr);
            gs::ScanJob job{(const uint32_t *)sums.ptr, (uint32_t *)offs.ptr, (uint32_t *)tot.ptr, nchunks};
            hipLaunchKernelGGL(gs::k_scan_chunks, dim3(1), dim3(1024), 0, st, job, job);
            hipLaunchKernelGGL(gs::k_chunk_scan_write, dim3(nchunks), dim3(gs::PP_THREADS), 0, st,
                               (const uint32_t *)din.ptr, (const uint32_t *)offs.ptr, n,
                               (uint32_t *)dout.ptr);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out, dout.ptr, count * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&total, tot.ptr, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(GS_ERR_HIP, (uint64_t)e, 0, 0, "scan failed: %s", hipGetErrorString(e));
    }
    if (total_out) *total_out = total;
    for (DevArray *a : {&din, &dout, &sums, &offs, &tot}) dev_free(*a);
    return rc;
}
