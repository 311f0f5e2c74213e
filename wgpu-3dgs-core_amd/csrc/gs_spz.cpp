// gs_spz.cpp — SPZ (Niantic) source format: gzip'd column-major quantised Gaussians, versions 1-3.
// Restates src/source_format/spz.rs:436-959 (header, column order, gzip framing) and
// src/gaussian.rs:126-352 (Gaussian::from_spz / to_spz, GaussianToSpzOptions).  Host side only.
// The reference's arithmetic is mirrored operation by operation, including two quirks that a
// drop-in must keep: quantize_sh only buckets when bucket_size < 8 (gaussian.rs:317-324, so the
// default [5,4,4] bits do not bucket at all), and the smallest-three quaternion is unpacked in
// ascending component order (gaussian.rs:171-190; Niantic's own decoder walks it descending).
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "gs_convert.h"
#include "gs_internal.h"

static_assert(sizeof(gs_spz_header) == 16, "SpzGaussiansHeaderPod is 16 bytes");

static const uint32_t k_magic = 0x5053474eu;   // "NGSP", spz.rs:458
static const float k_ab = 0.2820948f / 0.15f;  // SPZ_COLOR_TO_LINEAR_FRAC_A_B, gaussian.rs:127-128
static const float k_c = (1.0f - k_ab) * (0.5f * 255.0f);   // SPZ_COLOR_TO_LINEAR_C, :130-131

static uint32_t num_coefficients(uint32_t deg) { return deg == 0 ? 0 : deg == 1 ? 3 : deg == 2 ? 8 : 15; }

extern "C" void gs_spz_options_default(gs_spz_options *o) {   // spz.rs:985-999
    o->version = 3;
    o->sh_degree = 3;
    o->fractional_bits = 12;
    o->antialiased = 0;
    o->sh_quantize_bits[0] = 5;
    o->sh_quantize_bits[1] = 4;
    o->sh_quantize_bits[2] = 4;
}

// SpzGaussiansHeader::try_from_pod, spz.rs:487-512
static gs_status validate_header(const gs_spz_header &h) {
    if (h.magic != k_magic)
        return gs_fail(GS_ERR_SPZ, h.magic, k_magic, 0, "Invalid SPZ magic number: %X, expected %X", h.magic, k_magic);
    if (h.version < 1 || h.version > 3)
        return gs_fail(GS_ERR_SPZ, h.version, 0, 0, "Unsupported SPZ version: %u, expected one of 1..=3", h.version);
    if (h.sh_degree > 3)
        return gs_fail(GS_ERR_SPZ, h.sh_degree, 0, 0, "Unsupported SPZ SH degree: %u, expected one of 0..=3", h.sh_degree);
    return GS_OK;
}

// ---- f16 (half crate const conversions == IEEE RNE) ---------------------------------------------
static uint16_t f32_to_f16(float value) {
    union { uint32_t u; float f; } f, magic;
    f.f = value;
    const uint32_t f32infty = 255u << 23, f16max = (127u + 16u) << 23;
    magic.u = ((127u - 15u) + (23u - 10u) + 1u) << 23;
    uint32_t sign = f.u & 0x80000000u;
    f.u ^= sign;
    uint16_t o;
    if (f.u >= f16max) o = (f.u > f32infty) ? (uint16_t)0x7e00u : (uint16_t)0x7c00u;
    else if (f.u < (113u << 23)) { f.f += magic.f; o = (uint16_t)(f.u - magic.u); }
    else { uint32_t odd = (f.u >> 13) & 1u; f.u += ((uint32_t)(15 - 127) << 23) + 0xfffu; f.u += odd; o = (uint16_t)(f.u >> 13); }
    return (uint16_t)(o | (sign >> 16));
}
static float f16_to_f32(uint16_t h) {
    union { uint32_t u; float f; } o;
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) { o.f = std::ldexp((float)m, -24); o.u |= sign; }
    else if (e == 31) o.u = sign | 0x7f800000u | (m << 13);
    else o.u = sign | ((e + 112u) << 23) | (m << 13);
    return o.f;
}
static uint8_t sat_u8(float v) { return !(v > 0.0f) ? 0 : v >= 255.0f ? 255 : (uint8_t)v; }
static uint32_t sat_u32(float v) { return !(v > 0.0f) ? 0u : v >= 4294967295.0f ? 0xffffffffu : (uint32_t)v; }
static int32_t sat_i32(float v) { return v != v ? 0 : v >= 2147483647.0f ? 2147483647 : v <= -2147483648.0f ? (int32_t)0x80000000 : (int32_t)v; }

template <class F>
static void spz_parallel_for(size_t n, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    size_t threads = n < 32768 ? 1 : (hw ? (hw > 32 ? 32 : hw) : 4);
    if (const char *e = std::getenv("GS3D_HOST_THREADS")) threads = std::atoi(e) > 0 ? (size_t)std::atoi(e) : threads;
    if (threads <= 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    const size_t per = (n + threads - 1) / threads;
    for (size_t t = 0; t < threads; t++) {
        const size_t a = t * per, b = a + per < n ? a + per : n;
        if (a >= b) break;
        pool.emplace_back([=] { fn(a, b); });
    }
    for (auto &th : pool) th.join();
}

// column sizes (bytes per Gaussian)
static size_t pos_bytes(uint32_t version) { return version == 1 ? 6 : 9; }
static size_t rot_bytes(uint32_t version) { return version >= 3 ? 4 : 3; }

static size_t payload_bytes(const gs_spz_header &h) {
    return 16 + (size_t)h.num_points * (pos_bytes(h.version) + 1 + 3 + 3 + rot_bytes(h.version) + 3 * num_coefficients(h.sh_degree));
}

gs_status gs_spz_payload_layout(const void *bytes, size_t len, gs_spz_header *header_out, size_t col_offset[6],
                                uint32_t *ncoef_out) {
    if (!bytes) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    if (len < 16) return gs_fail(GS_ERR_SPZ, len, 0, 0, "failed to fill whole buffer");
    gs_spz_header h;
    std::memcpy(&h, bytes, 16);
    gs_status rc = validate_header(h);
    if (rc != GS_OK) return rc;
    if (len < payload_bytes(h)) return gs_fail(GS_ERR_SPZ, len, payload_bytes(h), 0, "failed to fill whole buffer");
    const size_t n = h.num_points;
    col_offset[0] = 16;
    col_offset[1] = col_offset[0] + n * pos_bytes(h.version);
    col_offset[2] = col_offset[1] + n;
    col_offset[3] = col_offset[2] + 3 * n;
    col_offset[4] = col_offset[3] + 3 * n;
    col_offset[5] = col_offset[4] + n * rot_bytes(h.version);
    if (header_out) *header_out = h;
    if (ncoef_out) *ncoef_out = num_coefficients(h.sh_degree);
    return GS_OK;
}

// ---- Gaussian::from_spz over a decompressed buffer (gaussian.rs:134-229, spz.rs:739-771) --------
extern "C" gs_status gs_spz_decode_decompressed(const void *bytes, size_t len, gs_spz_header *header_out,
                                                gs_gaussian *out, size_t capacity, size_t *count_out) {
    if (!bytes || !count_out) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    if (len < 16) return gs_fail(GS_ERR_SPZ, len, 0, 0, "failed to fill whole buffer");
    gs_spz_header h;
    std::memcpy(&h, bytes, 16);
    gs_status rc = validate_header(h);
    if (rc != GS_OK) return rc;
    if (header_out) *header_out = h;
    *count_out = h.num_points;
    if (!out) return GS_OK;
    if (len < payload_bytes(h)) return gs_fail(GS_ERR_SPZ, len, payload_bytes(h), 0, "failed to fill whole buffer");
    const size_t n = h.num_points, m = n < capacity ? n : capacity;
    const uint8_t *b = (const uint8_t *)bytes + 16;
    const uint8_t *pos = b;
    const uint8_t *alpha = pos + n * pos_bytes(h.version);
    const uint8_t *color = alpha + n;
    const uint8_t *scale = color + 3 * n;
    const uint8_t *rot = scale + 3 * n;
    const uint8_t *sh = rot + n * rot_bytes(h.version);
    const uint32_t ncoef = num_coefficients(h.sh_degree);
    gs::SpzView view{pos, alpha, color, scale, rot, sh, h.version, h.fractional_bits, ncoef};
    static_assert(sizeof(gs_gaussian) == gs::CV_GAUSSIAN_WORDS * 4, "gs_gaussian layout");
    // the per-Gaussian arithmetic is gs_convert.h's, shared with the device kernel (k_from_spz_pods)
    spz_parallel_for(m, [=](size_t a, size_t e) {
        for (size_t i = a; i < e; i++) {
            uint32_t gw[gs::CV_GAUSSIAN_WORDS];
            gs::spz_to_gaussian_words(view, i, gw, [](float x) { return std::sqrt(x); });
            std::memcpy(&out[i], gw, sizeof(gw));
        }
    });
    return GS_OK;
}

// ---- Gaussian::to_spz (gaussian.rs:240-352) + write_decompressed (spz.rs:776-794) --------------
static uint8_t quantize_sh(float x, uint32_t bucket) {   // gaussian.rs:317-324
    uint32_t q = sat_u32(std::round(x * 128.0f + 128.0f));
    if (!(bucket >= 8)) q = (q + bucket / 2) / bucket * bucket;
    return (uint8_t)(q > 255u ? 255u : q);
}

extern "C" gs_status gs_spz_encode_decompressed(const gs_gaussian *in, size_t n, const gs_spz_options *opt,
                                                void *out, size_t capacity, size_t *bytes_out) {
    if (!opt || !bytes_out || (n && !in)) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    gs_spz_header h;
    h.magic = k_magic;
    h.version = opt->version;
    h.num_points = (uint32_t)n;
    h.sh_degree = opt->sh_degree;
    h.fractional_bits = opt->fractional_bits;
    h.flags = opt->antialiased ? 1 : 0;
    h.reserved = 0;
    gs_status rc = validate_header(h);
    if (rc != GS_OK) return rc;
    if (h.sh_degree && opt->sh_quantize_bits[h.sh_degree - 1] > 8)
        return gs_fail(GS_ERR_INVALID_ARGUMENT, opt->sh_quantize_bits[h.sh_degree - 1], 8, 0,
                       "sh_quantize_bits must be in 0..=8");
    if (h.version > 1 && h.fractional_bits > 30)
        return gs_fail(GS_ERR_INVALID_ARGUMENT, h.fractional_bits, 30, 0, "fractional_bits must be below 31");
    const size_t total = payload_bytes(h);
    *bytes_out = total;
    if (!out) return GS_OK;
    if (capacity < total) return gs_fail(GS_ERR_INVALID_ARGUMENT, capacity, total, 0, "output buffer too small");
    uint8_t *b = (uint8_t *)out;
    std::memcpy(b, &h, 16);
    uint8_t *pos = b + 16;
    uint8_t *alpha = pos + n * pos_bytes(h.version);
    uint8_t *color = alpha + n;
    uint8_t *scale = color + 3 * n;
    uint8_t *rot = scale + 3 * n;
    uint8_t *sh = rot + n * rot_bytes(h.version);
    const uint32_t ncoef = num_coefficients(h.sh_degree);
    const uint32_t bits = h.sh_degree ? opt->sh_quantize_bits[h.sh_degree - 1] : 8;
    const uint32_t bucket = 1u << (8 - bits);
    for (size_t i = 0; i < n; i++) {
        const gs_gaussian &g = in[i];
        if (h.version == 1) {
            for (int c = 0; c < 3; c++) {
                uint16_t v = f32_to_f16(g.pos[c]);
                std::memcpy(pos + 6 * i + 2 * c, &v, 2);
            }
        } else {
            const float s = (float)(1 << h.fractional_bits);
            for (int c = 0; c < 3; c++) {
                int32_t fixed = sat_i32(std::round(g.pos[c] * s));
                uint8_t *p = pos + 9 * i + 3 * c;
                p[0] = (uint8_t)(fixed & 0xff);
                p[1] = (uint8_t)((fixed >> 8) & 0xff);
                p[2] = (uint8_t)((fixed >> 16) & 0xff);
            }
        }
        for (int c = 0; c < 3; c++) {
            float v = std::round((std::log(g.scale[c]) + 10.0f) * 16.0f);
            scale[3 * i + c] = sat_u8(std::fmin(std::fmax(v, 0.0f), 255.0f));
        }
        // Quat::normalize
        float q[4];
        {
            float len = std::sqrt(((g.rot[0] * g.rot[0] + g.rot[1] * g.rot[1]) + g.rot[2] * g.rot[2]) + g.rot[3] * g.rot[3]);
            for (int c = 0; c < 4; c++) q[c] = g.rot[c] / len;
        }
        if (h.version >= 3) {
            uint32_t largest = 0;
            for (uint32_t k = 1; k < 4; k++)
                if (std::fabs(q[k]) >= std::fabs(q[largest])) largest = k;   // max_by keeps the LAST maximum
            const uint32_t mask = (1u << 9) - 1u;
            uint32_t negate = q[largest] < 0.0f ? 1u : 0u;
            uint32_t comp = largest;
            for (uint32_t k = 0; k < 4; k++) {
                if (k == largest) continue;
                uint32_t neg = (q[k] < 0.0f ? 1u : 0u) ^ negate;
                float mv = (float)mask * (std::fabs(q[k]) * 1.41421356237309504880f) + 0.5f;
                uint32_t mag = sat_u32(std::fmin(std::fmax(mv, 0.0f), (float)mask - 1.0f));
                comp = (comp << 10) | (neg << 9) | mag;
            }
            uint8_t *r = rot + 4 * i;
            r[0] = (uint8_t)(comp & 0xff);
            r[1] = (uint8_t)((comp >> 8) & 0xff);
            r[2] = (uint8_t)((comp >> 16) & 0xff);
            r[3] = (uint8_t)((comp >> 24) & 0xff);
        } else {
            float sgn = q[3] < 0.0f ? -1.0f : 1.0f;
            for (int c = 0; c < 3; c++) {
                float v = std::round(((sgn < 0.0f ? -q[c] : q[c]) + 1.0f) * 127.5f);
                rot[3 * i + c] = sat_u8(std::fmin(std::fmax(v, 0.0f), 255.0f));
            }
        }
        alpha[i] = g.color[3];
        for (int c = 0; c < 3; c++) {
            float v = ((float)g.color[c] - k_c) / k_ab;
            color[3 * i + c] = sat_u8(std::fmin(std::fmax(v, 0.0f), 255.0f));
        }
        for (uint32_t k = 0; k < ncoef; k++)
            for (int c = 0; c < 3; c++) sh[(i * ncoef + k) * 3 + c] = quantize_sh(g.sh[3 * k + c], bucket);
    }
    return GS_OK;
}

// ---- gzip framing (flate2 GzDecoder / GzEncoder, spz.rs:945-959) ---------------------------------
// Inputs of any size are fed to zlib in chunks (avail_in is 32-bit).  The output is bounded: once
// the 16-byte SPZ header has been inflated the payload size is known (payload_bytes), and a stream
// that inflates past it (plus slack) is rejected instead of growing without limit; a stream whose
// first bytes are not an SPZ header is capped at 1 GiB.
static gs_status gunzip(const void *bytes, size_t len, std::vector<uint8_t> &out);
gs_status gs_spz_gunzip(const void *bytes, size_t len, std::vector<uint8_t> &out) { return gunzip(bytes, len, out); }
static gs_status gunzip(const void *bytes, size_t len, std::vector<uint8_t> &out) {
    constexpr size_t CHUNK = (size_t)1 << 30;
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return gs_fail(GS_ERR_SPZ, 0, 0, 0, "zlib init failed");
    const uint8_t *in = (const uint8_t *)bytes;
    size_t in_left = len, total = 0, limit = CHUNK;
    bool limit_from_header = false;
    int rc = Z_OK;
    try {
        out.resize(len < ((size_t)1 << 26) ? len * 4 + 1024 : len + (len >> 1));
        do {
            if (zs.avail_in == 0 && in_left) {
                const size_t take = in_left < CHUNK ? in_left : CHUNK;
                zs.next_in = (Bytef *)in;
                zs.avail_in = (uInt)take;
                in += take;
                in_left -= take;
            }
            if (total == out.size()) out.resize(out.size() < limit ? (out.size() * 2 < limit ? out.size() * 2 : limit + 1) : out.size() + 1);
            size_t room = out.size() - total;
            if (room > CHUNK) room = CHUNK;
            zs.next_out = out.data() + total;
            zs.avail_out = (uInt)room;
            rc = inflate(&zs, Z_NO_FLUSH);
            total += room - zs.avail_out;
            if (!limit_from_header && total >= 16) {
                gs_spz_header h;
                std::memcpy(&h, out.data(), 16);
                if (h.magic == k_magic && h.version >= 1 && h.version <= 3 && h.sh_degree <= 3) {
                    limit = payload_bytes(h) + 65536;
                    limit_from_header = true;
                }
            }
            if (total > limit) {
                inflateEnd(&zs);
                return gs_fail(GS_ERR_SPZ, total, limit, 0, "gzip stream inflates past the size its SPZ header declares");
            }
            if (rc == Z_BUF_ERROR && zs.avail_in == 0 && in_left == 0) break;    // truncated input
        } while (rc == Z_OK || (rc == Z_BUF_ERROR && (zs.avail_out == 0 || in_left)));
    } catch (const std::bad_alloc &) {
        inflateEnd(&zs);
        return gs_fail(GS_ERR_OUT_OF_MEMORY, out.size(), 0, 0, "out of memory inflating SPZ data");
    }
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) return gs_fail(GS_ERR_SPZ, (uint64_t)rc, 0, 0, "invalid gzip header");
    out.resize(total);
    return GS_OK;
}

static gs_status gzip_member(const void *bytes, size_t len, std::vector<uint8_t> &z) {
    constexpr size_t CHUNK = (size_t)1 << 30;
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 16 + MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK)
        return gs_fail(GS_ERR_SPZ, 0, 0, 0, "zlib init failed");
    int zr = Z_OK;
    size_t total = 0;
    try {
        // deflateBound takes a uLong; bound the whole input chunk by chunk
        size_t bound = 64;
        for (size_t left = len; ; left -= CHUNK) {
            bound += deflateBound(&zs, (uLong)(left < CHUNK ? left : CHUNK));
            if (left <= CHUNK) break;
        }
        z.resize(bound);
        const uint8_t *in = (const uint8_t *)bytes;
        size_t in_left = len;
        do {
            if (zs.avail_in == 0 && in_left) {
                const size_t take = in_left < CHUNK ? in_left : CHUNK;
                zs.next_in = (Bytef *)in;
                zs.avail_in = (uInt)take;
                in += take;
                in_left -= take;
            }
            size_t room = z.size() - total;
            if (room > CHUNK) room = CHUNK;
            zs.next_out = z.data() + total;
            zs.avail_out = (uInt)room;
            zr = deflate(&zs, in_left ? Z_NO_FLUSH : Z_FINISH);
            total += room - zs.avail_out;
        } while (zr == Z_OK && total < z.size());
    } catch (const std::bad_alloc &) {
        deflateEnd(&zs);
        return gs_fail(GS_ERR_OUT_OF_MEMORY, len, 0, 0, "out of memory compressing SPZ data");
    }
    deflateEnd(&zs);
    if (zr != Z_STREAM_END) return gs_fail(GS_ERR_SPZ, (uint64_t)zr, 0, 0, "gzip compression failed");
    z.resize(total);
    return GS_OK;
}

static gs_status copy_out(const std::vector<uint8_t> &v, void *out, size_t capacity, size_t *bytes_out) {
    *bytes_out = v.size();
    if (!out) return GS_OK;
    if (capacity < v.size()) return gs_fail(GS_ERR_INVALID_ARGUMENT, capacity, v.size(), 0, "output buffer too small");
    if (!v.empty()) std::memcpy(out, v.data(), v.size());
    return GS_OK;
}

// the gzip framing on its own: SpzGaussians::read_from = decompress + read_decompressed,
// write_to = write_decompressed + compress (spz.rs:945-959)
extern "C" gs_status gs_spz_decompress(const void *bytes, size_t len, void *out, size_t capacity, size_t *bytes_out) {
    if (!bytes || !bytes_out) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    std::vector<uint8_t> raw;
    gs_status rc = gunzip(bytes, len, raw);
    if (rc != GS_OK) return rc;
    return copy_out(raw, out, capacity, bytes_out);
}

extern "C" gs_status gs_spz_compress(const void *bytes, size_t len, void *out, size_t capacity, size_t *bytes_out) {
    if ((len && !bytes) || !bytes_out) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    std::vector<uint8_t> z;
    gs_status rc = gzip_member(bytes, len, z);
    if (rc != GS_OK) return rc;
    return copy_out(z, out, capacity, bytes_out);
}

extern "C" gs_status gs_spz_decode(const void *bytes, size_t len, gs_spz_header *header_out, gs_gaussian *out,
                                   size_t capacity, size_t *count_out) {
    if (!bytes || !count_out) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    std::vector<uint8_t> raw;
    gs_status rc = gunzip(bytes, len, raw);
    if (rc != GS_OK) return rc;
    return gs_spz_decode_decompressed(raw.data(), raw.size(), header_out, out, capacity, count_out);
}

extern "C" gs_status gs_spz_encode(const gs_gaussian *in, size_t n, const gs_spz_options *opt, void *out,
                                   size_t capacity, size_t *bytes_out) {
    if (!bytes_out) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    size_t raw_size = 0;
    gs_status rc = gs_spz_encode_decompressed(in, n, opt, nullptr, 0, &raw_size);
    if (rc != GS_OK) return rc;
    std::vector<uint8_t> raw;
    try {
        raw.resize(raw_size);
    } catch (const std::bad_alloc &) {
        return gs_fail(GS_ERR_OUT_OF_MEMORY, raw_size, 0, 0, "out of memory encoding SPZ data");
    }
    rc = gs_spz_encode_decompressed(in, n, opt, raw.data(), raw.size(), &raw_size);
    if (rc != GS_OK) return rc;
    std::vector<uint8_t> z;
    rc = gzip_member(raw.data(), raw.size(), z);
    if (rc != GS_OK) return rc;
    return copy_out(z, out, capacity, bytes_out);
}

// ---- Gaussians / GaussiansSource (gaussian.rs:394-548) over the two codecs -----------------------
extern "C" gs_status gs_gaussians_read(const void *bytes, size_t len, gs_gaussians_source source, gs_gaussian *out,
                                       size_t capacity, size_t *count_out) {
    if (!bytes || !count_out) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    switch (source) {
    case GS_SOURCE_PLY: {
        size_t n = 0;
        gs_status rc = gs_ply_read(bytes, len, nullptr, 0, &n, nullptr);
        if (rc != GS_OK) return rc;
        *count_out = n;
        if (!out) return GS_OK;
        const size_t m = n < capacity ? n : capacity;
        std::vector<gs_ply_gaussian_pod> pods;
        try {
            pods.resize(m);
        } catch (const std::bad_alloc &) {
            return gs_fail(GS_ERR_OUT_OF_MEMORY, m * sizeof(gs_ply_gaussian_pod), 0, 0, "out of memory reading PLY data");
        }
        rc = gs_ply_read(bytes, len, pods.data(), m, &n, nullptr);
        if (rc != GS_OK) return rc;
        gs_gaussian_from_ply(pods.data(), m, out);
        return GS_OK;
    }
    case GS_SOURCE_SPZ: return gs_spz_decode(bytes, len, nullptr, out, capacity, count_out);
    default: return gs_fail(GS_ERR_INVALID_ARGUMENT, (uint64_t)source, 0, 0, "cannot read Internal Gaussians from buffer");
    }
}

extern "C" gs_status gs_gaussians_write(const gs_gaussian *in, size_t n, gs_gaussians_source source, void *out,
                                        size_t capacity, size_t *bytes_out) {
    if ((n && !in) || !bytes_out) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    switch (source) {
    case GS_SOURCE_PLY: {
        std::vector<gs_ply_gaussian_pod> pods;
        try {
            pods.resize(n);
        } catch (const std::bad_alloc &) {
            return gs_fail(GS_ERR_OUT_OF_MEMORY, n * sizeof(gs_ply_gaussian_pod), 0, 0, "out of memory writing PLY data");
        }
        gs_gaussian_to_ply(in, n, pods.data());
        return gs_ply_write(pods.data(), n, out, capacity, bytes_out);
    }
    case GS_SOURCE_SPZ: {
        gs_spz_options o;
        gs_spz_options_default(&o);
        return gs_spz_encode(in, n, &o, out, capacity, bytes_out);
    }
    default: return gs_fail(GS_ERR_INVALID_ARGUMENT, (uint64_t)source, 0, 0, "cannot write Internal Gaussians to buffer");
    }
}
