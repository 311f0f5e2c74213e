// gs_kernel_lib.h — the device-function library (gfx950) that replaces the reference's WESL
// shader package `wgpu_3dgs_core` (src/shader.rs:8-56):
//   module gaussian            src/shader/gaussian.wesl
//   module gaussian_transform  src/shader/gaussian_transform.wesl
//   module model_transform     src/shader/model_transform.wesl
// WESL feature flags (sh_single / sh_half / sh_norm8 / sh_none, cov3d_rot_scale / cov3d_single /
// cov3d_half) become the template parameters SH and COV.
//
// A Gaussian is addressed as its POD words `w` (uint32_t[pod_words]): either a pointer into the
// AoS storage buffer or a register array filled from the chunk-planar mirror; with constant
// indices after unrolling both compile to direct register / immediate-offset accesses.
//
// Numerics: plain IEEE binary32 in the written order.  The translation unit is compiled with
// -ffp-contract=off, so nothing below fuses unless __builtin_fmaf is written.
#pragma once

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else   // compiled at run time by hiprtc (gs_bundle_create_from_source): no system headers
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
#endif

namespace gs {

enum : int { SH_SINGLE = 0, SH_HALF = 1, SH_NORM8 = 2, SH_NONE = 3 };
enum : int { COV_ROT_SCALE = 0, COV_SINGLE = 1, COV_HALF = 2 };

// ---- layouts: src/buffer/gaussian.rs:301-384, src/gaussian_config.rs:37,54,90,127,171,193,224
__host__ __device__ constexpr int sh_bytes(int sh) {
    return sh == SH_SINGLE ? 180 : sh == SH_HALF ? 92 : sh == SH_NORM8 ? 48 : 0;
}
__host__ __device__ constexpr int cov_bytes(int cov) {
    return cov == COV_ROT_SCALE ? 28 : cov == COV_SINGLE ? 24 : 12;
}
__host__ __device__ constexpr int pod_bytes(int sh, int cov) {
    return (16 + sh_bytes(sh) + cov_bytes(cov) + 15) / 16 * 16;
}
__host__ __device__ constexpr int pod_words(int sh, int cov) { return pod_bytes(sh, cov) / 4; }
__host__ __device__ constexpr int sh_word0(int) { return 4; }
__host__ __device__ constexpr int cov_word0(int sh) { return 4 + sh_bytes(sh) / 4; }

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

// ---- WGSL built-ins
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// unpack2x16float component: binary16 -> binary32, exact (v_cvt_f32_f16)
__device__ __forceinline__ float half_lo(uint32_t w) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
}
__device__ __forceinline__ float half_hi(uint32_t w) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
}
// unpack4x8snorm component: max(i8 / 127, -1)
__device__ __forceinline__ float snorm8(uint32_t w, int j) {
    int b = (int)(int8_t)((w >> (8 * j)) & 0xffu);
    return fmaxf((float)b / 127.0f, -1.0f);
}
// unpack4x8unorm component: u8 / 255
__device__ __forceinline__ float unorm8(uint32_t w, int j) {
    return (float)((w >> (8 * j)) & 0xffu) / 255.0f;
}

// ---- module gaussian -----------------------------------------------------------------------

// gaussian.wesl:24-26
__device__ __forceinline__ vec4 gaussian_unpack_color(const uint32_t *w) {
    uint32_t c = w[3];
    return {unorm8(c, 0), unorm8(c, 1), unorm8(c, 2), unorm8(c, 3)};
}

// gaussian.wesl:29-77
template <int SH>
__device__ __forceinline__ vec3 gaussian_unpack_sh(const uint32_t *w, uint32_t sh_index) {
    const uint32_t *s = w + sh_word0(SH);
    if constexpr (SH == SH_SINGLE) {
        return {u2f(s[sh_index * 3]), u2f(s[sh_index * 3 + 1]), u2f(s[sh_index * 3 + 2])};
    } else if constexpr (SH == SH_HALF) {
        uint32_t i = sh_index * 3;
        uint32_t xi = i / 2, yi = (i + 1) / 2, zi = (i + 2) / 2;
        if (xi == yi) {
            return {half_lo(s[xi]), half_hi(s[xi]), half_lo(s[zi])};
        } else {
            return {half_hi(s[xi]), half_lo(s[yi]), half_hi(s[yi])};
        }
    } else if constexpr (SH == SH_NORM8) {
        uint32_t i = sh_index * 3;
        return {snorm8(s[i / 4], i % 4), snorm8(s[(i + 1) / 4], (i + 1) % 4),
                snorm8(s[(i + 2) / 4], (i + 2) % 4)};
    } else {
        return {0.0f, 0.0f, 0.0f};
    }
}

// gaussian.wesl:80-149; returns (S00, S01, S02, S11, S12, S22); products summed ((k0+k1)+k2)
template <int SH, int COV>
__device__ __forceinline__ void gaussian_unpack_cov3d(const uint32_t *w, float out[6]) {
    const uint32_t *c = w + cov_word0(SH);
    if constexpr (COV == COV_ROT_SCALE) {
        float rx = u2f(c[0]), ry = u2f(c[1]), rz = u2f(c[2]), rw = u2f(c[3]);
        float sx = u2f(c[4]), sy = u2f(c[5]), sz = u2f(c[6]);
        float x2 = rx + rx, y2 = ry + ry, z2 = rz + rz;
        float xx = rx * x2, xy = rx * y2, xz = rx * z2;
        float yy = ry * y2, yz = ry * z2, zz = rz * z2;
        float wx = rw * x2, wy = rw * y2, wz = rw * z2;
        float m0[3] = {(1.0f - (yy + zz)) * sx, (xy + wz) * sx, (xz - wy) * sx};
        float m1[3] = {(xy - wz) * sy, (1.0f - (xx + zz)) * sy, (yz + wx) * sy};
        float m2[3] = {(xz + wy) * sz, (yz - wx) * sz, (1.0f - (xx + yy)) * sz};
#define GS_SIG(cc, rr) ((m0[rr] * m0[cc] + m1[rr] * m1[cc]) + m2[rr] * m2[cc])
        out[0] = GS_SIG(0, 0);
        out[1] = GS_SIG(0, 1);
        out[2] = GS_SIG(0, 2);
        out[3] = GS_SIG(1, 1);
        out[4] = GS_SIG(1, 2);
        out[5] = GS_SIG(2, 2);
#undef GS_SIG
    } else if constexpr (COV == COV_SINGLE) {
#pragma unroll
        for (int k = 0; k < 6; k++) out[k] = u2f(c[k]);
    } else {
        out[0] = half_lo(c[0]);
        out[1] = half_hi(c[0]);
        out[2] = half_lo(c[1]);
        out[3] = half_hi(c[1]);
        out[4] = half_lo(c[2]);
        out[5] = half_hi(c[2]);
    }
}

// ---- module gaussian_transform ---------------------------------------------------------------

struct GaussianTransform { float size; uint32_t flags; };  // gaussian_transform.wesl:4-7

constexpr uint32_t gaussian_display_mode_splat = 0u;
constexpr uint32_t gaussian_display_mode_ellipse = 1u;
constexpr uint32_t gaussian_display_mode_point = 2u;

// gaussian_transform.wesl:14-31 (unpack4xU8: byte i = component i)
__host__ __device__ __forceinline__ uint32_t gaussian_transform_display_mode(uint32_t flags) {
    return flags & 0xffu;
}
__host__ __device__ __forceinline__ uint32_t gaussian_transform_sh_deg(uint32_t flags) {
    return (flags >> 8) & 0xffu;
}
__host__ __device__ __forceinline__ bool gaussian_transform_no_sh0(uint32_t flags) {
    return ((flags >> 16) & 0xffu) != 0u;
}
__host__ __device__ __forceinline__ float gaussian_transform_max_std_dev(uint32_t flags) {
    return (float)((flags >> 24) & 0xffu) / 255.0f * 3.0f;
}

// ---- module model_transform --------------------------------------------------------------------

struct ModelTransform {  // model_transform.wesl:6-10 (uniform layout: 48 bytes)
    float pos[3];
    float _pad0;
    float rot[4];
    float scale[3];
    float _pad1;
};

struct QuatTerms { float xx, xy, xz, yy, yz, zz, wx, wy, wz; };

__host__ __device__ __forceinline__ QuatTerms quat_terms(const float r[4]) {
    float x2 = r[0] + r[0], y2 = r[1] + r[1], z2 = r[2] + r[2];
    QuatTerms t;
    t.xx = r[0] * x2; t.xy = r[0] * y2; t.xz = r[0] * z2;
    t.yy = r[1] * y2; t.yz = r[1] * z2; t.zz = r[2] * z2;
    t.wx = r[3] * x2; t.wy = r[3] * y2; t.wz = r[3] * z2;
    return t;
}

// model_transform.wesl:105-143 — column-major out[3*c + r]
__host__ __device__ __forceinline__ void model_scale_rot_mat(const ModelTransform &m, float out[9]) {
    QuatTerms t = quat_terms(m.rot);
    float sx = m.scale[0], sy = m.scale[1], sz = m.scale[2];
    out[0] = (1.0f - (t.yy + t.zz)) * sx;
    out[1] = (t.xy + t.wz) * sx;
    out[2] = (t.xz - t.wy) * sx;
    out[3] = (t.xy - t.wz) * sy;
    out[4] = (1.0f - (t.xx + t.zz)) * sy;
    out[5] = (t.yz + t.wx) * sy;
    out[6] = (t.xz + t.wy) * sz;
    out[7] = (t.yz - t.wx) * sz;
    out[8] = (1.0f - (t.xx + t.yy)) * sz;
}

// model_transform.wesl:64-102
__host__ __device__ __forceinline__ void model_transform_inv_sr_mat(const ModelTransform &m,
                                                                    float out[9]) {
    QuatTerms t = quat_terms(m.rot);
    float sx = m.scale[0], sy = m.scale[1], sz = m.scale[2];
    out[0] = (1.0f - (t.yy + t.zz)) / sx;
    out[1] = (t.xy - t.wz) / sy;
    out[2] = (t.xz + t.wy) / sz;
    out[3] = (t.xy + t.wz) / sx;
    out[4] = (1.0f - (t.xx + t.zz)) / sy;
    out[5] = (t.yz - t.wx) / sz;
    out[6] = (t.xz - t.wy) / sx;
    out[7] = (t.yz + t.wx) / sy;
    out[8] = (1.0f - (t.xx + t.yy)) / sz;
}

// model_transform.wesl:18-61 — column-major out[4*c + r]
__host__ __device__ __forceinline__ void model_transform_mat(const ModelTransform &m, float out[16]) {
    float sr[9];
    model_scale_rot_mat(m, sr);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        out[4 * c + 0] = sr[3 * c + 0];
        out[4 * c + 1] = sr[3 * c + 1];
        out[4 * c + 2] = sr[3 * c + 2];
        out[4 * c + 3] = 0.0f;
    }
    out[12] = m.pos[0];
    out[13] = m.pos[1];
    out[14] = m.pos[2];
    out[15] = 1.0f;
}

// mat4 * (p, 1), summed ((c0 + c1) + c2) + c3
__host__ __device__ __forceinline__ void mat4_mul_point(const float m[16], const float p[3],
                                                        float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++)
        out[r] = ((m[r] * p[0] + m[4 + r] * p[1]) + m[8 + r] * p[2]) + m[12 + r];
}

// model_transform.wesl:13-15
__host__ __device__ __forceinline__ void model_to_world(const ModelTransform &m, const float p[3],
                                                        float out[4]) {
    float mat[16];
    model_transform_mat(m, mat);
    mat4_mul_point(mat, p, out);
}

// ---- bind groups as seen by a kernel ---------------------------------------------------------------
// Every ComputeBundle kernel (built-in or compiled from source with gs_bundle_create_from_source)
// has the signature  __global__ void entry(gs::BundleArgs a, uint32_t count):
// the bind groups flattened group-major in binding order (buffer pointer + byte size).
constexpr int MAX_BINDINGS = 8;

struct BundleArgs {
    void *ptr[MAX_BINDINGS];
    uint64_t size[MAX_BINDINGS];
    uint32_t reg_second_group;
    uint32_t reg_has_constant;
    uint32_t reg_constant;
    uint32_t _pad;
};

}  // namespace gs
