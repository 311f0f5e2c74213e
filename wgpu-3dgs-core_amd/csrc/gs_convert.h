// gs_convert.h — source-format record -> struct Gaussian conversions as ONE set of host + device
// functions: Gaussian::from_ply (src/gaussian.rs:70-92).  The host entry point
// (gs_gaussian_from_ply, gs_ply.cpp) and the device kernel (k_from_ply_pods, gs3d.hip) both call
// ply_to_gaussian_words below, so a scene loaded on the device is bit-identical to one converted on
// the host by construction.
//
// exp: the reference calls f32::exp, which on Linux is glibc's expf.  A device `expf` (ocml) is a
// different algorithm and differs from it in the last bit, so BOTH sides use gs_expf below: the
// published algorithm glibc (>= 2.27), musl, newlib and bionic share — Szabolcs Nagy's expf from
// ARM optimized-routines (math/expf.c, math/exp2f_data.c; EXP2F_TABLE_BITS = 5): the argument is
// reduced in DOUBLE to k/32 + r, 2^(k/32) comes from a 32-entry table, a cubic in r finishes, and the
// double result is rounded to binary32 once (worst-case error 0.502 ulp).  Every operation is an IEEE
// double +, *, or a conversion, written without contraction, so host and device agree bit for bit,
// and both agree with libm wherever the double result is not within ~1e-9 ulp of a binary32 rounding
// boundary (tests/test_ply.py: 0 differences in 4 M samples; reference tolerance 1e-4,
// tests/common/assert.rs:4).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIP__) || defined(__HIPCC_RTC__)
#define GS_HD __host__ __device__
#else
#define GS_HD
#endif

namespace gs {

GS_HD inline uint32_t cv_f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
GS_HD inline float cv_u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
GS_HD inline uint64_t cv_d2u(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    return u;
}
GS_HD inline double cv_u2d(uint64_t u) {
    double d;
    memcpy(&d, &u, 8);
    return d;
}

// IEEE 754 leaves the sign and payload of a NaN that ARITHMETIC produces to the implementation: x86
// returns the negative quiet NaN (0xffc00000) for 0/0 and propagates operand payloads, gfx950 returns
// +qNaN (0x7fc00000).  Values computed from degenerate input (a zero or infinite quaternion) are
// therefore canonicalised to +qNaN wherever host and device must agree bit for bit.
GS_HD inline uint32_t cv_canon_nan_bits(float f) { return f != f ? 0x7fc00000u : cv_f2u(f); }

// T[i] = bits(2^(i/32)) - (i << 47), 2^(i/32) correctly rounded to binary64
GS_HD inline uint64_t expf_table(uint32_t i) {
    const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    return T[i & 31u];
}

// expf for every binary32 input (NaN -> NaN, +inf -> +inf, -inf -> 0, overflow -> +inf,
// underflow -> 0 / subnormals rounded once)
GS_HD inline float gs_expf(float x) {
    const uint32_t ux = cv_f2u(x);
    const uint32_t abstop = (ux >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {                         // |x| >= 88 or NaN / inf
        if (ux == 0xff800000u) return 0.0f;         // -inf
        if (abstop >= 0x7f8u) return x + x;         // NaN, +inf
        if (x > 88.72283172607421875f) return cv_u2f(0x7f800000u);   // x > log(2^128)
        if (x < -103.972076416015625f) return 0.0f;                  // x < log(2^-150)
    }
    const double N = 32.0;
    const double inv_ln2_n = 0x1.71547652b82fep+0 * N;
    const double shift = 0x1.8p+52;
    const double c0 = 0x1.c6af84b912394p-5 / N / N / N, c1 = 0x1.ebfce50fac4f3p-3 / N / N, c2 = 0x1.62e42ff0c52d6p-1 / N;
    const double z = inv_ln2_n * (double)x;
    double kd = z + shift;                          // round to nearest integer in the low bits
    const uint64_t ki = cv_d2u(kd);
    kd = kd - shift;
    const double r = z - kd;
    const uint64_t t = expf_table((uint32_t)ki) + (ki << 47);
    const double s = cv_u2d(t);
    const double p = c0 * r + c1;
    const double r2 = r * r;
    double y = c2 * r + 1.0;
    y = p * r2 + y;
    y = y * s;
    return (float)y;
}

// Rust `as u8` of an f32: truncating, saturating, NaN -> 0
GS_HD inline uint32_t cv_sat_u8(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 255.0f) return 255u;
    return (uint32_t)(int)v;
}
GS_HD inline float cv_clamp_0_255(float v) {       // glam Vec4::clamp = max(lo).min(hi): NaN -> lo (sat_u8 maps it to 0 anyway)
    float a = v > 0.0f ? v : 0.0f;
    if (v != v) a = 0.0f;
    return a < 255.0f ? a : 255.0f;
}

// struct Gaussian word offsets (include/gs3d.h gs_gaussian): rot xyzw @0, pos @4, color u8x4 @7,
// sh[45] @8, scale @53; PlyGaussianPod (62 f32): pos @0, normal @3, f_dc @6, f_rest @9, opacity @54,
// scale @55, rot wxyz @58
constexpr int PLY_WORDS = 62;
constexpr int CV_GAUSSIAN_WORDS = 56;

// Gaussian::from_ply (src/gaussian.rs:70-92): p = 62 words of one PlyGaussianPod, g = 56 words out.
// `sqrtf_cr` must be a correctly rounded binary32 square root (host sqrtf; device sqrtf, which hipcc rounds correctly by default — NOT __fsqrt_rn, the native approximation).
template <class Sqrt>
GS_HD inline void ply_to_gaussian_words(const uint32_t *p, uint32_t *g, Sqrt sqrtf_cr) {
    g[4] = p[0];
    g[5] = p[1];
    g[6] = p[2];
    // Quat::from_xyzw(rot[1], rot[2], rot[3], rot[0]).normalize()
    const float q[4] = {cv_u2f(p[59]), cv_u2f(p[60]), cv_u2f(p[61]), cv_u2f(p[58])};
    const float len = sqrtf_cr(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    for (int k = 0; k < 4; k++) g[k] = cv_canon_nan_bits(q[k] / len);
    for (int k = 0; k < 3; k++) g[53 + k] = cv_canon_nan_bits(gs_expf(cv_u2f(p[55 + k])));
    uint32_t color = 0;
    for (int k = 0; k < 3; k++) {
        const float v = (cv_u2f(p[6 + k]) * 0.2820948f + 0.5f) * 255.0f;        // SH0_TO_LINEAR_FACTOR
        color |= cv_sat_u8(cv_clamp_0_255(v)) << (8 * k);
    }
    const float a = (1.0f / (1.0f + gs_expf(-cv_u2f(p[54])))) * 255.0f;
    color |= cv_sat_u8(cv_clamp_0_255(a)) << 24;
    g[7] = color;
    for (int k = 0; k < 15; k++) {       // channel-planar f_rest -> RGB-interleaved
        g[8 + 3 * k + 0] = p[9 + k];
        g[8 + 3 * k + 1] = p[9 + k + 15];
        g[8 + 3 * k + 2] = p[9 + k + 30];
    }
}

// ---- Gaussian::from_spz (src/gaussian.rs:134-229) over the decompressed columns (spz.rs:739-771) ----

// where the columns of a decompressed SPZ payload start (positions -> alphas -> colors -> scales ->
// rotations -> sh) and how to read them; pointers may be host or device memory
struct SpzView {
    const uint8_t *pos, *alpha, *color, *scale, *rot, *sh;
    uint32_t version;           // 1..3: f16 vs 24-bit fixed positions; first-three vs smallest-three quaternions
    uint32_t fractional_bits;
    uint32_t ncoef;             // SH coefficients per channel: 0, 3, 8, 15
};

GS_HD inline float cv_f16_to_f32(uint32_t h) {
    const uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) return cv_u2f(cv_f2u((float)m * 5.9604644775390625e-08f) | sign);      // m * 2^-24, exact
    if (e == 31) return cv_u2f(sign | 0x7f800000u | (m << 13));
    return cv_u2f(sign | ((e + 112u) << 23) | (m << 13));
}

// SPZ_COLOR_TO_LINEAR_FRAC_A_B = 0.2820948 / 0.15, SPZ_COLOR_TO_LINEAR_C = (1 - A_B) * (0.5 * 255) — gaussian.rs:127-131
GS_HD inline float spz_color_ab() { return 0.2820948f / 0.15f; }
GS_HD inline float spz_color_c() { return (1.0f - spz_color_ab()) * (0.5f * 255.0f); }

template <class Sqrt>
GS_HD inline void spz_to_gaussian_words(const SpzView &v, size_t i, uint32_t *g, Sqrt sqrtf_cr) {
    if (v.version == 1) {
        for (int c = 0; c < 3; c++) {
            const uint8_t *p = v.pos + 6 * i + 2 * c;
            g[4 + c] = cv_f2u(cv_f16_to_f32((uint32_t)p[0] | ((uint32_t)p[1] << 8)));
        }
    } else {
        // `1 << fractional_bits` on i32 as Rust evaluates it in release builds (shift count masked to 5
        // bits); the header byte is untrusted, so the plain C shift would be undefined behaviour
        const float s = 1.0f / (float)(int32_t)(1u << (v.fractional_bits & 31u));
        for (int c = 0; c < 3; c++) {
            const uint8_t *p = v.pos + 9 * i + 3 * c;
            int32_t fixed = (int32_t)p[0] | ((int32_t)p[1] << 8) | ((int32_t)p[2] << 16);
            if (fixed & 0x800000) fixed |= (int32_t)0xff000000u;
            g[4 + c] = cv_f2u((float)fixed * s);
        }
    }
    for (int c = 0; c < 3; c++) g[53 + c] = cv_f2u(gs_expf((float)v.scale[3 * i + c] / 16.0f - 10.0f));
    if (v.version < 3) {
        const float x = (float)v.rot[3 * i] / 127.5f - 1.0f, y = (float)v.rot[3 * i + 1] / 127.5f - 1.0f,
                    z = (float)v.rot[3 * i + 2] / 127.5f - 1.0f;
        const float l2 = (x * x + y * y) + z * z;
        const float w2 = 1.0f - l2;
        g[0] = cv_f2u(x);
        g[1] = cv_f2u(y);
        g[2] = cv_f2u(z);
        g[3] = cv_f2u(sqrtf_cr(w2 > 0.0f ? w2 : 0.0f));        // f32::max(.., 0.0): NaN -> 0
    } else {
        const uint8_t *q = v.rot + 4 * i;
        uint32_t comp = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        const uint32_t mask = (1u << 9) - 1u;
        const uint32_t largest = comp >> 30;
        float sum = 0.0f, comps[4];
        for (uint32_t k = 0; k < 4; k++) {   // ascending, as the reference's array::from_fn
            if (k == largest) {
                comps[k] = 0.0f;
                continue;
            }
            const uint32_t mag = comp & mask, neg = (comp >> 9) & 1u;
            comp >>= 10;
            const float val = 0.70710678118654752440f * ((float)mag / (float)mask) * (neg ? -1.0f : 1.0f);
            sum += val * val;
            comps[k] = val;
        }
        const float w2 = 1.0f - sum;
        for (uint32_t k = 0; k < 4; k++) g[k] = cv_f2u(k == largest ? sqrtf_cr(w2 > 0.0f ? w2 : 0.0f) : comps[k]);
    }
    uint32_t color = 0;
    for (int c = 0; c < 3; c++) {
        const float val = (float)v.color[3 * i + c] * spz_color_ab() + spz_color_c();
        color |= cv_sat_u8(cv_clamp_0_255(val)) << (8 * c);
    }
    color |= (uint32_t)v.alpha[i] << 24;
    g[7] = color;
    for (int k = 0; k < 45; k++) g[8 + k] = 0u;
    for (uint32_t k = 0; k < v.ncoef; k++)
        for (int c = 0; c < 3; c++) g[8 + 3 * k + c] = cv_f2u(((float)v.sh[(i * v.ncoef + k) * 3 + c] - 128.0f) / 128.0f);
}

}  // namespace gs
