// gs_pack_kernels.h — G::from_gaussian for all 12 POD layouts (src/buffer/gaussian.rs:314-339,
// src/gaussian_config.rs:32-233), as ONE set of __host__ __device__ encoders used both by the host
// path (gs_pack) and by the device kernel (gs_pack_device / GaussiansBuffer::new(gaussians)): the
// two cannot drift apart, and the device result is bit-equal to the host's by construction (integer
// bit manipulation for f16 / snorm8, float arithmetic in one written order, no contraction).
#pragma once

#include "gs_convert.h"
#include "gs_kernel_lib.h"

namespace gs {

// struct Gaussian as the C ABI lays it out (include/gs3d.h gs_gaussian, 224 bytes = 56 words):
// rot xyzw @0, pos @4, color u8x4 @7, sh[45] @8, scale @53
constexpr int GAUSSIAN_WORDS = 56;
constexpr int GW_ROT = 0, GW_POS = 4, GW_COLOR = 7, GW_SH = 8, GW_SCALE = 53;

// IEEE binary32 -> binary16, round to nearest even (half 2.7.1 f16::from_f32; gaussian_config.rs:59,227)
__host__ __device__ inline uint16_t f32_to_f16_rtne_bits(uint32_t fbits) {
    const uint32_t f32infty = 255u << 23, f16max = (127u + 16u) << 23;
    const uint32_t magic = ((127u - 15u) + (23u - 10u) + 1u) << 23;
    const uint32_t sign = fbits & 0x80000000u;
    uint32_t u = fbits ^ sign;
    uint16_t o;
    if (u >= f16max) {
        o = (u > f32infty) ? (uint16_t)(0x7e00u | ((u >> 13) & 0x1ffu)) : (uint16_t)0x7c00u;
    } else if (u < (113u << 23)) {
        // subnormal / zero result: one float add against the magic constant does the rounding
        union { uint32_t u; float f; } a, m;
        a.u = u;
        m.u = magic;
        a.f += m.f;
        o = (uint16_t)(a.u - magic);
    } else {
        const uint32_t odd = (u >> 13) & 1u;
        u += ((uint32_t)(15 - 127) << 23) + 0xfffu;
        u += odd;
        o = (uint16_t)(u >> 13);
    }
    return (uint16_t)(o | (sign >> 16));
}

// GaussianShNorm8Config: (v * 127).clamp(-127, 127) as i8 — truncation toward zero, NaN -> 0
__host__ __device__ inline uint32_t sh_norm8_byte(float v) {
    float x = v * 127.0f;
    if (x != x) x = 0.0f;
    if (x < -127.0f) x = -127.0f;
    if (x > 127.0f) x = 127.0f;
    return (uint32_t)(uint8_t)(int8_t)(int)x;
}

// GaussianCov3dSingleConfig: (R S)(R S)^T with glam's Mat3::from_quat formulation, upper triangle
__host__ __device__ inline void cov3d_from_rot_scale(const float q[4], const float s[3], float out[6]) {
    ModelTransform mt{};
    for (int k = 0; k < 4; k++) mt.rot[k] = q[k];
    for (int k = 0; k < 3; k++) mt.scale[k] = s[k];
    float m[9];
    model_scale_rot_mat(mt, m);   // columns scaled by s
#define GS_PACK_SIG(r, c) ((m[0 + r] * m[0 + c] + m[3 + r] * m[3 + c]) + m[6 + r] * m[6 + c])
    out[0] = GS_PACK_SIG(0, 0);
    out[1] = GS_PACK_SIG(1, 0);
    out[2] = GS_PACK_SIG(2, 0);
    out[3] = GS_PACK_SIG(1, 1);
    out[4] = GS_PACK_SIG(2, 1);
    out[5] = GS_PACK_SIG(2, 2);
#undef GS_PACK_SIG
}

// One Gaussian (56 words) -> one POD (pod_words(sh, cov) words, padding zeroed).
__host__ __device__ inline void pack_words(int sh, int cov, const uint32_t *g, uint32_t *p) {
    const int nw = pod_words(sh, cov);
    for (int k = 0; k < nw; k++) p[k] = 0u;
    p[0] = g[GW_POS];
    p[1] = g[GW_POS + 1];
    p[2] = g[GW_POS + 2];
    p[3] = g[GW_COLOR];
    uint32_t *s = p + 4;
    if (sh == SH_SINGLE) {
        for (int k = 0; k < 45; k++) s[k] = g[GW_SH + k];
    } else if (sh == SH_HALF) {
        for (int k = 0; k < 45; k++) {
            const uint32_t h = f32_to_f16_rtne_bits(g[GW_SH + k]);
            s[k >> 1] |= h << (16 * (k & 1));
        }
    } else if (sh == SH_NORM8) {
        for (int k = 0; k < 45; k++) {
            union { uint32_t u; float f; } v;
            v.u = g[GW_SH + k];
            s[k >> 2] |= sh_norm8_byte(v.f) << (8 * (k & 3));
        }
    }
    uint32_t *c = p + cov_word0(sh);
    if (cov == COV_ROT_SCALE) {
        for (int k = 0; k < 4; k++) c[k] = g[GW_ROT + k];
        for (int k = 0; k < 3; k++) c[4 + k] = g[GW_SCALE + k];
    } else {
        union { uint32_t u; float f; } q[4], sc[3], o[6];
        for (int k = 0; k < 4; k++) q[k].u = g[GW_ROT + k];
        for (int k = 0; k < 3; k++) sc[k].u = g[GW_SCALE + k];
        float qf[4] = {q[0].f, q[1].f, q[2].f, q[3].f}, sf[3] = {sc[0].f, sc[1].f, sc[2].f}, c6[6];
        cov3d_from_rot_scale(qf, sf, c6);
        for (int k = 0; k < 6; k++) o[k].u = cv_canon_nan_bits(c6[k]);      // computed NaNs: one bit pattern on host and device
        if (cov == COV_SINGLE) {
            for (int k = 0; k < 6; k++) c[k] = o[k].u;
        } else {
            for (int k = 0; k < 6; k++) c[k >> 1] |= (uint32_t)f32_to_f16_rtne_bits(o[k].u) << (16 * (k & 1));
        }
    }
}

// Device pack: a workgroup converts PACK_GROUP Gaussians.  The 224-byte source records are read as
// one contiguous span into LDS (coalesced), each of the first PACK_GROUP threads encodes one record
// LDS -> LDS, and the PODs leave as one contiguous span again: both HBM sides stream, whatever the
// POD size.  The upload of a scene is then ONE host-to-device copy of the source records plus this
// kernel, instead of a host-side pack of every record.
constexpr uint32_t PACK_GROUP = 128;
template <int SH, int COV>
__global__ __launch_bounds__(256) void k_pack_pods(const uint32_t *__restrict__ gaussians, uint64_t count,
                                                   uint32_t *__restrict__ pods) {
    constexpr int NW = pod_words(SH, COV);
    __shared__ uint32_t s_in[PACK_GROUP * GAUSSIAN_WORDS];    // 28 KiB
    __shared__ uint32_t s_out[PACK_GROUP * NW];               // <= 28 KiB
    const uint64_t g0 = (uint64_t)blockIdx.x * PACK_GROUP;
    const uint32_t ng = (uint32_t)(count - g0 < PACK_GROUP ? count - g0 : PACK_GROUP);
    const uint32_t *src = gaussians + g0 * GAUSSIAN_WORDS;
    for (uint32_t q = threadIdx.x; q < ng * GAUSSIAN_WORDS; q += 256) s_in[q] = src[q];
    __syncthreads();
    if (threadIdx.x < ng) pack_words(SH, COV, s_in + threadIdx.x * GAUSSIAN_WORDS, s_out + threadIdx.x * NW);
    __syncthreads();
    uint32_t *dst = pods + g0 * NW;
    for (uint32_t q = threadIdx.x; q < ng * NW; q += 256) dst[q] = s_out[q];
}

// GaussianPod::into_gaussian (src/buffer/gaussian.rs:186-196, gaussian_config.rs:61-117): one POD ->
// one Gaussian (56 words); only for the lossless configurations (SH != None, Cov3dRotScale), which the
// callers check.  Shared by the host path (gs_unpack_to_gaussian) and the device kernel.
__host__ __device__ inline void unpack_words(int sh, const uint32_t *p, uint32_t *g) {
    g[GW_POS] = p[0];
    g[GW_POS + 1] = p[1];
    g[GW_POS + 2] = p[2];
    g[GW_COLOR] = p[3];
    const uint32_t *s = p + 4;
    for (int k = 0; k < 45; k++) {
        if (sh == SH_SINGLE) {
            g[GW_SH + k] = s[k];
        } else if (sh == SH_HALF) {
            g[GW_SH + k] = cv_f2u(cv_f16_to_f32((s[k >> 1] >> (16 * (k & 1))) & 0xffffu));
        } else {
            const float v = (float)(int8_t)((s[k >> 2] >> (8 * (k & 3))) & 0xffu) / 127.0f;
            g[GW_SH + k] = cv_f2u(v < -1.0f ? -1.0f : v);
        }
    }
    const uint32_t *c = p + cov_word0(sh);
    for (int k = 0; k < 4; k++) g[GW_ROT + k] = c[k];
    for (int k = 0; k < 3; k++) g[GW_SCALE + k] = c[4 + k];
}

// Device side of GaussiansBuffer::download::<Gaussian>: PODs -> struct Gaussian records, LDS-staged on
// both sides like k_pack_pods (COV is RotScale by construction).
template <int SH>
__global__ __launch_bounds__(256) void k_unpack_pods(const uint32_t *__restrict__ pods, uint64_t count,
                                                     uint32_t *__restrict__ gaussians) {
    constexpr int NW = pod_words(SH, COV_ROT_SCALE);
    __shared__ uint32_t s_in[PACK_GROUP * NW];
    __shared__ uint32_t s_out[PACK_GROUP * GAUSSIAN_WORDS];
    const uint64_t g0 = (uint64_t)blockIdx.x * PACK_GROUP;
    const uint32_t ng = (uint32_t)(count - g0 < PACK_GROUP ? count - g0 : PACK_GROUP);
    const uint32_t *src = pods + g0 * NW;
    for (uint32_t q = threadIdx.x; q < ng * NW; q += 256) s_in[q] = src[q];
    __syncthreads();
    if (threadIdx.x < ng) unpack_words(SH, s_in + threadIdx.x * NW, s_out + threadIdx.x * GAUSSIAN_WORDS);
    __syncthreads();
    uint32_t *dst = gaussians + g0 * GAUSSIAN_WORDS;
    for (uint32_t q = threadIdx.x; q < ng * GAUSSIAN_WORDS; q += 256) dst[q] = s_out[q];
}

// Device load path of a PLY scene: Gaussian::from_ply fused with G::from_gaussian.  Same structure as
// k_pack_pods (contiguous span in, LDS, one record per thread LDS -> LDS, contiguous span out); the
// per-record arithmetic is gs_convert.h's ply_to_gaussian_words, shared with the host path.
struct DeviceSqrt {
    __device__ float operator()(float v) const { return sqrtf(v); }   // correctly rounded (hipcc default); __fsqrt_rn is the native approximation
};
template <int SH, int COV>
__global__ __launch_bounds__(256) void k_from_ply_pods(const uint32_t *__restrict__ ply, uint64_t count,
                                                       uint32_t *__restrict__ pods) {
    constexpr int NW = pod_words(SH, COV);
    static_assert(CV_GAUSSIAN_WORDS == GAUSSIAN_WORDS, "one struct Gaussian layout");
    __shared__ uint32_t s_in[PACK_GROUP * PLY_WORDS];         // 31 KiB
    __shared__ uint32_t s_out[PACK_GROUP * NW];               // <= 28 KiB
    const uint64_t g0 = (uint64_t)blockIdx.x * PACK_GROUP;
    const uint32_t ng = (uint32_t)(count - g0 < PACK_GROUP ? count - g0 : PACK_GROUP);
    const uint32_t *src = ply + g0 * PLY_WORDS;
    for (uint32_t q = threadIdx.x; q < ng * PLY_WORDS; q += 256) s_in[q] = src[q];
    __syncthreads();
    if (threadIdx.x < ng) {
        uint32_t gw[GAUSSIAN_WORDS];
        ply_to_gaussian_words(s_in + threadIdx.x * PLY_WORDS, gw, DeviceSqrt());
        pack_words(SH, COV, gw, s_out + threadIdx.x * NW);
    }
    __syncthreads();
    uint32_t *dst = pods + g0 * NW;
    for (uint32_t q = threadIdx.x; q < ng * NW; q += 256) dst[q] = s_out[q];
}

// Device load path of an SPZ scene (after the host's inflate): Gaussian::from_spz over the
// decompressed columns fused with G::from_gaussian.  The columns are byte streams (9 + 1 + 3 + 3 + 4 +
// 3 ncoef bytes per Gaussian); a thread reads its own bytes — neighbouring threads read neighbouring
// bytes of every column, so the lines are used whole — and the PODs leave through LDS as one span.
template <int SH, int COV>
__global__ __launch_bounds__(256) void k_from_spz_pods(SpzView v, uint64_t count, uint32_t *__restrict__ pods) {
    constexpr int NW = pod_words(SH, COV);
    __shared__ uint32_t s_out[PACK_GROUP * NW];               // <= 28 KiB
    const uint64_t g0 = (uint64_t)blockIdx.x * PACK_GROUP;
    const uint32_t ng = (uint32_t)(count - g0 < PACK_GROUP ? count - g0 : PACK_GROUP);
    if (threadIdx.x < ng) {
        uint32_t gw[GAUSSIAN_WORDS];
        spz_to_gaussian_words(v, (size_t)(g0 + threadIdx.x), gw, DeviceSqrt());
        pack_words(SH, COV, gw, s_out + threadIdx.x * NW);
    }
    __syncthreads();
    uint32_t *dst = pods + g0 * NW;
    for (uint32_t q = threadIdx.x; q < ng * NW; q += 256) dst[q] = s_out[q];
}

}  // namespace gs
