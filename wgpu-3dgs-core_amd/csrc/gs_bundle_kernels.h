// gs_bundle_kernels.h — the kernels behind ComputeBundle's registry (gs_kernel_id).  Each one is
// the HIP equivalent of a compute entry point the reference's tests feed to ComputeBundleBuilder;
// they follow the reference's shader convention (src/compute_bundle.rs:27-40): one invocation per
// element, `if index >= n return`, any workgroup size from 1 to the device limit.
#pragma once

#include "gs_kernel_lib.h"

namespace gs {

// tests/common/shader/array_map_add.wesl:1-28
__global__ void k_array_map_add(BundleArgs a, uint32_t count) {
    (void)count;
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t *data = (uint32_t *)a.ptr[0];
    if (index >= (uint32_t)(a.size[0] / 4u)) return;
    uint32_t v = data[index] + 1u;
    if (a.reg_second_group) v += *(const uint32_t *)a.ptr[1];
    if (a.reg_has_constant) v += a.reg_constant;
    data[index] = v;
}

// tests/shader/gaussian.rs:26-59 — Output { color: vec4, sh: array<f32,45>, cov3d: array<f32,6> }
template <int SH, int COV>
__global__ void k_test_gaussian(BundleArgs a, uint32_t count) {
    (void)count;
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= 1u) return;
    const uint32_t *w = (const uint32_t *)a.ptr[0] + (uint64_t)index * pod_words(SH, COV);
    float *out = (float *)a.ptr[1];
    vec4 c = gaussian_unpack_color(w);
    out[0] = c.x;
    out[1] = c.y;
    out[2] = c.z;
    out[3] = c.w;
    for (uint32_t i = 0; i < 15u; i++) {
        vec3 s = gaussian_unpack_sh<SH>(w, i);
        out[4 + i * 3 + 0] = s.x;
        out[4 + i * 3 + 1] = s.y;
        out[4 + i * 3 + 2] = s.z;
    }
    float cov[6];
    gaussian_unpack_cov3d<SH, COV>(w, cov);
    for (int k = 0; k < 6; k++) out[49 + k] = cov[k];
    out[55] = 0.0f;
}

// tests/shader/gaussian_transform.rs:13-48 — Output { display_mode, sh_deg, no_sh0: u32, max_std_dev: f32 }
__global__ void k_test_gaussian_transform(BundleArgs a, uint32_t count) {
    (void)count;
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= 1u) return;
    const GaussianTransform *t = (const GaussianTransform *)a.ptr[0];
    uint32_t *out = (uint32_t *)a.ptr[1];
    out[0] = gaussian_transform_display_mode(t->flags);
    out[1] = gaussian_transform_sh_deg(t->flags);
    out[2] = gaussian_transform_no_sh0(t->flags) ? 1u : 0u;
    out[3] = f2u(gaussian_transform_max_std_dev(t->flags));
}

// tests/shader/model_transform.rs:14-51 — Output { transformed_pos: vec4, transform_mat: mat4x4,
// inv_sr_mat: mat3x3, scale_rot_mat: mat3x3 } with WGSL storage layout (mat3x3 columns padded to
// vec4): 4 + 16 + 12 + 12 = 44 floats.
__global__ void k_test_model_transform(BundleArgs a, uint32_t count) {
    (void)count;
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= 1u) return;
    ModelTransform m = *(const ModelTransform *)a.ptr[0];
    const float *tp = (const float *)a.ptr[1];
    float *out = (float *)a.ptr[2];
    float p[3] = {tp[0], tp[1], tp[2]};
    float world[4], mat[16], inv[9], sr[9];
    model_to_world(m, p, world);
    model_transform_mat(m, mat);
    model_transform_inv_sr_mat(m, inv);
    model_scale_rot_mat(m, sr);
    for (int k = 0; k < 4; k++) out[k] = world[k];
    for (int k = 0; k < 16; k++) out[4 + k] = mat[k];
    for (int c = 0; c < 3; c++) {
        for (int r = 0; r < 3; r++) {
            out[20 + 4 * c + r] = inv[3 * c + r];
            out[32 + 4 * c + r] = sr[3 * c + r];
        }
        out[20 + 4 * c + 3] = 0.0f;
        out[32 + 4 * c + 3] = 0.0f;
    }
}

// AoS PODs -> SoA f32 planes: plane p of `n` floats, p = 0..3 colour, 4..48 sh, 49..54 cov3d.
// n = number of Gaussians in the bound buffer (arrayLength); `count` invocations requested.
template <int SH, int COV>
__global__ void k_unpack_soa(BundleArgs a, uint32_t count) {
    (void)count;
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = (uint32_t)(a.size[0] / (uint64_t)pod_bytes(SH, COV));
    if (index >= n) return;
    const uint32_t *w = (const uint32_t *)a.ptr[0] + (uint64_t)index * pod_words(SH, COV);
    float *out = (float *)a.ptr[1];
    vec4 c = gaussian_unpack_color(w);
    out[0ull * n + index] = c.x;
    out[1ull * n + index] = c.y;
    out[2ull * n + index] = c.z;
    out[3ull * n + index] = c.w;
    for (uint32_t i = 0; i < 15u; i++) {
        vec3 s = gaussian_unpack_sh<SH>(w, i);
        out[(uint64_t)(4 + i * 3 + 0) * n + index] = s.x;
        out[(uint64_t)(4 + i * 3 + 1) * n + index] = s.y;
        out[(uint64_t)(4 + i * 3 + 2) * n + index] = s.z;
    }
    float cov[6];
    gaussian_unpack_cov3d<SH, COV>(w, cov);
    for (int k = 0; k < 6; k++) out[(uint64_t)(49 + k) * n + index] = cov[k];
}

}  // namespace gs
