// gs_render_kernels.h — hand-written gfx950 kernels of the render hot path:
//   repack      AoS PODs -> chunk-planar mirror (16-byte chunks, one plane per chunk index)
//   preprocess  unpack + model/view transform + SH evaluation + 3D->2D covariance projection
//               + cull + tile rect  (HBM-read bound: the roofline kernel)
//   scan        exclusive prefix of per-Gaussian tile counts (chunk sums fused into preprocess,
//               per-chunk scan fused into emit)
//   emit        64-bit (tile, depth) keys + u32 Gaussian index per overlapped tile
//   sort        device-wide stable LSD radix sort, 8-bit digits, wave64 ballot ranking
//   ranges      per-tile [start, end) from key boundaries
//   blend       one 256-thread workgroup per 16x16 tile, sorted splats staged through LDS,
//               wave64 ballot compaction of non-contributing splats, front-to-back alpha blend
// No MFMA anywhere: nothing here is a dense contraction.  wave = 64 lanes throughout.
#pragma once

#include "gs_kernel_lib.h"

namespace gs {

constexpr int WAVE = 64;
constexpr int PP_THREADS = 256;
constexpr int PP_ITEMS = 4;                       // Gaussians per thread in preprocess / emit
constexpr int PP_CHUNK = PP_THREADS * PP_ITEMS;   // scan chunk = one workgroup's Gaussians

// Per-frame constants, derived once on the host from the uniforms and passed by
// value as a kernel argument (scalar registers / constant cache).
struct FrameConsts {
    float M[16];    // model_transform_mat
    float V[16];    // camera view
    float ISR[9];   // model_transform_inv_sr_mat
    float WS[9];    // W * (R_m S_m), row-major [r][c]; W = view rotation with rows 1,2 negated
    float cam_pos[3];
    float fx, fy, cx, cy;
    float near_plane, far_plane;
    float size2, limx, limy, max_std_dev;
    float bg[3];
    uint32_t sh_deg, no_sh0;
    uint32_t width, height;
    uint32_t tiles_x, tiles_y;
    uint32_t band_ty0, band_ty1;   // already clamped to tiles_y
};

// ---------------------------------------------------------------------------------------------
// wave / block primitives
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t t = __shfl_up(v, d, WAVE);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

// Exclusive scan over a 256-thread block; `smem` holds >= 4 words; returns exclusive prefix and
// the block total.  Ends with the LDS reusable after the caller's next barrier.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *smem,
                                                             uint32_t &total) {
    uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63u) smem[wid] = inc;
    __syncthreads();
    uint32_t w0 = smem[0], w1 = smem[1], w2 = smem[2], w3 = smem[3];
    uint32_t wave_off = wid == 0 ? 0u : wid == 1 ? w0 : wid == 2 ? w0 + w1 : w0 + w1 + w2;
    total = w0 + w1 + w2 + w3;
    __syncthreads();
    return wave_off + inc - v;
}

// ---------------------------------------------------------------------------------------------
// repack: AoS (N x pod_bytes) -> chunk-planar (pod_bytes/16 planes of plane_stride x 16 B)
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_repack_planar(const uint4 *__restrict__ aos,
                                                       uint4 *__restrict__ planar, uint64_t first,
                                                       uint64_t count, uint32_t chunks,
                                                       uint64_t plane_stride) {
    uint64_t total = count * chunks;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
         q += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t i = first + q / chunks;
        uint32_t c = (uint32_t)(q % chunks);
        planar[(uint64_t)c * plane_stride + i] = aos[first * chunks + q];
    }
}

// ---------------------------------------------------------------------------------------------
// preprocess (rows x1, x2 of the hot-path table)
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    return fminf(fmaxf(v, lo), hi);
}

// Real SH basis, rest coefficients k = 0..14 (degrees 1..3); summation order per DESIGN.md §3.2.
template <int SH>
__device__ __forceinline__ void eval_sh(const uint32_t *w, uint32_t deg, bool no_sh0,
                                        const float d[3], float rgb[3]) {
    vec4 col = gaussian_unpack_color(w);
    float acc[3] = {no_sh0 ? 0.0f : col.x, no_sh0 ? 0.0f : col.y, no_sh0 ? 0.0f : col.z};
    if constexpr (SH != SH_NONE) {
        if (deg >= 1u) {
            float x = d[0], y = d[1], z = d[2];
            const float C1 = 0.4886025119029199f;
            vec3 s0 = gaussian_unpack_sh<SH>(w, 0), s1 = gaussian_unpack_sh<SH>(w, 1),
                 s2 = gaussian_unpack_sh<SH>(w, 2);
            float a0 = C1 * y, a1 = C1 * z, a2 = C1 * x;
            acc[0] = ((acc[0] - a0 * s0.x) + a1 * s1.x) - a2 * s2.x;
            acc[1] = ((acc[1] - a0 * s0.y) + a1 * s1.y) - a2 * s2.y;
            acc[2] = ((acc[2] - a0 * s0.z) + a1 * s1.z) - a2 * s2.z;
            if (deg >= 2u) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                float b0 = 1.0925484305920792f * xy;
                float b1 = -1.0925484305920792f * yz;
                float b2 = 0.31539156525252005f * ((2.0f * zz - xx) - yy);
                float b3 = -1.0925484305920792f * xz;
                float b4 = 0.5462742152960396f * (xx - yy);
                vec3 t0 = gaussian_unpack_sh<SH>(w, 3), t1 = gaussian_unpack_sh<SH>(w, 4),
                     t2 = gaussian_unpack_sh<SH>(w, 5), t3 = gaussian_unpack_sh<SH>(w, 6),
                     t4 = gaussian_unpack_sh<SH>(w, 7);
                acc[0] = ((((acc[0] + b0 * t0.x) + b1 * t1.x) + b2 * t2.x) + b3 * t3.x) + b4 * t4.x;
                acc[1] = ((((acc[1] + b0 * t0.y) + b1 * t1.y) + b2 * t2.y) + b3 * t3.y) + b4 * t4.y;
                acc[2] = ((((acc[2] + b0 * t0.z) + b1 * t1.z) + b2 * t2.z) + b3 * t3.z) + b4 * t4.z;
                if (deg >= 3u) {
                    float c0 = (-0.5900435899266435f * y) * (3.0f * xx - yy);
                    float c1 = (2.890611442640554f * xy) * z;
                    float c2 = (-0.4570457994644658f * y) * ((4.0f * zz - xx) - yy);
                    float c3 = (0.3731763325901154f * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
                    float c4 = (-0.4570457994644658f * x) * ((4.0f * zz - xx) - yy);
                    float c5 = (1.445305721320277f * z) * (xx - yy);
                    float c6 = (-0.5900435899266435f * x) * (xx - 3.0f * yy);
                    vec3 u0 = gaussian_unpack_sh<SH>(w, 8), u1 = gaussian_unpack_sh<SH>(w, 9),
                         u2 = gaussian_unpack_sh<SH>(w, 10), u3 = gaussian_unpack_sh<SH>(w, 11),
                         u4 = gaussian_unpack_sh<SH>(w, 12), u5 = gaussian_unpack_sh<SH>(w, 13),
                         u6 = gaussian_unpack_sh<SH>(w, 14);
                    acc[0] = ((((((acc[0] + c0 * u0.x) + c1 * u1.x) + c2 * u2.x) + c3 * u3.x) +
                               c4 * u4.x) + c5 * u5.x) + c6 * u6.x;
                    acc[1] = ((((((acc[1] + c0 * u0.y) + c1 * u1.y) + c2 * u2.y) + c3 * u3.y) +
                               c4 * u4.y) + c5 * u5.y) + c6 * u6.y;
                    acc[2] = ((((((acc[2] + c0 * u0.z) + c1 * u1.z) + c2 * u2.z) + c3 * u3.z) +
                               c4 * u4.z) + c5 * u5.z) + c6 * u6.z;
                }
            }
        }
    }
    rgb[0] = fmaxf(acc[0], 0.0f);
    rgb[1] = fmaxf(acc[1], 0.0f);
    rgb[2] = fmaxf(acc[2], 0.0f);
}

// One Gaussian: returns the number of tiles touched (0 = culled) and fills the 48-byte record.
template <int SH, int COV>
__device__ __forceinline__ uint32_t project_one(const uint32_t *w, const FrameConsts &fc,
                                                uint4 rec[3]) {
    float p[3] = {u2f(w[0]), u2f(w[1]), u2f(w[2])};
    float pw[4], t[4];
    mat4_mul_point(fc.M, p, pw);
    mat4_mul_point(fc.V, pw, t);
    float xv = t[0], yv = -t[1], zv = -t[2];
    if (!(zv > fc.near_plane) || !(zv < fc.far_plane)) return 0u;

    float S[6];
    gaussian_unpack_cov3d<SH, COV>(w, S);
    float txz = xv / zv, tyz = yv / zv;
    float xc = clampf(txz, -fc.limx, fc.limx) * zv;
    float yc = clampf(tyz, -fc.limy, fc.limy) * zv;
    float zz = zv * zv;
    float j00 = fc.fx / zv, j02 = -(fc.fx * xc) / zz;
    float j11 = fc.fy / zv, j12 = -(fc.fy * yc) / zz;
    float T0[3], T1[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        T0[c] = j00 * fc.WS[0 + c] + j02 * fc.WS[6 + c];
        T1[c] = j11 * fc.WS[3 + c] + j12 * fc.WS[6 + c];
    }
    float a0 = (S[0] * T0[0] + S[1] * T0[1]) + S[2] * T0[2];
    float a1 = (S[1] * T0[0] + S[3] * T0[1]) + S[4] * T0[2];
    float a2 = (S[2] * T0[0] + S[4] * T0[1]) + S[5] * T0[2];
    float b0 = (S[0] * T1[0] + S[1] * T1[1]) + S[2] * T1[2];
    float b1 = (S[1] * T1[0] + S[3] * T1[1]) + S[4] * T1[2];
    float b2 = (S[2] * T1[0] + S[4] * T1[1]) + S[5] * T1[2];
    float ca = fc.size2 * ((T0[0] * a0 + T0[1] * a1) + T0[2] * a2) + 0.3f;
    float cb = fc.size2 * ((T0[0] * b0 + T0[1] * b1) + T0[2] * b2);
    float cc = fc.size2 * ((T1[0] * b0 + T1[1] * b1) + T1[2] * b2) + 0.3f;
    float det = ca * cc - cb * cb;
    if (!(det > 0.0f)) return 0u;
    float inv = 1.0f / det;
    float mid = 0.5f * (ca + cc);
    float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float radius = ceilf(fc.max_std_dev * sqrtf(lam));
    if (!(radius > 0.0f)) return 0u;
    float mx = fc.fx * txz + fc.cx;
    float my = fc.fy * tyz + fc.cy;
    float lo_y = (float)fc.band_ty0, hi_y = (float)fc.band_ty1;
    float fx0 = clampf(floorf((mx - radius) * 0.0625f), 0.0f, (float)fc.tiles_x);
    float fx1 = clampf(floorf((mx + radius) * 0.0625f) + 1.0f, 0.0f, (float)fc.tiles_x);
    float fy0 = clampf(floorf((my - radius) * 0.0625f), lo_y, hi_y);
    float fy1 = clampf(floorf((my + radius) * 0.0625f) + 1.0f, lo_y, hi_y);
    if (!(fx1 > fx0) || !(fy1 > fy0)) return 0u;
    uint32_t tx0 = (uint32_t)fx0, tx1 = (uint32_t)fx1, ty0 = (uint32_t)fy0, ty1 = (uint32_t)fy1;

    float dw[3] = {pw[0] - fc.cam_pos[0], pw[1] - fc.cam_pos[1], pw[2] - fc.cam_pos[2]};
    float dl = sqrtf((dw[0] * dw[0] + dw[1] * dw[1]) + dw[2] * dw[2]);
    float dn[3] = {dw[0] / dl, dw[1] / dl, dw[2] / dl};
    float dm[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
        dm[r] = (fc.ISR[r] * dn[0] + fc.ISR[3 + r] * dn[1]) + fc.ISR[6 + r] * dn[2];
    float ml = sqrtf((dm[0] * dm[0] + dm[1] * dm[1]) + dm[2] * dm[2]);
    float d[3] = {dm[0] / ml, dm[1] / ml, dm[2] / ml};
    float rgb[3];
    eval_sh<SH>(w, fc.sh_deg, fc.no_sh0 != 0u, d, rgb);
    float opacity = unorm8(w[3], 3);

    rec[0] = make_uint4(f2u(mx), f2u(my), f2u(-0.5f * (cc * inv)), f2u(cb * inv));
    rec[1] = make_uint4(f2u(-0.5f * (ca * inv)), f2u(opacity), f2u(rgb[0]), f2u(rgb[1]));
    rec[2] = make_uint4(f2u(rgb[2]), f2u(zv), tx0 | (ty0 << 16), tx1 | (ty1 << 16));
    return (tx1 - tx0) * (ty1 - ty0);
}

// Grid: one workgroup per PP_CHUNK Gaussians.  Reads the chunk-planar mirror with one
// global_load_dwordx4 per (lane, chunk): a wave reads 1 KiB contiguous per instruction.
template <int SH, int COV>
__global__ __launch_bounds__(PP_THREADS) void k_preprocess(
    const uint4 *__restrict__ planar, uint64_t plane_stride, uint32_t n, FrameConsts fc,
    uint4 *__restrict__ proj, uint32_t *__restrict__ tiles, uint32_t *__restrict__ chunk_sums,
    uint32_t *__restrict__ visible_count) {
    constexpr int NW = pod_words(SH, COV);
    constexpr int NC = NW / 4;
    __shared__ uint32_t s_red[8];
    uint32_t base = blockIdx.x * PP_CHUNK;
    uint32_t local = 0, local_vis = 0;
#pragma unroll 1
    for (int k = 0; k < PP_ITEMS; k++) {
        uint32_t i = base + k * PP_THREADS + threadIdx.x;
        if (i < n) {
            uint32_t w[NW];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                uint4 v = planar[(uint64_t)c * plane_stride + i];
                w[4 * c + 0] = v.x;
                w[4 * c + 1] = v.y;
                w[4 * c + 2] = v.z;
                w[4 * c + 3] = v.w;
            }
            uint4 rec[3];
            uint32_t cnt = project_one<SH, COV>(w, fc, rec);
            tiles[i] = cnt;
            if (cnt) {
                uint4 *o = proj + (uint64_t)i * 3;
                o[0] = rec[0];
                o[1] = rec[1];
                o[2] = rec[2];
                local += cnt;
                local_vis += 1u;
            }
        }
    }
    // chunk sum (feeds the scan) and visible count
    local = wave_reduce_add(local);
    local_vis = wave_reduce_add(local_vis);
    uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    if (lane == 0) {
        s_red[wid] = local;
        s_red[4 + wid] = local_vis;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        chunk_sums[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        uint32_t v = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
        if (v) atomicAdd(visible_count, v);
    }
}

// ---------------------------------------------------------------------------------------------
// scan of chunk sums (single workgroup; <= ~50k chunks for 50 M Gaussians)
// ---------------------------------------------------------------------------------------------

// counters[0] = total (saturating to 0xffffffff on overflow is not needed: D < 2^32 is checked
// on the host against capacity)
__global__ __launch_bounds__(1024) void k_scan_chunks(const uint32_t *__restrict__ sums,
                                                      uint32_t *__restrict__ offsets,
                                                      uint32_t num, uint32_t *__restrict__ total_out) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < num; base += 1024u) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < num ? sums[i] : 0u;
        uint32_t inc = wave_inclusive_scan(v, lane);
        if (lane == 63u) s_wave[wid] = inc;
        __syncthreads();
        uint32_t wave_off = 0, tot = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            uint32_t x = s_wave[k];
            if (k < wid) wave_off += x;
            tot += x;
        }
        uint32_t carry = s_carry;
        if (i < num) offsets[i] = carry + wave_off + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = s_carry;
}

// ---------------------------------------------------------------------------------------------
// emit (row x3): per-chunk exclusive scan + (tile << 32 | depth bits, Gaussian index) pairs
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(PP_THREADS) void k_emit(const uint32_t *__restrict__ tiles,
                                                     const uint32_t *__restrict__ chunk_offsets,
                                                     const uint4 *__restrict__ proj, uint32_t n,
                                                     uint32_t tiles_x, uint64_t *__restrict__ keys,
                                                     uint32_t *__restrict__ idx, uint32_t capacity) {
    __shared__ uint32_t s_scan[4];
    uint32_t base = blockIdx.x * PP_CHUNK + threadIdx.x * PP_ITEMS;
    uint32_t cnt[PP_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < PP_ITEMS; k++) {
        uint32_t i = base + k;
        cnt[k] = i < n ? tiles[i] : 0u;
        sum += cnt[k];
    }
    uint32_t total;
    uint32_t off = chunk_offsets[blockIdx.x] + block_exclusive_scan_256(sum, s_scan, total);
#pragma unroll 1
    for (int k = 0; k < PP_ITEMS; k++) {
        if (cnt[k]) {
            uint32_t i = base + k;
            uint4 r2 = proj[(uint64_t)i * 3 + 2];
            uint32_t depth_bits = r2.y;
            uint32_t tx0 = r2.z & 0xffffu, ty0 = r2.z >> 16, tx1 = r2.w & 0xffffu, ty1 = r2.w >> 16;
            uint32_t o = off;
            for (uint32_t ty = ty0; ty < ty1; ty++)
                for (uint32_t tx = tx0; tx < tx1; tx++) {
                    if (o < capacity) {
                        keys[o] = ((uint64_t)(ty * tiles_x + tx) << 32) | depth_bits;
                        idx[o] = i;
                    }
                    o++;
                }
            off += cnt[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// radix sort (row x4): stable LSD, 8-bit digits.  Per pass: histogram -> row scan -> scatter.
// ---------------------------------------------------------------------------------------------

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 8;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;   // 2048 pairs per workgroup
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;

// ghist layout: [digit][block] (digit-major) so that the row scan reads contiguous memory.
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const uint64_t *__restrict__ keys,
                                                            uint32_t count, uint32_t shift,
                                                            uint32_t *__restrict__ ghist,
                                                            uint32_t num_blocks) {
    __shared__ uint32_t s_hist[RADIX];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * SORT_TILE;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        uint32_t i = base + k * SORT_THREADS + threadIdx.x;
        if (i < count) atomicAdd(&s_hist[(uint32_t)(keys[i] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    ghist[(uint64_t)threadIdx.x * num_blocks + blockIdx.x] = s_hist[threadIdx.x];
}

// One workgroup per digit: exclusive scan of its row (over blocks) in place; row total out.
__global__ __launch_bounds__(256) void k_sort_scan_rows(uint32_t *__restrict__ ghist,
                                                        uint32_t num_blocks,
                                                        uint32_t *__restrict__ digit_totals) {
    __shared__ uint32_t s_scan[4];
    uint32_t *row = ghist + (uint64_t)blockIdx.x * num_blocks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_blocks; base += 256u) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < num_blocks ? row[i] : 0u;
        uint32_t total;
        uint32_t ex = block_exclusive_scan_256(v, s_scan, total);
        if (i < num_blocks) row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) digit_totals[blockIdx.x] = carry;
}

// Stable scatter.  Element order inside a workgroup tile: wave w owns elements
// [w*512, (w+1)*512) of the tile, round k of the wave covers 64 consecutive elements, lane order
// inside a round; ranks are assigned in exactly that order, so equal digits keep their order.
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t count,
    uint32_t shift, const uint32_t *__restrict__ ghist, uint32_t num_blocks,
    const uint32_t *__restrict__ digit_totals) {
    __shared__ uint32_t s_wave_hist[4][RADIX];   // per-wave digit counters
    __shared__ uint32_t s_bin_start[RADIX];      // exclusive scan of block digit counts
    __shared__ uint32_t s_global[RADIX];         // global offset of this block's digit run
    __shared__ uint32_t s_scan[4];
    __shared__ uint64_t s_keys[SORT_TILE];
    __shared__ uint32_t s_vals[SORT_TILE];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
#pragma unroll
    for (int w = 0; w < 4; w++) s_wave_hist[w][tid] = 0;
    __syncthreads();

    const uint32_t tile_base = blockIdx.x * SORT_TILE;
    const uint32_t wave_base = tile_base + wid * (SORT_ITEMS * WAVE);
    uint64_t key[SORT_ITEMS];
    uint32_t val[SORT_ITEMS];
    uint32_t rank[SORT_ITEMS];
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        uint32_t i = wave_base + k * WAVE + lane;
        bool ok = i < count;
        key[k] = ok ? keys_in[i] : ~0ull;
        val[k] = ok ? vals_in[i] : 0u;
    }
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        uint32_t d = (uint32_t)(key[k] >> shift) & (RADIX - 1);
        // wave64 match-any on the digit: peers = lanes holding the same digit
        uint64_t peers = ~0ull;
#pragma unroll
        for (int b = 0; b < RADIX_BITS; b++) {
            uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t before = mbcnt(peers);             // same-digit lanes below me
        uint32_t old = s_wave_hist[wid][d];
        rank[k] = old + before;
        if (before == 0u) s_wave_hist[wid][d] = old + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // per-digit: offsets of each wave inside the digit run, block digit count
    uint32_t c0 = s_wave_hist[0][tid], c1 = s_wave_hist[1][tid], c2 = s_wave_hist[2][tid],
             c3 = s_wave_hist[3][tid];
    uint32_t digit_count = (c0 + c1) + (c2 + c3);
    uint32_t total;
    uint32_t bin_start = block_exclusive_scan_256(digit_count, s_scan, total);
    s_bin_start[tid] = bin_start;
    s_wave_hist[0][tid] = bin_start;
    s_wave_hist[1][tid] = bin_start + c0;
    s_wave_hist[2][tid] = bin_start + c0 + c1;
    s_wave_hist[3][tid] = bin_start + c0 + c1 + c2;
    // global base of digit `tid`: sum of totals of smaller digits + this block's row prefix
    {
        uint32_t tot = digit_totals[tid];
        uint32_t t2;
        uint32_t digit_base = block_exclusive_scan_256(tot, s_scan, t2);
        s_global[tid] = digit_base + ghist[(uint64_t)tid * num_blocks + blockIdx.x];
    }
    __syncthreads();

    // local reorder through LDS so that each digit run is written by consecutive lanes
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        uint32_t d = (uint32_t)(key[k] >> shift) & (RADIX - 1);
        uint32_t pos = s_wave_hist[wid][d] + rank[k];
        s_keys[pos] = key[k];
        s_vals[pos] = val[k];
    }
    __syncthreads();
    uint32_t valid = count - tile_base < (uint32_t)SORT_TILE ? count - tile_base : (uint32_t)SORT_TILE;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        uint32_t pos = k * SORT_THREADS + tid;
        if (pos < valid) {
            uint64_t kk = s_keys[pos];
            uint32_t d = (uint32_t)(kk >> shift) & (RADIX - 1);
            uint32_t dst = s_global[d] + (pos - s_bin_start[d]);
            keys_out[dst] = kk;
            vals_out[dst] = s_vals[pos];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tile ranges (row x5a)
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_tile_ranges(const uint64_t *__restrict__ keys,
                                                     uint32_t count, uint32_t *__restrict__ ranges) {
    uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= count) return;
    uint32_t tile = (uint32_t)(keys[j] >> 32);
    if (j == 0 || (uint32_t)(keys[j - 1] >> 32) != tile) ranges[2 * tile] = j;
    if (j + 1 == count || (uint32_t)(keys[j + 1] >> 32) != tile) ranges[2 * tile + 1] = j + 1;
}

// ---------------------------------------------------------------------------------------------
// blend (row x5b)
// ---------------------------------------------------------------------------------------------

// exp for x <= 0 exactly as DESIGN.md §3.6 defines it (explicit fma polynomial, no hardware
// transcendental, so the result is bit-reproducible on any IEEE machine): t = x*log2(e); n = rint(t); f = t - n; 2^f by a degree-5 polynomial.
__device__ __forceinline__ float gs_exp(float x) {
    float t = x * 1.44269504088896340736f;
    float n = rintf(t);
    float f = t - n;
    float p = 0x1.5f0896p-10f;
    p = __builtin_fmaf(p, f, 0x1.3cbf6cp-7f);
    p = __builtin_fmaf(p, f, 0x1.c6af6cp-5f);
    p = __builtin_fmaf(p, f, 0x1.ebfa4ap-3f);
    p = __builtin_fmaf(p, f, 0x1.62e430p-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}

// Maximum over t in [lo, hi] of the concave parabola q2*t^2 + q1*t + q0 (q2 < 0).
__device__ __forceinline__ float parabola_max(float q2, float q1, float q0, float lo, float hi) {
    float t = clampf(-0.5f * q1 / q2, lo, hi);
    return (q2 * t + q1) * t + q0;
}

// Conservative test "can this splat reach alpha >= 1/255 anywhere on the pixel-centre rectangle
// [rx0,rx1] x [ry0,ry1]?".  power(d) = ca*dx^2 + cc*dy^2 + cb*dx*dy is a concave quadratic in
// d = mean - pixel; its maximum over the rectangle is 0 when the mean lies inside, otherwise it is
// attained on one of the four edges (a clamped 1-D parabola each).  alpha >= 1/255 needs
// power >= ln(1/(255*opacity)) >= -5.5413 (opacity <= 1); the threshold -5.7 leaves > 0.15 of
// slack for rounding in this bound, so a dropped splat is one the pixel loop would skip at
// every pixel of the tile.
__device__ __forceinline__ bool splat_touches_rect(float mx, float my, float ca, float cb, float cc,
                                                   float rx0, float rx1, float ry0, float ry1) {
    float dx_lo = mx - rx1, dx_hi = mx - rx0, dy_lo = my - ry1, dy_hi = my - ry0;
    bool in_x = dx_lo <= 0.0f && dx_hi >= 0.0f, in_y = dy_lo <= 0.0f && dy_hi >= 0.0f;
    if (in_x && in_y) return true;
    float m0 = parabola_max(cc, cb * dx_lo, ca * dx_lo * dx_lo, dy_lo, dy_hi);
    float m1 = parabola_max(cc, cb * dx_hi, ca * dx_hi * dx_hi, dy_lo, dy_hi);
    float m2 = parabola_max(ca, cb * dy_lo, cc * dy_lo * dy_lo, dx_lo, dx_hi);
    float m3 = parabola_max(ca, cb * dy_hi, cc * dy_hi * dy_hi, dx_lo, dx_hi);
    float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    return !(m < -5.7f);
}

constexpr int BLEND_BATCH = 256;

// One workgroup = one 16x16 tile; wave w covers pixel rows 4w..4w+3.  The tile's sorted splat
// list is staged through LDS in batches of 256; while staging, each lane tests its splat against
// the tile rectangle (conservative bound of the 1/255 alpha iso-contour) and the batch is
// compacted with wave64 ballots + prefix counts, so the pixel loop only walks splats that can
// contribute.  The compaction never changes results: a removed splat has alpha < 1/255 at every
// pixel of the tile, which the pixel loop would skip anyway.
__global__ __launch_bounds__(256) void k_blend(const uint32_t *__restrict__ ranges,
                                               const uint32_t *__restrict__ idx,
                                               const uint4 *__restrict__ proj, FrameConsts fc,
                                               float4 *__restrict__ rgba) {
    __shared__ float4 s_a[BLEND_BATCH];   // mx, my, ca, cb
    __shared__ float4 s_b[BLEND_BATCH];   // cc, opacity, r, g
    __shared__ float s_c[BLEND_BATCH];    // b
    __shared__ uint32_t s_wave_cnt[4];
    __shared__ uint32_t s_done;

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t tile = fc.band_ty0 * fc.tiles_x + blockIdx.x;
    const uint32_t ty = tile / fc.tiles_x, tx = tile % fc.tiles_x;
    const uint32_t lx = tid & 15u, ly = tid >> 4;
    const uint32_t px = tx * 16u + lx, py = ty * 16u + ly;
    const bool inside = px < fc.width && py < fc.height;
    const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
    // tile rectangle in pixel-centre coordinates
    const float rx0 = (float)(tx * 16u) + 0.5f, rx1 = (float)(tx * 16u) + 15.5f;
    const float ry0 = (float)(ty * 16u) + 0.5f, ry1 = (float)(ty * 16u) + 15.5f;

    const uint32_t start = ranges[2 * tile], end = ranges[2 * tile + 1];
    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    bool done = !inside;

    for (uint32_t b0 = start; b0 < end; b0 += BLEND_BATCH) {
        // all pixels of the tile finished -> stop fetching
        if (tid == 0) s_done = 0;
        __syncthreads();
        if (!done) s_done = 1;   // benign race: any unfinished lane sets it
        __syncthreads();
        if (s_done == 0) break;

        // stage + cull + compact (order preserving)
        uint32_t j = b0 + tid;
        bool keep = false;
        uint4 r0, r1, r2;
        if (j < end) {
            const uint4 *rec = proj + (uint64_t)idx[j] * 3;
            r0 = rec[0];
            r1 = rec[1];
            r2 = rec[2];
            keep = splat_touches_rect(u2f(r0.x), u2f(r0.y), u2f(r0.z), u2f(r0.w), u2f(r1.x), rx0,
                                      rx1, ry0, ry1);
        }
        uint64_t mask = __ballot(keep);
        uint32_t before = mbcnt(mask);
        if (lane == 0) s_wave_cnt[wid] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t w0 = s_wave_cnt[0], w1 = s_wave_cnt[1], w2 = s_wave_cnt[2], w3 = s_wave_cnt[3];
        uint32_t wave_off = wid == 0 ? 0u : wid == 1 ? w0 : wid == 2 ? w0 + w1 : w0 + w1 + w2;
        uint32_t kept = w0 + w1 + w2 + w3;
        if (keep) {
            uint32_t pos = wave_off + before;
            s_a[pos] = make_float4(u2f(r0.x), u2f(r0.y), u2f(r0.z), u2f(r0.w));
            s_b[pos] = make_float4(u2f(r1.x), u2f(r1.y), u2f(r1.z), u2f(r1.w));
            s_c[pos] = u2f(r2.x);
        }
        __syncthreads();

        // pixel loop over the compacted batch (wave-uniform trip count, LDS broadcast reads)
        if (!__all(done)) {
            for (uint32_t s = 0; s < kept; s++) {
                float4 a = s_a[s];
                float dx = a.x - pxf, dy = a.y - pyf;
                float4 bq = s_b[s];
                float u = a.z * dx, v = bq.x * dy, wq = a.w * dx;
                float power = __builtin_fmaf(u, dx, __builtin_fmaf(v, dy, wq * dy));
                bool act = !done && !(power > 0.0f) && !(power < -5.6f);
                if (!__any(act)) continue;
                float alpha = fminf(0.99f, bq.y * gs_exp(fmaxf(power, -6.0f)));
                act = act && !(alpha < 1.0f / 255.0f);
                float test_T = T * (1.0f - alpha);
                if (act && test_T < 0.0001f) {
                    done = true;
                    act = false;
                }
                if (act) {
                    float wgt = alpha * T;
                    C0 = __builtin_fmaf(bq.z, wgt, C0);
                    C1 = __builtin_fmaf(bq.w, wgt, C1);
                    C2 = __builtin_fmaf(s_c[s], wgt, C2);
                    T = test_T;
                }
                if (__all(done)) break;
            }
        }
    }
    if (inside) {
        float4 o;
        o.x = __builtin_fmaf(T, fc.bg[0], C0);
        o.y = __builtin_fmaf(T, fc.bg[1], C1);
        o.z = __builtin_fmaf(T, fc.bg[2], C2);
        o.w = 1.0f - T;
        rgba[(uint64_t)py * fc.width + px] = o;
    }
}

}  // namespace gs
