// gs_render_kernels.h — hand-written gfx950 kernels of the render hot path:
//   repack      AoS PODs -> block-planar mirror (16-byte chunks, one plane per chunk index and block),
//               optionally in spatial (Morton) order
//   preprocess  unpack + model/view transform + SH evaluation + 3D->2D covariance projection
//               + cull + tile rect (HBM-read bound: the roofline kernel); writes dense per-slot
//               arrays: 36-byte blend record, depth key (all-ones = culled), tile rect
//   depth sort  stable LSD radix sort of the visible Gaussians on the bits of their view depth;
//               its first pass reads the dense keys and compacts (k_sort_* with COMPACT)
//   expand      k_expand_count: gather of the tile rects into depth order + per-chunk pair counts;
//               k_pairs_emit: the (tile id, Gaussian) pairs in depth order, cut by output slots,
//               with the histogram of the tile sort's first pass fused in
//   tile sort   stable LSD radix sort of the D pairs on the tile id alone (u16 keys up to 65536 tiles)
//   ranges      per-tile [start, end) from key boundaries
//   blend       one 128-thread workgroup per 16x16 tile (k_blend_grouped): sorted splats staged
//               through LDS, per-8x4-block index lists by wave64 ballot, front-to-back blend
// All counts that only the device knows (V, D) stay on the device: grids are sized from host-side
// upper bounds and surplus workgroups exit.  No MFMA anywhere: nothing here is a dense contraction.
// wave = 64 lanes throughout.
#pragma once

#include <type_traits>

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
    uint32_t mask_culled_records;  // k_preprocess_banded: culled lanes skip their 36-byte record store
    uint32_t nt_loads;             // preprocess reads the mirror with the non-temporal policy
    uint32_t clip_rect;            // display mode Splat: the tile rect is clipped to the splat's visible box
    float cull_gain;               // block culling: size^2 * |R_m S_m|_F^2 * (fx^2 (1+limx^2) + fy^2 (1+limy^2)); 0 = off
    float ellipse_pmin;            // display mode Ellipse: -max_std_dev^2 / 2
    uint32_t rect32;               // tile rects are stored packed in 4 bytes (at most 256 tiles along either axis, 32768 in all)
    uint32_t tile_masks;           // rect version 4: the exact tile test for rects of at most 3 x 3 tiles (needs clip_rect)
    uint32_t wt_stores;            // store16 mode of the frame's 16-byte-per-lane outputs (pairs, image): 1 = write-through
    uint32_t wt_pairs;             // ... of k_pairs_emit's pairs (host-side copy of the switch: the kernel takes it through ExpandIO)
    uint32_t wt_records;           // store16 mode of the preprocess kernel's blend records
};

// Tile rect of a visible Gaussian.  Since version 4 a rect of at most 3 x 3 tiles may have lost the
// tiles its splat cannot reach: `rows` = 0x8000 | 4 bits per tile row, (first kept column) | (kept columns) << 2.
// Two storage formats.  (1) uint2, any image: x = x0 | y0 << 16; y = x1 | y1 << 16, or — tiles dropped — (row code |
// (w - 1) << 12 | (h - 1) << 14) << 16 with the low half 0 (x1 is never 0).  (2) ONE word while the image has at
// most 256 tiles along either axis and 32768 in all (1080p, 4K): masked << 31 | origin << 16 | payload, origin = y0 *
// tiles_x + x0 (what the pair generator needs anyway), payload = (w - 1) | (h - 1) << 8, or row code | (w - 1) << 12 |
// (h - 1) << 14 — 4 bytes less per visible Gaussian written by preprocess, written again in depth order by
// k_expand_count and read by k_pairs_emit.  Culled slots are recognised by their depth key, not by their rect.
__host__ __device__ inline uint32_t rect_small_code(uint32_t r0, uint32_t r1, uint32_t rows) {
    const uint32_t w = (r1 & 0xffffu) - (r0 & 0xffffu), h = (r1 >> 16) - (r0 >> 16);
    return (rows & 0xfffu) | ((w - 1u) << 12) | ((h - 1u) << 14);
}
__host__ __device__ inline uint32_t rect_pack32(uint32_t r0, uint32_t r1, uint32_t rows, uint32_t tiles_x) {
    const uint32_t x0 = r0 & 0xffffu, y0 = r0 >> 16, x1 = r1 & 0xffffu, y1 = r1 >> 16;
    const uint32_t origin = y0 * tiles_x + x0;
    if (rows & 0x8000u) return 0x80000000u | (origin << 16) | rect_small_code(r0, r1, rows);
    return (origin << 16) | (x1 - x0 - 1u) | ((y1 - y0 - 1u) << 8);
}
__host__ __device__ inline uint2 rect_pack64(uint32_t r0, uint32_t r1, uint32_t rows) {
    uint2 r;
    r.x = r0;
    r.y = (rows & 0x8000u) ? rect_small_code(r0, r1, rows) << 16 : r1;
    return r;
}
// kept tiles of a row code (its low 12 bits)
__host__ __device__ inline uint32_t rect_rows_count(uint32_t code) {
    return ((code >> 2) & 3u) + ((code >> 6) & 3u) + ((code >> 10) & 3u);
}
// tiles of a stored rect
__host__ __device__ inline uint32_t rect_count32(uint32_t p) {
    return (p >> 31) ? rect_rows_count(p) : ((p & 0xffu) + 1u) * (((p >> 8) & 0xffu) + 1u);
}
__host__ __device__ inline uint32_t rect_count64(uint2 r) {
    return (r.y & 0xffffu) == 0u ? rect_rows_count(r.y >> 16)
                                 : ((r.y & 0xffffu) - (r.x & 0xffffu)) * ((r.y >> 16) - (r.x >> 16));
}
// back to (x0 | y0 << 16, x1 | y1 << 16, rows) for the parity taps
__host__ __device__ inline void rect_unpack32(uint32_t p, uint32_t tiles_x, uint32_t &r0, uint32_t &r1, uint32_t &rows) {
    const uint32_t origin = (p >> 16) & 0x7fffu, x0 = origin % tiles_x, y0 = origin / tiles_x;
    const uint32_t w = (p >> 31) ? ((p >> 12) & 3u) + 1u : (p & 0xffu) + 1u, h = (p >> 31) ? ((p >> 14) & 3u) + 1u : ((p >> 8) & 0xffu) + 1u;
    r0 = x0 | (y0 << 16);
    r1 = (x0 + w) | ((y0 + h) << 16);
    rows = (p >> 31) ? 0x8000u | (p & 0xfffu) : 0u;
}
__host__ __device__ inline void rect_unpack64(uint2 r, uint32_t &r0, uint32_t &r1, uint32_t &rows) {
    r0 = r.x;
    if ((r.y & 0xffffu) == 0u && r.y != 0u) {
        const uint32_t c = r.y >> 16, w = ((c >> 12) & 3u) + 1u, h = ((c >> 14) & 3u) + 1u;
        r1 = ((r.x & 0xffffu) + w) | (((r.x >> 16) + h) << 16);
        rows = 0x8000u | (c & 0xfffu);
    } else {
        r1 = r.y;
        rows = 0u;
    }
}

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

// Wave64 inclusive scans on the DPP cross-lane path (no LDS traffic, one VALU instruction per step):
// row_shr 1/2/4/8 scan each 16-lane row, row_bcast15 carries the row totals into the odd rows
// (row mask 0xa) and row_bcast31 carries the lower half's total into the upper half (row mask 0xc).
// Lanes without a source keep `identity`.  (A __shfl_up ladder costs a ds_bpermute round trip per
// step: ~6 dependent LDS operations per scan.)
#define GS_DPP_STEP(OP, CTRL, ROWMASK)                                                             \
    v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROWMASK, 0xf, false))
__device__ __forceinline__ uint32_t dpp_op_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t dpp_op_max(uint32_t a, uint32_t b) { return a > b ? a : b; }

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
    (void)lane;
    const uint32_t identity = 0u;
    GS_DPP_STEP(dpp_op_add, 0x111, 0xf);   // row_shr:1
    GS_DPP_STEP(dpp_op_add, 0x112, 0xf);   // row_shr:2
    GS_DPP_STEP(dpp_op_add, 0x114, 0xf);   // row_shr:4
    GS_DPP_STEP(dpp_op_add, 0x118, 0xf);   // row_shr:8
    GS_DPP_STEP(dpp_op_add, 0x142, 0xa);   // row_bcast:15
    GS_DPP_STEP(dpp_op_add, 0x143, 0xc);   // row_bcast:31
    return v;
}

__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v) {
    const uint32_t identity = 0u;
    GS_DPP_STEP(dpp_op_max, 0x111, 0xf);
    GS_DPP_STEP(dpp_op_max, 0x112, 0xf);
    GS_DPP_STEP(dpp_op_max, 0x114, 0xf);
    GS_DPP_STEP(dpp_op_max, 0x118, 0xf);
    GS_DPP_STEP(dpp_op_max, 0x142, 0xa);
    GS_DPP_STEP(dpp_op_max, 0x143, 0xc);
    return v;
}
#undef GS_DPP_STEP

__device__ __forceinline__ uint32_t dpp_op_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
// wave minimum / maximum, uniform in every lane (all 64 lanes must be active)
__device__ __forceinline__ uint32_t wave_reduce_min(uint32_t v) {
    const uint32_t identity = 0xffffffffu;
#define GS_DPP_MIN(CTRL, ROWMASK) \
    v = dpp_op_min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROWMASK, 0xf, false))
    GS_DPP_MIN(0x111, 0xf); GS_DPP_MIN(0x112, 0xf); GS_DPP_MIN(0x114, 0xf); GS_DPP_MIN(0x118, 0xf);
    GS_DPP_MIN(0x142, 0xa); GS_DPP_MIN(0x143, 0xc);
#undef GS_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_reduce_max(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(v), 63);
}

// wave total, uniform in every lane (all 64 lanes must be active)
__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v, 0u), 63);
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

// 16-byte store, optionally WRITE-THROUGH at agent scope (`sc1`).  A kernel that leaves B bytes dirty in the L2s makes
// the next dependent kernel wait for their write-back: tools/mb/mb_boundary.hip measures 1.1 us for a clean boundary and
// + B / 7 TB/s behind plain stores, up to + 2.3 us from 16 MB on; with sc1 stores the bytes leave while the kernel still
// runs (+ 0.5 us behind 16 MB; the storing kernel itself gets 0.2-0.7 us longer).  Only for 16-byte-per-lane stores: a
// 4-byte sc1 store is one fabric write each (MI355X_MICROARCH.md: 6 x the time per byte).
__device__ __forceinline__ void store16(void *p, uint4 v, uint32_t mode) {       // mode: 0 plain, 1 sc1, 2 nt, 3 sc0 sc1
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x4_a4_t __attribute__((ext_vector_type(4), aligned(4)));      // (records are 36 bytes apart)
    const u32x4_t w = {v.x, v.y, v.z, v.w};
    // s_nop 1 behind each asm store: a VMEM store of more than 64 bits followed by a VALU write of its data registers
    // needs wait states (ISA "manually inserted wait states"), and the compiler's hazard recognizer does not see a store
    // inside inline asm — without them the first data register was overwritten by the next instruction before the store
    // had read it (a record's fifth word came out as the value computed right after the store).
    if (mode == 1u) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
    else if (mode == 2u) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
    else if (mode == 3u) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
    else *(u32x4_a4_t *)p = u32x4_a4_t{v.x, v.y, v.z, v.w};
}

// ---------------------------------------------------------------------------------------------
// device-resident frame state
// ---------------------------------------------------------------------------------------------

// What only the device knows about the frame in flight.  The host never waits for it: grids are
// sized from upper bounds (V <= N, D <= pair capacity), kernels read the counts from here, and the
// results reach the host through `FrameResult` in pinned memory, read lazily.
struct FrameState {
    uint32_t visible;        // V: written by the first depth-sort pass (which also compacts)
    uint32_t pairs;          // min(D, pair capacity): what the tile sort / ranges / blend work on
    uint32_t overflow;       // D exceeded the pair capacity: the blend leaves the image untouched (frame skipped)
    uint32_t list_blocks;    // length of the block list (k_block_cull), written by the workgroup that drew the last ticket
    uint32_t list_slots;     // = list_blocks * 1024: the preprocess outputs of a list frame live in LIST space, [0, list_slots)
    uint32_t cull_ticket;    // k_block_cull: workgroups take their group of 256 blocks in ticket order; the last ticket resets it
    uint32_t rank_fault;     // watchdog of the LDS-atomic rank (scatter_ranked): set when block 0 of a radix pass finds a rank
                             // that is not the ballot-based one; published with the frame flags, cleared by the host
    uint32_t rank_inject;    // test hook (GS3D_TEST_RANK_FAULT=1): added to the expected rank, so the watchdog fires
    uint32_t pairs_round1;       // two-round frames: the pairs of round 1 (round 2's publication adds them)
    uint32_t round2_visible;     // ... the Gaussians k_round2_write kept for round 2
    uint32_t tiles_done;         // ... tiles round 1 finished (counted by k_round2_count; 0 in a single-round frame)
    uint32_t tiles_open;         // ... tiles round 1 left unfinished although it had pairs for them
    uint32_t round2_skip;        // ... 1: round 1 finished EVERY tile of the band — round 2 has nothing to do (k_round2_gate)
    uint32_t round2_dense;       // ... slots the compacting pass of round 2's depth sort walks (0 when round 2 is skipped)
    uint32_t round1_visible;     // partitioned two-round frames: the Gaussians in front of the depth threshold (round 1 sorts only them)
    uint32_t depth_tau;          // ... the threshold: round 1 = keys < tau, round 2 = keys >= tau (k_round_threshold)
    uint32_t depth_bucket_max;   // largest top-digit bucket of the depth sort (k_bucket_sort, or the LSD sort's last pass)
    uint32_t tile_bucket_max;    // the same for the tile sort (k_bucket_sort)
};
constexpr uint32_t FRAME_FLAG_PAIR_OVERFLOW = 1u;   // D exceeded the pair capacity
constexpr uint32_t FRAME_FLAG_SKIPPED = 2u;         // ... so the frame was skipped: the image was NOT written
constexpr uint32_t FRAME_FLAG_RANK_FAULT = 4u;      // the LDS-atomic rank's order assumption failed in a pass of this (or, for the
                                                    // tile sort, of the previous) frame: blend order possibly wrong; the host
                                                    // switches the device to the ballot-based rank

// pinned host memory, one per frame parity; written by workgroup 0 of k_pairs_emit
struct FrameResult {
    uint32_t visible;
    uint32_t flags;
    uint64_t pairs_total;    // true D, also when it exceeded the capacity
    uint32_t gen;            // frame generation this result belongs to
    uint32_t depth_bucket_max;   // FrameState::depth_bucket_max: feeds the host's choice of the depth sort (MSD-first / LSD)
    uint32_t tile_bucket_max;    // FrameState::tile_bucket_max as the PREVIOUS frame of the renderer left it (the tile sort
                                 // runs behind the kernel that publishes this block); 0 = that frame's tile sort was LSD
    uint32_t tiles_done;         // two-round frames: tiles finished by round 1 (feeds the host's choice of round 1's length)
    uint32_t tiles_open;         // ... tiles round 1 had pairs for and did not finish
    uint32_t round_pairs_max;    // ... the larger of the two rounds' pair counts (what the next two-round frame's grids are sized for)
    uint32_t pad[2];
};

// One thread publishes a frame's result to pinned host memory.  `gen` goes LAST, behind a
// system-scope release: a host that polls the block without a fencing event (hipEventQuery on an
// event created with hipEventDisableSystemFence, then a read) and finds the expected generation
// also finds that generation's counts.
__device__ __forceinline__ void publish_result(FrameResult *r, uint32_t visible, uint64_t pairs_total, uint32_t flags,
                                               uint32_t gen, uint32_t *flags_dev = nullptr, uint32_t depth_bucket_max = 0u,
                                               uint32_t tile_bucket_max = 0u, uint32_t tiles_done = 0u, uint32_t tiles_open = 0u,
                                               uint32_t round_pairs_max = 0u) {
    // optional copy of the flags in DEVICE memory (gs_renderer_set_frame_flags_target): a sharded frame
    // carries it inside its band's gather chunk, so every rank learns from the one all-gather whether
    // any band was skipped
    if (flags_dev) *flags_dev = flags;
    r->visible = visible;
    r->pairs_total = pairs_total;
    r->flags = flags;
    r->depth_bucket_max = depth_bucket_max;
    r->tile_bucket_max = tile_bucket_max;
    r->tiles_done = tiles_done;
    r->tiles_open = tiles_open;
    r->round_pairs_max = round_pairs_max;
    __hip_atomic_store(&r->gen, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the frame without Gaussians: its (empty) result is published IN STREAM ORDER like every other
// frame's, so an older frame still in flight on the same result block cannot overwrite it later
__global__ void k_publish_result(FrameResult *r, FrameState *state, uint32_t gen, uint32_t *flags_dev) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state->visible = 0;
        state->pairs = 0;
        state->overflow = 0;
        publish_result(r, 0u, 0ull, 0u, gen, flags_dev);
    }
}

// wave sum of a 64-bit value, uniform in every lane (all 64 lanes active)
__device__ __forceinline__ uint64_t wave_reduce_add64(uint64_t v) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor((unsigned long long)v, d, WAVE);
    return v;
}

// ---------------------------------------------------------------------------------------------
// repack: AoS (N x pod_bytes) -> block-planar mirror
// ---------------------------------------------------------------------------------------------

// Mirror layout ("block-planar"): Gaussians are grouped in blocks of PLANAR_BLOCK = 1024; inside a
// block the nc = pod_bytes/16 chunk planes follow each other, each PLANAR_BLOCK x 16 B.  A
// preprocess workgroup (one block) therefore streams one contiguous nc x 16 KiB span, and a wave
// still reads 1 KiB contiguous per load instruction.  (Whole-array planes — plane stride N x 16 B —
// measured 2 % slower at 10 M: 14 streams 160 MB apart instead of one 224 KiB span per workgroup.)
constexpr uint32_t PLANAR_BLOCK = 1024;
__device__ __forceinline__ uint64_t planar_at(uint32_t c, uint64_t i, uint32_t nc) {
    return (((i / PLANAR_BLOCK) * nc + c) * PLANAR_BLOCK) | (i % PLANAR_BLOCK);
}

// One workgroup transposes REPACK_GROUP Gaussians through LDS: the AoS side is read as one
// contiguous span (REPACK_GROUP x pod_bytes), the mirror side is written 1 KiB contiguous per wave
// and plane; both directions are fully coalesced.  (A direct per-chunk copy scatters 16-byte
// writes over the planes: measured 1.66 ms vs the transposed kernel for 10 M x 224 B.)
constexpr uint32_t REPACK_GROUP = 128;
constexpr uint32_t REPACK_MAX_CHUNKS = 14;   // pod_bytes / 16 <= 224 / 16
__global__ __launch_bounds__(256) void k_repack_planar(const uint4 *__restrict__ aos,
                                                       uint4 *__restrict__ planar, uint64_t first,
                                                       uint64_t count, uint32_t chunks) {
    __shared__ uint4 s_t[REPACK_GROUP * REPACK_MAX_CHUNKS];   // 28 KiB
    const uint64_t g0 = (uint64_t)blockIdx.x * REPACK_GROUP;   // first Gaussian of this group, relative to `first`
    const uint32_t ng = (uint32_t)(count - g0 < REPACK_GROUP ? count - g0 : REPACK_GROUP);
    const uint32_t total = ng * chunks;
    const uint4 *src = aos + (first + g0) * chunks;
    for (uint32_t q = threadIdx.x; q < total; q += 256) s_t[q] = src[q];
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < total; e += 256) {
        const uint32_t c = e / ng, t = e - c * ng;   // consecutive threads -> consecutive Gaussians of one plane
        planar[planar_at(c, first + g0 + t, chunks)] = s_t[t * chunks + c];
    }
}

// ---------------------------------------------------------------------------------------------
// spatial mirror order: slot -> Gaussian index by the 30-bit Morton code of the
// position (10 bits per axis over the bounding box of all positions), ties by index.  Built when
// the whole buffer is (re)mirrored.  Neighbouring slots are then neighbours in space, so frustum-
// and band-culled Gaussians cluster into whole 128-byte lines whose SH chunks are never fetched.
// ---------------------------------------------------------------------------------------------

// position = first 12 bytes of every AoS record
__global__ __launch_bounds__(256) void k_bbox_partial(const uint32_t *__restrict__ aos, uint32_t pod_words,
                                                      uint32_t n, float *__restrict__ partial) {
    __shared__ float s_lo[4][3], s_hi[4][3];
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const uint32_t *w = aos + i * pod_words;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float v = u2f(w[a]);
            lo[a] = fminf(lo[a], v);   // minNum / maxNum: a NaN coordinate is ignored
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int d = WAVE / 2; d > 0; d >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, WAVE));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, WAVE));
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            s_lo[wid][a] = lo[a];
            s_hi[wid][a] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3u) {
        const uint32_t a = threadIdx.x;
        partial[blockIdx.x * 6u + a] = fminf(fminf(s_lo[0][a], s_lo[1][a]), fminf(s_lo[2][a], s_lo[3][a]));
        partial[blockIdx.x * 6u + 3u + a] = fmaxf(fmaxf(s_hi[0][a], s_hi[1][a]), fmaxf(s_hi[2][a], s_hi[3][a]));
    }
}

// one workgroup: partial[count][6] -> bbox[6] = lo xyz, hi xyz
__global__ __launch_bounds__(256) void k_bbox_final(const float *__restrict__ partial, uint32_t count,
                                                    float *__restrict__ bbox) {
    __shared__ float s_v[256][6];
    float v[6] = {__builtin_inff(), __builtin_inff(), __builtin_inff(),
                  -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint32_t i = threadIdx.x; i < count; i += 256u) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            v[a] = fminf(v[a], partial[i * 6u + a]);
            v[3 + a] = fmaxf(v[3 + a], partial[i * 6u + 3u + a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) s_v[threadIdx.x][a] = v[a];
    __syncthreads();
    if (threadIdx.x < 6u) {
        const uint32_t a = threadIdx.x;
        float r = s_v[0][a];
        for (uint32_t t = 1; t < 256u; t++) r = a < 3u ? fminf(r, s_v[t][a]) : fmaxf(r, s_v[t][a]);
        bbox[a] = r;
    }
}

__device__ __forceinline__ uint32_t morton_axis(float p, float lo, float hi) {
    float s = ((p - lo) / (hi - lo)) * 1024.0f;
    return s >= 1023.0f ? 1023u : (s > 0.0f ? (uint32_t)s : 0u);   // NaN (incl. 0/0 of a flat axis) -> 0
}

__global__ __launch_bounds__(256) void k_morton_keys(const uint32_t *__restrict__ aos, uint32_t pod_words,
                                                     uint32_t n, const float *__restrict__ bbox,
                                                     uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t *w = aos + (uint64_t)i * pod_words;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        uint32_t q = morton_axis(u2f(w[a]), bbox[a], bbox[3 + a]);
#pragma unroll
        for (int b = 0; b < 10; b++) code |= ((q >> b) & 1u) << (3 * b + a);
    }
    keys[i] = code;
    vals[i] = i;
}

__global__ __launch_bounds__(256) void k_invert_order(const uint32_t *__restrict__ order, uint32_t n,
                                                      uint32_t *__restrict__ inv) {
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot < n) inv[order[slot]] = slot;
}

// mirror slots [0, count) <- records order[slot] (whole-buffer rebuild in spatial order)
__global__ __launch_bounds__(256) void k_repack_planar_ordered(const uint4 *__restrict__ aos,
                                                               uint4 *__restrict__ planar,
                                                               const uint32_t *__restrict__ order,
                                                               uint64_t count, uint32_t chunks) {
    __shared__ uint4 s_t[REPACK_GROUP * REPACK_MAX_CHUNKS];
    const uint64_t g0 = (uint64_t)blockIdx.x * REPACK_GROUP;
    const uint32_t ng = (uint32_t)(count - g0 < REPACK_GROUP ? count - g0 : REPACK_GROUP);
    const uint32_t total = ng * chunks;
    for (uint32_t q = threadIdx.x; q < total; q += 256) {
        const uint32_t t = q / chunks, c = q - t * chunks;   // consecutive threads -> one record's consecutive chunks
        s_t[q] = aos[(uint64_t)order[g0 + t] * chunks + c];
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < total; e += 256) {
        const uint32_t c = e / ng, t = e - c * ng;
        planar[planar_at(c, g0 + t, chunks)] = s_t[t * chunks + c];
    }
}

// partial update in spatial order: records [first, first+count) go to their existing slots
__global__ __launch_bounds__(256) void k_repack_planar_scatter(const uint4 *__restrict__ aos,
                                                               uint4 *__restrict__ planar,
                                                               const uint32_t *__restrict__ inv,
                                                               uint64_t first, uint64_t count, uint32_t chunks) {
    const uint64_t total = count * chunks;
    for (uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x; q < total; q += (uint64_t)gridDim.x * 256u) {
        const uint64_t id = first + q / chunks;
        const uint32_t c = (uint32_t)(q % chunks);
        planar[planar_at(c, inv[id], chunks)] = aos[first * chunks + q];
    }
}

// ---------------------------------------------------------------------------------------------
// block bounds and conservative block culling.  With the mirror in spatial order a block of 1024
// slots is a compact region of space: bb[block] = {lo xyz, hi xyz of the positions, L, 0} with
// L = max Frobenius norm of the 3D covariances (>= their largest eigenvalue).  A preprocess
// workgroup whose block provably holds no visible Gaussian returns before reading anything.
// The test must never drop a Gaussian the exact per-Gaussian tests of project_geom would keep:
//   * depth: every centre lies in the box, so its view depth lies between the corners' depths;
//   * screen position: x/z and y/z are ratios of affine functions, so over a box in front of the
//     camera their extremes are attained at corners;
//   * radius: Sigma' = size^2 (J W R_m S_m) Sigma (..)^T + 0.3 I, so lambda_max(Sigma') <=
//     size^2 |J|_2^2 |W R_m S_m|_2^2 lambda_max(Sigma) + 0.3 with SPECTRAL norms (round 4; rounds 2-3 used
//     Frobenius norms and the Jacobian's worst case over the whole image, which inflated the radius bound up
//     to 4.7 x at the image centre and kept half again as many blocks of a band as needed):
//       - |W R_m S_m|_2^2: largest eigenvalue of (W S)^T (W S), computed on the host (cull_gain);
//       - |J|_2^2 = lambda_max(J J^T) / z^2 with J J^T z^2 = [[fx^2 (1 + u^2), fx fy u v], [fx fy u v, fy^2 (1 +
//         v^2)]], u, v the CLAMPED x/z, y/z: the largest eigenvalue of a PSD 2x2 matrix grows with its diagonal
//         and with |off-diagonal|, all of which grow with |u|, |v|, so it is bounded by its value at the
//         block's largest |u|, |v| (corner extremes, cut at limx / limy) and smallest z;
//       - lambda_max(Sigma) <= min(Frobenius norm, largest absolute row sum) per Gaussian (k_block_bounds).
//     The spec's lambda (mid + sqrt(max(0.1, ..))) exceeds the true one by at most 0.32 and
//     radius = ceil(k sqrt(lambda)).  All comparisons carry explicit slack for f32 rounding and
//     are written so that NaN / inf bounds never cull.
// ---------------------------------------------------------------------------------------------

template <int SH, int COV>
__global__ __launch_bounds__(PP_THREADS) void k_block_bounds(const uint4 *__restrict__ planar, uint32_t n,
                                                             float *__restrict__ bb) {
    constexpr int NW = pod_words(SH, COV);
    constexpr int NC = NW / 4;
    constexpr int G0 = cov_word0(SH) / 4;
    constexpr int G1 = (cov_word0(SH) + cov_bytes(COV) / 4 - 1) / 4;
    __shared__ float s_v[4][7];
    const uint32_t base = blockIdx.x * PP_CHUNK;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    float L = 0.0f;
    for (int k = 0; k < PP_ITEMS; k++) {
        const uint32_t i = base + k * PP_THREADS + threadIdx.x;
        if (i < n) {
            uint32_t w[NW];
            uint4 v0 = planar[planar_at(0, i, NC)];
            w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
#pragma unroll
            for (int c = G0; c <= G1; c++) {
                uint4 v = planar[planar_at(c, i, NC)];
                w[4 * c + 0] = v.x; w[4 * c + 1] = v.y; w[4 * c + 2] = v.z; w[4 * c + 3] = v.w;
            }
            float S[6];
            gaussian_unpack_cov3d<SH, COV>(w, S);
            float f2 = ((S[0] * S[0] + S[3] * S[3]) + S[5] * S[5]) + 2.0f * ((S[1] * S[1] + S[2] * S[2]) + S[4] * S[4]);
            float f = sqrtf(f2);
            // two upper bounds of the largest eigenvalue of the (symmetric) covariance: its Frobenius norm and its
            // largest absolute row sum (Gershgorin); the smaller one is kept.  A NaN in either makes the
            // comparison false, so f — NaN or not — stays; a NaN f never replaces L below, +inf does.
            const float g0 = (fabsf(S[0]) + fabsf(S[1])) + fabsf(S[2]);
            const float g1 = (fabsf(S[1]) + fabsf(S[3])) + fabsf(S[4]);
            const float g2 = (fabsf(S[2]) + fabsf(S[4])) + fabsf(S[5]);
            const float gr = fmaxf(fmaxf(g0, g1), g2) * 1.000001f;
            if (gr < f && !(g0 != g0) && !(g1 != g1) && !(g2 != g2)) f = gr;
            L = f > L ? f : L;                 // NaN never replaces L; +inf does (=> never culled)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                float p = u2f(w[a]);
                lo[a] = fminf(lo[a], p);       // NaN positions are ignored: such Gaussians are always culled
                hi[a] = fmaxf(hi[a], p);
            }
        }
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, WAVE));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, WAVE));
        }
        float o = __shfl_xor(L, d, WAVE);
        L = o > L ? o : L;
    }
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            s_v[wid][a] = lo[a];
            s_v[wid][3 + a] = hi[a];
        }
        s_v[wid][6] = L;
    }
    __syncthreads();
    if (threadIdx.x < 8u) {
        const uint32_t a = threadIdx.x;
        float r = 0.0f;
        if (a < 3u) r = fminf(fminf(s_v[0][a], s_v[1][a]), fminf(s_v[2][a], s_v[3][a]));
        else if (a < 6u) r = fmaxf(fmaxf(s_v[0][a], s_v[1][a]), fmaxf(s_v[2][a], s_v[3][a]));
        else if (a == 6u) {
            r = s_v[0][6];
            for (int w = 1; w < 4; w++) r = s_v[w][6] > r ? s_v[w][6] : r;
        }
        bb[blockIdx.x * 8u + a] = r;
    }
}

// true = no Gaussian of the block can pass project_geom's visibility tests for this frame / band
__device__ __forceinline__ bool block_is_culled(const float *__restrict__ bb, const FrameConsts &fc) {
    const float lo[3] = {bb[0], bb[1], bb[2]}, hi[3] = {bb[3], bb[4], bb[5]};
    const float L = bb[6];
    float z0 = __builtin_inff(), z1 = -__builtin_inff();
    float u0 = __builtin_inff(), u1 = -__builtin_inff(), v0 = __builtin_inff(), v1 = -__builtin_inff();
    float mag = 0.0f;   // largest |term| sum of the two matrix products: scales their rounding error
    bool finite = true;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        float p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
        float pw[4], t[4];
        mat4_mul_point(fc.M, p, pw);
        mat4_mul_point(fc.V, pw, t);
        float xv = t[0], yv = -t[1], zv = -t[2];
        finite = finite && (fabsf(xv) < 1e30f) && (fabsf(yv) < 1e30f) && (fabsf(zv) < 1e30f);   // false for NaN / inf
        // |V| (|M| |p| + |translation|): an upper bound of every intermediate magnitude
        float a[3];
#pragma unroll
        for (int r = 0; r < 3; r++)
            a[r] = ((fabsf(fc.M[r]) * fabsf(p[0]) + fabsf(fc.M[4 + r]) * fabsf(p[1])) + fabsf(fc.M[8 + r]) * fabsf(p[2])) +
                   fabsf(fc.M[12 + r]);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            float b = ((fabsf(fc.V[r]) * a[0] + fabsf(fc.V[4 + r]) * a[1]) + fabsf(fc.V[8 + r]) * a[2]) + fabsf(fc.V[12 + r]);
            mag = b > mag ? b : mag;
        }
        z0 = fminf(z0, zv);
        z1 = fmaxf(z1, zv);
        float u = xv / zv, v = yv / zv;
        u0 = fminf(u0, u); u1 = fmaxf(u1, u);
        v0 = fminf(v0, v); v1 = fmaxf(v1, v);
    }
    if (!finite || !(mag < 1e30f)) return false;
    // slack for f32 rounding of the transforms (a centre inside the box may come out slightly
    // outside the corners' hull): ~40 ulp of the largest intermediate, plus a relative part
    const float ea = 5e-6f * mag + 1e-6f;
    const float ez = ea + 1e-5f * (fabsf(z0) + fabsf(z1));
    if (z1 + ez <= fc.near_plane) return true;    // every centre is at or behind the near plane
    if (z0 - ez >= fc.far_plane) return true;
    if (!(z0 - ez > 0.0f)) return false;          // box reaches the camera plane: x/z is unbounded
    const float zl = fmaxf(z0 - ez, fc.near_plane);
    const float um = fmaxf(fabsf(u0), fabsf(u1)), vm = fmaxf(fabsf(v0), fabsf(v1));
    // |J|_2^2 z^2 at the block's largest clamped |x/z|, |y/z| (with their rounding error; see the header comment)
    const float ub = fminf(um + (ea * (1.0f + um) / zl) * 2.0f + 1e-4f * um, fc.limx);
    const float vb = fminf(vm + (ea * (1.0f + vm) / zl) * 2.0f + 1e-4f * vm, fc.limy);
    const float ja = (fc.fx * fc.fx) * (1.0f + ub * ub), jc = (fc.fy * fc.fy) * (1.0f + vb * vb);
    const float jb = fabsf(fc.fx * fc.fy) * (ub * vb);
    const float jn = (0.5f * (ja + jc) + sqrtf(0.25f * ((ja - jc) * (ja - jc)) + jb * jb)) * 1.0001f;
    const float lam = fc.cull_gain * jn * L / (zl * zl) + 0.7f;
    const float r = fc.max_std_dev * sqrtf(lam) * 1.001f + 1.01f;
    if (!(r < 1e30f)) return false;
    const float mxa = fc.fx * u0 + fc.cx, mxb = fc.fx * u1 + fc.cx;
    const float mya = fc.fy * v0 + fc.cy, myb = fc.fy * v1 + fc.cy;
    const float mx0 = fminf(mxa, mxb), mx1 = fmaxf(mxa, mxb), my0 = fminf(mya, myb), my1 = fmaxf(mya, myb);
    // error of x/z: (ea + |x/z| ea) / z, in pixels times the focal length; plus a relative part
    const float ex = fabsf(fc.fx) * (ea * (1.0f + um) / zl) * 2.0f + 1e-4f * (fabsf(mx0) + fabsf(mx1)) + 0.05f;
    const float ey = fabsf(fc.fy) * (ea * (1.0f + vm) / zl) * 2.0f + 1e-4f * (fabsf(my0) + fabsf(my1)) + 0.05f;
    if (mx1 + ex + r < 0.0f) return true;
    if (mx0 - ex - r >= 16.0f * (float)fc.tiles_x) return true;
    if (my1 + ey + r < 16.0f * (float)fc.band_ty0) return true;
    if (my0 - ey - r >= 16.0f * (float)fc.band_ty1) return true;
    return false;
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

// streaming read of the mirror: every byte is read once per frame.  nt = the non-temporal cache policy
typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
// NT is a COMPILE-TIME choice: as a run-time flag every load sat in its own if / else, and hipcc put an
// s_waitcnt vmcnt(0) behind each of them — the record's 14 chunk loads ran as 14 dependent round trips
// (rounds 1-2 believed them "back to back"; the ISA said otherwise, round 3).
template <bool NT>
__device__ __forceinline__ uint4 load_planar(const uint4 *__restrict__ p) {
    if constexpr (NT) {
        const u32x4_nt t = __builtin_nontemporal_load((const u32x4_nt *)p);
        return make_uint4(t.x, t.y, t.z, t.w);
    } else {
        return *p;
    }
}

// ln k, correctly rounded to binary32, k = opacity byte (alpha = (k / 255) exp(power)
// reaches 1/255 only where power >= -ln k).  Entry 0 is unused (opacity 0: never visible).
__device__ const float k_ln_opacity_byte[256] = {
    0.0f, 0.0f, 0.693147182f, 1.09861231f, 1.38629436f, 1.60943794f, 1.79175949f, 1.9459101f,
    2.07944155f, 2.19722462f, 2.30258512f, 2.39789534f, 2.48490667f, 2.56494927f, 2.6390574f, 2.70805025f,
    2.77258873f, 2.83321333f, 2.8903718f, 2.94443893f, 2.99573231f, 3.04452252f, 3.09104252f, 3.13549423f,
    3.17805386f, 3.21887589f, 3.25809646f, 3.29583693f, 3.33220458f, 3.36729574f, 3.40119743f, 3.43398714f,
    3.46573591f, 3.49650764f, 3.52636051f, 3.55534816f, 3.58351898f, 3.61091781f, 3.63758612f, 3.66356158f,
    3.68887949f, 3.71357203f, 3.73766971f, 3.76120019f, 3.7841897f, 3.80666256f, 3.82864141f, 3.85014749f,
    3.87120104f, 3.89182019f, 3.91202307f, 3.93182564f, 3.95124364f, 3.97029185f, 3.98898411f, 4.00733328f,
    4.02535152f, 4.04305124f, 4.06044292f, 4.07753754f, 4.09434462f, 4.1108737f, 4.12713432f, 4.14313459f,
    4.15888309f, 4.17438745f, 4.18965483f, 4.20469284f, 4.21950769f, 4.23410654f, 4.2484951f, 4.26268005f,
    4.27666616f, 4.29045963f, 4.30406523f, 4.31748819f, 4.3307333f, 4.34380531f, 4.356709f, 4.36944771f,
    4.38202667f, 4.39444923f, 4.40671921f, 4.41884041f, 4.43081665f, 4.44265127f, 4.45434713f, 4.46590805f,
    4.47733688f, 4.48863649f, 4.49980974f, 4.51085949f, 4.5217886f, 4.53259945f, 4.54329491f, 4.55387688f,
    4.56434822f, 4.57471085f, 4.58496761f, 4.59511995f, 4.60517025f, 4.61512041f, 4.62497282f, 4.63472891f,
    4.64439106f, 4.65396023f, 4.66343927f, 4.67282867f, 4.68213129f, 4.69134808f, 4.70048046f, 4.70953035f,
    4.71849871f, 4.72738791f, 4.73619843f, 4.74493217f, 4.75359011f, 4.76217413f, 4.77068472f, 4.77912331f,
    4.7874918f, 4.79579067f, 4.80402088f, 4.81218433f, 4.82028151f, 4.82831383f, 4.83628178f, 4.84418726f,
    4.85203028f, 4.85981226f, 4.86753464f, 4.87519741f, 4.88280201f, 4.89034891f, 4.89784002f, 4.90527487f,
    4.91265488f, 4.919981f, 4.92725372f, 4.93447399f, 4.94164228f, 4.94876003f, 4.95582724f, 4.96284485f,
    4.96981335f, 4.97673368f, 4.98360682f, 4.99043274f, 4.99721241f, 5.0039463f, 5.01063538f, 5.01727962f,
    5.02388048f, 5.03043795f, 5.0369525f, 5.04342508f, 5.04985619f, 5.0562458f, 5.06259489f, 5.0689044f,
    5.07517385f, 5.08140421f, 5.08759642f, 5.09375f, 5.09986639f, 5.10594559f, 5.11198759f, 5.11799383f,
    5.12396383f, 5.12989855f, 5.13579845f, 5.14166355f, 5.14749432f, 5.1532917f, 5.15905523f, 5.16478586f,
    5.17048407f, 5.17614985f, 5.18178368f, 5.18738604f, 5.19295692f, 5.19849682f, 5.20400667f, 5.20948601f,
    5.21493578f, 5.22035599f, 5.22574663f, 5.23110867f, 5.23644209f, 5.2417469f, 5.24702406f, 5.25227356f,
    5.2574954f, 5.26269007f, 5.26785803f, 5.27299976f, 5.2781148f, 5.2832036f, 5.28826714f, 5.29330492f,
    5.29831743f, 5.30330467f, 5.30826759f, 5.3132062f, 5.31812f, 5.32300997f, 5.32787609f, 5.33271885f,
    5.33753824f, 5.34233427f, 5.34710741f, 5.35185814f, 5.35658646f, 5.36129236f, 5.36597586f, 5.37063789f,
    5.37527847f, 5.37989712f, 5.38449526f, 5.38907194f, 5.39362764f, 5.39816284f, 5.40267754f, 5.40717173f,
    5.41164589f, 5.4161005f, 5.42053509f, 5.42495012f, 5.42934561f, 5.43372202f, 5.43807936f, 5.44241762f,
    5.44673729f, 5.45103836f, 5.45532131f, 5.45958567f, 5.4638319f, 5.46806002f, 5.47227049f, 5.47646332f,
    5.48063898f, 5.484797f, 5.48893785f, 5.49306154f, 5.49716806f, 5.50125837f, 5.50533152f, 5.50938845f,
    5.51342869f, 5.51745272f, 5.52146101f, 5.52545309f, 5.52942896f, 5.53338957f, 5.53733444f, 5.54126358f,
};

// One Gaussian: returns the number of tiles touched (0 = culled) and fills the 48-byte record.
// Written without early exits: every cull test only clears `ok`, so all the record's loads are
// unconditional and the compiler can issue them back to back at the top (memory-level
// parallelism is what this HBM-bound kernel needs); arithmetic on culled lanes is discarded.
// The geometry half reads only the position / colour chunk and the covariance words; the colour
// half (shade_one) reads the SH words.  project_one = both, in the order the spec gives.
template <int SH, int COV>
__device__ __forceinline__ uint32_t project_geom(const uint32_t *w, const FrameConsts &fc,
                                                 uint4 rec[3], float d[3], const float *ln_tab, uint32_t &rows) {
    float p[3] = {u2f(w[0]), u2f(w[1]), u2f(w[2])};
    float pw[4], t[4];
    mat4_mul_point(fc.M, p, pw);
    mat4_mul_point(fc.V, pw, t);
    float xv = t[0], yv = -t[1], zv = -t[2];
    bool ok = (zv > fc.near_plane) && (zv < fc.far_plane);

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
    ok = ok && (det > 0.0f);
    float inv = 1.0f / det;
    float mid = 0.5f * (ca + cc);
    float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float radius = ceilf(fc.max_std_dev * sqrtf(lam));
    ok = ok && (radius > 0.0f);
    float mx = fc.fx * txz + fc.cx;
    float my = fc.fy * tyz + fc.cy;
    float lo_y = (float)fc.band_ty0, hi_y = (float)fc.band_ty1;
    float fx0 = clampf(floorf((mx - radius) * 0.0625f), 0.0f, (float)fc.tiles_x);
    float fx1 = clampf(floorf((mx + radius) * 0.0625f) + 1.0f, 0.0f, (float)fc.tiles_x);
    float fy0 = clampf(floorf((my - radius) * 0.0625f), lo_y, hi_y);
    float fy1 = clampf(floorf((my + radius) * 0.0625f) + 1.0f, lo_y, hi_y);
    // the record's quadratic form: power(dx, dy) = qa dx^2 + qb dx dy + qc dy^2
    const float qa = -0.5f * (cc * inv), qb = cb * inv, qc = -0.5f * (ca * inv);
    bool masks_ok = false;
    float mask_lim = 0.0f;
    if (fc.clip_rect) {
        // DESIGN.md §3.3 (rect, second step): a pixel receives alpha >= 1/255 from this splat only where
        // power >= -ln k.  The bounding box of {power >= -(ln k + 0.1)} (0.1: head room for the blend's
        // own rounding) therefore holds every pixel the splat can colour; tiles outside it are dropped.
        const uint32_t kop = w[3] >> 24;
        ok = ok && (kop != 0u);
        // Version 3 (round 3): the blend evaluates `power` in binary32; over the pixels of the radius
        // square (|dx|, |dy| <= radius + 16) its rounding error is at most
        // E = 5 u (|qa| + |qb| + |qc|) (radius + 16)^2 with 5 u = 3e-7.  Only where E <= 0.05 (half the head
        // room; the other half covers the rounding of ex / ey) is the clip provably invisible; needles
        // hundreds of pixels long and thinner than a pixel keep the radius square.  NaN E: no clip.
        const float dd = radius + 16.0f;
        const float err = (3.0e-7f * ((fabsf(qa) + fabsf(qb)) + fabsf(qc))) * (dd * dd);
        if (err <= 0.05f) {
            // (ln_tab: the workgroup's LDS copy of k_ln_opacity_byte — from global memory this was a dependent
            // round trip in the middle of every visible Gaussian's projection)
            const float lim = ln_tab[kop] + 0.1f;
            const float ex = sqrtf(lim / -(qa - (qb * qb) / (4.0f * qc)));
            const float ey = sqrtf(lim / -(qc - (qb * qb) / (4.0f * qa)));
            // tile t holds the pixel centres 16 t + 0.5 ... 16 t + 15.5; NaN extents change nothing
            fx0 = fmaxf(fx0, floorf(((mx - ex) - 15.5f) * 0.0625f) + 1.0f);
            fx1 = fminf(fx1, floorf(((mx + ex) - 0.5f) * 0.0625f) + 1.0f);
            fy0 = fmaxf(fy0, floorf(((my - ey) - 15.5f) * 0.0625f) + 1.0f);
            fy1 = fminf(fy1, floorf(((my + ey) - 0.5f) * 0.0625f) + 1.0f);
            // version 4: the tile test below evaluates `power` itself and claims half of the remaining head room
            masks_ok = fc.tile_masks != 0u && err <= 0.025f;
            mask_lim = lim;
        }
    }
    ok = ok && (fx1 > fx0) && (fy1 > fy0);
    // NaN-safe conversions: culled lanes may carry garbage; their values are never used
    uint32_t tx0 = ok ? (uint32_t)fx0 : 0u, tx1 = ok ? (uint32_t)fx1 : 0u;
    uint32_t ty0 = ok ? (uint32_t)fy0 : 0u, ty1 = ok ? (uint32_t)fy1 : 0u;
    uint32_t count = (tx1 - tx0) * (ty1 - ty0);
    rows = 0u;
    {
        // DESIGN.md §3.3, third step (version 4): a rect of 2..3 x 2..3 tiles loses the CORNER tiles whose pixel-centre
        // box [16 t + 0.5, 16 t + 15.5]^2 the region {power >= -(ln k + 0.1)} does not reach.  power is concave and peaks (0)
        // at the mean; only a box lying diagonally off the mean is examined: its maximum sits on the two edges FACING the
        // mean, one clamped parabola each — along dx = const the exponent peaks at dy = ry dx, along dy = const at dx =
        // rx dy.  The same operations in the same order as the tests' CPU restatement.  (A first version
        // examined all nine tiles: ~250 instructions per Gaussian, +8 us at 1 M and +120..200 us at 50 M for 0.4 % more
        // pairs dropped than the corners alone.)
        const uint32_t rw = tx1 - tx0, rh = ty1 - ty0;
        const bool small = ok && masks_ok && rw >= 2u && rw <= 3u && rh >= 2u && rh <= 3u;
        if (__any(small)) {
            const float ry = (-0.5f * qb) / qc, rx = (-0.5f * qb) / qa, nlim = -mask_lim;
            float dl[2][2], dh[2][2], df[2][2];      // [x / y][first / last column or row]: d range of the tile box, facing d
            bool out[2][2];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float m = a == 0 ? mx : my;
                    const uint32_t t = a == 0 ? (e == 0 ? tx0 : tx1 - 1u) : (e == 0 ? ty0 : ty1 - 1u);
                    const float b0 = (float)(16u * t) + 0.5f;
                    dl[a][e] = m - (b0 + 15.0f);
                    dh[a][e] = m - b0;
                    out[a][e] = !(dl[a][e] <= 0.0f && dh[a][e] >= 0.0f);
                    df[a][e] = dh[a][e] < 0.0f ? dh[a][e] : dl[a][e];
                }
            bool drop[2][2];                          // [first / last row][first / last column]
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const float dx = df[0][i], dy = df[1][j];
                    const float t0 = clampf(ry * dx, dl[1][j], dh[1][j]);
                    const float m0 = (qc * t0 + qb * dx) * t0 + (qa * dx) * dx;
                    const float t1 = clampf(rx * dy, dl[0][i], dh[0][i]);
                    const float m1 = (qa * t1 + qb * dy) * t1 + (qc * dy) * dy;
                    drop[j][i] = out[0][i] && out[1][j] && fmaxf(m0, m1) < nlim;
                }
            uint32_t code = 0, kept = 0;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const bool edge_row = j == 0 || (uint32_t)j == rh - 1u;
                const int jj = j == 0 ? 0 : 1;
                uint32_t first = edge_row && drop[jj][0] ? 1u : 0u;
                const uint32_t last = edge_row && drop[jj][1] ? rw - 2u : rw - 1u;
                uint32_t cnt = (uint32_t)j < rh && last + 1u > first ? last + 1u - first : 0u;
                if (!cnt) first = 0u;
                code |= (first | (cnt << 2)) << (4 * j);
                kept += cnt;
            }
            if (small) {
                ok = ok && kept != 0u;              // nothing reached: culled
                if (kept != count) {
                    count = kept;
                    rows = 0x8000u | code;
                }
            }
        }
    }
    if (!ok) {
        tx0 = tx1 = ty0 = ty1 = 0u;
        count = 0u;
        rows = 0u;
    }

    float dw[3] = {pw[0] - fc.cam_pos[0], pw[1] - fc.cam_pos[1], pw[2] - fc.cam_pos[2]};
    float dl = sqrtf((dw[0] * dw[0] + dw[1] * dw[1]) + dw[2] * dw[2]);
    float dn[3] = {dw[0] / dl, dw[1] / dl, dw[2] / dl};
    float dm[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
        dm[r] = (fc.ISR[r] * dn[0] + fc.ISR[3 + r] * dn[1]) + fc.ISR[6 + r] * dn[2];
    float ml = sqrtf((dm[0] * dm[0] + dm[1] * dm[1]) + dm[2] * dm[2]);
    d[0] = dm[0] / ml;
    d[1] = dm[1] / ml;
    d[2] = dm[2] / ml;
    float opacity = unorm8(w[3], 3);

    rec[0] = make_uint4(f2u(mx), f2u(my), f2u(qa), f2u(qb));
    rec[1] = make_uint4(f2u(qc), f2u(opacity), 0u, 0u);
    rec[2] = make_uint4(0u, f2u(zv), tx0 | (ty0 << 16), tx1 | (ty1 << 16));
    return count;
}

template <int SH>
__device__ __forceinline__ void shade_one(const uint32_t *w, const FrameConsts &fc, const float d[3],
                                          uint4 rec[3]) {
    float rgb[3];
    eval_sh<SH>(w, fc.sh_deg, fc.no_sh0 != 0u, d, rgb);
    rec[1].z = f2u(rgb[0]);
    rec[1].w = f2u(rgb[1]);
    rec[2].x = f2u(rgb[2]);
}

template <int SH, int COV>
__device__ __forceinline__ uint32_t project_one(const uint32_t *w, const FrameConsts &fc,
                                                uint4 rec[3], const float *ln_tab, uint32_t &rows) {
    float d[3];
    uint32_t cnt = project_geom<SH, COV>(w, fc, rec, d, ln_tab, rows);
    shade_one<SH>(w, fc, d, rec);
    return cnt;
}

// 4-byte aligned 16-byte vector: lets 36-byte records be moved with two dwordx4 + one dword
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
constexpr int REC_WORDS = 9;   // blend record: mx, my, ca, cb, cc, opacity, r, g, b  (36 bytes)

// Everything a preprocess workgroup writes (passed by value).
struct PreOut {
    uint32_t *recs;                  // [N][9] blend records, by mirror slot
    uint2 *rect;                     // [N] tile rects (0 = culled), by mirror slot
    uint32_t *depth;                 // [N] depth bits - key_bias, 0xffffffff = culled, by mirror slot
    uint32_t *chunk_tiles;           // [chunks] sum of tiles touched (sizing pass)
    uint32_t *chunk_vis;             // [chunks] visible count; 0 = the chunk was (possibly) block-culled and its
                                     //          per-slot arrays are stale: consumers must skip it
    uint32_t *zero_ptr;              // per-frame clear job spread over the grid (tile ranges, expansion sums)
    uint32_t zero_words;
    uint32_t key_bias;
    const float *block_bounds;
    const uint32_t *block_list;      // non-null: workgroup i takes block block_list[i], i < *block_count (k_block_cull ran) AND
                                     // writes its outputs in LIST space: per-slot arrays at i * 1024 + lane, chunk scalars at i.
                                     // The list is ascending, so list-space order == mirror order among the surviving blocks
                                     // (ties of the depth sort break the same way), and everything behind this kernel — the
                                     // compacting depth pass, its histogram, the sizing scan — walks list_slots, not N.
    const uint32_t *block_count;
    uint32_t *chunk_hist;            // [chunks][hist_words] words: histogram of the chunk's keys on the FIRST digit of the depth
                                     // sort ((key >> digit_shift) & digit_mask), two 16-bit counts per word (a chunk holds <= 1024 keys).
                                     // Counted here, where the keys sit in registers: the compacting pass's histogram
                                     // kernel then sums 1 KB rows instead of re-reading 4 KB of keys per chunk.
    uint32_t digit_mask;
    uint32_t digit_shift;            // 0: the LSD sort's lowest digit; the MSD-first sort counts its TOP digit (key >> shift)
    uint32_t hist_words;             // words per chunk row: 256 (512 bins: the LSD sort's digits) or 512 (1024 bins: the
                                     // MSD-first sort's 10-bit top digit)
};
constexpr int MSD_TOP_BITS = 10;              // top digit of the MSD-first sorts: 1024 buckets
constexpr int PRE_HIST_BINS = 1 << MSD_TOP_BITS;

// this workgroup's share of the per-frame clear job
__device__ __forceinline__ void pre_begin(const PreOut &io) {
    for (uint32_t i = blockIdx.x * PP_THREADS + threadIdx.x; i < io.zero_words; i += gridDim.x * PP_THREADS)
        io.zero_ptr[i] = 0u;
}
// a workgroup that is going to project Gaussians clears its digit histogram first and takes its LDS copy of the
// ln(opacity byte) table of the rect clip
__device__ __forceinline__ void pre_hist_clear(uint32_t *s_dhist, float *s_ln) {
#pragma unroll
    for (int q = 0; q < PRE_HIST_BINS / PP_THREADS; q++) s_dhist[threadIdx.x + q * PP_THREADS] = 0u;
    static_assert(PP_THREADS == 256, "one table entry per thread");
    s_ln[threadIdx.x] = k_ln_opacity_byte[threadIdx.x];
    __syncthreads();
}

// per-chunk sums (tiles touched, visible count)
__device__ __forceinline__ void pre_finish(const PreOut &io, uint32_t block, uint32_t local_tiles, uint32_t local_vis,
                                           uint32_t *s_red, const uint32_t *s_dhist) {
    local_tiles = wave_reduce_add(local_tiles);
    local_vis = wave_reduce_add(local_vis);
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    if (lane == 0) {
        s_red[wid] = local_tiles;
        s_red[4 + wid] = local_vis;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        io.chunk_tiles[block] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        io.chunk_vis[block] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    }
    // the chunk's digit histogram (every LDS atomic of the workgroup is behind the barrier above): 1 KB, coalesced
    static_assert(PRE_HIST_BINS == 4 * PP_THREADS, "up to two words = four bins per thread");
    io.chunk_hist[(uint64_t)block * io.hist_words + threadIdx.x] = s_dhist[2u * threadIdx.x] | (s_dhist[2u * threadIdx.x + 1u] << 16);
    if (io.hist_words > (uint32_t)PP_THREADS) {
        const uint32_t t = threadIdx.x + PP_THREADS;
        io.chunk_hist[(uint64_t)block * io.hist_words + t] = s_dhist[2u * t] | (s_dhist[2u * t + 1u] << 16);
    }
}

// whole block provably invisible: nothing is read, nothing is written but the two chunk scalars
__device__ __forceinline__ void pre_finish_culled(const PreOut &io) {
    if (threadIdx.x == 0) {
        io.chunk_tiles[blockIdx.x] = 0u;
        io.chunk_vis[blockIdx.x] = 0u;
    }
}

// Grid: one workgroup per PP_CHUNK Gaussians (= one block of the block-planar mirror).  One
// global_load_dwordx4 per (lane, chunk): a wave reads 1 KiB contiguous per instruction.
// Outputs per Gaussian, written DENSELY (culled lanes store too): the 36-byte blend record, the
// depth key (0xffffffff when culled) and the 8-byte tile rect (0 when culled).  Masking the stores
// of culled lanes would leave holes in every 64-byte sector, which turns the writes into
// read-modify-writes and costs 0.15 ms at 10 M Gaussians on a randomly ordered mirror (measured); a
// dense store of a few don't-care bytes is cheaper.  There is no compaction step: the first pass of
// the depth sort reads the dense keys and simply does not rank the culled ones.
//
// What bounds it (tools/mb/mb_rw.hip, 10 M x 224 B on MI355X): the read pattern alone streams at
// 6.3 TB/s (0.354 ms).  On the view that culls nothing the two-phase kernel below moves its 2.24 GB
// of reads + 0.48 GB of writes in 0.45-0.50 ms = 5.4-6.0 TB/s of physical traffic: the mixed
// read / write ceiling of the part, not a property of the store pattern (rounds 1-2 quoted a
// "0.52 ms read + write floor" from a microbenchmark whose loads were issued as dependent round
// trips; withdrawn in round 3).
// (Tried in round 2 and rejected: compacting the visible (key, slot) pairs here with a decoupled
// look-back over the workgroups.  The inclusive frontier advances one look-back window per status
// round trip across the XCDs, and every waiting workgroup keeps its registers: 0.43 -> 0.54 ms with
// a 64-wide window, 0.60 ms with a 256-wide one.)
template <int SH, int COV, bool NT = false>
__global__ __launch_bounds__(PP_THREADS) void k_preprocess(const uint4 *__restrict__ planar, uint32_t n,
                                                           FrameConsts fc, PreOut io) {
    __shared__ uint32_t s_red[8];
    __shared__ uint32_t s_dhist[PRE_HIST_BINS];
    __shared__ float s_ln[256];
    pre_begin(io);
    if (fc.cull_gain > 0.0f && block_is_culled(io.block_bounds + (uint64_t)blockIdx.x * 8u, fc)) {
        pre_finish_culled(io);
        return;
    }
    pre_hist_clear(s_dhist, s_ln);
    constexpr int NW = pod_words(SH, COV);
    constexpr int NC = NW / 4;
    const uint32_t base = blockIdx.x * PP_CHUNK;
    uint32_t local = 0, local_vis = 0;
#pragma unroll 1
    for (int k = 0; k < PP_ITEMS; k++) {
        const uint32_t i = base + k * PP_THREADS + threadIdx.x;
        if (i < n) {
            // Issue ALL of the record's loads back to back (NC x 1 KiB per wave in flight), then
            // pin them with empty asm statements: without this the compiler sinks each load next
            // to its first use (SH chunks end up behind the projection arithmetic and behind the
            // uniform sh_deg branches), which serialises 4-5 HBM round trips per Gaussian.
            uint4 v[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) v[c] = load_planar<NT>(planar + planar_at(c, i, NC));
            __builtin_amdgcn_sched_barrier(0);   // no load may sink below this point, no use may rise above it
            uint32_t w[NW];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                asm volatile("" : "+v"(v[c].x), "+v"(v[c].y), "+v"(v[c].z), "+v"(v[c].w));
                w[4 * c + 0] = v[c].x;
                w[4 * c + 1] = v[c].y;
                w[4 * c + 2] = v[c].z;
                w[4 * c + 3] = v[c].w;
            }
            uint4 rec[3];
            uint32_t rows;
            const uint32_t cnt = project_one<SH, COV>(w, fc, rec, s_ln, rows);
            uint32_t *o = io.recs + (uint64_t)i * REC_WORDS;
            store16(o, rec[0], fc.wt_records);
            store16(o + 4, rec[1], fc.wt_records);
            o[8] = rec[2].x;
            const uint32_t key = cnt ? rec[2].y - io.key_bias : 0xffffffffu;
            io.depth[i] = key;
            if (cnt) atomicAdd(&s_dhist[(key >> io.digit_shift) & io.digit_mask], 1u);
            if (fc.rect32) ((uint32_t *)io.rect)[i] = cnt ? rect_pack32(rec[2].z, rec[2].w, rows, fc.tiles_x) : 0u;
            else io.rect[i] = cnt ? rect_pack64(rec[2].z, rec[2].w, rows) : make_uint2(0u, 0u);
            local += cnt;
            local_vis += cnt ? 1u : 0u;
        }
    }
    pre_finish(io, blockIdx.x, local, local_vis, s_red, s_dhist);
}

// Two-phase variant for records with SH: phase 1 loads only the chunks that hold position, colour
// and covariance, projects and culls; only the surviving lanes then load their SH chunks
// (EXEC-masked loads: a 128-byte line none of whose lanes survived is not fetched).  With the
// mirror in spatial order the Gaussians a view (or a rank's tile-row band) culls fill whole lines.
// Same arithmetic, same outputs as k_preprocess.
template <int SH, int COV, bool PIPELINED = true, bool NT = false>
__global__ __launch_bounds__(PP_THREADS) void k_preprocess_banded(const uint4 *__restrict__ planar, uint32_t n,
                                                                  FrameConsts fc, PreOut io) {
    __shared__ uint32_t s_red[8];
    __shared__ uint32_t s_dhist[PRE_HIST_BINS];
    __shared__ float s_ln[256];
    pre_begin(io);
    uint32_t block = blockIdx.x;
    if (io.block_list) {
        // block list (k_block_cull): the surviving blocks are taken by the FIRST *block_count workgroups,
        // back to back; the rest of the grid leaves without a single vector instruction.  (With the test
        // in here, a rank's band — five of six blocks culled — spent a third of the CUs' workgroup slots
        // on workgroups that load their bounds, test them on all 256 threads and leave.)
        if (blockIdx.x >= *io.block_count) return;
        block = io.block_list[blockIdx.x];
    } else if (fc.cull_gain > 0.0f && block_is_culled(io.block_bounds + (uint64_t)blockIdx.x * 8u, fc)) {
        pre_finish_culled(io);
        return;
    }
    pre_hist_clear(s_dhist, s_ln);
    constexpr int NW = pod_words(SH, COV);
    constexpr int NC = NW / 4;
    constexpr int G0 = cov_word0(SH) / 4;                              // first chunk holding covariance words
    constexpr int G1 = (cov_word0(SH) + cov_bytes(COV) / 4 - 1) / 4;   // last one
    constexpr int NG = G1 - G0 + 1;
    const uint32_t base = block * PP_CHUNK;
    // where this workgroup's outputs go: list space when it was handed its block by the list
    const uint32_t out_chunk = io.block_list ? blockIdx.x : block;
    const uint32_t obase = out_chunk * PP_CHUNK;
    uint32_t local = 0, local_vis = 0;
    // geometry chunks (position / colour + covariance) of one Gaussian
    auto load_geom = [&](uint32_t i, uint4 &v0, uint4 (&vg)[NG]) {
        v0 = load_planar<NT>(planar + planar_at(0, i, NC));
#pragma unroll
        for (int c = G0; c <= G1; c++)
            if (c != 0) vg[c - G0] = load_planar<NT>(planar + planar_at(c, i, NC));
    };
    // PIPELINED (round 3): the geometry chunks of Gaussian k + 1 are requested BEFORE Gaussian k is
    // projected, shaded and stored, so that (a) a wave has two dependent round trips per Gaussian in
    // flight instead of one after the other, and (b) waiting for them does not also wait for Gaussian
    // k's stores, which on gfx950 retire through the same in-order counter as the loads issued after them.
    uint4 v0_next = make_uint4(0u, 0u, 0u, 0u), vg_next[NG];
#pragma unroll
    for (int c = 0; c < NG; c++) vg_next[c] = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (PIPELINED) {
        const uint32_t i0 = base + threadIdx.x;
        if (i0 < n) load_geom(i0, v0_next, vg_next);
    }
#pragma unroll 1
    for (int k = 0; k < PP_ITEMS; k++) {
        const uint32_t i = base + k * PP_THREADS + threadIdx.x;
        uint4 v0, vg[NG];
        if constexpr (PIPELINED) {
            v0 = v0_next;
#pragma unroll
            for (int c = 0; c < NG; c++) vg[c] = vg_next[c];
            asm volatile("" : "+v"(v0.x), "+v"(v0.y), "+v"(v0.z), "+v"(v0.w));
#pragma unroll
            for (int c = 0; c < NG; c++)
                asm volatile("" : "+v"(vg[c].x), "+v"(vg[c].y), "+v"(vg[c].z), "+v"(vg[c].w));
            if (k + 1 < PP_ITEMS) {
                const uint32_t inext = i + PP_THREADS;
                if (inext < n) load_geom(inext, v0_next, vg_next);
            }
        }
        // the key of every lane of the chunk is stored, also past N (the buffer's last, partial block): in
        // list space those lanes sit INSIDE [0, list_slots) and the compacting depth pass reads them.  ONE
        // store after the branch — a second store inside it landed between the prefetch loads of the next
        // Gaussian and made their wait cover it too (stores and loads retire through one in-order counter).
        uint32_t key = 0xffffffffu;
        if (i < n) {
            uint32_t w[NW];
            if constexpr (!PIPELINED) {
                load_geom(i, v0, vg);
                asm volatile("" : "+v"(v0.x), "+v"(v0.y), "+v"(v0.z), "+v"(v0.w));
#pragma unroll
                for (int c = 0; c < NG; c++)
                    asm volatile("" : "+v"(vg[c].x), "+v"(vg[c].y), "+v"(vg[c].z), "+v"(vg[c].w));
            }
            w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
#pragma unroll
            for (int c = G0; c <= G1; c++) {
                if (c == 0) continue;
                w[4 * c + 0] = vg[c - G0].x;
                w[4 * c + 1] = vg[c - G0].y;
                w[4 * c + 2] = vg[c - G0].z;
                w[4 * c + 3] = vg[c - G0].w;
            }
            uint4 rec[3];
            float d[3];
            uint32_t rows;
            const uint32_t cnt = project_geom<SH, COV>(w, fc, rec, d, s_ln, rows);
            if (cnt) {
                constexpr int S0 = 1, S1 = G0 - 1;   // SH-only chunks (G0.. were loaded above)
                if constexpr (S1 >= S0) {
                    uint4 vs[S1 - S0 + 1];
#pragma unroll
                    for (int c = S0; c <= S1; c++) vs[c - S0] = load_planar<NT>(planar + planar_at(c, i, NC));
#pragma unroll
                    for (int c = S0; c <= S1; c++) {
                        asm volatile("" : "+v"(vs[c - S0].x), "+v"(vs[c - S0].y), "+v"(vs[c - S0].z), "+v"(vs[c - S0].w));
                        w[4 * c + 0] = vs[c - S0].x;
                        w[4 * c + 1] = vs[c - S0].y;
                        w[4 * c + 2] = vs[c - S0].z;
                        w[4 * c + 3] = vs[c - S0].w;
                    }
                }
                shade_one<SH>(w, fc, d, rec);
            }
            const uint32_t oi = obase + (i - base);
            if (cnt || !fc.mask_culled_records) {
                // (the rect of a culled Gaussian is never read: its key says "culled")
                uint32_t *o = io.recs + (uint64_t)oi * REC_WORDS;
                store16(o, rec[0], fc.wt_records);
                store16(o + 4, rec[1], fc.wt_records);
                o[8] = rec[2].x;
                if (fc.rect32) ((uint32_t *)io.rect)[oi] = cnt ? rect_pack32(rec[2].z, rec[2].w, rows, fc.tiles_x) : 0u;
                else io.rect[oi] = rect_pack64(rec[2].z, rec[2].w, rows);
            }
            if (cnt) {
                key = rec[2].y - io.key_bias;
                atomicAdd(&s_dhist[(key >> io.digit_shift) & io.digit_mask], 1u);
            }
            local += cnt;
            local_vis += cnt ? 1u : 0u;
        }
        io.depth[obase + (i - base)] = key;
    }
    pre_finish(io, out_chunk, local, local_vis, s_red, s_dhist);
}

// The block test of a whole frame, one thread per block (instead of 256 threads of every preprocess
// workgroup testing the same block).  The surviving blocks are written to `list` in ASCENDING order
// (round 4): a workgroup tests a group of 256 blocks, publishes its count and adds up the counts of all
// groups in front of it — a decoupled look-back over one status word per group, (tag << 10) | count
// written and polled with agent-scope (sc1) accesses; data and tag share the word, so no fence is
// needed.  Groups are handed out by a TICKET (the order in which workgroups actually start), so a
// workgroup only ever waits for workgroups that started before it: forward progress does not depend on
// the dispatch order or on the whole grid being resident.  The workgroup that draws the last ticket
// resets the ticket for the next frame; the one that owns the last group publishes the list length.
// Ascending matters because the preprocess kernel writes its outputs in LIST order (PreOut::block_list):
// list order must be mirror order for the depth sort's ties to break as they do without the list.
__global__ __launch_bounds__(256) void k_block_cull(const float *__restrict__ bb, uint32_t nblocks, FrameConsts fc,
                                                    uint32_t *__restrict__ list, FrameState *state,
                                                    uint32_t *__restrict__ status, uint32_t tag, uint32_t groups) {
    __shared__ uint32_t s_group, s_cnt[4], s_sum[4];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0u) {
        const uint32_t t = atomicAdd(&state->cull_ticket, 1u);
        if (t + 1u == groups) atomicExch(&state->cull_ticket, 0u);   // every ticket of this launch is out
        s_group = t;
    }
    __syncthreads();
    const uint32_t g = s_group;
    const uint32_t i = g * 256u + threadIdx.x;
    const bool alive = i < nblocks && !block_is_culled(bb + (uint64_t)i * 8u, fc);
    const uint64_t m = __ballot(alive);
    if (lane == 0u) s_cnt[wid] = (uint32_t)__popcll(m);
    __syncthreads();
    const uint32_t tot = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
    if (threadIdx.x == 0u) __hip_atomic_store(&status[g], (tag << 10) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // look back: thread t adds up groups t, t + 256, ... in front of this one (each published by a workgroup
    // whose ticket is smaller, i.e. one that is already running)
    uint32_t before = 0;
    for (uint32_t p = threadIdx.x; p < g; p += 256u) {
        uint32_t v = __hip_atomic_load(&status[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((v >> 10) != tag) {
            __builtin_amdgcn_s_sleep(2);
            v = __hip_atomic_load(&status[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        before += v & 1023u;
    }
    before = wave_reduce_add(before);
    if (lane == 0u) s_sum[wid] = before;
    __syncthreads();
    uint32_t base = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    if (g + 1u == groups && threadIdx.x == 0u) {
        state->list_blocks = base + tot;
        state->list_slots = (base + tot) * (uint32_t)PP_CHUNK;
    }
    for (uint32_t w = 0; w < wid; w++) base += s_cnt[w];
    if (alive) list[base + mbcnt(m)] = i;
}

// ---------------------------------------------------------------------------------------------
// scan of per-chunk sums: one workgroup per array (blockIdx.x selects it).  Used by the stand-alone
// gs_exclusive_scan_u32 and by the frame's sizing pass (exact 64-bit-safe grand total of the tiles).
// ---------------------------------------------------------------------------------------------

struct ScanJob {
    const uint32_t *sums;
    uint32_t *offsets;   // exclusive prefix per chunk
    uint32_t *total;     // grand total (may point into pinned host memory)
    uint32_t num;
    const uint32_t *num_dev = nullptr;   // optional device word: the real length (<= num)
};

__global__ __launch_bounds__(1024) void k_scan_chunks(ScanJob j0, ScanJob j1) {
    constexpr uint32_t PER = 8;   // consecutive values per thread -> 8192 per iteration
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    __shared__ unsigned long long s_total64;   // exact grand total: past 32 bits it is reported as 0xffffffff
    ScanJob job = blockIdx.x == 0 ? j0 : j1;
    if (job.num_dev) {
        const uint32_t real = *job.num_dev;
        if (real < job.num) job.num = real;
    }
    uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    __shared__ uint32_t s_half[2];
    if (threadIdx.x == 0) {
        s_carry = 0;
        s_total64 = 0;
        s_half[0] = 0;
        s_half[1] = 0;
    }
    __syncthreads();
    for (uint32_t base = 0; base < job.num; base += 1024u * PER) {
        uint32_t i0 = base + threadIdx.x * PER;
        uint32_t v[PER];
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            v[k] = i0 + k < job.num ? job.sums[i0 + k] : 0u;
            sum += v[k];
        }
        // exact (non-wrapping) total for the overflow check: the low and high 16-bit halves of the
        // values are summed separately (8192 halves of < 2^16 stay below 2^29), reduced per wave on
        // the DPP path, and only lane 0 of each wave touches the two LDS accumulators
        uint32_t half_lo = 0, half_hi = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            half_lo += v[k] & 0xffffu;
            half_hi += v[k] >> 16;
        }
        half_lo = wave_reduce_add(half_lo);
        half_hi = wave_reduce_add(half_hi);
        if (lane == 0) {
            atomicAdd(&s_half[0], half_lo);
            atomicAdd(&s_half[1], half_hi);
        }
        uint32_t inc = wave_inclusive_scan(sum, lane);
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
        uint32_t run = carry + wave_off + inc - sum;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            if (i0 + k < job.num) job.offsets[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_carry = carry + tot;
            s_total64 += (unsigned long long)s_half[0] + ((unsigned long long)s_half[1] << 16);
            s_half[0] = 0;
            s_half[1] = 0;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *job.total = s_total64 > 0xfffffff0ull ? 0xffffffffu : s_carry;
}

// ---------------------------------------------------------------------------------------------
// expand (row x3): walk the Gaussians in depth order and emit one (tile id, Gaussian index) pair
// per overlapped tile.  Within a tile the pairs are therefore already depth-ordered; the
// following stable sort on the tile id alone yields exactly the order of a stable sort on the
// 64-bit (tile << 32 | depth) key.
// ---------------------------------------------------------------------------------------------

// Workgroup -> tile of the pass.  The hardware deals consecutive workgroup ids round-robin to the 8
// XCDs, each with its own L2.  Tile b's run of a digit is followed in memory by tile b + 1's, and the
// runs are short (16-128 elements): with tiles dealt in order, the two halves of almost every
// 128-byte line are written through two different L2s and reach HBM as partial lines.  With
// xcd_chunk = C, XCD x takes C consecutive tiles of every group of 8 C tiles: neighbouring runs meet
// in one L2, while all XCDs still work on the same region of the output at any time (giving each
// XCD one contiguous eighth of the pass instead lost more on DRAM page locality than it gained).
__device__ __forceinline__ uint32_t scatter_tile_of(uint32_t wg, uint32_t xcd_chunk) {
    if (!xcd_chunk) return wg;
    const uint32_t slot = wg >> 3;
    return (slot / xcd_chunk) * 8u * xcd_chunk + (wg & 7u) * xcd_chunk + slot % xcd_chunk;   // may be past the live tiles
}

constexpr int EXP_CHUNK = 256;   // Gaussians per workgroup in the expansion kernels
constexpr uint32_t EXP_SB = 128;   // chunks per super-chunk of the expansion's offset sums

struct PairCursorRec;
struct ExpandIO {
    const uint32_t *order;           // [V] mirror slots in depth order (values of the depth sort)
    const uint2 *rect;               // [N] tile rects by slot
    uint2 *sorted_rect;              // [V] tile rects in depth order (count -> emit)
    uint32_t *sums;                  // [grid] tile count of every chunk
    unsigned long long *sb_sums;     // [grid / EXP_SB + 1] tile count of every super-chunk (zeroed per frame)
    uint32_t *tvals;                 // [capacity] out: slot of every pair
    FrameState *state;
    FrameResult *result;             // pinned host memory
    uint32_t capacity;               // pair capacity (pairs beyond it are dropped and flagged)
    uint32_t tiles_x;
    uint32_t gen;
    uint32_t sb_bound;               // host bound of the number of super-chunks (entries past the real one are zero)
    struct PairCursorRec *cursors;   // [capacity / CURSOR_SLOTS + 1] where the pairs of every 1024-slot span start
    uint32_t rect32;                 // rect / sorted_rect hold packed 4-byte rects (rect_pack32)
    uint32_t *flags_dev;             // optional device copy of the frame flags (null: none)
    uint32_t xcd_chunk;              // k_expand_count: workgroup -> span order (0: dispatch order)
    uint32_t wt_stores;              // k_pairs_emit stores its pairs write-through (store16)
    // How many Gaussians `order` holds: *count_dev (FrameState::visible, or ::round2_visible for the second round of a
    // two-round frame), at most `limit` (round 1 of a two-round frame: the nearest K).
    const uint32_t *count_dev;
    uint32_t limit;
    uint32_t round;                  // 0: the frame's only round; 1: first of two (publishes nothing); 2: second (publishes both)
    __device__ __forceinline__ uint32_t visible() const {      // Gaussians of this round
        const uint32_t v = *count_dev;
        return v < limit ? v : limit;
    }
};

// Expansion, part 1: gather the tile rects into depth order (the only random access of the key
// path: 8 bytes per visible Gaussian from a compact array) and sum the tile counts per chunk of
// EXP_CHUNK Gaussians and per super-chunk of EXP_SB chunks.  A workgroup covers EXP_COUNT_CHUNKS
// chunks (8 independent gathers in flight per thread) and issues ONE 64-bit atomic for all of them,
// so a super-chunk's word sees EXP_SB / EXP_COUNT_CHUNKS = 16 adds (one add per chunk: 128 adds
// serialised on one address, 26 us instead of 9 at 1 M).
// The grid covers the host's upper bound of V (= N); workgroups past the real V exit at once.
constexpr uint32_t EXP_COUNT_CHUNKS = 8;
static_assert(EXP_SB % EXP_COUNT_CHUNKS == 0, "a count workgroup must not straddle super-chunks");
template <bool RECT32>
__global__ __launch_bounds__(EXP_CHUNK) void k_expand_count(ExpandIO io) {
    __shared__ uint32_t s_red[EXP_COUNT_CHUNKS][4];
    const uint32_t v_count = io.visible();
    // XCD-aware order (io.xcd_chunk, as in the radix passes): consecutive workgroups go round-robin to the 8 XCDs,
    // each with its own L2; neighbours in depth order are neighbours in space often enough (one 64-byte sector holds
    // the rects of 16 consecutive mirror slots) that giving one XCD a run of consecutive spans lets its L2 serve part
    // of the gather
    const uint32_t first_chunk = scatter_tile_of(blockIdx.x, io.xcd_chunk) * EXP_COUNT_CHUNKS;
    if ((uint64_t)first_chunk * EXP_CHUNK >= v_count) return;
    uint32_t slot[EXP_COUNT_CHUNKS];
#pragma unroll
    for (uint32_t c = 0; c < EXP_COUNT_CHUNKS; c++) {
        const uint64_t j = (uint64_t)(first_chunk + c) * EXP_CHUNK + threadIdx.x;
        slot[c] = j < v_count ? io.order[j] : 0xffffffffu;
    }
    // the gather: 8 (4) bytes per visible Gaussian from a compact array, EXP_COUNT_CHUNKS in flight per thread
    uint2 r[EXP_COUNT_CHUNKS];
    uint32_t rp[EXP_COUNT_CHUNKS];
#pragma unroll
    for (uint32_t c = 0; c < EXP_COUNT_CHUNKS; c++) {
        // no branch around a gather (a load behind a branch gets its own wait, and that wait also covers every
        // store issued before it): elements past V gather slot 0 and are masked below
        const uint32_t sl = slot[c] != 0xffffffffu ? slot[c] : 0u;
        if constexpr (RECT32) rp[c] = ((const uint32_t *)io.rect)[sl];
        else r[c] = io.rect[sl];
    }
#pragma unroll
    for (uint32_t c = 0; c < EXP_COUNT_CHUNKS; c++) {
        if constexpr (RECT32) rp[c] = slot[c] != 0xffffffffu ? rp[c] : 0u;
        else r[c] = slot[c] != 0xffffffffu ? r[c] : make_uint2(0u, 0u);
    }
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    // all stores first, back to back, then the reductions: with each store next to its reduction hipcc put an
    // s_waitcnt vmcnt(0) behind every one of the eight stores (eight store round trips in a row per thread)
#pragma unroll
    for (uint32_t c = 0; c < EXP_COUNT_CHUNKS; c++) {
        const uint64_t j = (uint64_t)(first_chunk + c) * EXP_CHUNK + threadIdx.x;
        if constexpr (RECT32) {
            // unconditional: the array is padded past V by more than a workgroup's span (gs3d.hip reserves
            // (N + 1024) x 8 bytes for 4-byte entries), and k_pairs_emit masks what it reads past V
            ((uint32_t *)io.sorted_rect)[j] = rp[c];
        } else {
            if (j < v_count) io.sorted_rect[j] = r[c];
        }
    }
#pragma unroll
    for (uint32_t c = 0; c < EXP_COUNT_CHUNKS; c++) {
        const uint64_t j = (uint64_t)(first_chunk + c) * EXP_CHUNK + threadIdx.x;
        uint32_t v;
        if constexpr (RECT32) v = j < v_count ? rect_count32(rp[c]) : 0u;
        else v = rect_count64(r[c]);
        v = wave_reduce_add(v);
        if (lane == 0) s_red[c][wid] = v;
    }
    __syncthreads();
    if (threadIdx.x < EXP_COUNT_CHUNKS) {
        const uint32_t c = threadIdx.x;
        const uint32_t total = (s_red[c][0] + s_red[c][1]) + (s_red[c][2] + s_red[c][3]);   // <= 256 * 2^22
        if ((uint64_t)(first_chunk + c) * EXP_CHUNK < v_count) io.sums[first_chunk + c] = total;
        uint64_t all = total;      // the 8 chunk totals live in lanes 0..7 of wave 0
#pragma unroll
        for (int d = 1; d < (int)EXP_COUNT_CHUNKS; d <<= 1) all += __shfl_xor((unsigned long long)all, d, WAVE);
        if (c == 0 && all) atomicAdd(io.sb_sums + first_chunk / EXP_SB, (unsigned long long)all);
    }
}

// Round 2 of a two-round frame (DESIGN.md §4.2 "rounds"): which of the visible Gaussians behind the nearest K can still
// colour a pixel?  A Gaussian whose rect (at most 3 x 3 tiles; of a rect that lost tiles to the exact test, version 4,
// only the kept ones) lies entirely in tiles that round 1 finished cannot, and is dropped; larger rects are few and stay.
// k_round2_box_table first condenses the finished-tile bits into one entry per tile: "is the w x h box at this origin
// finished?".  k_round2_slot_bits then answers the question for EVERY output slot, in slot order — a stream over the rect
// array instead of a gather in depth order (a first version gathered: 495 us at 50 M, a 200 MB array read by random
// 4-byte accesses) — and leaves one bit per slot (N / 8 bytes: 6 MB at 50 M, which the caches hold); slots of culled
// Gaussians hold stale rects and get a meaningless bit that nobody reads.  k_round2_count / _write then walk the depth
// order behind K, gather the bits and write the survivors' slots to `order_out`, still in depth order.
struct Round2IO {
    const uint32_t *order;           // [V - K] mirror slots in depth order behind round 1's
    const uint2 *rect;               // [slots] tile rects by slot
    const uint32_t *done;            // one bit per tile, + 1 word
    const uint32_t *open;            // one bit per tile: round 1 had pairs for it and did not finish it
    const uint16_t *box_table;       // [tiles, padded to 8] k_round2_box_table
    uint32_t *keep_bits;             // [slots / 32 + 2] one bit per slot: this Gaussian can still colour a pixel
    uint32_t *order_out;             // [<= V - K]
    FrameState *state;
    unsigned long long *masks;       // [groups * 32] ballots of the kept Gaussians, by group, chunk and wave
    uint32_t *counts, *offsets;      // [groups] kept Gaussians per group, and their exclusive prefix (k_scan_chunks)
    uint32_t first;                  // K
    uint32_t groups;                 // = grid size of k_round2_count / _write
    uint32_t tiles_x, num_tiles;
    uint32_t slots;                  // entries of `rect` (a multiple of 1024)
};
constexpr uint32_t R2_CHUNKS = 8;
constexpr uint32_t R2_GROUP = R2_CHUNKS * 256u;
constexpr uint32_t R2_SLOT_ITEMS = 8;                         // k_round2_slot_bits: slots per thread (2048 per workgroup)
constexpr uint32_t R2_LDS_TILES = 32768;                      // ... tiles whose box table it keeps in LDS (the packed rects' limit)

// Per tile t, nine bits: bit (h - 1) * 3 + (w - 1) = "every tile of the w x h box whose origin is t is finished" (tiles
// past the image edge count as finished: no rect reaches them).  One thread per tile; 16 bits per tile.
__global__ __launch_bounds__(256) void k_round2_box_table(const uint32_t *__restrict__ done, uint16_t *__restrict__ table,
                                                          uint32_t tiles_x, uint32_t tiles_y, const FrameState *state) {
    if (state->round2_skip) return;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= tiles_x * tiles_y) return;
    const uint32_t x0 = t % tiles_x, y0 = t / tiles_x;
    uint32_t row[3];          // per row j: bit i = tiles (x0 .. x0 + i, y0 + j) all finished
#pragma unroll
    for (uint32_t j = 0; j < 3u; j++) {
        uint32_t run = 1u, bits = 0u;
#pragma unroll
        for (uint32_t i = 0; i < 3u; i++) {
            const bool inside = x0 + i < tiles_x && y0 + j < tiles_y;
            const uint32_t q = inside ? (y0 + j) * tiles_x + x0 + i : 0u;
            const uint32_t d = inside ? (done[q >> 5] >> (q & 31u)) & 1u : 1u;
            run &= d;
            bits |= run << i;
        }
        row[j] = bits;
    }
    const uint32_t h1 = row[0], h2 = h1 & row[1], h3 = h2 & row[2];
    table[t] = (uint16_t)(h1 | (h2 << 3) | (h3 << 6));
}

// One bit per output slot: "this Gaussian can still colour a pixel" = its rect is larger than 3 x 3 tiles or its box is
// not finished.  (A rect that lost tiles to the exact test, version 4, is judged by its whole box: conservative — it may
// keep a Gaussian whose open tile is one it does not touch.)  LDS: the workgroup first copies the table into LDS.  A
// first version tested the tiles' bits row by row (3 divergent 8-byte loads and ~100 instructions per slot: 114 us at
// 50 M, from global memory or LDS alike); the table makes it one 2-byte lookup.  A stale rect may point anywhere: its
// origin is clamped into the table.
template <bool RECT32, bool LDS>
__global__ __launch_bounds__(256) void k_round2_slot_bits(Round2IO io) {
    // (dynamic LDS: 2 bytes per tile, padded to 16 — 16 KB at 1080p, 64 KB at 4K; the workgroups are persistent, so the
    // table is copied a few hundred times per frame, not once per 2048 slots)
    extern __shared__ __attribute__((aligned(16))) uint16_t s_table[];
    const uint32_t lane = threadIdx.x & 63u;
    if (io.state->round2_skip) return;
    if constexpr (LDS) {
        const uint4 *src = (const uint4 *)io.box_table;
        uint4 *dst = (uint4 *)s_table;
        for (uint32_t q = threadIdx.x; q < (io.num_tiles + 7u) / 8u; q += 256u) dst[q] = src[q];
        __syncthreads();
    }
    const uint16_t *table = LDS ? s_table : io.box_table;
    const uint32_t span = 256u * R2_SLOT_ITEMS;
    for (uint32_t first = blockIdx.x * span; first < io.slots; first += gridDim.x * span) {
    const uint32_t base = first + (threadIdx.x & ~63u) * R2_SLOT_ITEMS;      // first slot of this wave
    uint32_t p[R2_SLOT_ITEMS];
    uint2 pr[R2_SLOT_ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < R2_SLOT_ITEMS; i++) {
        uint32_t slot = base + i * 64u + lane;
        slot = slot < io.slots ? slot : io.slots - 1u;               // (the last span: slots is a multiple of 1024)
        if constexpr (RECT32) p[i] = ((const uint32_t *)io.rect)[slot];
        else pr[i] = io.rect[slot];
    }
#pragma unroll
    for (uint32_t i = 0; i < R2_SLOT_ITEMS; i++) {
        uint32_t origin, w, h;
        if constexpr (RECT32) {
            const bool masked = (p[i] >> 31) != 0u;
            origin = (p[i] >> 16) & 0x7fffu;
            w = masked ? ((p[i] >> 12) & 3u) + 1u : (p[i] & 0xffu) + 1u;
            h = masked ? ((p[i] >> 14) & 3u) + 1u : ((p[i] >> 8) & 0xffu) + 1u;
        } else {
            uint32_t r0, r1, rows;
            rect_unpack64(pr[i], r0, r1, rows);
            origin = __umul24(r0 >> 16, io.tiles_x) + (r0 & 0xffffu);
            w = (r1 & 0xffffu) - (r0 & 0xffffu);
            h = (r1 >> 16) - (r0 >> 16);
        }
        const bool small = w >= 1u && h >= 1u && w <= 3u && h <= 3u;
        origin = origin < io.num_tiles ? origin : io.num_tiles - 1u;
        const uint32_t e = table[origin];
        const bool finished = small && ((e >> (((h - 1u) & 3u) * 3u + ((w - 1u) & 3u))) & 1u) != 0u;
        const uint64_t m = __builtin_amdgcn_ballot_w64(!finished);
        if (lane == 0u && base + i * 64u < io.slots)
            *(uint2 *)(io.keep_bits + ((base + i * 64u) >> 5)) = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
    }
    }
}

// The survivors in depth order: an ordered compaction in three launches — k_round2_count gathers the bits of a group of
// 2048 Gaussians, leaves their ballots (one 64-bit word per wave and chunk) and the group's count; k_scan_chunks turns the
// counts into offsets and the total (FrameState::round2_visible); k_round2_write places the slots.  (A single pass with a
// decoupled look-back over one status word per group was 325 us at 50 M: the 2048 resident workgroups finish together,
// so every one of them walks back over ~2000 words, 64 per dependent round trip, before it meets an inclusive prefix.)
// tiles finished by round 1, and tiles it had pairs for and left open (bits past the last tile are never set): one
// workgroup of 256 threads, for the host's feedback (FrameState::tiles_done / tiles_open)
__device__ __forceinline__ void round2_count_tiles(const Round2IO &io, uint32_t *s_cnt, uint32_t *s_open) {
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t n = 0, o = 0;
    for (uint32_t q = threadIdx.x; q < (io.num_tiles + 31u) / 32u; q += 256u) {
        n += (uint32_t)__popc(io.done[q]);
        o += (uint32_t)__popc(io.open[q]);
    }
    n = wave_reduce_add(n);
    o = wave_reduce_add(o);
    if (lane == 0u) {
        s_cnt[wid] = n;
        s_open[wid] = o;
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        io.state->tiles_done = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        io.state->tiles_open = (s_open[0] + s_open[1]) + (s_open[2] + s_open[3]);
    }
}

__global__ __launch_bounds__(256) void k_round2_count(Round2IO io) {
    __shared__ uint32_t s_cnt[4];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x;
    if (io.state->round2_skip) {          // (k_round2_gate) nothing is left: count 0, no ballots
        if (threadIdx.x == 0u) io.counts[g] = 0u;
        return;
    }
    const uint32_t v = io.state->visible;
    const uint32_t count = v > io.first ? v - io.first : 0u;
    const uint64_t j0 = (uint64_t)g * R2_GROUP + threadIdx.x;
    uint32_t slot[R2_CHUNKS];
#pragma unroll
    for (uint32_t c = 0; c < R2_CHUNKS; c++) {
        const uint64_t j = j0 + c * 256u;
        slot[c] = j < count ? io.order[j] : 0xffffffffu;
    }
    uint32_t word[R2_CHUNKS];
#pragma unroll
    for (uint32_t c = 0; c < R2_CHUNKS; c++)      // (no branch around a gather: see k_expand_count)
        word[c] = io.keep_bits[(slot[c] != 0xffffffffu ? slot[c] : 0u) >> 5];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t c = 0; c < R2_CHUNKS; c++) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(slot[c] != 0xffffffffu && ((word[c] >> (slot[c] & 31u)) & 1u) != 0u);
        mine += (uint32_t)__popcll(m);
        if (lane == 0u) io.masks[((uint64_t)g * R2_CHUNKS + c) * 4u + wid] = m;      // item order: chunk-major, wave-minor
    }
    if (lane == 0u) s_cnt[wid] = mine;
    __syncthreads();
    if (threadIdx.x == 0u) io.counts[g] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
}

// The gate of round 2, one workgroup, right behind round 1's blend: counts the tiles (FrameState::tiles_done / tiles_open, for
// the host's feedback) and — when round 1 finished EVERY tile of the band, as it does for a view the scene covers —
// declares round 2 empty: the box table, the slot bits and the compaction return at once, round 2's Gaussian count
// comes out as 0 and every kernel behind it exits on that count (10 M: ~110 us of round 2 become ~13 empty launches).
__global__ __launch_bounds__(256) void k_round2_gate(Round2IO io, uint32_t band_tiles, uint32_t dense, const uint32_t *dense_dev) {
    __shared__ uint32_t s_cnt[4], s_open[4];
    round2_count_tiles(io, s_cnt, s_open);
    if (threadIdx.x == 0u) {
        const uint32_t done = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        const uint32_t skip = done == band_tiles ? 1u : 0u;
        io.state->round2_skip = skip;
        io.state->round2_dense = skip ? 0u : (dense_dev ? *dense_dev : dense);
    }
}

__global__ __launch_bounds__(256) void k_round2_write(Round2IO io) {
    if (io.state->round2_skip) return;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x;
    const uint32_t v = io.state->visible;
    const uint32_t count = v > io.first ? v - io.first : 0u;
    if ((uint64_t)g * R2_GROUP >= count) return;
    const uint64_t j0 = (uint64_t)g * R2_GROUP + threadIdx.x;
    // every wave scans the group's 32 ballots for itself (lane l: chunk l / 4, wave l % 4)
    const uint64_t mall = lane < R2_CHUNKS * 4u ? io.masks[(uint64_t)g * (R2_CHUNKS * 4u) + lane] : 0ull;
    const uint32_t base = io.offsets[g];
    const uint32_t pc = (uint32_t)__popcll(mall);
    const uint32_t excl = wave_inclusive_scan(pc, lane) - pc;
    uint32_t slot[R2_CHUNKS];
#pragma unroll
    for (uint32_t c = 0; c < R2_CHUNKS; c++) {       // all loads first (the padded array is readable past `count`)
        const uint64_t j = j0 + c * 256u;
        slot[c] = io.order[j < count ? j : 0u];
    }
#pragma unroll
    for (uint32_t c = 0; c < R2_CHUNKS; c++) {
        const uint32_t src = c * 4u + wid;
        const uint64_t m = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(mall >> 32), (int)src, WAVE) << 32) |
                           (uint32_t)__shfl((int)(uint32_t)mall, (int)src, WAVE);
        const uint32_t off = (uint32_t)__shfl((int)excl, (int)src, WAVE);
        if ((m >> lane) & 1ull) io.order_out[base + off + mbcnt(m)] = slot[c];
    }
}

// Partitioned two-round frames: where to cut the depth order.  The preprocess kernel counted the TOP 10 bits of every
// visible Gaussian's depth key per 1024-slot chunk (PreOut::chunk_hist in its MSD-first layout); k_sort_hist_chunks and
// the row scan of the MSD-first sort turn the rows into the 1024 digit totals (a first version added the rows up with one
// global atomic per bin and workgroup: 230 us at 50 M — 763 atomics on each of 1024 addresses), and this kernel — one
// workgroup — turns the totals into the threshold: the first digit boundary with at least `target` Gaussians in front of
// it (FrameState::depth_tau = that digit << low_bits; everything in front: round 1, sorted and rendered first; the rest
// waits for k_round2_slot_bits).  It also publishes V (FrameState::visible), which no depth sort of such a frame counts.
struct ThresholdIO {
    const uint32_t *totals;          // [1024] Gaussians per top digit
    FrameState *state;
    uint32_t target;                 // Gaussians wanted in round 1
    uint32_t low_bits;               // key bits below the top digit
};
__global__ __launch_bounds__(512) void k_round_threshold(ThresholdIO io) {
    __shared__ uint32_t s_tot[1024];
    __shared__ uint32_t s_wave[8];
    const uint32_t tid = threadIdx.x;
    const uint32_t t0 = io.totals[2u * tid], t1 = io.totals[2u * tid + 1u];
    const uint32_t lane = tid & 63u, wid = tid >> 6;
    const uint32_t incl = wave_inclusive_scan(t0 + t1, lane);
    if (lane == 63u) s_wave[wid] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t q = 0; q < 8u; q++) {
        before += q < wid ? s_wave[q] : 0u;
        all += s_wave[q];
    }
    const uint32_t i1 = before + incl, i0 = i1 - t1;      // Gaussians with digit <= 2t / <= 2t + 1
    s_tot[2u * tid] = i0;
    s_tot[2u * tid + 1u] = i1;
    __syncthreads();
    // the cut: the smallest digit d with (Gaussians of digits <= d) >= target -> tau = (d + 1) << low_bits
    const bool hit0 = i0 >= io.target && (tid == 0u || s_tot[2u * tid - 1u] < io.target);
    const bool hit1 = i1 >= io.target && i0 < io.target;
    if (hit0 || hit1) {
        const uint32_t d = 2u * tid + (hit0 ? 0u : 1u);
        const uint64_t tau = (uint64_t)(d + 1u) << io.low_bits;
        io.state->depth_tau = tau >= 0xffffffffull ? 0xffffffffu : (uint32_t)tau;
    }
    if (tid == 0u) {
        if (all < io.target) io.state->depth_tau = 0xffffffffu;      // fewer visible Gaussians than wanted: all of them in round 1
        io.state->visible = all;
    }
}

// ---------------------------------------------------------------------------------------------
// radix sort (row x4): stable LSD, up to 8 bits per pass (the host balances the digit widths over
// the passes: 13 tile bits -> 7 + 6, which doubles / quadruples the length of the digit runs a
// workgroup writes and so the share of whole 64-byte sectors); per pass: histogram -> row scan ->
// scatter.
// Templated on the key type: u32 depth keys (4 passes), u16 / u32 tile keys (2 / 3 passes), and
// u64 keys for the stand-alone gs_sort_pairs_u64.
// ---------------------------------------------------------------------------------------------

constexpr int SORT_THREADS = 256;
constexpr int RADIX_BITS = 8;          // default digit width; the depth sort may use 9 (see sort_pairs_device)
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int RADIX_BITS_MAX = 9;

// Keys per thread (tile = 256 x ITEMS keys per workgroup).  Larger tiles write longer digit runs and
// amortise the per-block histogram rows; smaller tiles give a small sort more workgroups than CUs.
// Measured (round 2, depth sort of V = 0.71 N): 16 -> 32 keys per thread is 0.212 -> 0.180 ms at
// N = 10 M but 50 -> 68 us at N = 1 M; 16-bit tile keys: 32 per thread beats 16 at both sizes, and
// 64 per thread (96 KiB of LDS, one workgroup per CU) is slower (10 M: 0.30 -> 0.39 ms).
template <typename K> struct SortCfg;
template <> struct SortCfg<uint64_t> { static constexpr int ITEMS = 8, ITEMS_LARGE = 8; };
template <> struct SortCfg<uint32_t> { static constexpr int ITEMS = 16, ITEMS_LARGE = 32; };
template <> struct SortCfg<uint16_t> { static constexpr int ITEMS = 16, ITEMS_LARGE = 32; };
template <typename K> constexpr int sort_tile() { return SORT_THREADS * SortCfg<K>::ITEMS; }

// Where a sort's element count comes from: a host value, or (count_dev != null) a device word
// clamped to the host-side bound `count` that sized the grid and the histogram rows.
struct SortCount {
    uint32_t count;
    const uint32_t *count_dev;
    __device__ __forceinline__ uint32_t get() const {
        if (!count_dev) return count;
        const uint32_t c = *count_dev;
        return c < count ? c : count;
    }
};

// COMPACT passes (the first pass of the frame's depth sort, u32 keys only): the input is the DENSE
// per-slot key array of preprocess — key 0xffffffff = culled — plus the per-chunk visible counts
// (0 = the chunk's keys are stale, skip it); the value of an element is its index (the mirror
// slot).  Culled elements are simply not counted and not ranked, so the pass performs the ordered
// compaction of the visible Gaussians for free: the output of the pass is the dense, stably
// partitioned (key, slot) list of the V visible ones, and workgroup 0 publishes V.
constexpr uint32_t SORT_INVALID_KEY = 0xffffffffu;

// Which elements of a COMPACT pass count (partitioned two-round frames, DESIGN.md §4.2 "rounds"): all visible ones
// (tau_dev == null), those in front of the depth threshold (side 0: key < *tau_dev), or those behind it that can
// still colour a pixel (side 1: key >= *tau_dev and the slot's bit in keep_bits, k_round2_slot_bits).  The rest is treated
// exactly like a culled Gaussian, so each round sorts only what it renders.
struct CompactPred {
    const uint32_t *tau_dev = nullptr;
    const uint32_t *keep_bits = nullptr;
    uint32_t side = 0;
};

// ghist layout: [digit][block] (digit-major, row stride = the grid size) so that the row scan reads
// contiguous memory.  Tile ids and depth exponents are highly repetitive, so neighbouring lanes
// often hit the same bin; the private copies (lane & (COPIES-1)) cut the same-address LDS atomic
// serialisation.
template <typename K, int RB, bool COMPACT, int ITEMS, bool PRED = false>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const K *__restrict__ keys, SortCount sc,
                                                            uint32_t shift, uint32_t digit_mask,
                                                            uint32_t *__restrict__ ghist,
                                                            const uint32_t *__restrict__ chunk_vis, uint32_t num_blocks,
                                                            uint32_t xcd_chunk, CompactPred pred = CompactPred()) {
    constexpr uint32_t TILE = SORT_THREADS * ITEMS;
    constexpr int R = 1 << RB;
    constexpr int COPIES = 2048 / R;   // 8 KiB of private copies: 8 x 256 or 4 x 512 bins
    static_assert(!PRED || COMPACT, "the predicate belongs to the compacting pass");
    uint32_t tau = 0;
    if constexpr (PRED) tau = *pred.tau_dev;
    constexpr int DPT = R / SORT_THREADS;
    static_assert(!COMPACT || (sizeof(K) == 4 && TILE % PP_CHUNK == 0), "compacting pass: u32 keys, whole chunks per tile");
    const uint32_t count = sc.get();
    // same workgroup -> tile order as the scatter: the row entries of neighbouring tiles (4 bytes each,
    // adjacent in ghist) are then written through one L2
    const uint32_t block = scatter_tile_of(blockIdx.x, xcd_chunk);
    if ((uint64_t)block * TILE >= count) return;   // past the real count: the row entries are never read
    __shared__ uint32_t s_hist[COPIES][R];
#pragma unroll
    for (int c = 0; c < COPIES; c++)
#pragma unroll
        for (int q = 0; q < DPT; q++) s_hist[c][threadIdx.x + q * SORT_THREADS] = 0;
    __syncthreads();
    const uint32_t base = block * TILE;
    const uint32_t copy = threadIdx.x & (uint32_t)(COPIES - 1);
    // 16-byte loads: the order of the keys does not matter for a histogram
    constexpr int PER_VEC = 16 / sizeof(K);
    constexpr int VECS = ITEMS / PER_VEC;
    static_assert(ITEMS % PER_VEC == 0, "tile must be a whole number of 16-byte vectors per thread");
    if (count - base >= TILE) {
        const uint4 *src = (const uint4 *)(keys + base);
        // ALL loads of the tile are issued back to back before the first LDS atomic, and pinned there:
        // hipcc neither hoists a global load above a ds_add nor keeps hoisted ones apart from their
        // uses — left alone it emitted load, s_waitcnt vmcnt(0), four atomics, eight times in a row
        // (the kernel ran at 2 TB/s).  No branch sits between the loads either: a chunk the compacting
        // pass must skip (block-culled: its keys are stale) is redirected to vector 0's address, whose
        // line is being fetched anyway, and ignored below.
        bool ok[VECS];
        uint4 qv[VECS];
#pragma unroll
        for (int v = 0; v < VECS; v++) {
            // vector v of the tile covers elements [v * 1024, (v + 1) * 1024) for u32 keys = one preprocess chunk
            ok[v] = true;
            if constexpr (COMPACT) ok[v] = chunk_vis[(base >> 10) + v] != 0u;
        }
#pragma unroll
        for (int v = 0; v < VECS; v++) qv[v] = src[(ok[v] ? v : 0) * SORT_THREADS + threadIdx.x];
        uint32_t kb[PRED ? VECS : 1];        // a partitioned frame: the words holding the four slots' keep bits (loaded
                                             // unconditionally — keep_bits is always a readable array —, used by side 1)
        if constexpr (PRED) {
#pragma unroll
            for (int v = 0; v < VECS; v++) kb[v] = pred.keep_bits[(base + ((uint32_t)v * SORT_THREADS + threadIdx.x) * 4u) >> 5];
        }
        __builtin_amdgcn_sched_barrier(0);   // every load is issued before the first pin / use
#pragma unroll
        for (int v = 0; v < VECS; v++) asm volatile("" : "+v"(qv[v].x), "+v"(qv[v].y), "+v"(qv[v].z), "+v"(qv[v].w));
#pragma unroll
        for (int v = 0; v < VECS; v++) {
            if (COMPACT && !ok[v]) continue;
            const uint4 q = qv[v];
            uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < PER_VEC; e++) {
                K k;
                if constexpr (sizeof(K) == 2) k = (K)(w[e >> 1] >> (16 * (e & 1)));
                else if constexpr (sizeof(K) == 4) k = (K)w[e];
                else k = (K)(((uint64_t)w[2 * e + 1] << 32) | w[2 * e]);
                bool live = true;
                if constexpr (COMPACT) live = (uint32_t)k != SORT_INVALID_KEY;
                if constexpr (PRED) {
                    const uint32_t slot = base + ((uint32_t)v * SORT_THREADS + threadIdx.x) * 4u + (uint32_t)e;
                    live = pred.side == 0u ? (uint32_t)k < tau : live && (uint32_t)k >= tau && ((kb[v] >> (slot & 31u)) & 1u) != 0u;
                }
                if (live) atomicAdd(&s_hist[copy][(uint32_t)(k >> shift) & digit_mask], 1u);
            }
        }
    } else {
        const uint32_t valid = count - base;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            uint32_t i = k * SORT_THREADS + threadIdx.x;
            if (i < valid) {
                const K key = keys[base + i];
                bool ok = true;
                if constexpr (COMPACT) {
                    ok = (uint32_t)key != SORT_INVALID_KEY && chunk_vis[(base + i) >> 10] != 0u;
                    if constexpr (PRED)
                        ok = pred.side == 0u ? ok && (uint32_t)key < tau
                                             : ok && (uint32_t)key >= tau && ((pred.keep_bits[(base + i) >> 5] >> ((base + i) & 31u)) & 1u) != 0u;
                }
                if (ok) atomicAdd(&s_hist[copy][(uint32_t)(key >> shift) & digit_mask], 1u);
            }
        }
    }
    __syncthreads();
    // only the LIVE rows (digits that can occur: digit <= digit_mask) are written, scanned and read — a pass
    // of 6 bits run by the 8-bit instantiation used to write, scan and read 256 rows, 192 of them all zero.
    // (All sums first, then the stores: with each store next to its sum hipcc waited for the first store
    // to complete before the LDS reads of the second.)
    uint32_t sum[DPT];
#pragma unroll
    for (int q = 0; q < DPT; q++) {
        sum[q] = 0;
#pragma unroll
        for (int c = 0; c < COPIES; c++) sum[q] += s_hist[c][threadIdx.x + q * SORT_THREADS];
    }
#pragma unroll
    for (int q = 0; q < DPT; q++) {
        const uint32_t digit = threadIdx.x + q * SORT_THREADS;
        if (digit <= digit_mask) ghist[(uint64_t)digit * num_blocks + block] = sum[q];
    }
}

// The histogram of the frame's COMPACTING first pass without touching the keys: the preprocess kernel
// counted the first digit per 1024-slot chunk while the keys were in its registers (PreOut::chunk_hist, two
// 16-bit counts per word, 1 KB per chunk); a tile of the pass is TILE / 1024 whole chunks, so its row entries
// are the sums of that many 1 KB rows.  Same output and same workgroup -> tile order as k_sort_hist<.., COMPACT>
// (which read 4 KB of keys per chunk behind 1024 LDS atomics: 21 us at 10 M, 95 us at 50 M).  Chunks whose
// visible count is 0 may be block-culled: their rows are stale and skipped, like their keys.
template <int RB, int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist_chunks(const uint32_t *__restrict__ chunk_hist, SortCount sc,
                                                                   uint32_t digit_mask, uint32_t *__restrict__ ghist,
                                                                   const uint32_t *__restrict__ chunk_vis, uint32_t num_blocks,
                                                                   uint32_t xcd_chunk, uint32_t hist_words) {
    constexpr uint32_t TILE = SORT_THREADS * ITEMS;
    constexpr uint32_t C = TILE / PP_CHUNK;
    constexpr uint32_t R = 1u << RB;
    constexpr uint32_t WPT = R > 2u * SORT_THREADS ? R / (2u * SORT_THREADS) : 1u;   // words of a chunk's row per thread (2 bins each)
    static_assert(TILE % PP_CHUNK == 0 && SORT_THREADS == PP_THREADS && R <= (uint32_t)PRE_HIST_BINS, "whole chunks per tile");
    const uint32_t count = sc.get();
    const uint32_t block = scatter_tile_of(blockIdx.x, xcd_chunk);
    if ((uint64_t)block * TILE >= count) return;
    const uint32_t chunk0 = block * C;
    const uint32_t live = (uint32_t)(((uint64_t)count - (uint64_t)block * TILE + PP_CHUNK - 1) / PP_CHUNK);   // chunks of this tile that hold slots
    uint32_t w[WPT][C];
    bool ok[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) ok[c] = c < live && chunk_vis[chunk0 + (c < live ? c : 0u)] != 0u;
#pragma unroll
    for (uint32_t q = 0; q < WPT; q++)
#pragma unroll
        for (uint32_t c = 0; c < C; c++)
            w[q][c] = chunk_hist[(uint64_t)(chunk0 + (ok[c] ? c : 0u)) * hist_words + q * SORT_THREADS + threadIdx.x];
#pragma unroll
    for (uint32_t q = 0; q < WPT; q++) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            lo += ok[c] ? w[q][c] & 0xffffu : 0u;
            hi += ok[c] ? w[q][c] >> 16 : 0u;
        }
        const uint32_t d = 2u * (q * SORT_THREADS + threadIdx.x);
        if (d <= digit_mask) {     // live rows only (k_sort_hist); the mask is 2^b - 1: row d + 1 is live with row d
            ghist[(uint64_t)d * num_blocks + block] = lo;
            if (digit_mask) ghist[(uint64_t)(d + 1u) * num_blocks + block] = hi;
        }
    }
}

// One workgroup per digit: exclusive scan of its row (over the blocks that hold data) in place; row
// total out.  1024 threads x 4 consecutive values = 4096 blocks per step: the rows of every frame
// size of the bench (up to 50 M Gaussians: 27 k blocks) take a few dependent steps — with 256 values
// per step the 10 M tile sort's rows took 21 (A/B on one box: 10 M frame -0.5 %, 1 M unchanged).
constexpr int SCAN_ROWS_THREADS = 1024;
template <int TILE>
__global__ __launch_bounds__(SCAN_ROWS_THREADS) void k_sort_scan_rows(uint32_t *__restrict__ ghist, uint32_t row_stride,
                                                                      SortCount sc,
                                                                      uint32_t *__restrict__ digit_totals) {
    constexpr uint32_t PER = 4;
    __shared__ uint32_t s_wave[SCAN_ROWS_THREADS / WAVE];
    const uint32_t count = sc.get();
    const uint32_t num_blocks = (uint32_t)(((uint64_t)count + TILE - 1) / TILE);
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t *row = ghist + (uint64_t)blockIdx.x * row_stride;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_blocks; base += SCAN_ROWS_THREADS * PER) {
        const uint32_t i0 = base + threadIdx.x * PER;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            v[k] = i0 + k < num_blocks ? row[i0 + k] : 0u;
            sum += v[k];
        }
        const uint32_t inc = wave_inclusive_scan(sum, lane);
        if (lane == 63u) s_wave[wid] = inc;
        __syncthreads();
        uint32_t wave_off = 0, tot = 0;
#pragma unroll
        for (uint32_t k = 0; k < SCAN_ROWS_THREADS / WAVE; k++) {
            const uint32_t x = s_wave[k];
            if (k < wid) wave_off += x;
            tot += x;
        }
        uint32_t run = carry + wave_off + inc - sum;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            if (i0 + k < num_blocks) row[i0 + k] = run;
            run += v[k];
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) digit_totals[blockIdx.x] = carry;
}

// The same scan for SHORT rows (up to 256 blocks = one step: the depth sort's rows at 1 M): one WAVE
// per row, four rows per workgroup, no LDS and no barrier.  Worth 0.8 us per launch at 1 M (A/B on
// one box: depth sort 60.9 -> 58.6 us); rows that need several dependent steps per wave are slower
// this way than with the 4096-wide workgroup (tile sort at 1 M, 780 blocks: 50.2 -> 52.4 us).
constexpr uint32_t SCAN_ROWS_SMALL_MAX = 256;      // blocks per row up to which the wave-per-row kernel is used: one step
template <int TILE>
__global__ __launch_bounds__(256) void k_sort_scan_rows_small(uint32_t *__restrict__ ghist, uint32_t row_stride,
                                                              SortCount sc, uint32_t *__restrict__ digit_totals,
                                                              uint32_t rows) {
    constexpr uint32_t PER = 4;
    const uint32_t lane = threadIdx.x & 63u, r = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r >= rows) return;
    const uint32_t count = sc.get();
    const uint32_t num_blocks = (uint32_t)(((uint64_t)count + TILE - 1) / TILE);
    uint32_t *row = ghist + (uint64_t)r * row_stride;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_blocks; base += WAVE * PER) {
        const uint32_t i0 = base + lane * PER;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            v[k] = i0 + k < num_blocks ? row[i0 + k] : 0u;
            sum += v[k];
        }
        const uint32_t inc = wave_inclusive_scan(sum, lane);
        uint32_t run = carry + inc - sum;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            if (i0 + k < num_blocks) row[i0 + k] = run;
            run += v[k];
        }
        carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    if (lane == 0) digit_totals[r] = carry;
}

// Stable scatter.  Element order inside a workgroup tile: wave w owns ITEMS*64 consecutive
// elements, round k of the wave covers 64 consecutive elements, lane order inside a round; ranks
// are assigned in exactly that order, so equal digits keep their order.
// FAST_RANK: on gfx950 a returning LDS atomic (ds_add_rtn_u32) issued by one wave instruction
// hands out its pre-add values in increasing lane order when several lanes hit the same address
// (measured: tools/mb/mb_ldsatomic.hip, 0 violations in 5e7 operations).  One such atomic per key
// then IS the stable rank.  This is not an architectural guarantee, so gs_device_create probes it
// with this kernel's own access pattern for both digit widths and with partially masked waves
// (k_probe_lds_atomic_order<RB>) and the host falls back to the ballot-based ranking if the probe
// ever fails.

// LDS of a scatter workgroup
template <typename K, int RB, int ITEMS>
struct ScatterShared {
    static constexpr int TILE = SORT_THREADS * ITEMS;
    static constexpr int R = 1 << RB;
    uint32_t wave_hist[4][R];      // per-wave digit counters
    uint32_t delta[R];             // (global offset of this block's digit run) - (its start in the LDS tile)
    uint32_t scan[4];
    K keys[TILE];
    uint32_t vals[TILE];
};

template <typename K, int RB, int ITEMS>
__device__ __forceinline__ void scatter_clear(ScatterShared<K, RB, ITEMS> &sh) {
    constexpr int DPT = (1 << RB) / SORT_THREADS;
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int q = 0; q < DPT; q++) sh.wave_hist[w][threadIdx.x + q * SORT_THREADS] = 0;
    __syncthreads();
}

// Everything after a workgroup holds its tile in registers (key[k], val[k] = element
// wave_off + k * 64 + lane of the tile; padding = all-ones key): rank, local reorder, coalesced store.
// KO / ko_shift: the keys this pass WRITES may be narrower than the keys it sorts — the pass before the
// last one of the frame's depth sort stores only the bits the last pass still needs (key >> ko_shift
// as u16: 9 of 27 bits are left), which saves 2 bytes per element written and 2 x 2 bytes read.
template <typename K, bool FAST_RANK, int RB, bool COMPACT, int ITEMS, typename KO = K>
__device__ __forceinline__ void scatter_ranked(ScatterShared<K, RB, ITEMS> &sh, K (&key)[ITEMS], uint32_t (&val)[ITEMS],
                                               uint32_t in_tile, uint32_t block, uint32_t num_blocks,
                                               KO *__restrict__ keys_out, uint32_t ko_shift,
                                               uint32_t *__restrict__ vals_out, uint32_t shift, uint32_t digit_mask,
                                               const uint32_t *__restrict__ ghist,
                                               const uint32_t *__restrict__ digit_totals,
                                               uint32_t *__restrict__ visible_out, uint32_t *rank_fault = nullptr,
                                               uint32_t watch_round = 0u, uint32_t *bucket_max_out = nullptr,
                                               uint32_t *bucket_start_out = nullptr) {
    constexpr int R = 1 << RB;
    constexpr int DPT = R / SORT_THREADS;        // digits per thread: thread t owns digits [t*DPT, t*DPT+DPT)
    auto &s_wave_hist = sh.wave_hist;
    auto &s_delta = sh.delta;
    auto &s_scan = sh.scan;
    auto &s_keys = sh.keys;
    auto &s_vals = sh.vals;
    const uint32_t tid = threadIdx.x, wid = tid >> 6;
    uint32_t rank[ITEMS];
    if constexpr (FAST_RANK) {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            uint32_t d = (uint32_t)(key[k] >> shift) & digit_mask;
            if (!COMPACT || (uint32_t)key[k] != SORT_INVALID_KEY) rank[k] = atomicAdd(&s_wave_hist[wid][d], 1u);
        }
        // Watchdog of the bet.  What the bet claims is the order INSIDE one wave instruction: the
        // lanes that hit one counter receive consecutive values in lane order, i.e. rank = (value the lowest such lane
        // got) + (number of such lanes below me).  ONE workgroup of every pass (the caller picks it from the frame
        // generation and hands it the word: every tile is sampled over a few hundred frames, not just tile 0 as in
        // round 4) checks exactly that for ONE of its rounds (`watch_round`, likewise rotating) against the ballot.
        // A mismatch sets FrameState.rank_fault; the frame result carries it to the host, which switches the device
        // to the ballot-based rank for good.  (rank_fault[1]: test hook, makes the expectation wrong.)
        if (rank_fault) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                if ((uint32_t)k != watch_round) continue;      // uniform: static register indexing
                const uint32_t dk = (uint32_t)(key[k] >> shift) & digit_mask;
                const bool livek = !COMPACT || (uint32_t)key[k] != SORT_INVALID_KEY;
                uint64_t peers = __ballot(livek);
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    const uint64_t m = __ballot((dk >> b) & 1u);
                    peers &= ((dk >> b) & 1u) ? m : ~m;
                }
                const uint32_t first = (uint32_t)__shfl((int)rank[k], peers ? (int)__builtin_ctzll(peers) : 0, WAVE);
                const bool bad = livek && rank[k] != first + mbcnt(peers) + rank_fault[1];
                if (__any(bad) && (threadIdx.x & 63u) == 0u) atomicOr(rank_fault, 1u);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            uint32_t d = (uint32_t)(key[k] >> shift) & digit_mask;
            const bool live = !COMPACT || (uint32_t)key[k] != SORT_INVALID_KEY;
            // wave64 match-any on the digit: peers = live lanes holding the same digit
            uint64_t peers = __ballot(live);
#pragma unroll
            for (int b = 0; b < RB; b++) {
                uint64_t m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            uint32_t before = mbcnt(peers);             // same-digit live lanes below me
            uint32_t old = s_wave_hist[wid][d];
            rank[k] = old + before;
            __builtin_amdgcn_wave_barrier();
            if (live && before == 0u) s_wave_hist[wid][d] = old + (uint32_t)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    // per digit: offsets of each wave inside the digit run, block digit count
    uint32_t live_total;      // elements this block scatters
    uint32_t bin0[DPT];       // start of my digits' runs in the LDS tile
    {
        uint32_t c[DPT][4], dc[DPT], mine = 0;
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            const uint32_t digit = tid * DPT + q;
#pragma unroll
            for (int w = 0; w < 4; w++) c[q][w] = s_wave_hist[w][digit];
            dc[q] = (c[q][0] + c[q][1]) + (c[q][2] + c[q][3]);
            mine += dc[q];
        }
        uint32_t bin_start = block_exclusive_scan_256(mine, s_scan, live_total);
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            const uint32_t digit = tid * DPT + q;
            bin0[q] = bin_start;
            s_wave_hist[0][digit] = bin_start;
            s_wave_hist[1][digit] = bin_start + c[q][0];
            s_wave_hist[2][digit] = bin_start + c[q][0] + c[q][1];
            s_wave_hist[3][digit] = bin_start + c[q][0] + c[q][1] + c[q][2];
            bin_start += dc[q];
        }
    }
    // global base of every digit: sum of totals of smaller digits + this block's row prefix
    {
        uint32_t tot[DPT], mine = 0;
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            // rows past the digit mask do not exist this pass (nobody wrote, scanned or totalled them): the load
            // goes to a clamped row and is masked afterwards — no branch around it (a load behind a branch gets
            // its own s_waitcnt: DESIGN.md §4.2)
            const uint32_t d = tid * DPT + q;
            const uint32_t t = digit_totals[d <= digit_mask ? d : digit_mask];
            tot[q] = d <= digit_mask ? t : 0u;
            mine += tot[q];
        }
        uint32_t gh[DPT];
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            const uint32_t d = tid * DPT + q;
            gh[q] = ghist[(uint64_t)(d <= digit_mask ? d : digit_mask) * num_blocks + block];
        }
        uint32_t all;
        uint32_t digit_base = block_exclusive_scan_256(mine, s_scan, all);
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            const uint32_t digit = tid * DPT + q;
            s_delta[digit] = digit_base + (digit <= digit_mask ? gh[q] : 0u) - bin0[q];
            // the MSD-first sorts: where every digit's bucket starts, for k_bucket_sort (tile 0 writes the table)
            if (bucket_start_out && block == 0u && digit <= digit_mask) bucket_start_out[digit] = digit_base;
            digit_base += tot[q];
        }
        if constexpr (COMPACT)
            if (block == 0 && tid == 0) *visible_out = all;     // V = everything the pass ranked
        // the LSD sort's pass on the TOP digit reports the largest digit total = the largest bucket an MSD-first
        // sort of these keys would hand to k_bucket_sort (the host's choice between the two, gs3d.hip)
        if (bucket_max_out && block == 0u) {
            uint32_t m = 0;
#pragma unroll
            for (int q = 0; q < DPT; q++) m = tot[q] > m ? tot[q] : m;
            m = wave_reduce_max(m);
            if ((tid & 63u) == 0u) s_scan[wid] = m;
            __syncthreads();
            if (tid == 0u) {
                const uint32_t a = s_scan[0] > s_scan[1] ? s_scan[0] : s_scan[1], b = s_scan[2] > s_scan[3] ? s_scan[2] : s_scan[3];
                *bucket_max_out = a > b ? a : b;
            }
        }
    }
    __syncthreads();

    // Local reorder through LDS so that each digit run is written by consecutive lanes.  Both loops
    // below are written in BATCHES of CH elements — all LDS reads of a batch, a scheduling barrier, then
    // what depends on them — because the ISA of rounds 1-2 (tools/isa_audit.py) ran every element as its
    // own chain: `ds_read base; s_waitcnt; ds_write` sixteen times here (hipcc will not move a read of
    // wave_hist above a write to keys / vals of the same LDS struct), and in the store loop, behind a
    // branch per element, `ds_read key; wait; ds_read delta; wait; store; ds_read val; wait; store` —
    // 3 dependent LDS round trips x ITEMS per thread, in kernels that have 2-5 waves per SIMD to hide them.
    // A/B on one box (gpurun_out/r03n/ab.log): depth sort 0.157 -> 0.152 ms at 10 M, 0.575 -> 0.551 ms at
    // 50 M; tile sort 0.571 -> 0.538 ms at 4K, 0.247 -> 0.244 ms at 1080p — occupancy had hidden most of it.
    constexpr int CH = ITEMS < 16 ? ITEMS : 16;
#pragma unroll
    for (int k0 = 0; k0 < ITEMS; k0 += CH) {
        uint32_t wbase[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) wbase[j] = s_wave_hist[wid][(uint32_t)(key[k0 + j] >> shift) & digit_mask];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; j++) {
            if (COMPACT && (uint32_t)key[k0 + j] == SORT_INVALID_KEY) continue;
            const uint32_t pos = wbase[j] + rank[k0 + j];
            s_keys[pos] = key[k0 + j];
            s_vals[pos] = val[k0 + j];
        }
    }
    __syncthreads();
    // without COMPACT the padding of a partial tile carries the all-ones key and sorts to the end
    const uint32_t live = COMPACT ? live_total : in_tile;
#pragma unroll
    for (int k0 = 0; k0 < ITEMS; k0 += CH) {
        K kk[CH];
        uint32_t vv[CH], dl[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const uint32_t pos = (k0 + j) * SORT_THREADS + tid;
            const uint32_t p = pos < live ? pos : 0u;          // (no branch around the reads: slot 0 is always there)
            kk[j] = s_keys[p];
            vv[j] = s_vals[p];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; j++) dl[j] = s_delta[(uint32_t)(kk[j] >> shift) & digit_mask];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const uint32_t pos = (k0 + j) * SORT_THREADS + tid;
            if (pos < live) {
                const uint32_t dst = dl[j] + pos;
                if (keys_out) keys_out[dst] = (KO)(kk[j] >> ko_shift);      // null: nobody reads the keys of this pass (last depth pass)
                vals_out[dst] = vv[j];
            }
        }
    }
}

template <typename K, bool FAST_RANK, int RB, bool COMPACT, int ITEMS, typename KO = K, bool PRED = false>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(
    const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, KO *__restrict__ keys_out, uint32_t ko_shift,
    uint32_t *__restrict__ vals_out, SortCount sc, uint32_t shift, uint32_t digit_mask,
    const uint32_t *__restrict__ ghist, const uint32_t *__restrict__ digit_totals,
    const uint32_t *__restrict__ chunk_vis, uint32_t *__restrict__ visible_out, uint32_t num_tiles,
    uint32_t xcd_chunk_nt, uint32_t *rank_fault, uint32_t watch, uint32_t *bucket_max_out, uint32_t *bucket_start_out,
    CompactPred pred = CompactPred()) {
    constexpr int TILE = SORT_THREADS * ITEMS;
    __shared__ ScatterShared<K, RB, ITEMS> sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t count = sc.get();
    const uint32_t live_tiles = (uint32_t)(((uint64_t)count + TILE - 1) / TILE);   // the grid comes from an upper bound
    const bool nt_inputs = (xcd_chunk_nt >> 31) != 0u;
    const uint32_t xcd_chunk = xcd_chunk_nt & 0x7fffffffu;
    const uint32_t block = scatter_tile_of(blockIdx.x, xcd_chunk);
    if (block >= live_tiles) {
        // a list frame whose list is empty (every block culled) has no tile at all: V = 0 must still be
        // published, or the frame state keeps the previous frame's count
        if constexpr (COMPACT)
            if (live_tiles == 0u && blockIdx.x == 0u && tid == 0u) *visible_out = 0u;
        return;
    }
    scatter_clear(sh);

    const uint32_t tile_base = block * TILE;
    const uint32_t in_tile = count - tile_base < (uint32_t)TILE ? count - tile_base : (uint32_t)TILE;
    const uint32_t wave_off = wid * (ITEMS * WAVE);
    // COMPACT: 16 rounds of a wave (1024 elements) are exactly one preprocess chunk
    static_assert(!COMPACT || ITEMS % 16 == 0, "compacting pass: whole chunks per wave");
    bool chunk_ok[COMPACT ? ITEMS / 16 : 1];
    if constexpr (COMPACT) {
#pragma unroll
        for (int c = 0; c < ITEMS / 16; c++)
            chunk_ok[c] = wave_off + c * 1024u < in_tile && chunk_vis[(tile_base + wave_off + c * 1024u) >> 10] != 0u;
    }
    K key[ITEMS];
    uint32_t val[ITEMS];
    // Inputs larger than the L2s are read with the non-temporal policy (xcd_chunk's top bit, set by the
    // host from its bound of the count): they are read exactly once, and left alone the lines of this
    // pass's OUTPUT — the next pass's input — survive in the caches instead (10 M: tile sort 0.237 ->
    // 0.219 ms; 50 M: depth sort 0.552 -> 0.523 ms).  One uniform branch around the whole batch of
    // loads, not a flag per load (that put a branch and a wait behind every load).
    if (!COMPACT && nt_inputs) {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t e = wave_off + k * WAVE + lane;
            const bool ok = e < in_tile;
            key[k] = ok ? __builtin_nontemporal_load(&keys_in[tile_base + e]) : (K)~(K)0;
            val[k] = ok ? __builtin_nontemporal_load(&vals_in[tile_base + e]) : 0u;
        }
    } else
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t e = wave_off + k * WAVE + lane;     // element of the tile (no 32-bit wrap near 2^32)
        const bool ok = e < in_tile && (!COMPACT || chunk_ok[COMPACT ? k / 16 : 0]);
        if constexpr (COMPACT) {
            // no branch around the load: a skipped chunk (block-culled: stale keys) reads the tile's first
            // element instead — always there, its line is fetched anyway — and is masked afterwards.  With
            // the load inside `ok ? load : ~0` hipcc emitted a branch and an s_waitcnt vmcnt(0) per key:
            // 32 dependent round trips per thread in the frame's first depth pass.
            // (non-temporal: the dense keys are dead after this read; 50 M: depth sort -25 us)
            const K raw = __builtin_nontemporal_load(&keys_in[tile_base + (ok ? e : 0u)]);
            key[k] = ok ? raw : (K)~(K)0;
            val[k] = tile_base + e;

        } else {
            key[k] = ok ? keys_in[tile_base + e] : (K)~(K)0;
            val[k] = ok ? vals_in[tile_base + e] : 0u;
        }
    }
    if constexpr (PRED) {
        // A partitioned frame: this round's side of the depth threshold only — the other elements become "culled".  The
        // keep bits of the wave's 64 slots per round are two consecutive words (loaded unconditionally, all rounds back
        // to back: keep_bits is always a readable array; side 0 ignores them).
        static_assert(COMPACT, "the predicate belongs to the compacting pass");
        const uint32_t tau = *pred.tau_dev;
        uint32_t word[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            uint32_t e = wave_off + (uint32_t)k * WAVE + lane;
            e = e < in_tile ? e : 0u;
            word[k] = pred.keep_bits[(tile_base + e) >> 5];
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t kk = (uint32_t)key[k];
            const bool mine = pred.side == 0u ? kk < tau
                                              : kk >= tau && kk != SORT_INVALID_KEY && ((word[k] >> (lane & 31u)) & 1u) != 0u;
            key[k] = mine ? key[k] : (K)~(K)0;
        }
    }
    // the watchdog's sample of this pass: tile (watch mod live tiles), round (watch / live tiles) mod ITEMS
    uint32_t *rf = rank_fault && block == watch % live_tiles ? rank_fault : (uint32_t *)nullptr;
    scatter_ranked<K, FAST_RANK, RB, COMPACT, ITEMS, KO>(sh, key, val, in_tile, block, num_tiles, keys_out, ko_shift, vals_out,
                                                         shift, digit_mask, ghist, digit_totals, visible_out, rf,
                                                         (watch / live_tiles) % (uint32_t)ITEMS, bucket_max_out, bucket_start_out);
}

// ---------------------------------------------------------------------------------------------
// Bucket sort: the second half of an MSD-first sort (round 5).  A scatter pass on the TOP digit (its
// histogram comes for free: the preprocess kernel / k_pairs_emit count it while the keys sit in
// registers) leaves the elements partitioned into at most 1024 buckets, each contiguous and in input
// order; one 1024-thread workgroup per bucket then sorts its bucket on the remaining low bits
// entirely on the CU — keys and values in registers, one returning LDS atomic per key and pass for
// the rank (as in scatter_ranked), one exchange through LDS per pass — and writes the final order.
// No global histogram, no row scan and no trip through HBM for the low digits: the LSD sort's
// 3 x (histogram, row scan, scatter) become (histogram sum, row scan, scatter, bucket sort), 9
// dependent launches -> 4 at 1 M Gaussians.
// A bucket of up to BKT_CAP = 30 720 elements takes the register path.  A larger one (depths that
// collapse into a few top digits: a wall seen head-on, `narrow_range`, `same_depth`) is still sorted
// correctly, by the same workgroup in chunks through global scratch (bucket_sort_chunked: two sweeps
// per pass, one workgroup's bandwidth) — the device never needs the host to notice.  The host does
// notice, one or two frames later (FrameResult::depth_bucket_max), and goes back to the LSD passes
// while buckets do not fit.
// ---------------------------------------------------------------------------------------------
// Diagnostic build only (tools/mb/mb_bucket.hip defines GS3D_BKT_STAMPS): s_memrealtime (100 MHz) at the phase boundaries of the
// register path, written by thread 0 of every bucket to a buffer nothing else reads.  Absent from the product.
#ifdef GS3D_BKT_STAMPS
#define GS_BKT_STAMP(io, bucket, i)                                                                  \
    do {                                                                                              \
        if (threadIdx.x == 0u && (io).stamps) (io).stamps[(bucket) * 16u + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define GS_BKT_STAMP(io, bucket, i) do { } while (0)
#endif

// Two workgroup shapes (template parameter T = threads).  T = 1024, 30 keys per lane: few, large buckets — the depth
// sort, where the largest bucket is the critical path and wants a whole CU.  T = 256, 32 keys per lane: many small
// buckets — the tile sort's ~1000 buckets of a few thousand pairs, where a workgroup's fixed costs (clearing and
// scanning its counters, barriers) dominate and four workgroups per CU hide them (33 KB of LDS instead of 157).
constexpr int BKT_THREADS = 1024;          // the large shape: gs_sort_info::bucket_capacity refers to it
constexpr int BKT_THREADS_SMALL = 256;
template <int T> struct BucketCfg {
    static_assert(T == 1024 || T == 256, "workgroup shapes of k_bucket_sort");
    static constexpr int THREADS = T, WAVES = T / WAVE;
    static constexpr int ITEMS_MAX = T == 1024 ? 30 : 32;
    static constexpr uint32_t CAP = (uint32_t)T * ITEMS_MAX;      // elements of the register path
    static constexpr int CHUNK_ITEMS = 16;                         // chunked path: 16 T elements per step
};
constexpr uint32_t BKT_CAP = BucketCfg<BKT_THREADS>::CAP;
constexpr uint32_t BKT_CAP_SMALL = BucketCfg<BKT_THREADS_SMALL>::CAP;

struct BucketSortIO {
    const uint32_t *totals;     // [nb] bucket sizes = digit totals of the scatter pass that made the buckets
    const uint32_t *starts;     // [nb] where every bucket starts (written by tile 0 of that scatter pass)
    uint32_t nb;                // buckets (<= BKT_THREADS = 1024: the 10-bit top digit) = grid size
    const void *keys_in;        // [count] keys after the top-digit scatter (u32, or u16 tile ids)
    const uint32_t *vals_in;
    void *keys_tmp;             // scratch of the chunked path when two passes are left (in -> tmp -> out)
    uint32_t *vals_tmp;
    void *keys_out;             // may be null: nobody reads the sorted keys (depth sort)
    uint32_t *vals_out;
    uint32_t low_bits;          // key bits [0, low_bits) are still unsorted; <= 2 * RB
    uint32_t *bucket_max;       // may be null: receives the largest bucket size (workgroup 0)
    uint32_t *ranges;           // may be null: per-tile [start, end) (tile sort: tile = bucket << low_bits | low digit)
    uint32_t num_tiles;
    uint32_t *rank_fault;       // may be null: watchdog word of the LDS-atomic rank (FrameState::rank_fault, [1] = test hook)
    uint32_t watch;             // which (bucket, round) the watchdog samples this frame (frame generation)
#ifdef GS3D_BKT_STAMPS
    unsigned long long *stamps; // [nb][16]
#endif
};

template <int RB, int T>
struct BucketShared {
    static constexpr int R = 1 << RB;
    uint32_t wave_hist[BucketCfg<T>::WAVES][R];   // per-wave digit counters, then each wave's base per digit
    uint32_t xbuf[BucketCfg<T>::CAP];             // exchange buffer of the register path
    uint32_t dbase[R];                            // chunked path: running output offset of every digit
    uint32_t scan[BucketCfg<T>::WAVES];
    uint32_t bcast[4];
};

// exclusive scan over the T threads of a bucket workgroup; ends with `smem` reusable
template <int T>
__device__ __forceinline__ uint32_t block_exclusive_scan_t(uint32_t v, uint32_t *smem, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63u) smem[wid] = inc;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)(T / WAVE); k++) {
        const uint32_t x = smem[k];
        if (k < wid) wave_off += x;
        tot += x;
    }
    total = tot;
    __syncthreads();
    return wave_off + inc - v;
}

// Stable rank of the wave's ITEMS x 64 elements inside their digit: element (k, lane) precedes (k', lane') when
// k < k' or k == k' and lane < lane'.  `hist` is the wave's private row of counters, zero on entry; element
// (k, lane) is live when k * 64 + lane < nlive (dead elements are neither counted nor ranked); its digit is
// (key[k] >> shift) & mask — recomputed where needed instead of kept: the register path holds 28 keys, 28 values
// and 28 ranks per lane as it is.
template <int RB, int ITEMS, bool FAST_RANK>
__device__ __forceinline__ void bucket_rank(uint32_t *hist, const uint32_t (&key)[ITEMS], uint32_t shift, uint32_t mask,
                                            uint32_t nlive, uint32_t (&rank)[ITEMS], uint32_t *rank_fault) {
    const uint32_t lane = threadIdx.x & 63u;
    if constexpr (FAST_RANK) {
#pragma unroll
        for (int k = 0; k < ITEMS; k++)
            if (k * WAVE + lane < nlive) rank[k] = atomicAdd(&hist[(key[k] >> shift) & mask], 1u);
        // the bet's watchdog (scatter_ranked): the bucket the frame generation picks checks round 0 of every wave and
        // pass — the round whose expected value needs no second set of counters — against the ballot-based rank
        if (rank_fault) {
            const uint32_t d0 = (key[0] >> shift) & mask;
            const bool live0 = lane < nlive;
            uint64_t peers = __ballot(live0);
#pragma unroll
            for (int b = 0; b < RB; b++) {
                const uint64_t m = __ballot((d0 >> b) & 1u);
                peers &= ((d0 >> b) & 1u) ? m : ~m;
            }
            const bool bad = live0 && rank[0] != mbcnt(peers) + rank_fault[1];
            if (__any(bad) && lane == 0u) atomicOr(rank_fault, 1u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t d = (key[k] >> shift) & mask;
            const bool live = k * WAVE + lane < nlive;
            uint64_t peers = __ballot(live);
#pragma unroll
            for (int b = 0; b < RB; b++) {
                const uint64_t m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            const uint32_t before = mbcnt(peers);
            const uint32_t old = hist[d];
            rank[k] = old + before;
            __builtin_amdgcn_wave_barrier();
            if (live && before == 0u) hist[d] = old + (uint32_t)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// digit widths balanced over the passes, as run_sort_items does (18 bits -> 9 + 9, 11 -> 6 + 5)
__device__ __forceinline__ uint32_t bucket_pass_bits(uint32_t low_bits, uint32_t shift, uint32_t passes, uint32_t p) {
    return (low_bits - shift + (passes - p) - 1u) / (passes - p);
}

// per-tile ranges of a tile sort's last pass: thread `digit` holds its digit's offset and count inside the bucket
__device__ __forceinline__ void bucket_ranges(const BucketSortIO &io, uint32_t bucket, uint32_t start, uint32_t digit,
                                              uint32_t base, uint32_t tot) {
    const uint32_t tile = (bucket << io.low_bits) | digit;
    if (tot != 0u && tile < io.num_tiles) {
        io.ranges[2u * tile] = start + base;
        io.ranges[2u * tile + 1u] = start + base + tot;
    }
}

// counts of the 16 waves -> every wave's base per digit (in place); returns through `base` / `tot` the digit's
// offset inside the bucket's order of this pass and its count, for the thread that owns the digit (tid < R);
// `dbase` (chunked path): running offsets carried from chunk to chunk instead of a scan.  Barriers inside.
template <int RB, int T, bool RUNNING>
__device__ __forceinline__ void bucket_wave_bases(BucketShared<RB, T> &sh, uint32_t &base, uint32_t &tot) {
    constexpr int R = 1 << RB, WAVES = BucketCfg<T>::WAVES;
    static_assert(R <= T, "one thread per digit");
    const uint32_t tid = threadIdx.x;
    uint32_t c[WAVES];
    tot = 0;
    if (tid < (uint32_t)R) {
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            c[w] = sh.wave_hist[w][tid];
            tot += c[w];
        }
    }
    if constexpr (RUNNING) {
        base = tid < (uint32_t)R ? sh.dbase[tid] : 0u;
    } else {
        uint32_t all;
        base = block_exclusive_scan_t<T>(tid < (uint32_t)R ? tot : 0u, sh.scan, all);
    }
    if (tid < (uint32_t)R) {
        uint32_t run = base;
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            sh.wave_hist[w][tid] = run;
            run += c[w];
        }
        if constexpr (RUNNING) sh.dbase[tid] = run;
    }
    __syncthreads();
}

// The register path: the whole bucket (size <= 1024 * ITEMS) lives in the workgroup's registers; wave w owns
// elements [w * ITEMS * 64, (w + 1) * ITEMS * 64), round k of a wave 64 consecutive ones.  Only the KEYS travel: what
// is sorted is (remaining key bits, position in the bucket), and the values are gathered once at the end through the
// sorted positions (the bucket's values are 4 bytes x size, L2-resident: the top-digit scatter has just written
// them).  Pass 1 of 2 drops the digit it has sorted and packs (next digit << 15 | position) into one word — a
// position needs 15 bits, BKT_CAP <= 2^15 — so a lane holds ITEMS keys and ITEMS ranks and nothing else (with the
// values in registers too the 28-item instantiation spilled 4.6 KB per lane), and a pass is ONE exchange through LDS.
template <typename K, int RB, int T, int ITEMS, bool FAST_RANK>
__device__ __forceinline__ void bucket_sort_fast(BucketShared<RB, T> &sh, const BucketSortIO &io, uint32_t bucket,
                                                 uint32_t start, uint32_t size, uint32_t *rank_fault) {
    constexpr int R = 1 << RB, WAVES = BucketCfg<T>::WAVES;
    static_assert(BucketCfg<T>::CAP <= (1u << 15) && RB + 15 <= 32, "(digit << 15 | position) in one word");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t wave_off = wid * (uint32_t)(ITEMS * WAVE);
    const uint32_t nlive = size > wave_off ? size - wave_off : 0u;     // live elements of this wave (the rest is padding)
    const K *kin = (const K *)io.keys_in + start;
    const uint32_t *vin = io.vals_in + start;
    K *kout = io.keys_out ? (K *)io.keys_out + start : (K *)nullptr;
    uint32_t *vout = io.vals_out + start;
    const uint32_t passes = (io.low_bits + RB - 1) / RB;    // 0, 1 or 2
    if (passes == 0u) {   // the top digit was the whole key: the bucket is in its final order
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t i = k * T + tid;
            if (i < size) {
                vout[i] = vin[i];
                if (kout) kout[i] = kin[i];
            }
        }
        if (io.ranges && tid == 0u) bucket_ranges(io, bucket, start, 0u, 0u, size);
        return;
    }
    uint32_t key[ITEMS], rank[ITEMS];
    GS_BKT_STAMP(io, bucket, 0);
    // (no branch around a load: a load behind a branch gets its own wait; element 0 exists, size > 0)
#pragma unroll
    for (int k = 0; k < ITEMS; k++) key[k] = (uint32_t)kin[k * WAVE + lane < nlive ? wave_off + k * WAVE + lane : 0u];
    // one counting pass over the registers: ranks, wave bases, final place of every element in `rank`
    auto pass = [&](uint32_t shift, uint32_t mask, bool ranges, uint32_t stamp0) {
        (void)stamp0;
#pragma unroll
        for (int q = 0; q < WAVES * R / T; q++) (&sh.wave_hist[0][0])[tid + q * T] = 0u;
        __syncthreads();
        GS_BKT_STAMP(io, bucket, stamp0);
        bucket_rank<RB, ITEMS, FAST_RANK>(sh.wave_hist[wid], key, shift, mask, nlive, rank, rank_fault);
        __syncthreads();
        GS_BKT_STAMP(io, bucket, stamp0 + 1u);
        uint32_t base, tot;
        bucket_wave_bases<RB, T, false>(sh, base, tot);
        GS_BKT_STAMP(io, bucket, stamp0 + 2u);
        if (ranges && tid < (uint32_t)R) bucket_ranges(io, bucket, start, tid, base, tot);
        // all LDS reads of the batch first, then what depends on them (scatter_ranked's note on hipcc's chains)
#pragma unroll
        for (int k = 0; k < ITEMS; k++) rank[k] += sh.wave_hist[wid][(key[k] >> shift) & mask];
        __builtin_amdgcn_sched_barrier(0);
    };
    const uint32_t bits1 = bucket_pass_bits(io.low_bits, 0u, passes, 0u), bits2 = io.low_bits - bits1;
    pass(0u, (1u << bits1) - 1u, passes == 1u && io.ranges != nullptr, 1u);     // (ranges: the tile sort, one pass by contract)
    // what an element still needs: its position in the input bucket and, if a pass follows, that pass's digit
    const uint32_t mask2 = (1u << bits2) - 1u;
#pragma unroll
    for (int k = 0; k < ITEMS; k++)
        if (k * WAVE + lane < nlive) sh.xbuf[rank[k]] = (((key[k] >> bits1) & mask2) << 15) | (wave_off + k * WAVE + lane);
    __syncthreads();
    GS_BKT_STAMP(io, bucket, 4);
    if (passes == 2u) {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) key[k] = sh.xbuf[k * WAVE + lane < nlive ? wave_off + k * WAVE + lane : 0u];
        __syncthreads();
        pass(15u, mask2, false, 5u);
#pragma unroll
        for (int k = 0; k < ITEMS; k++)
            if (k * WAVE + lane < nlive) sh.xbuf[rank[k]] = key[k];
        __syncthreads();
    }
    GS_BKT_STAMP(io, bucket, 8);
    // xbuf[i] (low 15 bits) = position in the INPUT bucket of the element that belongs at place i.  The values follow
    // through LDS: read coalesced, parked at their input positions, picked up by the sorted positions, written
    // coalesced.  (Gathering them from global memory by position — 64 different sectors per wave instruction — made
    // this kernel three times slower: 33 us for a 21 000-key bucket.)
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = k * T + tid;
        rank[k] = sh.xbuf[i < size ? i : 0u] & 0x7fffu;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = k * T + tid;
        key[k] = vin[i < size ? i : 0u];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = k * T + tid;
        if (i < size) sh.xbuf[i] = key[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) key[k] = sh.xbuf[rank[k]];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = k * T + tid;
        if (i < size) vout[i] = key[k];
    }
    GS_BKT_STAMP(io, bucket, 9);
    if (kout) {     // the sorted keys likewise (the tile sort: its parity tap rebuilds the 64-bit keys from them)
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t i = k * T + tid;
            key[k] = (uint32_t)kin[i < size ? i : 0u];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t i = k * T + tid;
            if (i < size) sh.xbuf[i] = key[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; k++) key[k] = sh.xbuf[rank[k]];
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t i = k * T + tid;
            if (i < size) kout[i] = (K)key[k];
        }
    }
}

// The chunked path: a bucket of any size, by one workgroup, through global memory.  Per pass: sweep 1 counts the
// digits of the whole bucket, sweep 2 walks the bucket in order, 16 384 elements at a time, ranks each chunk like
// the register path and writes every element to its final place of the pass (running per-digit offsets in LDS).
// Slow — one CU's bandwidth, scattered stores — and only ever a fallback (see the section comment).
template <typename K, int RB, int T, bool FAST_RANK>
__device__ __forceinline__ void bucket_sort_chunked(BucketShared<RB, T> &sh, const BucketSortIO &io, uint32_t bucket,
                                                    uint32_t start, uint32_t size) {
    constexpr int R = 1 << RB, WAVES = BucketCfg<T>::WAVES;
    constexpr int ITEMS = BucketCfg<T>::CHUNK_ITEMS;
    constexpr uint32_t CH = (uint32_t)T * ITEMS;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t wave_off = wid * (uint32_t)(ITEMS * WAVE);
    const uint32_t passes = (io.low_bits + RB - 1) / RB;   // <= 2 (host)
    if (passes == 0u) {
        for (uint32_t i = tid; i < size; i += T) {
            io.vals_out[start + i] = io.vals_in[start + i];
            if (io.keys_out) ((K *)io.keys_out)[start + i] = ((const K *)io.keys_in)[start + i];
        }
        if (io.ranges && tid == 0u) bucket_ranges(io, bucket, start, 0u, 0u, size);
        return;
    }
    uint32_t shift = 0;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t bits = bucket_pass_bits(io.low_bits, shift, passes, p);
        const uint32_t mask = (1u << bits) - 1u;
        const bool last = p + 1u == passes;
        const K *src_k = (p == 0u ? (const K *)io.keys_in : (const K *)io.keys_tmp) + start;
        const uint32_t *src_v = (p == 0u ? io.vals_in : (const uint32_t *)io.vals_tmp) + start;
        K *dst_k = last ? (io.keys_out ? (K *)io.keys_out + start : (K *)nullptr) : (K *)io.keys_tmp + start;
        uint32_t *dst_v = (last ? io.vals_out : io.vals_tmp) + start;
        // sweep 1: digit counts of the bucket
        if (tid < (uint32_t)R) sh.dbase[tid] = 0u;
        __syncthreads();
        for (uint32_t c0 = 0; c0 < size; c0 += CH) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const uint32_t e = c0 + k * T + tid;
                if (e < size) atomicAdd(&sh.dbase[((uint32_t)src_k[e] >> shift) & mask], 1u);
            }
        }
        __syncthreads();
        {
            const uint32_t tot = tid < (uint32_t)R ? sh.dbase[tid] : 0u;
            uint32_t all;
            const uint32_t base = block_exclusive_scan_t<T>(tot, sh.scan, all);
            if (tid < (uint32_t)R) {
                sh.dbase[tid] = base;
                if (last && io.ranges) bucket_ranges(io, bucket, start, tid, base, tot);
            }
        }
        __syncthreads();
        // sweep 2: stable placement, chunk by chunk
        for (uint32_t c0 = 0; c0 < size; c0 += CH) {
#pragma unroll
            for (int q = 0; q < WAVES * R / T; q++) (&sh.wave_hist[0][0])[tid + q * T] = 0u;
            __syncthreads();
            const uint32_t w0 = c0 + wave_off;
            const uint32_t nlive = size > w0 ? size - w0 : 0u;
            uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const uint32_t e = k * WAVE + lane < nlive ? w0 + k * WAVE + lane : 0u;
                key[k] = (uint32_t)src_k[e];
                val[k] = src_v[e];
            }
            bucket_rank<RB, ITEMS, FAST_RANK>(sh.wave_hist[wid], key, shift, mask, nlive, rank, nullptr);
            __syncthreads();
            {
                uint32_t base, tot;
                bucket_wave_bases<RB, T, true>(sh, base, tot);
            }
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                if (k * WAVE + lane < nlive) {
                    const uint32_t dst = sh.wave_hist[wid][(key[k] >> shift) & mask] + rank[k];
                    if (dst_k) dst_k[dst] = (K)key[k];
                    dst_v[dst] = val[k];
                }
            }
            __syncthreads();
        }
        shift += bits;
        // (the next pass reads what this one wrote: the barrier above orders the workgroup's own global accesses)
    }
}

template <typename K, int RB, int T, bool FAST_RANK>
__global__ __launch_bounds__(T) void k_bucket_sort(BucketSortIO io) {
    constexpr int WAVES = BucketCfg<T>::WAVES;
    constexpr uint32_t PER = 1024u / (uint32_t)T;      // totals per thread: the top digit has at most 1024 values
    static_assert(RB >= 6 && RB <= 9 && (WAVES << RB) % T == 0 && (1 << RB) <= T,
                  "the waves' counters are cleared with whole rounds of T threads; one thread per digit");
    __shared__ BucketShared<RB, T> sh;
    const uint32_t tid = threadIdx.x, bucket = blockIdx.x;
    // (round 5, first version: every workgroup scanned the bucket sizes itself — two block scans over 1024 values, ~1 us
    // in front of every bucket; the scatter pass has the starts in registers anyway and writes them)
    const uint32_t size = io.totals[bucket], start = io.starts[bucket];
    if (bucket == 0u && io.bucket_max) {
        uint32_t m = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) {
            const uint32_t i = tid * PER + q;
            const uint32_t v = i < io.nb ? io.totals[i] : 0u;
            m = v > m ? v : m;
        }
        m = wave_reduce_max(m);
        if ((tid & 63u) == 0u) sh.scan[tid >> 6] = m;
        __syncthreads();
        if (tid == 0u) {
            uint32_t mm = 0;
#pragma unroll
            for (int w = 0; w < WAVES; w++) mm = sh.scan[w] > mm ? sh.scan[w] : mm;
            *io.bucket_max = mm;
        }
        __syncthreads();
    }
    if (size == 0u) return;
    // the watchdog samples every 16th bucket, a different sixteenth every frame
    uint32_t *rf = io.rank_fault && ((bucket + io.watch) & 15u) == 0u ? io.rank_fault : (uint32_t *)nullptr;
    if (size <= 4u * T) bucket_sort_fast<K, RB, T, 4, FAST_RANK>(sh, io, bucket, start, size, rf);
    else if (size <= 8u * T) bucket_sort_fast<K, RB, T, 8, FAST_RANK>(sh, io, bucket, start, size, rf);
    else if (size <= 16u * T) bucket_sort_fast<K, RB, T, 16, FAST_RANK>(sh, io, bucket, start, size, rf);
    else if (size <= BucketCfg<T>::CAP) bucket_sort_fast<K, RB, T, BucketCfg<T>::ITEMS_MAX, FAST_RANK>(sh, io, bucket, start, size, rf);
    else bucket_sort_chunked<K, RB, T, FAST_RANK>(sh, io, bucket, start, size);
}

// ---------------------------------------------------------------------------------------------
// Expansion, part 2: the (tile id, Gaussian) pairs in depth order, cut by OUTPUT slots.  Pair p
// belongs to the Gaussian j with prefix(j) <= p < prefix(j + 1) and is tile (p - prefix(j)) of its
// rect, row-major.  A workgroup of k_pairs_emit produces exactly the pairs of one tile of the tile
// sort's first pass (a wave owns ITEMS * 64 consecutive slots), so
//   * the work is balanced whatever the splat sizes: the few hundred nearest splats that cover
//     hundreds of tiles each no longer form a critical path (Gaussian-cut workgroups: 25 us at 1 M
//     for 3.5 M pairs, of which 5 us are average work);
//   * the digit histogram of the sort's first pass is counted while the pairs are written, which
//     removes that pass's histogram kernel (one read of all keys).
// A wave finds its starting point with two cooperative steps over k_expand_count's sums (super-chunk
// sums, then the chunk sums of one super-chunk; above 8.4 M Gaussians from the table that
// k_pairs_cursors writes instead) and then walks the Gaussians one chunk of 256 at a time, 4 per lane
// (next batch prefetched).  Inside a batch every Gaussian drops ONE marker at its first slot; slots
// are produced 256 at a time, 4 consecutive per lane: the owner of each slot comes from one
// inclusive max-scan over the lanes, and the tile id follows from the owner's table entry without a
// per-pair integer division.
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t wave_inclusive_scan64(uint64_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint64_t o = __shfl_up((unsigned long long)v, d, WAVE);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}

struct PairCursor {
    uint32_t chunk;      // first chunk (of EXP_CHUNK Gaussians in depth order) holding pairs at or after slot o0
    uint64_t prefix;     // pairs in front of that chunk
    uint64_t total;      // D: all pairs of the frame
};

// One level of the search: PER consecutive values per lane starting at element `first` (limit `n`),
// `run` = pairs in front of `first`.  Returns true and (index, prefix in front of it) of the element
// whose range holds slot o0; else adds the level's total to run.  Wave-uniform.
template <int PER, bool ALWAYS_TOTAL = false, typename T>
__device__ __forceinline__ bool cursor_level(const T *__restrict__ vals, uint32_t first, uint32_t n, uint64_t o0,
                                             uint32_t lane, uint64_t &run, uint32_t &index, uint64_t &prefix) {
    uint64_t v[PER], mine = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t q = first + lane * PER + k;
        v[k] = q < n ? (uint64_t)vals[q] : 0ull;
        mine += v[k];
    }
    const uint64_t incl = wave_inclusive_scan64(mine, lane);
    const uint64_t hit = __ballot(run + incl > o0);
    const uint64_t level_total = __shfl((unsigned long long)incl, WAVE - 1, WAVE);
    if (hit) {
        const uint32_t l = (uint32_t)__builtin_ctzll(hit);
        uint64_t before = run + __shfl((unsigned long long)(incl - mine), l, WAVE);
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint64_t x = __shfl((unsigned long long)v[k], l, WAVE);
            if (k == PER - 1 || before + x > o0) {
                index = first + l * PER + k;
                prefix = before;
                break;
            }
            before += x;
        }
        if (ALWAYS_TOTAL) run += level_total;
        return true;
    }
    run += level_total;
    return false;
}

// wave-uniform; every lane of the wave must call it
__device__ __forceinline__ PairCursor pair_cursor(const ExpandIO &io, uint32_t v_count, uint64_t o0, uint32_t lane) {
    constexpr int SB_PER = 4, CH_PER = (int)(EXP_SB / WAVE);
    static_assert(EXP_SB % WAVE == 0, "one chunk-level step per super-chunk");
    const uint32_t nchunks = (uint32_t)(((uint64_t)v_count + EXP_CHUNK - 1) / EXP_CHUNK);
    PairCursor cur;
    uint64_t run = 0, sb_prefix = 0;
    uint32_t sb = 0;
    bool found = false;
    // (the loop bound comes from the host, so these loads do not wait for V; entries past the real
    // number of super-chunks are zero)
    for (uint32_t q0 = 0; q0 < io.sb_bound; q0 += WAVE * SB_PER) {
        // one pass serves both the search and D: after the hit only the totals are still needed
        uint64_t r = run;
        uint32_t i = 0;
        uint64_t p = 0;
        const bool hit = cursor_level<SB_PER, true>(io.sb_sums, q0, io.sb_bound, o0, lane, r, i, p);
        if (hit && !found) {
            found = true;
            sb = i;
            sb_prefix = p;
        }
        run = r;
    }
    cur.total = run;
    cur.chunk = nchunks;
    cur.prefix = run;
    if (!found) return cur;                // o0 >= D
    run = sb_prefix;
    const uint32_t c_end = nchunks < (sb + 1u) * EXP_SB ? nchunks : (sb + 1u) * EXP_SB;
    cursor_level<CH_PER>(io.sums, sb * EXP_SB, c_end, o0, lane, run, cur.chunk, cur.prefix);
    return cur;
}

// Where the pairs of every CURSOR_SLOTS-slot span of the output start: chunk (of EXP_CHUNK Gaussians
// in depth order) whose pairs hold the span's first slot, and the number of pairs in front of that
// chunk.  Written by k_pairs_cursors, one workgroup per super-chunk, one thread per chunk: O(chunks)
// work in all, instead of a search over the sums in every wave of k_pairs_emit.
constexpr uint32_t CURSOR_SLOTS = 1024;
struct PairCursorRec {
    uint32_t chunk;
    uint32_t pad;
    unsigned long long prefix;
};

// One thread publishes the round's pair count: FrameState::pairs (clamped to the capacity) and ::overflow for the kernels
// behind it and — unless this is round 1 of a two-round frame — the frame result.  A two-round frame is skipped as a
// whole when either round exceeds the capacity (round 1's blend has then left the image untouched, round 2's does too).
__device__ __forceinline__ void publish_pairs(const ExpandIO &io, uint64_t d) {
    uint32_t over = d > (uint64_t)io.capacity ? 1u : 0u;
    io.state->pairs = over ? io.capacity : (uint32_t)d;
    if (io.round == 2u) over |= io.state->overflow;          // round 1 did not fit
    io.state->overflow = over;
    if (io.round == 1u) {
        io.state->pairs_round1 = d > 0xffffffffull ? 0xffffffffu : (uint32_t)d;
        return;
    }
    const uint64_t total = d + (io.round == 2u ? (uint64_t)io.state->pairs_round1 : 0ull);
    publish_result(io.result, io.state->visible, total,
                   (over ? FRAME_FLAG_PAIR_OVERFLOW | FRAME_FLAG_SKIPPED : 0u) | (io.state->rank_fault ? FRAME_FLAG_RANK_FAULT : 0u), io.gen,
                   io.flags_dev, io.state->depth_bucket_max, io.state->tile_bucket_max, io.round == 2u ? io.state->tiles_done : 0u, io.round == 2u ? io.state->tiles_open : 0u,
                   io.round == 2u ? (d > (uint64_t)io.state->pairs_round1 ? (d > 0xffffffffull ? 0xffffffffu : (uint32_t)d) : io.state->pairs_round1) : 0u);
}

// Grid: sb_bound workgroups of EXP_SB threads.  Workgroup 0 also publishes D (clamped to the pair
// capacity), the overflow flag and the frame result.
__global__ __launch_bounds__(EXP_SB) void k_pairs_cursors(ExpandIO io) {
    static_assert(EXP_SB == 2 * WAVE, "two waves per super-chunk");
    __shared__ unsigned long long s_before[2], s_all[2], s_wave0;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t sb = blockIdx.x;
    const uint32_t v_count = io.visible();
    const uint32_t nchunks = (uint32_t)(((uint64_t)v_count + EXP_CHUNK - 1) / EXP_CHUNK);
    uint64_t before = 0, all = 0;
    for (uint32_t q = threadIdx.x; q < io.sb_bound; q += EXP_SB) {
        const uint64_t v = io.sb_sums[q];
        if (q < sb) before += v;
        all += v;
    }
    before = wave_reduce_add64(before);
    all = wave_reduce_add64(all);
    const uint32_t c = sb * EXP_SB + threadIdx.x;
    const uint64_t v = c < nchunks ? (uint64_t)io.sums[c] : 0ull;
    const uint64_t incl = wave_inclusive_scan64(v, lane);
    if (lane == 63u) {
        s_before[wid] = before;
        s_all[wid] = all;
        if (wid == 0u) s_wave0 = incl;
    }
    __syncthreads();
    const uint64_t d = s_all[0] + s_all[1];
    if (sb == 0u && threadIdx.x == 0u) publish_pairs(io, d);
    if (v == 0ull) return;
    const uint64_t p = s_before[0] + s_before[1] + (wid ? s_wave0 : 0ull) + incl - v;   // pairs in front of chunk c
    const uint64_t end = p + v < (uint64_t)io.capacity ? p + v : (uint64_t)io.capacity;
    // spans whose first slot lies in [p, end): bounded by capacity / CURSOR_SLOTS, ends for every thread
    for (uint64_t k = (p + CURSOR_SLOTS - 1) / CURSOR_SLOTS; k * CURSOR_SLOTS < end; k++) {
        io.cursors[k].chunk = c;
        io.cursors[k].prefix = p;
    }
}

constexpr int GEN_PER = 4;                    // Gaussians per lane per batch (consecutive: vector loads)
constexpr int GEN_BATCH = WAVE * GEN_PER;     // = EXP_CHUNK: a batch is one chunk of the count kernel
static_assert(GEN_BATCH == EXP_CHUNK, "the cursor hands out chunk starts");

// per Gaussian of the batch: first slot (signed, relative to the wave's first), id, tile id of the rect origin, rect
// width — or the row code of a small rect with dropped tiles.  PACKED (rect_pack32: width <= 256, origin < 2^15): 12 bytes instead of 16 — with the staging below a
// workgroup of k_pairs_emit then needs 38 KB of LDS instead of 52, four per CU instead of three (the kernel is
// bound by the latency of its dependent loads times its occupancy, NOTES.md Part II).
template <bool PACKED>
struct PairGenTab {
    uint4 rec[GEN_BATCH];
    __device__ __forceinline__ void put(uint32_t i, uint32_t start, uint32_t gid, uint32_t origin, uint32_t w) {
        rec[i] = make_uint4(start, gid, origin, w);
    }
    __device__ __forceinline__ uint4 get(uint32_t i) const { return rec[i]; }
};
template <>
struct PairGenTab<true> {
    uint2 so[GEN_BATCH];               // first slot, origin (15 bits: the packed rects' limit) | shape << 15
    uint32_t gid[GEN_BATCH];
    // shape: the rect's width (<= 256), or 0x10000 | row code for a small rect that lost tiles (rect version 4)
    __device__ __forceinline__ void put(uint32_t i, uint32_t start, uint32_t g, uint32_t origin, uint32_t shape) {
        so[i] = make_uint2(start, origin | (shape << 15));
        gid[i] = g;
    }
    __device__ __forceinline__ uint4 get(uint32_t i) const {
        const uint2 a = so[i];
        return make_uint4(a.x, gid[i], a.y & 0x7fffu, a.y >> 15);
    }
};
constexpr uint32_t PAIR_SHAPE_ROWS = 0x10000u;       // PairGenTab shape: a row code, not a width

template <typename K, int NSLOTS, bool RECT32>
struct PairGenShared {                 // LDS private to one wave
    uint32_t vals[NSLOTS];             // output staging (Gaussian id per slot); until a slot is produced it holds
                                       // the marker: 1 + batch index of the Gaussian whose first pair it is, 0 = none
                                       // (markers and groups stay below hi <= n_slots <= NSLOTS, a multiple of 256)
    K keys[NSLOTS];                    // output staging (tile id per slot)
    PairGenTab<RECT32> tab;
};

// The wave produces its output slots [0, n_slots) (absolute: o0 + slot) into sh.keys / sh.vals and
// calls count(tile) once per pair.  Nothing is stored to global memory in here: stores retire
// through the same in-order counter as the loads of the next batch, and waiting for that load would
// wait for every store issued after it (measured: 25 us instead of 10 at 1 M).
// A lane owns GEN_PER consecutive slots of a 256-slot group, so one max-scan over the lanes serves
// 256 pairs.  Every iteration of the outer loop advances j (bounded by v_count): the wave always
// terminates.
template <typename K, int NSLOTS, bool RECT32, typename Count>
__device__ __forceinline__ void pair_generate(const ExpandIO &io, uint32_t v_count, uint64_t o0, uint32_t n_slots,
                                              const PairCursor &cur, uint32_t lane, PairGenShared<K, NSLOTS, RECT32> &sh,
                                              Count count) {
    static_assert(sizeof(K) == 2 || sizeof(K) == 4, "tile keys are u16 or u32");
    static_assert(NSLOTS % GEN_BATCH == 0, "whole groups of 256 slots");
    {
        uint4 *z = (uint4 *)sh.vals;
#pragma unroll
        for (uint32_t q0 = 0; q0 < NSLOTS / 4; q0 += WAVE) z[q0 + lane] = make_uint4(0u, 0u, 0u, 0u);
    }
    // slots relative to o0 from here on: the chunk in front of o0 holds < 2^30 pairs, a batch < 2^30
    int32_t rel = (int32_t)(int64_t)(cur.prefix - o0);      // <= 0: first slot of the batch
    const int32_t n = (int32_t)n_slots;
    uint32_t j = cur.chunk * EXP_CHUNK;
    // the next batch's 48 bytes per lane are requested before the current batch is worked on (the
    // arrays are padded: reading past v_count is in bounds, the values are masked below)
    const uint4 *order4 = (const uint4 *)io.order;
    const uint4 *rect4 = (const uint4 *)io.sorted_rect;
    uint4 g_next = order4[(j >> 2) + lane];
    // RECT32: the four rects of a lane are one 16-byte vector (rect_pack32); otherwise two
    uint4 ra_next, rb_next = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (RECT32) {
        ra_next = rect4[(j >> 2) + lane];
    } else {
        ra_next = rect4[(j >> 1) + 2 * lane];
        rb_next = rect4[(j >> 1) + 2 * lane + 1];
    }
    __builtin_amdgcn_wave_barrier();
    while (rel < n && j < v_count) {
        const uint32_t g[GEN_PER] = {g_next.x, g_next.y, g_next.z, g_next.w};
        const uint4 ra = ra_next, rb = rb_next;
        if (j + GEN_BATCH < v_count) {
            g_next = order4[((j + GEN_BATCH) >> 2) + lane];
            if constexpr (RECT32) {
                ra_next = rect4[((j + GEN_BATCH) >> 2) + lane];
            } else {
                ra_next = rect4[((j + GEN_BATCH) >> 1) + 2 * lane];
                rb_next = rect4[((j + GEN_BATCH) >> 1) + 2 * lane + 1];
            }
        }
        // per Gaussian: rect width, tile count, tile id of the rect's origin
        uint32_t w[GEN_PER], cnt[GEN_PER], origin[GEN_PER], mine = 0;
#pragma unroll
        for (int k = 0; k < GEN_PER; k++) {
            const bool live = j + GEN_PER * lane + k < v_count;
            // w[k]: the rect's width, or PAIR_SHAPE_ROWS | row code when tiles of a small rect were dropped (rect version 4)
            if constexpr (RECT32) {
                const uint32_t pk = k == 0 ? ra.x : k == 1 ? ra.y : k == 2 ? ra.z : ra.w;
                const bool masked = (pk >> 31) != 0u;
                w[k] = masked ? PAIR_SHAPE_ROWS | (pk & 0xfffu) : (pk & 0xffu) + 1u;
                cnt[k] = live ? rect_count32(pk) : 0u;
                origin[k] = (pk >> 16) & 0x7fffu;
            } else {
                const uint32_t r0 = k == 0 ? ra.x : k == 1 ? ra.z : k == 2 ? rb.x : rb.z;
                const uint32_t r1 = k == 0 ? ra.y : k == 1 ? ra.w : k == 2 ? rb.y : rb.w;
                const bool masked = (r1 & 0xffffu) == 0u;          // (x1 is never 0; an all-zero entry past V counts 0 tiles)
                w[k] = masked ? PAIR_SHAPE_ROWS | ((r1 >> 16) & 0xfffu) : (r1 & 0xffffu) - (r0 & 0xffffu);
                cnt[k] = live ? rect_count64(make_uint2(r0, r1)) : 0u;
                origin[k] = __umul24(r0 >> 16, io.tiles_x) + (r0 & 0xffffu);
            }
            mine += cnt[k];
        }
        const uint32_t incl = wave_inclusive_scan(mine, lane);
        const int32_t bend = rel + (int32_t)__builtin_amdgcn_readlane(incl, WAVE - 1);
        const int32_t lo = rel > 0 ? rel : 0, hi = bend < n ? bend : n;
        if (hi > lo) {
            int32_t start = rel + (int32_t)(incl - mine);
            uint32_t last = 0;        // 1 + batch index of the last Gaussian (with pairs) starting at or before lo
#pragma unroll
            for (int k = 0; k < GEN_PER; k++) {
                const uint32_t idx = GEN_PER * lane + k;
                sh.tab.put(idx, (uint32_t)start, g[k], origin[k], w[k] ? w[k] : 1u);
                if (cnt[k] != 0u) {
                    if (start <= lo) last = idx + 1u;
                    else if (start < hi) sh.vals[start] = idx + 1u;
                }
                start += (int32_t)cnt[k];
            }
            uint32_t carry = wave_reduce_max(last);       // >= 1: the batch reaches past lo
            __builtin_amdgcn_wave_barrier();
            // one group = 256 consecutive slots, 4 per lane.  FULL: the whole group belongs to this batch
            // (no per-slot range tests, vector stores)
            auto group = [&](int32_t wb, auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
                const int32_t s0 = wb + GEN_PER * (int32_t)lane;
                const uint4 mraw = *(const uint4 *)&sh.vals[s0];
                const uint32_t raw[GEN_PER] = {mraw.x, mraw.y, mraw.z, mraw.w};
                bool act[GEN_PER];
                uint32_t m[GEN_PER], lane_max = 0;
#pragma unroll
                for (int k = 0; k < GEN_PER; k++) {
                    act[k] = FULL || (s0 + k >= lo && s0 + k < hi);
                    m[k] = act[k] ? raw[k] : 0u;       // slots below lo already hold their output
                    lane_max = m[k] > lane_max ? m[k] : lane_max;
                }
                const uint32_t incl_max = wave_inclusive_max(lane_max);
                uint32_t own = __shfl_up(incl_max, 1, WAVE);      // markers of the lanes below me
                own = lane == 0u ? carry : (own > carry ? own : carry);
                const uint32_t top = __builtin_amdgcn_readlane(incl_max, WAVE - 1);
                carry = top > carry ? top : carry;
                // owners first, then the four table reads back to back, then the arithmetic, then the
                // histogram adds: LDS atomics in between would serialise the reads behind them
                uint32_t tile[GEN_PER], gid[GEN_PER];
                uint4 ot[GEN_PER];
#pragma unroll
                for (int k = 0; k < GEN_PER; k++) {
                    own = m[k] > own ? m[k] : own;
                    ot[k] = sh.tab.get(own - 1u);
                }
#pragma unroll
                for (int k = 0; k < GEN_PER; k++) {
                    const uint32_t local = (uint32_t)(s0 + k - (int32_t)ot[k].x);
                    const uint32_t shape = ot[k].w, width = shape & 0xffffu;
                    // row = local / width without the integer division: local < 2^22 and width < 2^16 are
                    // exact in f32, the estimate is off by at most one either way, fixed up exactly
                    uint32_t row = (uint32_t)((float)local * __builtin_amdgcn_rcpf((float)width));
                    const uint32_t rem = local - __umul24(row, width);
                    if ((int32_t)rem < 0) row--;
                    else if (rem >= width) row++;
                    // tile = origin + row * tiles_x + col, col = local - row * width
                    uint32_t t = ot[k].z + local + __umul24(row, io.tiles_x - width);
                    // a small rect that lost tiles: row j keeps cnt_j columns from first_j on (4 bits per row)
                    const uint32_t c0 = (shape >> 2) & 3u, c01 = c0 + ((shape >> 6) & 3u);
                    const uint32_t mrow = (local >= c0 ? 1u : 0u) + (local >= c01 ? 1u : 0u);
                    const uint32_t before = mrow == 0u ? 0u : mrow == 1u ? c0 : c01;
                    const uint32_t first = (shape >> (4u * mrow)) & 3u;
                    const uint32_t mt = ot[k].z + __umul24(mrow, io.tiles_x) + first + (local - before);
                    tile[k] = (shape & PAIR_SHAPE_ROWS) ? mt : t;
                    gid[k] = ot[k].y;
                }
#pragma unroll
                for (int k = 0; k < GEN_PER; k++)
                    if (act[k]) count(tile[k]);
                if constexpr (FULL) {
                    *(uint4 *)&sh.vals[s0] = make_uint4(gid[0], gid[1], gid[2], gid[3]);
                    if constexpr (sizeof(K) == 2)
                        *(uint2 *)&sh.keys[s0] = make_uint2(tile[0] | (tile[1] << 16), tile[2] | (tile[3] << 16));
                    else
                        *(uint4 *)&sh.keys[s0] = make_uint4(tile[0], tile[1], tile[2], tile[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < GEN_PER; k++)
                        if (act[k]) {
                            sh.vals[s0 + k] = gid[k];
                            sh.keys[s0 + k] = (K)tile[k];
                        }
                }
            };
            for (int32_t wb = lo & ~(GEN_BATCH - 1); wb < hi; wb += GEN_BATCH) {
                if (wb >= lo && wb + GEN_BATCH <= hi) group(wb, std::true_type());
                else group(wb, std::false_type());
            }
            __builtin_amdgcn_wave_barrier();
        }
        rel = bend;
        j += GEN_BATCH;
    }
}

// Pairs of one sort tile + the histogram of their first digit; workgroup 0 also publishes D (clamped
// to the pair capacity), the overflow flag and the frame result.  Grid: capacity / TILE workgroups.
template <typename K, int RB, int ITEMS, bool RECT32 = false>
__global__ __launch_bounds__(SORT_THREADS) void k_pairs_emit(ExpandIO io, uint32_t digit_mask,
                                                             uint32_t *__restrict__ ghist, K *__restrict__ tkeys,
                                                             uint32_t num_blocks, uint32_t xcd_chunk, uint32_t digit_shift) {
    constexpr uint32_t TILE = SORT_THREADS * ITEMS;
    constexpr uint32_t NSLOTS = ITEMS * WAVE;
    constexpr int R = 1 << RB;
    // two copies of the histogram (lanes alternate): neighbouring slots of a lane are neighbouring tiles, and the
    // Gaussians of one wave instruction are neighbours in depth, not in space — same-address adds are rare
    constexpr int COPIES = 512 / R > 1 ? 512 / R : 1;
    constexpr int DPT = R / SORT_THREADS;
    __shared__ uint32_t s_hist[COPIES][R];
    __shared__ PairGenShared<K, NSLOTS, RECT32> s_gen[4];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t v_count = io.visible();
    const uint32_t block = scatter_tile_of(blockIdx.x, xcd_chunk);   // as the scatter and the histograms
    const uint64_t o0 = (uint64_t)block * TILE + wid * NSLOTS;
    PairCursor cur;
    uint32_t count;
    if (io.cursors) {
        static_assert(NSLOTS % CURSOR_SLOTS == 0, "a wave starts on a cursor");
        // the cursor record is requested together with the counts (the table covers the whole grid: a
        // record past D is stale but in bounds and unused), so that the wave's first batch is two
        // dependent round trips away instead of three
        const uint64_t span = o0 / CURSOR_SLOTS, last_span = (uint64_t)io.capacity / CURSOR_SLOTS;   // (the grid is rounded up)
        const PairCursorRec rec = io.cursors[span < last_span ? span : last_span];
        count = io.state->pairs < io.capacity ? io.state->pairs : io.capacity;   // published by k_pairs_cursors
        cur.chunk = 0;
        cur.prefix = 0;
        if (o0 < count) {
            cur.chunk = rec.chunk;
            cur.prefix = rec.prefix;
        }
    } else {
        cur = pair_cursor(io, v_count, o0, lane);
        const uint64_t d = cur.total;
        count = d > (uint64_t)io.capacity ? io.capacity : (uint32_t)d;
        if (block == 0u && threadIdx.x == 0u) publish_pairs(io, d);
    }
    if ((uint64_t)block * TILE >= count) return;      // the same D in every wave: block-uniform
#pragma unroll
    for (int c = 0; c < COPIES; c++)
#pragma unroll
        for (int q = 0; q < DPT; q++) s_hist[c][threadIdx.x + q * SORT_THREADS] = 0;
    __syncthreads();
    if (o0 < count) {
        const uint32_t n_slots = count - o0 < (uint64_t)NSLOTS ? (uint32_t)(count - o0) : NSLOTS;
        uint32_t *hist = s_hist[threadIdx.x & (uint32_t)(COPIES - 1)];
        PairGenShared<K, NSLOTS, RECT32> &sh = s_gen[wid];
        pair_generate<K, NSLOTS, RECT32>(io, v_count, o0, n_slots, cur, lane, sh,
                         [&](uint32_t tile) { atomicAdd(&hist[(tile >> digit_shift) & digit_mask], 1u); });
        // the wave's slots leave in whole 16-byte vectors (o0 is a multiple of NSLOTS: aligned)
        uint32_t *vout = io.tvals + o0;
        K *kout = tkeys + o0;
        if (n_slots == NSLOTS) {
#pragma unroll
            for (uint32_t q0 = 0; q0 < NSLOTS / 4; q0 += WAVE)
                store16((uint4 *)vout + q0 + lane, ((const uint4 *)sh.vals)[q0 + lane], io.wt_stores);
#pragma unroll
            for (uint32_t q0 = 0; q0 < NSLOTS * sizeof(K) / 16; q0 += WAVE)
                store16((uint4 *)kout + q0 + lane, ((const uint4 *)sh.keys)[q0 + lane], io.wt_stores);
        } else {
            for (uint32_t q = lane; q < n_slots; q += WAVE) {
                vout[q] = sh.vals[q];
                kout[q] = sh.keys[q];
            }
        }
    }
    __syncthreads();
    uint32_t sum[DPT];
#pragma unroll
    for (int q = 0; q < DPT; q++) {
        sum[q] = 0;
#pragma unroll
        for (int c = 0; c < COPIES; c++) sum[q] += s_hist[c][threadIdx.x + q * SORT_THREADS];
    }
#pragma unroll
    for (int q = 0; q < DPT; q++) {
        const uint32_t digit = threadIdx.x + q * SORT_THREADS;
        if (digit <= digit_mask) ghist[(uint64_t)digit * num_blocks + block] = sum[q];   // live rows only (k_sort_hist)
    }
}

// Probe for FAST_RANK, run by gs_device_create for both digit widths the sorts use (RB = 8: 256
// counters, RB = 9: 512 counters per wave, the depth sort's layout): every wave issues returning
// adds to its own row of LDS counters exactly as k_sort_scatter does — ITEMS back-to-back atomics
// per lane on digits taken from registers — and then checks every returned value against the
// ballot-based rank.  bad[0] counts the lanes whose value is not (count before this round) +
// (lower lanes with the same digit).  Digits mix uniform values, few values, runs and all-equal rounds.
template <int RB>
__global__ __launch_bounds__(SORT_THREADS) void k_probe_lds_atomic_order(uint32_t rounds, uint32_t seed,
                                                                         uint32_t *__restrict__ bad) {
    constexpr int R = 1 << RB;
    constexpr int ITEMS = 16;
    __shared__ uint32_t s_cnt[4][R];      // what the atomics hit
    __shared__ uint32_t s_ref[4][R];      // reference counters advanced by the ballot ranking
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < 4 * R; q += SORT_THREADS) {
        (&s_cnt[0][0])[q] = 0;
        (&s_ref[0][0])[q] = 0;
    }
    __syncthreads();
    uint32_t errors = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        uint32_t d[ITEMS], got[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            uint32_t x = ((blockIdx.x * rounds + r) * ITEMS + k) * 256u + threadIdx.x + seed;
            x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
            const uint32_t mode = (r + k) & 3u;
            d[k] = (mode == 0 ? x : mode == 1 ? (x & 3u) * 37u : mode == 2 ? (lane >> 3) + 250u : 7u + 256u) & (R - 1);
        }
        // odd rounds mask out a data-dependent subset of the lanes, as the compacting pass does
        bool live[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; k++) live[k] = !(r & 1u) || ((d[k] * 2654435761u) >> 29) != 0u;
#pragma unroll
        for (int k = 0; k < ITEMS; k++)
            if (live[k]) got[k] = atomicAdd(&s_cnt[wid][d[k]], 1u);   // back to back, as in the sort
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            uint64_t peers = __ballot(live[k]);
#pragma unroll
            for (int b = 0; b < RB; b++) {
                uint64_t m = __ballot((d[k] >> b) & 1u);
                peers &= ((d[k] >> b) & 1u) ? m : ~m;
            }
            const uint32_t before = mbcnt(peers);
            const uint32_t old = s_ref[wid][d[k]];
            errors += live[k] && got[k] != old + before;
            __builtin_amdgcn_wave_barrier();
            if (live[k] && before == 0u) s_ref[wid][d[k]] = old + (uint32_t)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (errors) atomicAdd(bad, errors);
}

// ---------------------------------------------------------------------------------------------
// tile ranges (row x5a)
// ---------------------------------------------------------------------------------------------

// consecutive keys per thread = one 16-byte load
template <typename TK> constexpr int range_items() { return 16 / sizeof(TK); }

template <typename TK>
__global__ __launch_bounds__(256) void k_tile_ranges(const TK *__restrict__ tkeys, SortCount sc,
                                                     uint32_t *__restrict__ ranges) {
    constexpr int N = range_items<TK>();
    const uint32_t count = sc.get();
    const uint64_t j64 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * N;
    if (j64 >= count) return;
    const uint32_t j0 = (uint32_t)j64;
    uint32_t tile[N];
    if (count - j0 >= (uint32_t)N) {
        uint4 q = *(const uint4 *)(tkeys + j0);
        uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < N; e++) tile[e] = sizeof(TK) == 2 ? (w[e >> 1] >> (16 * (e & 1))) & 0xffffu : w[e];
    } else {
#pragma unroll
        for (int e = 0; e < N; e++) tile[e] = j0 + e < count ? (uint32_t)tkeys[j0 + e] : 0u;
    }
    uint32_t prev = j0 ? (uint32_t)tkeys[j0 - 1] : 0xffffffffu;
#pragma unroll
    for (int e = 0; e < N; e++) {
        uint32_t j = j0 + e;
        if (j < count) {
            if (tile[e] != prev) {
                ranges[2 * tile[e]] = j;
                if (prev != 0xffffffffu) ranges[2 * prev + 1] = j;
            }
            if (j + 1 == count) ranges[2 * tile[e] + 1] = j + 1;
            prev = tile[e];
        }
    }
}

// The same ranges by SEARCH: one wave per tile, its two halves find the first key >= tile and the first
// key >= tile + 1 with a 32-ary search (32 probes per round and half; 127 M keys: 6 rounds of dependent
// loads, whatever D is), instead of every key being read once more.  Neighbouring tiles probe the same
// sectors in the early rounds, so the traffic stays far below the key array's size.  Same output as
// k_tile_ranges, including (0, 0) for a tile without pairs.
template <typename TK>
__global__ __launch_bounds__(256) void k_tile_ranges_search(const TK *__restrict__ tkeys, SortCount sc,
                                                            uint32_t *__restrict__ ranges, uint32_t num_tiles) {
    const uint32_t count = sc.get();
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * 4u + wid;
    if (t >= num_tiles) return;                      // wave-uniform
    const uint32_t h = lane >> 5, l = lane & 31u;
    const uint32_t target = t + h;
    uint32_t lo = 0, len = count;                    // the answer lies in [lo, lo + len]; uniform per half
    while (__any(len != 0u)) {
        // probes at lo + (i + 1) * step - 1, i = 0..31 (those inside the interval): sorted keys make "key <
        // target" true for the first k of them, so the answer moves to [lo + k * step, + step - 1]
        const uint32_t step = len / 32u + 1u;
        const uint64_t p = (uint64_t)lo + (uint64_t)(l + 1u) * step - 1u;
        const bool in = p < (uint64_t)lo + len;      // false for every lane of a half that is done (len == 0)
        const bool less = in && (uint32_t)tkeys[in ? p : 0u] < target;
        const uint64_t m = __ballot(less);
        const uint32_t k = (uint32_t)__popc((uint32_t)(m >> (32u * h)));
        const uint32_t end = lo + len;
        lo += k * step;                              // <= end: only floor(len / step) probes are inside
        const uint32_t rem = end - lo;
        len = len == 0u ? 0u : (rem < step - 1u ? rem : step - 1u);
    }
    const uint32_t lb0 = __shfl(lo, 0, WAVE), lb1 = __shfl(lo, 32, WAVE);
    if (lane == 0u) {
        const bool any = lb1 > lb0;
        ranges[2u * t] = any ? lb0 : 0u;
        ranges[2u * t + 1u] = any ? lb1 : 0u;
    }
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

// Slack of the staging cull for one (splat, tile).  The pixel loop evaluates `power` in binary32 from
// three terms that may cancel (a needle 1000 px long: terms of 1e6, sum of -5): its rounding error over
// the pixels of the tile is at most 5 u S, S <= (|ca| + |cb| + |cc|) D^2 with D the largest offset from
// the splat centre to a pixel centre of the tile, 5 u = 3e-7; the cull's own evaluation of the block
// maximum errs by as much again.  A constant 0.1 (rounds 1-2) is enough for ordinary splats only: an
// 8K frame with an 800-px needle lost threshold-level pixels against the oracle (round 3).  The
// threshold handed to the block tests is therefore  pmin - 0.1 - 2 * 3e-7 * (|ca|+|cb|+|cc|) * D^2.
__device__ __forceinline__ float cull_rounding_slack(float mx, float my, float ca, float cb, float cc, float x0,
                                                     float y0) {
    const float ax = fmaxf(fabsf(mx - x0), fabsf(mx - (x0 + 15.0f)));
    const float ay = fmaxf(fabsf(my - y0), fabsf(my - (y0 + 15.0f)));
    const float d = fmaxf(ax, ay);
    return (6.0e-7f * ((fabsf(ca) + fabsf(cb)) + fabsf(cc))) * (d * d);
}

// Maximum over t in [lo, hi] of the concave parabola q2*t^2 + q1*t + q0 (q2 < 0).  Used only by
// the conservative cull below, so the hardware reciprocal (1 ulp) is fine here.
__device__ __forceinline__ float parabola_max(float q2, float q1, float q0, float lo, float hi) {
    float t = clampf(-0.5f * q1 * __builtin_amdgcn_rcpf(q2), lo, hi);
    return (q2 * t + q1) * t + q0;
}

// Conservative test "can this splat reach alpha >= 1/255 anywhere on the pixel-centre rectangle
// [rx0,rx1] x [ry0,ry1]?".  power(d) = ca*dx^2 + cc*dy^2 + cb*dx*dy is a concave quadratic in
// d = mean - pixel; its maximum over the rectangle is 0 when the mean lies inside, otherwise it is
// attained on one of the four edges (a clamped 1-D parabola each).  alpha >= 1/255 needs
// power >= ln(1/(255*opacity)); the caller passes `thr` = that bound minus 0.1 of slack for the
// rounding in this test, so a dropped splat is one the pixel loop would skip at every pixel of
// the rectangle.
__device__ __forceinline__ bool splat_touches_rect(float mx, float my, float ca, float cb, float cc,
                                                   float rx0, float rx1, float ry0, float ry1,
                                                   float thr) {
    float dx_lo = mx - rx1, dx_hi = mx - rx0, dy_lo = my - ry1, dy_hi = my - ry0;
    bool in_x = dx_lo <= 0.0f && dx_hi >= 0.0f, in_y = dy_lo <= 0.0f && dy_hi >= 0.0f;
    float m0 = parabola_max(cc, cb * dx_lo, ca * dx_lo * dx_lo, dy_lo, dy_hi);
    float m1 = parabola_max(cc, cb * dx_hi, ca * dx_hi * dx_hi, dy_lo, dy_hi);
    float m2 = parabola_max(ca, cb * dy_lo, cc * dy_lo * dy_lo, dx_lo, dx_hi);
    float m3 = parabola_max(ca, cb * dy_hi, cc * dy_hi * dy_hi, dx_lo, dx_hi);
    float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    return (in_x && in_y) || !(m < thr);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    return __builtin_elementwise_fma(a, b, c);
}

// A tile's [start, end) found by the blend workgroup ITSELF (round 4): the same 32-ary search as
// k_tile_ranges_search, run by each of the workgroup's waves on its own (no LDS, no barrier; the second
// wave's probes hit the sectors the first one just pulled in).  The stand-alone range kernel was a
// dependent launch of 5-35 us in front of the blend (4.8 us at 1 M: five dependent probe rounds and nothing
// else on the chip; 35 us at 4K, where it read every key); inside the blend only the first wave of
// workgroups waits for its probes with nothing to overlap — every later workgroup searches while the other
// six workgroups of its CU blend.  keys == null: read the range from `ranges` (the frame without Gaussians;
// GS3D_RANGES_IN_BLEND=0).  Lane 0 of wave 0 stores the range for the taps (gs_renderer_download_ranges).
struct TileKeys {
    const void *keys;             // sorted tile ids of the D pairs (u16, or u32 when `wide`)
    const uint32_t *count_dev;    // D (FrameState.pairs)
    uint32_t count_bound;         // pair capacity
    uint32_t wide;
    // Two-round frames (DESIGN.md §4.2 "rounds"): round 1 blends the nearest Gaussians and leaves (C, +-T) in the image for the
    // tiles that are not finished (-T: the pixel itself is), a bit in `done` for those that are (their pixels are final);
    // round 2 returns at once for the done tiles and resumes the others.  0: the frame's only round.
    uint32_t round;
    uint32_t *done;               // [tiles / 32 + 1], zeroed per frame
    uint32_t *open;               // [tiles / 32 + 1], zeroed per frame: tiles round 1 had pairs for and did not finish
};
__device__ __forceinline__ void blend_tile_range(const TileKeys &tk, uint32_t *__restrict__ ranges, uint32_t tile,
                                                 uint32_t lane, bool writer, uint32_t &start, uint32_t &end) {
    if (!tk.keys) {
        start = ranges[2 * tile];
        end = ranges[2 * tile + 1];
        return;
    }
    uint32_t count = *tk.count_dev;
    count = count < tk.count_bound ? count : tk.count_bound;
    const uint32_t h = lane >> 5, l = lane & 31u;
    const uint32_t target = tile + h;
    uint32_t lo = 0, len = count;                    // the answer lies in [lo, lo + len]; uniform per half
    while (__any(len != 0u)) {
        const uint32_t step = len / 32u + 1u;
        const uint64_t p = (uint64_t)lo + (uint64_t)(l + 1u) * step - 1u;
        const bool in = p < (uint64_t)lo + len;
        const uint64_t q = in ? p : 0u;
        const uint32_t key = tk.wide ? ((const uint32_t *)tk.keys)[q] : (uint32_t)((const uint16_t *)tk.keys)[q];
        const bool less = in && key < target;
        const uint64_t m = __ballot(less);
        const uint32_t k = (uint32_t)__popc((uint32_t)(m >> (32u * h)));
        const uint32_t stop = lo + len;
        lo += k * step;
        const uint32_t rem = stop - lo;
        len = len == 0u ? 0u : (rem < step - 1u ? rem : step - 1u);
    }
    const uint32_t lb0 = __shfl(lo, 0, WAVE), lb1 = __shfl(lo, 32, WAVE);
    const bool any = lb1 > lb0;
    start = any ? lb0 : 0u;
    end = any ? lb1 : 0u;
    if (writer && lane == 0u) {
        ranges[2u * tile] = start;
        ranges[2u * tile + 1u] = end;
    }
}
// the workgroup form: wave 0 searches, the other wave takes the result from LDS behind one barrier (both
// waves searching doubled the probes and cost the blend 4-7 us of occupancy)
__device__ __forceinline__ void blend_tile_range_wg(const TileKeys &tk, uint32_t *__restrict__ ranges, uint32_t tile,
                                                    uint32_t lane, uint32_t wid, uint32_t *s_range, uint32_t &start,
                                                    uint32_t &end) {
    if (!tk.keys) {
        start = ranges[2 * tile];
        end = ranges[2 * tile + 1];
        return;
    }
    if (wid == 0u) {
        blend_tile_range(tk, ranges, tile, lane, true, start, end);
        if (lane == 0u) {
            s_range[0] = start;
            s_range[1] = end;
        }
    }
    __syncthreads();
    start = s_range[0];
    end = s_range[1];
}

constexpr int BLEND_THREADS = 128;
constexpr int BLEND_BATCH = 128;

// One workgroup (2 waves) = one 16x16 tile.  Wave h owns the half-tile of pixel rows 8h..8h+7;
// each lane owns TWO pixels (x, y) and (x, y+4), so the per-splat arithmetic runs on packed f32
// (v_pk_mul/fma/add_f32: two IEEE operations per lane per instruction, bit-identical to the scalar
// form).  The tile's sorted splat list is staged through LDS in batches of 128: each lane fetches
// one splat, tests it against both half-tiles (exact maximum of the concave exponent over the
// half-tile rectangle, conservative threshold), and the survivors are compacted per half-tile with
// wave64 ballots + mbcnt prefix counts (order preserving).  A wave therefore only walks splats
// that can contribute to its own 16x8 pixels.  The compaction never changes results: a removed
// splat has alpha < 1/255 at every pixel of the half-tile, which the pixel loop would skip.
// MODE = GaussianDisplayMode: 0 splat (the Gaussian falloff), 1 ellipse (flat
// alpha = min(0.99, opacity) inside the max_std_dev ellipse), 2 point (flat alpha inside a 1.5-px dot).
template <int MODE>
__global__ __launch_bounds__(BLEND_THREADS) void k_blend(uint32_t *__restrict__ ranges,
                                                         const uint32_t *__restrict__ idx,
                                                         const uint32_t *__restrict__ recs,
                                                         FrameConsts fc, float4 *__restrict__ rgba,
                                                         const FrameState *__restrict__ state, TileKeys tk) {
    // A frame that outgrew its pair capacity has lost its FARTHEST pairs: blending the rest would show
    // holes.  It is skipped instead — the image keeps what it held — and flagged.
    if (state->overflow) return;
    __shared__ float4 s_a[2][BLEND_BATCH];   // mx, my, ca, cb
    __shared__ float4 s_b[2][BLEND_BATCH];   // cc, opacity, r, g
    __shared__ float2 s_c[2][BLEND_BATCH];   // b, pmin (see below)
    __shared__ uint32_t s_cnt[2][2];         // [staging wave][half]
    __shared__ uint32_t s_alive[2];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t tile = fc.band_ty0 * fc.tiles_x + blockIdx.x;
    const uint32_t ty = tile / fc.tiles_x, tx = tile % fc.tiles_x;
    const uint32_t px = tx * 16u + (lane & 15u);
    const uint32_t py0 = ty * 16u + wid * 8u + (lane >> 4), py1 = py0 + 4u;
    const float pxf = (float)px + 0.5f;
    // tile / half-tile rectangles in pixel-centre coordinates
    const float rx0 = (float)(tx * 16u) + 0.5f, rx1 = rx0 + 15.0f;
    const float ry0 = (float)(ty * 16u) + 0.5f;

    uint32_t start, end;
    __shared__ uint32_t s_range[2];
    blend_tile_range_wg(tk, ranges, tile, lane, wid, s_range, start, end);
    f32x2 T = {1.0f, 1.0f}, C0 = {0.0f, 0.0f}, C1 = {0.0f, 0.0f}, C2 = {0.0f, 0.0f};
    const bool in0 = px < fc.width && py0 < fc.height, in1 = px < fc.width && py1 < fc.height;
    // A finished (or out-of-image) pixel is parked at y = DEAD: its exponent becomes hugely
    // negative, so the ordinary "power >= -5.6" test rejects it and the loop carries no per-pixel
    // "done" masks (those cost scalar-ALU mask arithmetic every iteration).
    constexpr float DEAD = 1.0e15f;
    f32x2 pyf = {in0 ? (float)py0 + 0.5f : DEAD, in1 ? (float)py1 + 0.5f : DEAD};
    // wave-uniform count of unfinished pixels
    uint32_t remaining = __builtin_amdgcn_readfirstlane(
        (uint32_t)__popcll(__ballot(in0)) + (uint32_t)__popcll(__ballot(in1)));

    for (uint32_t b0 = start; b0 < end; b0 += BLEND_BATCH) {
        // stop fetching once every pixel of the tile is finished
        if (lane == 0) s_alive[wid] = remaining;
        __syncthreads();
        if ((s_alive[0] | s_alive[1]) == 0u) break;

        // stage + cull + compact per half-tile (order preserving)
        uint32_t j = b0 + tid;
        bool keep0 = false, keep1 = false;
        float pmin = 0.0f;
        u32x4_a4 r0, r1;
        uint32_t r2x = 0;
        if (j < end) {
            const uint32_t *rec = recs + (uint64_t)idx[j] * REC_WORDS;
            r0 = *(const u32x4_a4 *)(rec);
            r1 = *(const u32x4_a4 *)(rec + 4);
            r2x = rec[8];
            float mx = u2f(r0.x), my = u2f(r0.y), ca = u2f(r0.z), cb = u2f(r0.w), cc = u2f(r1.x);
            // alpha = opacity * exp(power) >= 1/255  <=>  power >= -ln(255 * opacity).  pmin is that
            // bound minus 1e-3 (covers the hardware log's and the exp polynomial's error), so a
            // pixel with power < pmin is one the exact alpha test would reject anyway: using pmin
            // in place of the constant -5.6 changes no result, it only skips more work.
            if constexpr (MODE == 0) {
                pmin = fmaxf(-__logf(255.0f * u2f(r1.y)) - 1.0e-3f, -5.6f);
                const float thr = (pmin - 0.1f) - cull_rounding_slack(mx, my, ca, cb, cc, rx0, ry0);
                keep0 = splat_touches_rect(mx, my, ca, cb, cc, rx0, rx1, ry0, ry0 + 7.0f, thr);
                keep1 = splat_touches_rect(mx, my, ca, cb, cc, rx0, rx1, ry0 + 8.0f, ry0 + 15.0f, thr);
            } else if constexpr (MODE == 1) {
                pmin = fc.ellipse_pmin;   // exact bound of the pixel test; the cull below gets slack
                const float thr = (pmin - 0.1f - 1.0e-3f * fabsf(pmin)) - cull_rounding_slack(mx, my, ca, cb, cc, rx0, ry0);
                keep0 = splat_touches_rect(mx, my, ca, cb, cc, rx0, rx1, ry0, ry0 + 7.0f, thr);
                keep1 = splat_touches_rect(mx, my, ca, cb, cc, rx0, rx1, ry0 + 8.0f, ry0 + 15.0f, thr);
            } else {
                // distance from the centre to the half-tile's pixel-centre rectangle vs the 1.5-px dot
                const float ex = mx - clampf(mx, rx0, rx1);
                const float ey0 = my - clampf(my, ry0, ry0 + 7.0f), ey1 = my - clampf(my, ry0 + 8.0f, ry0 + 15.0f);
                keep0 = ex * ex + ey0 * ey0 <= 2.26f;
                keep1 = ex * ex + ey1 * ey1 <= 2.26f;
            }
        }
        uint64_t m0 = __ballot(keep0), m1 = __ballot(keep1);
        if (lane == 0) {
            s_cnt[wid][0] = (uint32_t)__popcll(m0);
            s_cnt[wid][1] = (uint32_t)__popcll(m1);
        }
        __syncthreads();
        const uint32_t base0 = wid ? s_cnt[0][0] : 0u, base1 = wid ? s_cnt[0][1] : 0u;
        const uint32_t kept =   // list length of MY half-tile (wave-uniform -> scalar loop control)
            __builtin_amdgcn_readfirstlane(s_cnt[0][wid] + s_cnt[1][wid]);
        if (keep0) {
            uint32_t pos = base0 + mbcnt(m0);
            s_a[0][pos] = make_float4(u2f(r0.x), u2f(r0.y), u2f(r0.z), u2f(r0.w));
            s_b[0][pos] = make_float4(u2f(r1.x), u2f(r1.y), u2f(r1.z), u2f(r1.w));
            s_c[0][pos] = make_float2(u2f(r2x), pmin);
        }
        if (keep1) {
            uint32_t pos = base1 + mbcnt(m1);
            s_a[1][pos] = make_float4(u2f(r0.x), u2f(r0.y), u2f(r0.z), u2f(r0.w));
            s_b[1][pos] = make_float4(u2f(r1.x), u2f(r1.y), u2f(r1.z), u2f(r1.w));
            s_c[1][pos] = make_float2(u2f(r2x), pmin);
        }
        __syncthreads();

        // pixel loop over this half-tile's compacted list (wave-uniform trip count, LDS
        // broadcast reads, packed f32 over the lane's two pixels)
        if (remaining != 0u) {
            for (uint32_t s = 0; s < kept; s++) {
                const float4 a = s_a[wid][s];
                const float4 bq = s_b[wid][s];
                const float dx = a.x - pxf;
                const f32x2 dy = f32x2{a.y, a.y} - pyf;
                const float u = a.z * dx, wq = a.w * dx;
                const f32x2 v = f32x2{bq.x, bq.x} * dy;
                f32x2 t = f32x2{wq, wq} * dy;
                t = pk_fma(v, dy, t);
                const f32x2 power = pk_fma(f32x2{u, u}, f32x2{dx, dx}, t);
                const float2 cq = s_c[wid][s];   // b, pmin
                bool p0, p1;
                if constexpr (MODE == 2) {
                    const f32x2 d2 = f32x2{dx * dx, dx * dx} + dy * dy;
                    p0 = d2.x <= 2.25f;
                    p1 = d2.y <= 2.25f;
                } else {
                    p0 = power.x <= 0.0f && power.x >= cq.y;
                    p1 = power.y <= 0.0f && power.y >= cq.y;
                }
                if (!__any(p0 || p1)) continue;
                f32x2 alpha;
                if constexpr (MODE == 0) {
                // exp on both pixels.  n = rint(t) is taken with the 1.5*2^23 magic
                // constant (t + M - M == rint(t) for |t| < 2^22, ties to even like rintf), and the
                // final ldexp(p, n) is an integer add of n into the exponent field: bits(t + M) =
                // bits(M) + n and bits(M) has its low 9 bits clear, so (bits(t + M) << 23) is
                // exactly n << 23 (mod 2^32).  Both equal the spec's rintf / ldexpf bit for bit
                // wherever the result is used (power in [-5.6, 0] => normal numbers).
                const f32x2 tt = power * f32x2{1.44269504088896340736f, 1.44269504088896340736f};
                const f32x2 tm = tt + f32x2{12582912.0f, 12582912.0f};
                const f32x2 n = tm - f32x2{12582912.0f, 12582912.0f};
                const f32x2 f = tt - n;
                f32x2 p = {0x1.5f0896p-10f, 0x1.5f0896p-10f};
                p = pk_fma(p, f, f32x2{0x1.3cbf6cp-7f, 0x1.3cbf6cp-7f});
                p = pk_fma(p, f, f32x2{0x1.c6af6cp-5f, 0x1.c6af6cp-5f});
                p = pk_fma(p, f, f32x2{0x1.ebfa4ap-3f, 0x1.ebfa4ap-3f});
                p = pk_fma(p, f, f32x2{0x1.62e430p-1f, 0x1.62e430p-1f});
                p = pk_fma(p, f, f32x2{1.0f, 1.0f});
                const f32x2 e = {u2f(f2u(p.x) + (f2u(tm.x) << 23)), u2f(f2u(p.y) + (f2u(tm.y) << 23))};
                const f32x2 oe = f32x2{bq.y, bq.y} * e;
                alpha = f32x2{fminf(0.99f, oe.x), fminf(0.99f, oe.y)};
                } else {
                    const float flat = fminf(0.99f, bq.y);
                    alpha = f32x2{flat, flat};
                }
                const bool act0 = p0 && alpha.x >= (1.0f / 255.0f);
                const bool act1 = p1 && alpha.y >= (1.0f / 255.0f);
                // a pixel that skips this splat blends it with alpha 0: T * (1 - 0) == T and
                // fma(rgb, 0 * T, C) == C exactly, so no selects are needed on T and C
                f32x2 alpha_eff = {act0 ? alpha.x : 0.0f, act1 ? alpha.y : 0.0f};
                f32x2 test_T = T * (f32x2{1.0f, 1.0f} - alpha_eff);
                const bool fin0 = act0 && test_T.x < 0.0001f, fin1 = act1 && test_T.y < 0.0001f;
                if (__any(fin0 || fin1)) {   // rare: some pixel reached T < 1e-4 -> it stops here
                    if (fin0) { pyf.x = DEAD; alpha_eff.x = 0.0f; test_T.x = T.x; }
                    if (fin1) { pyf.y = DEAD; alpha_eff.y = 0.0f; test_T.y = T.y; }
                    remaining = __builtin_amdgcn_readfirstlane(
                        remaining - ((uint32_t)__popcll(__ballot(fin0)) + (uint32_t)__popcll(__ballot(fin1))));
                }
                const f32x2 wgt = alpha_eff * T;
                C0 = pk_fma(f32x2{bq.z, bq.z}, wgt, C0);
                C1 = pk_fma(f32x2{bq.w, bq.w}, wgt, C1);
                C2 = pk_fma(f32x2{cq.x, cq.x}, wgt, C2);
                T = test_T;
                if (remaining == 0u) break;
            }
        }
    }
    if (in0) {
        float4 o;
        o.x = __builtin_fmaf(T.x, fc.bg[0], C0.x);
        o.y = __builtin_fmaf(T.x, fc.bg[1], C1.x);
        o.z = __builtin_fmaf(T.x, fc.bg[2], C2.x);
        o.w = 1.0f - T.x;
        store16(rgba + (uint64_t)py0 * fc.width + px, make_uint4(f2u(o.x), f2u(o.y), f2u(o.z), f2u(o.w)), fc.wt_stores);
    }
    if (in1) {
        float4 o;
        o.x = __builtin_fmaf(T.y, fc.bg[0], C0.y);
        o.y = __builtin_fmaf(T.y, fc.bg[1], C1.y);
        o.z = __builtin_fmaf(T.y, fc.bg[2], C2.y);
        o.w = 1.0f - T.y;
        store16(rgba + (uint64_t)py1 * fc.width + px, make_uint4(f2u(o.x), f2u(o.y), f2u(o.z), f2u(o.w)), fc.wt_stores);
    }
}

// ---------------------------------------------------------------------------------------------
// blend, grouped form: culling at sub-block granularity with the packed arithmetic kept
// ---------------------------------------------------------------------------------------------

// Exact maximum of the concave exponent over a pixel-centre rectangle, from the two edges that
// face the splat centre (the maximum of a concave quadratic over a rectangle that does not contain
// its apex lies on the boundary visible from the apex: the nearer vertical and the nearer
// horizontal edge; evaluating an edge that is not visible only adds a value attained on the
// rectangle, so the result never exceeds the true maximum).  Two clamped parabolas instead of four.
__device__ __forceinline__ bool splat_touches_rect2(float mx, float my, float ca, float cb, float cc,
                                                    float rx0, float rx1, float ry0, float ry1, float thr) {
    const float dx_lo = mx - rx1, dx_hi = mx - rx0, dy_lo = my - ry1, dy_hi = my - ry0;
    const bool in_x = dx_lo <= 0.0f && dx_hi >= 0.0f, in_y = dy_lo <= 0.0f && dy_hi >= 0.0f;
    const float dxn = fabsf(dx_lo) < fabsf(dx_hi) ? dx_lo : dx_hi;   // offset to the nearer vertical edge
    const float dyn = fabsf(dy_lo) < fabsf(dy_hi) ? dy_lo : dy_hi;
    const float mv = parabola_max(cc, cb * dxn, ca * dxn * dxn, dy_lo, dy_hi);
    const float mh = parabola_max(ca, cb * dyn, cc * dyn * dyn, dx_lo, dx_hi);
    return (in_x && in_y) || !(fmaxf(mv, mh) < thr);
}

// One workgroup (2 waves) = one 16x16 tile, wave h = pixel rows 8h..8h+7, two pixels per lane on
// packed f32 — as k_blend — but the wave's 64 lanes are split into G lane groups that own
// different sub-blocks of the half-tile (G = 2: two 8x8 blocks, G = 4: four 8x4 blocks) and walk
// DIFFERENT splat lists in lock step: one wave instruction stream serves G (splat, block) pairs.
// A splat of a few pixels radius touches far fewer 8x4 blocks x 32 pixels than 16x8 half-tiles x
// 128 pixels, so the number of loop iterations drops (1 M scene: x0.65 for G = 4, x0.77 for
// G = 2, tools/blend_sim.py) at an unchanged instruction count per iteration.  The staged splat
// records are stored once per batch; the per-block lists hold 16-bit byte offsets of them, padded
// with the index of a null record whose exponent test never passes, so lanes whose list is shorter
// than the wave's longest simply idle.  Results are bit-identical to k_blend: culling only removes
// (splat, block) pairs whose alpha is below 1/255 at every pixel of the block.
// ROUNDS: the instantiation of two-round frames (TileKeys::round = 1 / 2); the single-round one carries none of it.
template <int MODE, int G, bool ROUNDS = false>
__global__ __launch_bounds__(BLEND_THREADS) void k_blend_grouped(uint32_t *__restrict__ ranges,
                                                                 const uint32_t *__restrict__ idx,
                                                                 const uint32_t *__restrict__ recs,
                                                                 FrameConsts fc, float4 *__restrict__ rgba,
                                                                 const FrameState *__restrict__ state, TileKeys tk) {
    static_assert(G == 2 || G == 4, "lane groups per wave");
    if (state->overflow) return;              // frame skipped: see k_blend
    constexpr int GL = WAVE / G;              // lanes per group
    constexpr int BH = 16 / G;                // block height: G = 2 -> 8, G = 4 -> 4 (block width is 8; GL lanes x 2 pixels)
    constexpr int NL = 2 * G;                 // lists per tile
    constexpr uint32_t NULL_REC = BLEND_BATCH;
    // One 48-byte LDS record per staged splat: {mx, my, ca, cb | cc, pmin, opacity, r | g, b, -, -}.  The
    // lists hold the records' BYTE OFFSETS (16 bits each), so a step's three LDS reads share one address
    // register and differ only in the instruction's immediate offset: no shift / mask per step.
    constexpr uint32_t RS = 48;
    __shared__ __attribute__((aligned(16))) float s_rec[(BLEND_BATCH + 1) * (RS / 4)];
    __shared__ __attribute__((aligned(16))) uint16_t s_list[NL][BLEND_BATCH];
    __shared__ uint32_t s_cnt[2][NL];         // [staging wave][list]
    __shared__ uint32_t s_alive[2];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t tile = fc.band_ty0 * fc.tiles_x + blockIdx.x;
    const uint32_t ty = tile / fc.tiles_x, tx = tile % fc.tiles_x;
    const uint32_t gi = lane / GL, lg = lane % GL;
    const uint32_t bx = gi & 1u, by = G == 4 ? gi >> 1 : 0u;
    const uint32_t px = tx * 16u + bx * 8u + (lg & 7u);
    const uint32_t py0 = ty * 16u + wid * 8u + by * BH + (lg >> 3), py1 = py0 + BH / 2;
    const float pxf = (float)px + 0.5f;
    const uint32_t my_list = wid * G + gi;
    const float tx0 = (float)(tx * 16u) + 0.5f, ty0 = (float)(ty * 16u) + 0.5f;   // tile origin, pixel centres

    if (ROUNDS && tk.round == 2u && ((tk.done[tile >> 5] >> (tile & 31u)) & 1u)) return;      // finished in round 1, pixels final
    uint32_t start, end;
    __shared__ uint32_t s_range[2];
    blend_tile_range_wg(tk, ranges, tile, lane, wid, s_range, start, end);
    f32x2 T = {1.0f, 1.0f}, C0 = {0.0f, 0.0f}, C1 = {0.0f, 0.0f}, C2 = {0.0f, 0.0f};
    const bool in0 = px < fc.width && py0 < fc.height, in1 = px < fc.width && py1 < fc.height;
    constexpr float DEAD = 1.0e15f;           // finished / out-of-image pixels are parked far away (see k_blend)
    bool live0 = in0, live1 = in1;
    if (ROUNDS && tk.round == 2u) {
        // resume: the pixel's state as round 1 left it
        if (in0) {
            const float4 s = rgba[(uint64_t)py0 * fc.width + px];
            C0.x = s.x; C1.x = s.y; C2.x = s.z; T.x = fabsf(s.w);
            live0 = s.w > 0.0f;
        }
        if (in1) {
            const float4 s = rgba[(uint64_t)py1 * fc.width + px];
            C0.y = s.x; C1.y = s.y; C2.y = s.z; T.y = fabsf(s.w);
            live1 = s.w > 0.0f;
        }
    }
    f32x2 pyf = {live0 ? (float)py0 + 0.5f : DEAD, live1 ? (float)py1 + 0.5f : DEAD};
    uint32_t remaining = __builtin_amdgcn_readfirstlane(
        (uint32_t)__popcll(__ballot(live0)) + (uint32_t)__popcll(__ballot(live1)));
    if (tid == 0) {   // the null record: power = 0 everywhere, pmin = 1 -> "power >= pmin" never holds
        *(float4 *)(s_rec + NULL_REC * (RS / 4)) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *(float4 *)(s_rec + NULL_REC * (RS / 4) + 4) = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
        *(float2 *)(s_rec + NULL_REC * (RS / 4) + 8) = make_float2(0.0f, 0.0f);
    }

    for (uint32_t b0 = start; b0 < end; b0 += BLEND_BATCH) {
        // stop fetching once every pixel of the tile is finished
        if (lane == 0) s_alive[wid] = remaining;
        __syncthreads();
        if ((s_alive[0] | s_alive[1]) == 0u) break;

        // lists start out as all-null (the previous batch's loops ended before the barrier above)
        {
            uint32_t *l32 = (uint32_t *)&s_list[0][0];
            constexpr uint32_t fill = (NULL_REC * RS) * 0x00010001u;
            for (uint32_t q = tid; q < NL * BLEND_BATCH / 2; q += BLEND_THREADS) l32[q] = fill;
        }
        // stage one splat per thread, test it against every block of the tile; the verdicts live as
        // wave ballots in scalar registers (a per-lane array of 2G flags cost 40 vector registers)
        const uint32_t j = b0 + tid;
        float mx = 0.0f, my = 0.0f, ca = 0.0f, cb = 0.0f, cc = 0.0f, thr = 0.0f;
        const bool have = j < end;
        if (have) {
            const uint32_t *rec = recs + (uint64_t)idx[j] * REC_WORDS;
            const u32x4_a4 r0 = *(const u32x4_a4 *)(rec);
            const u32x4_a4 r1 = *(const u32x4_a4 *)(rec + 4);
            const uint32_t r2x = rec[8];
            mx = u2f(r0.x); my = u2f(r0.y); ca = u2f(r0.z); cb = u2f(r0.w); cc = u2f(r1.x);
            float pmin;
            if constexpr (MODE == 0) pmin = fmaxf(-__logf(255.0f * u2f(r1.y)) - 1.0e-3f, -5.6f);
            else pmin = fc.ellipse_pmin;
            thr = (MODE == 0 ? pmin - 0.1f : pmin - 0.1f - 1.0e-3f * fabsf(pmin)) -
                  cull_rounding_slack(mx, my, ca, cb, cc, tx0, ty0);
            *(float4 *)(s_rec + tid * (RS / 4)) = make_float4(mx, my, ca, cb);
            *(float4 *)(s_rec + tid * (RS / 4) + 4) = make_float4(cc, pmin, u2f(r1.y), u2f(r1.z));
            *(float2 *)(s_rec + tid * (RS / 4) + 8) = make_float2(u2f(r1.w), u2f(r2x));
        }
        // The tile's 2G blocks form a grid of 2 columns x G rows (8 wide, BH high).  The exact
        // maximum of the concave exponent over a block comes from the block's two edges facing the
        // splat centre (splat_touches_rect2): the facing vertical edge depends only on the column,
        // the facing horizontal edge only on the row, so the parabola coefficients and their
        // unconstrained optima are set up once per column / row and a block costs two clamps
        // (v_med3), four fmas and a compare.  (fma is fine here: the cull is conservative, not part
        // of the bit-exact result.)
        constexpr int NROW = G;                            // block rows of the tile (2 columns each)
        uint64_t m[NL];
        if constexpr (MODE == 2) {
#pragma unroll
            for (int l = 0; l < NL; l++) {
                const int lw = l / G, lgi = l % G;
                const float rx0 = tx0 + 8.0f * (float)(lgi & 1);
                const float ry0 = ty0 + 8.0f * (float)lw + (G == 4 ? (float)BH * (float)(lgi >> 1) : 0.0f);
                const float ex = mx - clampf(mx, rx0, rx0 + 7.0f), ey = my - clampf(my, ry0, ry0 + (float)(BH - 1));
                m[l] = __builtin_amdgcn_ballot_w64(have && ex * ex + ey * ey <= 2.26f);
            }
        } else {
            const float rcc = __builtin_amdgcn_rcpf(cc), rca = __builtin_amdgcn_rcpf(ca);
            float c_lo[2], c_hi[2], c_q1[2], c_q0[2], c_t[2];
            bool c_in[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float x0c = tx0 + 8.0f * (float)c;
                c_lo[c] = mx - (x0c + 7.0f);
                c_hi[c] = mx - x0c;
                c_in[c] = c_lo[c] <= 0.0f && c_hi[c] >= 0.0f;
                const float dn = fabsf(c_lo[c]) < fabsf(c_hi[c]) ? c_lo[c] : c_hi[c];   // nearer vertical edge
                c_q1[c] = cb * dn;
                c_q0[c] = ca * dn * dn;
                c_t[c] = -0.5f * c_q1[c] * rcc;            // optimum of cc t^2 + q1 t + q0 along the edge
            }
#pragma unroll
            for (int r = 0; r < NROW; r++) {
                const float y0r = ty0 + (float)(BH * r);
                const float r_lo = my - (y0r + (float)(BH - 1)), r_hi = my - y0r;
                const bool r_in = r_lo <= 0.0f && r_hi >= 0.0f;
                const float dn = fabsf(r_lo) < fabsf(r_hi) ? r_lo : r_hi;               // nearer horizontal edge
                const float r_q1 = cb * dn, r_q0 = cc * dn * dn;
                const float r_t = -0.5f * r_q1 * rca;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float tv = __builtin_amdgcn_fmed3f(c_t[c], r_lo, r_hi);
                    const float mv = __builtin_fmaf(__builtin_fmaf(cc, tv, c_q1[c]), tv, c_q0[c]);
                    const float th = __builtin_amdgcn_fmed3f(r_t, c_lo[c], c_hi[c]);
                    const float mh = __builtin_fmaf(__builtin_fmaf(ca, th, r_q1), th, r_q0);
                    const bool keep = (c_in[c] && r_in) || !(fmaxf(mv, mh) < thr);
                    // list of block (column c, row r): rows 0..G/2-1 belong to wave 0
                    const int l = (r / (NROW / 2)) * G + (r % (NROW / 2)) * 2 + c;
                    m[l] = __builtin_amdgcn_ballot_w64(have && keep);
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int l = 0; l < NL; l++) s_cnt[wid][l] = (uint32_t)__popcll(m[l]);
        }
        __syncthreads();
        {   // (the other wave's counts are read as one batch: behind a branch per list hipcc emitted eight
            // ds_read + s_waitcnt pairs in a row)
            uint32_t lbase[NL];
#pragma unroll
            for (int l = 0; l < NL; l++) lbase[l] = s_cnt[0][l];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 0; l < NL; l++)
                if ((m[l] >> lane) & 1ull) s_list[l][(wid ? lbase[l] : 0u) + mbcnt(m[l])] = (uint16_t)(tid * RS);
        }
        // trip count of MY wave: its longest list (wave-uniform)
        uint32_t trip = 0;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint32_t c = s_cnt[0][wid * G + g] + s_cnt[1][wid * G + g];
            trip = c > trip ? c : trip;
        }
        trip = __builtin_amdgcn_readfirstlane(trip);
        __syncthreads();

        if (remaining != 0u) {
            // Two splats per trip, in registers A and B: while A is blended B's record is already
            // on its way from LDS and the offsets of the next pair are being read (one 32-bit read:
            // lists are rows of 16-bit offsets, the trip count is rounded up to even and the padding is the
            // null index, so the extra step of an odd list blends nothing).  One structured loop,
            // one conditional block per step: no continue / break inside (hipcc turned those into a
            // scalar state machine of ~25 instructions and 5 branches per iteration).
            const uint32_t *mine = (const uint32_t *)s_list[my_list];
            const char *rbase = (const char *)s_rec;
            const uint32_t trips = (trip + 1u) >> 1;
            uint32_t pair = mine[0];
            uint32_t id_A = pair & 0xffffu;
            float4 a_A = *(const float4 *)(rbase + id_A), b_A = *(const float4 *)(rbase + id_A + 16);
#define GS_BLEND_STEP(AREC, BREC, ID)                                                                          \
    {                                                                                                         \
        const float dx = AREC.x - pxf;                                                                        \
        const f32x2 dy = f32x2{AREC.y, AREC.y} - pyf;                                                         \
        const float u = AREC.z * dx, wq = AREC.w * dx;                                                        \
        const f32x2 v = f32x2{BREC.x, BREC.x} * dy;                                                           \
        f32x2 t = f32x2{wq, wq} * dy;                                                                         \
        t = pk_fma(v, dy, t);                                                                                 \
        const f32x2 power = pk_fma(f32x2{u, u}, f32x2{dx, dx}, t);                                            \
        bool p0, p1;                                                                                          \
        if constexpr (MODE == 2) {                                                                            \
            const f32x2 d2 = f32x2{dx * dx, dx * dx} + dy * dy;                                               \
            p0 = d2.x <= 2.25f && BREC.y <= 0.0f; /* the null record carries pmin = 1 */                      \
            p1 = d2.y <= 2.25f && BREC.y <= 0.0f;                                                             \
        } else {                                                                                              \
            /* "power >= pmin" (pmin = -ln(255 opacity) - 1e-3 >= -5.6) is more than a pre-test of          */ \
            /* "alpha >= 1/255": it is the DOMAIN GUARD of the exp below (the integer-add ldexp is only      */ \
            /* valid for arguments in [-5.6, 0]; finished pixels are parked at y = 1e15, power = -1e30), and */ \
            /* the any-lane branch it feeds skips most steps of a deep tile, whose pixels are long finished: */ \
            /* round 4 removed both for one build — wrong pixels, and the 10 M blend 0.157 -> 1.80 ms.       */ \
            p0 = power.x <= 0.0f && power.x >= BREC.y;                                                        \
            p1 = power.y <= 0.0f && power.y >= BREC.y;                                                        \
        }                                                                                                     \
        if (__builtin_amdgcn_ballot_w64(p0 || p1) != 0ull) {                                                  \
            const float2 cq = *(const float2 *)(rbase + ID + 32); /* g, b */                                  \
            f32x2 alpha;                                                                                      \
            if constexpr (MODE == 0) {                                                                        \
                /* exp exactly as in k_blend */                                              \
                const f32x2 tt = power * f32x2{1.44269504088896340736f, 1.44269504088896340736f};             \
                const f32x2 tm = tt + f32x2{12582912.0f, 12582912.0f};                                        \
                const f32x2 n = tm - f32x2{12582912.0f, 12582912.0f};                                         \
                const f32x2 f = tt - n;                                                                       \
                f32x2 p = {0x1.5f0896p-10f, 0x1.5f0896p-10f};                                                 \
                p = pk_fma(p, f, f32x2{0x1.3cbf6cp-7f, 0x1.3cbf6cp-7f});                                      \
                p = pk_fma(p, f, f32x2{0x1.c6af6cp-5f, 0x1.c6af6cp-5f});                                      \
                p = pk_fma(p, f, f32x2{0x1.ebfa4ap-3f, 0x1.ebfa4ap-3f});                                      \
                p = pk_fma(p, f, f32x2{0x1.62e430p-1f, 0x1.62e430p-1f});                                      \
                p = pk_fma(p, f, f32x2{1.0f, 1.0f});                                                          \
                const f32x2 e = {u2f(f2u(p.x) + (f2u(tm.x) << 23)), u2f(f2u(p.y) + (f2u(tm.y) << 23))};       \
                const f32x2 oe = f32x2{BREC.z, BREC.z} * e;                                                   \
                alpha = f32x2{fminf(0.99f, oe.x), fminf(0.99f, oe.y)};                                        \
            } else {                                                                                          \
                const float flat = fminf(0.99f, BREC.z);                                                      \
                alpha = f32x2{flat, flat};                                                                    \
            }                                                                                                 \
            const bool act0 = p0 && alpha.x >= (1.0f / 255.0f);                                               \
            const bool act1 = p1 && alpha.y >= (1.0f / 255.0f);                                               \
            /* a pixel that skips this splat blends it with alpha 0: T * (1 - 0) == T and */                  \
            /* fma(rgb, 0 * T, C) == C exactly, so no selects are needed on T and C */                        \
            f32x2 alpha_eff = {act0 ? alpha.x : 0.0f, act1 ? alpha.y : 0.0f};                                 \
            f32x2 test_T = T * (f32x2{1.0f, 1.0f} - alpha_eff);                                               \
            /* (no "act &&": a pixel that skips the splat has test_T == T, and T of a live, finished or */     \
            /* out-of-image pixel is never below 1e-4) */                                                     \
            const bool fin0 = test_T.x < 0.0001f, fin1 = test_T.y < 0.0001f;                                  \
            const uint64_t f0 = __builtin_amdgcn_ballot_w64(fin0), f1 = __builtin_amdgcn_ballot_w64(fin1);    \
            if ((f0 | f1) != 0ull) { /* rare: some pixel reached T < 1e-4 -> it stops here */                 \
                if (fin0) { pyf.x = DEAD; alpha_eff.x = 0.0f; test_T.x = T.x; }                               \
                if (fin1) { pyf.y = DEAD; alpha_eff.y = 0.0f; test_T.y = T.y; }                               \
                remaining -= (uint32_t)__popcll(f0) + (uint32_t)__popcll(f1);                                 \
            }                                                                                                 \
            const f32x2 wgt = alpha_eff * T;                                                                  \
            C0 = pk_fma(f32x2{BREC.w, BREC.w}, wgt, C0);                                                      \
            C1 = pk_fma(f32x2{cq.x, cq.x}, wgt, C1);                                                          \
            C2 = pk_fma(f32x2{cq.y, cq.y}, wgt, C2);                                                          \
            T = test_T;                                                                                       \
        }                                                                                                     \
    }
            for (uint32_t q = 0; q < trips && remaining != 0u; q++) {
                const uint32_t id_B = pair >> 16;
                const float4 a_B = *(const float4 *)(rbase + id_B), b_B = *(const float4 *)(rbase + id_B + 16);
                pair = mine[q + 1 < (uint32_t)(BLEND_BATCH / 2) ? q + 1 : q];      // offsets of the next trip
                GS_BLEND_STEP(a_A, b_A, id_A)
                id_A = pair & 0xffffu;
                a_A = *(const float4 *)(rbase + id_A);
                b_A = *(const float4 *)(rbase + id_A + 16);
                GS_BLEND_STEP(a_B, b_B, id_B)
            }
#undef GS_BLEND_STEP
        }
    }
    if (ROUNDS && tk.round == 1u) {
        // (s_alive: a wave that left the loop through its break wrote 0 and writes 0 again; otherwise two barriers
        // lie between the loop's last read and this write)
        if (lane == 0) s_alive[wid] = remaining;
        __syncthreads();
        if ((s_alive[0] | s_alive[1]) != 0u) {
            // not finished: the raw state, for round 2 to resume from (T >= 1e-4 > 0 always; -T = this pixel is finished)
            if (in0)
                store16(rgba + (uint64_t)py0 * fc.width + px,
                        make_uint4(f2u(C0.x), f2u(C1.x), f2u(C2.x), f2u(pyf.x == DEAD ? -T.x : T.x)), fc.wt_stores);
            if (in1)
                store16(rgba + (uint64_t)py1 * fc.width + px,
                        make_uint4(f2u(C0.y), f2u(C1.y), f2u(C2.y), f2u(pyf.y == DEAD ? -T.y : T.y)), fc.wt_stores);
            if (tid == 0 && end > start) atomicOr(tk.open + (tile >> 5), 1u << (tile & 31u));
            return;
        }
        if (tid == 0) atomicOr(tk.done + (tile >> 5), 1u << (tile & 31u));
    }
    if (in0) {
        float4 o;
        o.x = __builtin_fmaf(T.x, fc.bg[0], C0.x);
        o.y = __builtin_fmaf(T.x, fc.bg[1], C1.x);
        o.z = __builtin_fmaf(T.x, fc.bg[2], C2.x);
        o.w = 1.0f - T.x;
        store16(rgba + (uint64_t)py0 * fc.width + px, make_uint4(f2u(o.x), f2u(o.y), f2u(o.z), f2u(o.w)), fc.wt_stores);
    }
    if (in1) {
        float4 o;
        o.x = __builtin_fmaf(T.y, fc.bg[0], C0.y);
        o.y = __builtin_fmaf(T.y, fc.bg[1], C1.y);
        o.z = __builtin_fmaf(T.y, fc.bg[2], C2.y);
        o.w = 1.0f - T.y;
        store16(rgba + (uint64_t)py1 * fc.width + px, make_uint4(f2u(o.x), f2u(o.y), f2u(o.z), f2u(o.w)), fc.wt_stores);
    }
}

}  // namespace gs
