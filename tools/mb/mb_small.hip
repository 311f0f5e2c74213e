// What do the product's SMALL kernels cost outside a frame?  One radix pass of the 1 M frame's depth sort —
// k_sort_hist<u32, 9, false, 16>, k_sort_scan_rows_small<4096>, k_sort_scatter<u32, FAST_RANK, 9, false, 16, u32> on
// 708 615 random 27-bit keys, the very kernels of gs_render_kernels.h — launched back to back on one stream 300 times:
// hist alone, hist + scan, hist + scan + scatter; the differences are the marginal cost of each kernel in a hot loop.
// In the frame's kernel trace the three take 5.05 + 4.70 + 9.14 us (profiles/r04_1m_summary.md); tools/mb/mb_fold.hip's
// skeleton of the first two takes 2.6 + 2.6 us.  This tool tells which of the two the real kernels follow.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -o tools/mb/mb_small tools/mb/mb_small.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../wgpu-3dgs-core_amd/csrc/gs_render_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// What a frame does between two launches of the same small kernel, in two kernels: ~200 MB streamed through the L2s,
// and 64 KB of straight-line code through every CU's instruction cache.
__global__ __launch_bounds__(256) void k_thrash_data(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(64) void k_thrash_code(float *out, float seed) {
    float a = seed + threadIdx.x, b = 1.0f;
#pragma unroll
    for (int i = 0; i < 8192; i++) {           // 2 x 8192 VALU instructions with literal operands: ~128 KB of code
        a = a * (1.0f + i * 1e-7f) + b;
        b = b * 0.999f + (float)i;
    }
    if (a == 12345.678f) out[0] = a + b;      // never true: keeps the chain alive
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    const uint32_t n = 708615, RB = 9, ITEMS = 16, TILE = 256 * ITEMS, R = 1u << RB;
    const uint32_t nb = (n + TILE - 1) / TILE;
    std::vector<uint32_t> h(n), v(n);
    srand(7);
    for (uint32_t i = 0; i < n; i++) {
        h[i] = ((uint32_t)rand() * 2654435761u) >> 5;      // 27 bits
        v[i] = i;
    }
    uint32_t *keys[2], *vals[2], *ghist, *totals, *count_dev;
    for (int s = 0; s < 2; s++) {
        CK(hipMalloc(&keys[s], (size_t)(nb + 1) * TILE * 4));
        CK(hipMalloc(&vals[s], (size_t)(nb + 1) * TILE * 4));
    }
    CK(hipMalloc(&ghist, (size_t)nb * R * 4));
    CK(hipMalloc(&totals, R * 4));
    CK(hipMalloc(&count_dev, 4));
    CK(hipMemcpy(keys[0], h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vals[0], v.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(count_dev, &n, 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const gs::SortCount sc{n, count_dev};
    const uint32_t mask = R - 1u;
    auto hist = [&] {
        hipLaunchKernelGGL((gs::k_sort_hist<uint32_t, RB, false, ITEMS>), dim3(nb), dim3(gs::SORT_THREADS), 0, st, keys[0], sc, 9u, mask,
                           ghist, (const uint32_t *)nullptr, nb, 0u);
    };
    auto scan = [&] {
        hipLaunchKernelGGL((gs::k_sort_scan_rows_small<(int)TILE>), dim3((R + 3u) / 4u), dim3(256), 0, st, ghist, nb, sc, totals, R);
    };
    auto scatter = [&] {
        hipLaunchKernelGGL((gs::k_sort_scatter<uint32_t, true, RB, false, ITEMS, uint32_t>), dim3(nb), dim3(gs::SORT_THREADS), 0, st,
                           (const uint32_t *)keys[0], (const uint32_t *)vals[0], keys[1], 0u, vals[1], sc, 9u, mask,
                           (const uint32_t *)ghist, (const uint32_t *)totals, (const uint32_t *)nullptr, (uint32_t *)nullptr, nb, 0u,
                           (uint32_t *)nullptr);
    };
    uint4 *ta, *tb;
    const size_t n16 = (96u << 20) / 16;
    CK(hipMalloc(&ta, n16 * 16));
    CK(hipMalloc(&tb, n16 * 16));
    CK(hipMemset(ta, 1, n16 * 16));
    float *fout;
    CK(hipMalloc(&fout, 4));
    const bool cold = argc > 2 && atoi(argv[2]) != 0;
    auto thrash = [&] {
        hipLaunchKernelGGL(k_thrash_data, dim3(2048), dim3(256), 0, st, (const uint4 *)ta, tb, n16);
        hipLaunchKernelGGL(k_thrash_code, dim3(512), dim3(64), 0, st, fout, 1.0f);
    };
    double us[4];
    for (int rep = 0; rep < 2; rep++)
        for (int variant = cold ? -1 : 0; variant < 3; variant++) {
            auto run = [&] {
                if (cold) thrash();
                if (variant >= 0) hist();
                if (variant >= 1) scan();
                if (variant >= 2) scatter();
            };
            if (variant < 0) {
                for (int i = 0; i < 5; i++) run();
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; i++) run();
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                us[3] = ms * 1e3 / iters;
                continue;
            }
            for (int i = 0; i < 20; i++) run();
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; i++) run();
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            us[variant] = ms * 1e3 / iters - (cold ? us[3] : 0.0);
            if (rep && variant == 2 && cold)
                printf("COLD (192 MB copied + 128 KB of code run between passes, %.1f us, subtracted): ", us[3]);
            if (rep && variant == 2)
                printf("708 615 keys, %u tiles: hist %.2f us | + scan %.2f us | + scatter %.2f us | pass %.2f us "
                       "(frame's kernel trace: 5.05 + 4.70 + 9.14 = 18.89)\n", nb, us[0], us[1] - us[0], us[2] - us[1], us[2]);
        }
    // the pass must have sorted by bits 9..17, stably
    std::vector<uint32_t> ok(n), ov(n);
    CK(hipMemcpy(ok.data(), keys[1], n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ov.data(), vals[1], n * 4, hipMemcpyDeviceToHost));
    uint64_t bad = 0;
    for (uint32_t i = 1; i < n; i++) {
        const uint32_t a = (ok[i - 1] >> 9) & mask, b = (ok[i] >> 9) & mask;
        if (a > b || (a == b && ov[i - 1] > ov[i])) bad++;
    }
    for (uint32_t i = 0; i < n; i++)
        if (ok[i] != h[ov[i]]) bad++;
    printf("order check: %llu violations\n", (unsigned long long)bad);
    return bad != 0;
}
