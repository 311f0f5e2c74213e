// Where does gs::k_bucket_sort spend its time?  The product kernel of gs_render_kernels.h, compiled here with
// GS3D_BKT_STAMPS (s_memtime at the phase boundaries of the register path, thread 0 of every bucket), on the bucket
// sizes of the 1 M frame's depth sort: 116 live buckets of 27-bit depth keys of the synthetic scene (z uniform in
// 2..26), and on single buckets of 4 000 / 8 000 / 16 000 / 21 000 / 30 000 keys alone on the chip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGS3D_BKT_STAMPS -o tools/mb/mb_bucket tools/mb/mb_bucket.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../wgpu-3dgs-core_amd/csrc/gs_render_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static hipStream_t st;
static hipEvent_t e0, e1;

template <typename K, int RB, int T>
static void run(const char *name, const std::vector<uint32_t> &keys, uint32_t low_bits, int iters) {
    // stable partition by the top digit on the host (what the scatter pass leaves)
    const uint32_t n = (uint32_t)keys.size(), nb = 1024;
    std::vector<uint32_t> totals(nb, 0);
    for (uint32_t i = 0; i < n; i++) totals[keys[i] >> low_bits]++;
    std::vector<uint32_t> start(nb + 1, 0);
    for (uint32_t b = 0; b < nb; b++) start[b + 1] = start[b] + totals[b];
    std::vector<K> pk(n);
    std::vector<uint32_t> pv(n), cur(start.begin(), start.end() - 1);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t b = keys[i] >> low_bits;
        pk[cur[b]] = (K)keys[i];
        pv[cur[b]++] = i;
    }
    uint32_t *d_tot, *d_start, *d_v, *d_out;
    K *d_k;
    unsigned long long *d_st;
    CK(hipMalloc(&d_tot, nb * 4));
    CK(hipMalloc(&d_start, nb * 4));
    CK(hipMalloc(&d_k, (size_t)n * sizeof(K) + 64));
    CK(hipMalloc(&d_v, (size_t)n * 4 + 64));
    CK(hipMalloc(&d_out, (size_t)n * 4 + 64));
    CK(hipMalloc(&d_st, nb * 16 * 8));
    CK(hipMemset(d_st, 0, nb * 16 * 8));
    CK(hipMemcpy(d_tot, totals.data(), nb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_start, start.data(), nb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_k, pk.data(), (size_t)n * sizeof(K), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_v, pv.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    gs::BucketSortIO io;
    memset(&io, 0, sizeof(io));
    io.totals = d_tot;
    io.starts = d_start;
    io.nb = nb;
    io.keys_in = d_k;
    io.vals_in = d_v;
    io.vals_out = d_out;
    io.low_bits = low_bits;
    io.stamps = d_st;
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((gs::k_bucket_sort<K, RB, T, true>), dim3(nb), dim3(T), 0, st, io);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((gs::k_bucket_sort<K, RB, T, true>), dim3(nb), dim3(T), 0, st, io);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // check against a stable sort of the original
    std::vector<uint32_t> out(n), ref(n);
    CK(hipMemcpy(out.data(), d_out, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) ref[i] = i;
    std::stable_sort(ref.begin(), ref.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    const bool ok = out == ref;
    std::vector<unsigned long long> stp(nb * 16);
    CK(hipMemcpy(stp.data(), d_st, nb * 16 * 8, hipMemcpyDeviceToHost));
    uint32_t big = 0, live = 0;
    for (uint32_t b = 0; b < nb; b++) {
        if (totals[b] > totals[big]) big = b;
        live += totals[b] != 0;
    }
    printf("%-30s T %4d  n %7u  live buckets %4u  largest %5u  %7.2f us per launch  %s\n", name, T, n, live, totals[big], ms * 1000.0f / iters,
           ok ? "sorted OK" : "WRONG ORDER");
    const char *phase[10] = {"", "load keys+clear", "rank 1", "wave bases 1", "place 1", "reload+clear", "rank 2", "wave bases 2", "place 2", "values via LDS"};
    const unsigned long long *q = &stp[big * 16];
    printf("    largest bucket:");
    for (int i = 1; i <= 9; i++)
        if (q[i] && q[i - 1]) printf("  %s %.2f", phase[i], (double)(q[i] - q[i - 1]) * 0.01);
    unsigned long long last = q[9] ? q[9] : q[4];
    printf("   total %.2f us\n", (double)(last - q[0]) * 0.01);
    CK(hipFree(d_tot)); CK(hipFree(d_start)); CK(hipFree(d_k)); CK(hipFree(d_v)); CK(hipFree(d_out)); CK(hipFree(d_st));
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t near_bits = fbits(0.1f);
    srand(11);
    {   // the 1 M frame: 708 615 visible, depth roughly as the synthetic scene's, 27-bit keys, 10-bit top digit
        std::vector<uint32_t> keys(708615);
        for (auto &k : keys) k = fbits(2.0f + 24.0f * (float)rand() / (float)RAND_MAX * (0.5f + 0.5f * (float)rand() / (float)RAND_MAX)) - near_bits;
        run<uint32_t, 9, 1024>("1 M frame's depth keys", keys, 17, iters);
    }
    for (uint32_t n : {4000u, 8000u, 16000u, 21000u, 30000u}) {
        std::vector<uint32_t> keys(n);
        for (auto &k : keys) k = (400u << 17) | (((uint32_t)rand() * 2654435761u) >> 15);
        char name[64];
        snprintf(name, sizeof(name), "one bucket of %u", n);
        run<uint32_t, 9, 1024>(name, keys, 17, iters);
    }
    {   // the 1 M frame's tile sort: 2.55 M pairs, 8160 tiles (13 bits), centre tiles denser
        std::vector<uint32_t> keys(2550276);
        for (auto &k : keys) {
            const float u = (float)rand() / (float)RAND_MAX, v = (float)rand() / (float)RAND_MAX, w = (float)rand() / (float)RAND_MAX;
            const uint32_t tx = (uint32_t)(60.0f + (u - 0.5f) * 120.0f * (0.4f + 0.6f * w)) % 120u, ty = (uint32_t)(34.0f + (v - 0.5f) * 68.0f * (0.4f + 0.6f * w)) % 68u;
            k = ty * 120u + tx;
        }
        run<uint16_t, 6, 256>("1 M frame's tile ids", keys, 3, iters);
        run<uint16_t, 6, 1024>("1 M frame's tile ids", keys, 3, iters);
    }
    return 0;
}
