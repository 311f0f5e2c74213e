// Micro-benchmark 2: preprocess-like streaming with capped occupancy and synthetic ALU work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NC, int ITEMS, int ALU>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ planar, uint64_t stride, uint32_t n,
                                              uint4* __restrict__ out, uint32_t* __restrict__ out4, float seed) {
    extern __shared__ char dummy[];
    uint32_t base = blockIdx.x * 256 * ITEMS;
#pragma unroll 1
    for (int k = 0; k < ITEMS; k++) {
        uint32_t i = base + k * 256 + threadIdx.x;
        if (i < n) {
            uint4 v[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) v[c] = planar[(uint64_t)c * stride + i];
            float acc = seed;
#pragma unroll
            for (int c = 0; c < NC; c++) acc += __uint_as_float(v[c].x ^ v[c].y ^ v[c].z ^ v[c].w);
            float a = acc, b = acc * 0.5f, c2 = acc + 1.0f, d = acc - 1.0f;
#pragma unroll 8
            for (int t = 0; t < ALU / 4; t++) {  // 4 independent chains
                a = __builtin_fmaf(a, 1.0001f, 0.5f); b = __builtin_fmaf(b, 0.9999f, 0.25f);
                c2 = __builtin_fmaf(c2, 1.0002f, 0.125f); d = __builtin_fmaf(d, 0.9998f, 0.0625f);
            }
            uint32_t r = __float_as_uint(a + b + c2 + d);
            uint4* o = out + (uint64_t)i * 3;
            o[0] = make_uint4(r, r, r, r); o[1] = v[0]; o[2] = v[1];
            out4[i] = r; out4[n + i] = r + 1;
        }
    }
}

template <int ITEMS, int ALU>
void run(const char* name, const uint4* planar, uint64_t stride, uint32_t n, uint4* out, uint32_t* out4, int lds) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    uint32_t grid = (n + 256 * ITEMS - 1) / (256 * ITEMS);
    auto kern = k_read<14, ITEMS, ALU>;
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, planar, stride, n, out, out4, 1.0f);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, planar, stride, n, out, out4, 1.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    double rb = (double)n * 16 * 14, wb = (double)n * 56;
    printf("%-40s lds %6d  %8.3f ms  read %.2f TB/s  total %.2f TB/s\n", name, lds, ms, rb / ms / 1e9, (rb + wb) / ms / 1e9);
}

int main() {
    const uint32_t n = 10000000;
    uint64_t stride = (n + 63) / 64 * 64;
    uint4 *planar, *out; uint32_t* out4;
    CK(hipMalloc(&planar, stride * 16 * 14)); CK(hipMalloc(&out, (size_t)n * 48)); CK(hipMalloc(&out4, (size_t)n * 8));
    CK(hipMemset(planar, 1, stride * 16 * 14));
    int caps[] = {0, 20000, 32000, 40000, 53000};   // -> 8, 8, 5, 4, 3 blocks per CU
    for (int lds : caps) run<4, 0>("alu0 items4", planar, stride, n, out, out4, lds);
    for (int lds : caps) run<4, 700>("alu700 items4", planar, stride, n, out, out4, lds);
    for (int lds : caps) run<1, 700>("alu700 items1", planar, stride, n, out, out4, lds);
    run<4, 1400>("alu1400 items4", planar, stride, n, out, out4, 32000);
    run<4, 2800>("alu2800 items4", planar, stride, n, out, out4, 32000);
    return 0;
}
