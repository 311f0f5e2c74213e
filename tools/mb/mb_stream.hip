// Micro-benchmark: what does the preprocess access pattern reach without the arithmetic?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NC, int ITEMS, int WRITE_MODE>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ planar, uint64_t stride, uint32_t n,
                                              uint4* __restrict__ out, uint32_t* __restrict__ out4) {
    uint32_t base = blockIdx.x * 256 * ITEMS;
#pragma unroll 1
    for (int k = 0; k < ITEMS; k++) {
        uint32_t i = base + k * 256 + threadIdx.x;
        if (i < n) {
            uint4 v[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) v[c] = planar[(uint64_t)c * stride + i];
            uint32_t acc = 0;
#pragma unroll
            for (int c = 0; c < NC; c++) acc += v[c].x ^ v[c].y ^ v[c].z ^ v[c].w;
            if (WRITE_MODE == 0) { out4[i] = acc; }
            if (WRITE_MODE == 1) {  // AoS 48 B strided + 2 x u32
                uint4* o = out + (uint64_t)i * 3;
                o[0] = make_uint4(acc, acc, acc, acc); o[1] = v[0]; o[2] = v[1];
                out4[i] = acc; out4[n + i] = acc + 1;
            }
            if (WRITE_MODE == 2) {  // planar 3 x 16 B + 2 x u32
                out[i] = make_uint4(acc, acc, acc, acc); out[(uint64_t)n + i] = v[0]; out[2ull * n + i] = v[1];
                out4[i] = acc; out4[n + i] = acc + 1;
            }
        }
    }
}

template <int NC, int ITEMS, int WM>
void run(const char* name, const uint4* planar, uint64_t stride, uint32_t n, uint4* out, uint32_t* out4, double rbytes, double wbytes) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    uint32_t grid = (n + 256 * ITEMS - 1) / (256 * ITEMS);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k_read<NC, ITEMS, WM>), dim3(grid), dim3(256), 0, 0, planar, stride, n, out, out4);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_read<NC, ITEMS, WM>), dim3(grid), dim3(256), 0, 0, planar, stride, n, out, out4);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("%-34s %8.3f ms  read %.2f TB/s  total %.2f TB/s\n", name, ms, rbytes / ms / 1e9, (rbytes + wbytes) / ms / 1e9);
}

int main() {
    const uint32_t n = 10000000; const int NC = 14;
    uint64_t stride = (n + 63) / 64 * 64;
    uint4 *planar, *out; uint32_t* out4;
    CK(hipMalloc(&planar, stride * 16 * NC)); CK(hipMalloc(&out, (size_t)n * 48)); CK(hipMalloc(&out4, (size_t)n * 8));
    CK(hipMemset(planar, 1, stride * 16 * NC));
    double rb = (double)n * 16 * NC;
    run<14, 4, 0>("read14 items4 write4B", planar, stride, n, out, out4, rb, n * 4.0);
    run<14, 1, 0>("read14 items1 write4B", planar, stride, n, out, out4, rb, n * 4.0);
    run<14, 4, 1>("read14 items4 write AoS48+8", planar, stride, n, out, out4, rb, n * 56.0);
    run<14, 4, 2>("read14 items4 write planar48+8", planar, stride, n, out, out4, rb, n * 56.0);
    run<14, 1, 2>("read14 items1 write planar48+8", planar, stride, n, out, out4, rb, n * 56.0);
    run<14, 8, 2>("read14 items8 write planar48+8", planar, stride, n, out, out4, rb, n * 56.0);
    return 0;
}
