// Micro-benchmark: 14-plane block-planar read (the preprocess pattern) with different WRITE patterns,
// to find how the preprocess outputs should be stored.  10 M records x 224 B read in every mode.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int NC = 14;

template <int MODE, bool NTL = false, bool NTS = false>
__global__ __launch_bounds__(256) void k_rw(const uint4* __restrict__ planar, uint32_t n, uint4* __restrict__ out,
                                             uint32_t* __restrict__ out4, uint2* __restrict__ out8, uint32_t* __restrict__ sums) {
    __shared__ uint4 s_out[MODE == 4 ? 3 * 1024 : 1];
    uint32_t base = blockIdx.x * 1024u, acc_all = 0;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        uint32_t i = base + k * 256 + threadIdx.x;
        uint32_t il = i < n ? i : n - 1;
        uint4 v[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint4* p = planar + ((((uint64_t)(il >> 10) * NC + c) << 10) | (il & 1023u));
            if (NTL) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 t = __builtin_nontemporal_load((const u4*)p); v[c] = make_uint4(t.x, t.y, t.z, t.w); }
            else v[c] = *p;
        }
        uint4 r0 = v[0], r1 = v[1], r2 = v[2];
#pragma unroll
        for (int c = 3; c < NC; c++) { r0.x ^= v[c].x; r1.y += v[c].y; r2.z ^= v[c].z; r0.w += v[c].w; }
        float a = __uint_as_float(r0.x & 0x3fffffffu);
#pragma unroll 8
        for (int t = 0; t < 200; t++) a = __builtin_fmaf(a, 1.0001f, 0.5f);
        r1.x ^= __float_as_uint(a);
        acc_all += r0.x + r1.x + r2.x;
        if (i < n) {
            if (MODE == 1) out4[i] = r0.x;
            if (MODE == 2) out[i] = r0;
            if (MODE == 3) { out[i] = r0; out[(uint64_t)n + i] = r1; out[2ull * n + i] = r2; }
            if (MODE == 4) { s_out[k * 256 + threadIdx.x] = r0; s_out[1024 + k * 256 + threadIdx.x] = r1; s_out[2048 + k * 256 + threadIdx.x] = r2; }
            if (MODE == 5) {
                uint32_t* o = (uint32_t*)out + (uint64_t)i * 9;
                typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));
                if (NTS) {
                    __builtin_nontemporal_store(u4a{r0.x, r0.y, r0.z, r0.w}, (u4a*)o); __builtin_nontemporal_store(u4a{r1.x, r1.y, r1.z, r1.w}, (u4a*)(o + 4));
                    __builtin_nontemporal_store(r2.x, o + 8); __builtin_nontemporal_store(r2.y, out4 + i);
                    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store(u2{r2.z, r2.w}, (u2*)(out8 + i));
                } else {
                *(u4a*)o = u4a{r0.x, r0.y, r0.z, r0.w}; *(u4a*)(o + 4) = u4a{r1.x, r1.y, r1.z, r1.w}; o[8] = r2.x;
                out4[i] = r2.y; out8[i] = make_uint2(r2.z, r2.w); }
            }
            if (MODE == 6) {   // block-planar output: [block][3][1024] uint4, one contiguous 48 KiB span per WG
                uint64_t b = (uint64_t)blockIdx.x * 3 * 1024 + (k * 256 + threadIdx.x);
                out[b] = r0; out[b + 1024] = r1; out[b + 2048] = r2;
            }
            if (MODE == 7) { out4[i] = r2.y; out8[i] = make_uint2(r2.z, r2.w); }
            // round 5: today's outputs with the packed rect — 36 B record + 4 B key + 4 B rect — against
            // a 32-byte record (two aligned 16-byte stores, never straddling a 128-byte line) + key + rect + a 1-byte
            // opacity plane
            if (MODE == 8) {
                uint32_t* o = (uint32_t*)out + (uint64_t)i * 9;
                typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));
                *(u4a*)o = u4a{r0.x, r0.y, r0.z, r0.w}; *(u4a*)(o + 4) = u4a{r1.x, r1.y, r1.z, r1.w}; o[8] = r2.x;
                out4[i] = r2.y; ((uint32_t*)out8)[i] = r2.z;
            }
            if (MODE == 9) {
                out[2ull * i] = r0; out[2ull * i + 1] = r1;
                out4[i] = r2.y; ((uint32_t*)out8)[i] = r2.z; ((uint8_t*)out8)[4ull * n + i] = (uint8_t)r2.x;
            }
        }
    }
    if (MODE == 4) {
        __syncthreads();
        uint64_t b = (uint64_t)blockIdx.x * 3 * 1024;
        for (int q = threadIdx.x; q < 3 * 1024; q += 256) out[b + q] = s_out[q];
    }
    if (acc_all == 0x12345u) sums[blockIdx.x] = acc_all;
}

template <int MODE, bool NTL = false, bool NTS = false>
static void run(const char* name, double wbytes_per, const uint4* planar, uint32_t n, uint4* out, uint32_t* out4, uint2* out8, uint32_t* sums) {
    uint32_t nb = (n + 1023) / 1024;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_rw<MODE, NTL, NTS>), dim3(nb), dim3(256), 0, 0, planar, n, out, out4, out8, sums);
    CK(hipEventRecord(a));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k_rw<MODE, NTL, NTS>), dim3(nb), dim3(256), 0, 0, planar, n, out, out4, out8, sums);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("%-44s %7.3f ms  read %.2f TB/s  total %.2f TB/s\n", name, ms, n * 224.0 / ms / 1e9, n * (224.0 + wbytes_per) / ms / 1e9);
}

int main() {
    const uint32_t n = 10000000;
    uint64_t cap = ((uint64_t)n + 1023) / 1024 * 1024;
    uint4 *planar, *out; uint32_t *out4, *sums; uint2* out8;
    CK(hipMalloc(&planar, cap * 16 * NC)); CK(hipMemset(planar, 1, cap * 16 * NC));
    CK(hipMalloc(&out, cap * 48)); CK(hipMalloc(&out4, cap * 4)); CK(hipMalloc(&out8, cap * 8)); CK(hipMalloc(&sums, cap / 1024 * 4 + 4));
    run<0>("W0 no writes", 0, planar, n, out, out4, out8, sums);
    run<1>("W1 dense 4 B", 4, planar, n, out, out4, out8, sums);
    run<7>("W7 dense 4 B + 8 B (two streams)", 12, planar, n, out, out4, out8, sums);
    run<2>("W2 dense 16 B", 16, planar, n, out, out4, out8, sums);
    run<3>("W3 48 B as 3 whole-array planes", 48, planar, n, out, out4, out8, sums);
    run<6>("W6 48 B block-planar (48 KiB span per WG)", 48, planar, n, out, out4, out8, sums);
    run<4>("W4 48 B via LDS, burst at WG end", 48, planar, n, out, out4, out8, sums);
    run<5>("W5 36 B AoS + 4 B + 8 B (current)", 48, planar, n, out, out4, out8, sums);
    run<0, true>("W0 no writes, nt loads", 0, planar, n, out, out4, out8, sums);
    run<5, true, false>("W5 nt loads", 48, planar, n, out, out4, out8, sums);
    run<5, false, true>("W5 nt stores", 48, planar, n, out, out4, out8, sums);
    run<5, true, true>("W5 nt loads + nt stores", 48, planar, n, out, out4, out8, sums);
    run<7, true, false>("W7 nt loads", 12, planar, n, out, out4, out8, sums);
    run<8, true>("W8 36 B AoS + 4 B + 4 B (round 4/5), nt loads", 44, planar, n, out, out4, out8, sums);
    run<9, true>("W9 32 B AoS + 4 B + 4 B + 1 B, nt loads", 41, planar, n, out, out4, out8, sums);
    run<8, true>("W8 again", 44, planar, n, out, out4, out8, sums);
    run<9, true>("W9 again", 41, planar, n, out, out4, out8, sums);
    run<0>("W0 no writes (again)", 0, planar, n, out, out4, out8, sums);
    return 0;
}
