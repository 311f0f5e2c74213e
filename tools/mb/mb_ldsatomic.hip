// Probe: does ds_add_rtn_u32 return pre-add values in increasing lane order when several lanes of
// one wave instruction hit the same LDS address?  (needed for a 1-instruction stable rank)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_probe(const uint32_t* digits, uint32_t* olds, int rounds) {
    __shared__ uint32_t s[4][256];
    uint32_t wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 1024; i += 256) ((uint32_t*)s)[i] = 0;
    __syncthreads();
    for (int r = 0; r < rounds; r++) {
        uint32_t idx = (blockIdx.x * rounds + r) * 256 + threadIdx.x;
        uint32_t d = digits[idx];
        olds[idx] = atomicAdd(&s[wid][d], 1u);
    }
}

int main() {
    const int blocks = 2048, rounds = 16, n = blocks * rounds * 256;
    std::vector<uint32_t> h(n);
    uint32_t *dd, *dold; CK(hipMalloc(&dd, n * 4)); CK(hipMalloc(&dold, n * 4));
    long long bad_total = 0;
    for (int pattern = 0; pattern < 6; pattern++) {
        srand(1234 + pattern);
        for (int i = 0; i < n; i++) {
            switch (pattern) {
            case 0: h[i] = rand() & 255; break;          // uniform
            case 1: h[i] = rand() & 3; break;            // few bins
            case 2: h[i] = 7; break;                     // all same
            case 3: h[i] = (i & 1) ? 5 : (rand() & 255); break;
            case 4: h[i] = (i >> 3) & 255; break;        // runs of 8
            default: h[i] = (rand() % 10 == 0) ? rand() & 255 : 33; break;
            }
        }
        CK(hipMemcpy(dd, h.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, dd, dold, rounds);
        std::vector<uint32_t> o(n);
        CK(hipMemcpy(o.data(), dold, n * 4, hipMemcpyDeviceToHost));
        long long bad = 0;
        // within each wave-round: same digit => old strictly increasing with lane; and equals
        // (count before this round) + (number of lower lanes with the same digit)
        for (int b = 0; b < blocks; b++) {
            uint32_t cnt[4][256] = {};
            for (int r = 0; r < rounds; r++)
                for (int w = 0; w < 4; w++) {
                    uint32_t base = (b * rounds + r) * 256 + w * 64;
                    uint32_t seen[256] = {};
                    for (int l = 0; l < 64; l++) {
                        uint32_t d = h[base + l];
                        if (o[base + l] != cnt[w][d] + seen[d]) bad++;
                        seen[d]++;
                    }
                    for (int l = 0; l < 64; l++) cnt[w][h[base + l]]++;
                }
        }
        printf("pattern %d: %lld violations of lane-ordered return values (of %d)\n", pattern, bad, n);
        bad_total += bad;
    }
    printf("%s\n", bad_total ? "NOT ORDERED" : "ORDERED: ds_add_rtn returns lane-ordered ranks");
    return 0;
}
