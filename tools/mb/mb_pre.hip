// Stand-alone timing harness for k_preprocess<ShSingle,RotScale> on a device-generated scene.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../wgpu-3dgs-core_amd/csrc/gs_render_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ inline float u01(uint32_t i, uint32_t k) { return (hash(i * 64u + k) >> 8) * (1.0f / 16777216.0f); }

// fills the chunk-planar mirror of ShSingle/RotScale records (224 B = 14 chunks)
__global__ void k_gen(uint4* planar, uint64_t stride, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[56];
    float px = -14.f + 28.f * u01(i, 0), py = -8.f + 16.f * u01(i, 1), pz = -(2.f + 24.f * u01(i, 2));
    w[0] = __float_as_uint(px); w[1] = __float_as_uint(py); w[2] = __float_as_uint(pz);
    w[3] = hash(i) | 0x20000000u;
    for (int k = 0; k < 45; k++) w[4 + k] = __float_as_uint(-0.25f + 0.5f * u01(i, 16 + k));
    float q[4] = {u01(i, 3) - .5f, u01(i, 4) - .5f, u01(i, 5) - .5f, u01(i, 6) - .5f};
    float l = sqrtf(q[0]*q[0]+q[1]*q[1]+q[2]*q[2]+q[3]*q[3]) + 1e-9f;
    for (int k = 0; k < 4; k++) w[49 + k] = __float_as_uint(q[k] / l);
    for (int k = 0; k < 3; k++) w[53 + k] = __float_as_uint(expf(-3.6f + (u01(i, 7 + k) - 0.5f) * 1.7f));
    for (int c = 0; c < 14; c++) planar[gs::planar_at(c, i, 14)] = make_uint4(w[4*c], w[4*c+1], w[4*c+2], w[4*c+3]);
}

int main(int argc, char** argv) {
    const uint32_t n = 10000000;
    uint64_t stride = (n + 1023) / 1024 * 1024;
    uint4 *planar; uint32_t *proj, *depth, *ct, *cv; uint2* rect;
    uint32_t nchunks = (n + gs::PP_CHUNK - 1) / gs::PP_CHUNK;
    CK(hipMalloc(&planar, stride * 16 * 14)); CK(hipMalloc(&proj, (size_t)n * 36 + 16));
    CK(hipMalloc(&depth, n * 4ull)); CK(hipMalloc(&rect, n * 8ull));
    CK(hipMalloc(&ct, nchunks * 4ull)); CK(hipMalloc(&cv, nchunks * 4ull)); uint2* cr; CK(hipMalloc(&cr, nchunks * 8ull));
    hipLaunchKernelGGL(k_gen, dim3((n + 255) / 256), dim3(256), 0, 0, planar, stride, n);
    gs::FrameConsts fc; memset(&fc, 0, sizeof(fc));
    for (int i = 0; i < 4; i++) { fc.M[5*i] = 1.f; fc.V[5*i] = 1.f; }
    for (int i = 0; i < 3; i++) { fc.ISR[4*i] = 1.f; }
    fc.WS[0] = 1.f; fc.WS[4] = -1.f; fc.WS[8] = -1.f;
    fc.fx = fc.fy = 935.3f; fc.cx = 960.f; fc.cy = 540.f; fc.near_plane = 0.1f; fc.far_plane = 100.f;
    fc.size2 = 1.f; fc.limx = 1.3f * 960.f / 935.3f; fc.limy = 1.3f * 540.f / 935.3f; fc.max_std_dev = 3.f;
    fc.sh_deg = argc > 1 ? atoi(argv[1]) : 3; fc.width = 1920; fc.height = 1080; fc.tiles_x = 120; fc.tiles_y = 68; fc.band_ty1 = 68;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch = [&]() { hipLaunchKernelGGL((gs::k_preprocess<0, 0>), dim3(nchunks), dim3(gs::PP_THREADS), 0, 0,
        (const uint4*)planar, n, fc, proj, depth, rect, ct, cv, cr, (const float*)nullptr); };
    for (int w = 0; w < 3; w++) launch();
    CK(hipEventRecord(a));
    for (int r = 0; r < 10; r++) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    std::vector<uint32_t> h(nchunks); CK(hipMemcpy(h.data(), cv, nchunks * 4ull, hipMemcpyDeviceToHost));
    uint64_t vis = 0; for (auto x : h) vis += x;
    printf("k_preprocess<0,0> sh_deg=%u: %.3f ms  read %.2f TB/s  visible %llu\n", fc.sh_deg, ms, n * 224.0 / ms / 1e9, (unsigned long long)vis);
    return 0;
}
