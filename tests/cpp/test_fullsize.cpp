// Full-size parity WITHOUT Python or torch: the product library on the HIP runtime it
// is compiled and linked for (/opt/rocm, what a Rust or C++ host gets), driven through the C++ mirror
// include/gs3d.hpp.  Generates a BASELINE.json workload with tools/gs_synth.c (plain C, linked in),
// renders it and compares the sha256 of the packed scene and of the f32 RGBA frame with the hashes the
// CPU oracle produced offline (tests/golden/fullsize_v2.json; handed over on the command line by
// tests/test_cpp_mirror.py so that this file needs no JSON reader):
//
//   test_fullsize <n> <sh> <cov> <sh_deg> <width> <height> <scene_sha256> <frame_sha256> <frame_sha256_index_order>
//
// Prints the HIP version of the headers next to the version of the runtime in use.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "gs3d.hpp"

extern "C" void gs_synth_scene(uint64_t seed, uint64_t first, uint64_t count, gs_gaussian *out);   // tools/gs_synth.c

using namespace gs3d;
#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

// FIPS 180-4 SHA-256 (test-local; the product has no use for a hash)
struct Sha256 {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint8_t block[64];
    size_t fill = 0;
    uint64_t total = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void compress(const uint8_t *p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u,
            0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu,
            0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u,
            0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
            0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u,
            0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
            0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const void *data, size_t n) {
        const uint8_t *p = (const uint8_t *)data;
        total += n;
        while (n) {
            if (fill == 0 && n >= 64) { compress(p); p += 64; n -= 64; continue; }
            size_t k = 64 - fill < n ? 64 - fill : n;
            std::memcpy(block + fill, p, k);
            fill += k; p += k; n -= k;
            if (fill == 64) { compress(block); fill = 0; }
        }
    }
    std::string hex() {
        uint64_t bits = total * 8;
        uint8_t pad[72] = {0x80};
        size_t padn = (fill < 56 ? 56 - fill : 120 - fill);
        update(pad, padn);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(len, 8);
        char out[65];
        for (int i = 0; i < 8; i++) std::snprintf(out + 8 * i, 9, "%08x", h[i]);
        return std::string(out, 64);
    }
};

template <class G>
static int run(Device &dev, Stream &s, size_t n, uint32_t sh_deg, uint32_t W, uint32_t H, const char *scene_sha, const char *frame_sha,
               const char *frame_sha_index) {
    // scene in slices of 1 M Gaussians, exactly as tests/test_gpu_fullsize.py::_upload builds it
    GaussiansBuffer<G> buf = GaussiansBuffer<G>::new_empty(dev, n);
    Sha256 scene;
    const size_t step = 1000000;
    std::vector<Gaussian> g;
    for (size_t first = 0; first < n; first += step) {
        const size_t cnt = n - first < step ? n - first : step;
        g.resize(cnt);
        gs_synth_scene(0x3D650001ull, first, cnt, g.data());
        std::vector<uint8_t> pods = G::from_gaussians(g);
        scene.update(pods.data(), pods.size());
        buf.update_range_with_pod(s, first, pods);
    }
    s.synchronize();
    REQUIRE(scene.hex() == scene_sha);
    gs_camera cam;
    const float eye[3] = {0, 0, 0}, target[3] = {0, 0, -1}, up[3] = {0, 1, 0};
    gs_camera_look_at(eye, target, up, (float)(60.0 * 3.14159265358979323846 / 180.0), W, H, 0.1f, 100.0f, &cam);
    auto gt = gaussian_transform_pod(1.0f, GS_DISPLAY_SPLAT, (uint8_t)sh_deg, false, 3.0f);
    REQUIRE(gt.has_value());
    gs_model_transform_pod mt;
    gs_model_transform_pod_default(&mt);
    Buffer img(dev, (size_t)W * H * 16);
    Renderer r(dev);
    for (int pass = 0; pass < 2; pass++) {
        const bool spatial = pass == 0;
        buf.set_spatial_order(spatial);
        r.render(s, buf, *gt, mt, cam, (float *)img.device_ptr());
        auto fr = r.wait_frame();
        REQUIRE(fr.flags == 0 && fr.gaussians == n);
        // steady state: 5 untimed frames (clocks, the renderer's choice of sorts), then 20 pipelined frames timed on
        // the host around a stream synchronise
        for (int i = 0; i < 5; i++) r.render(s, buf, *gt, mt, cam, (float *)img.device_ptr());
        fr = r.wait_frame();
        const int frames = 20;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < frames; i++) r.render(s, buf, *gt, mt, cam, (float *)img.device_ptr());
        fr = r.wait_frame();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
        auto px = img.download<float>(s);
        Sha256 frame;
        frame.update(px.data(), px.size() * 4);
        const std::string got = frame.hex();
        const gs_sort_info si = r.sort_info();
        std::printf("%s order: visible %llu pairs %llu launches %u rounds %u (first %u, %u tiles finished)  %.4f ms/frame  sha256 %s\n",
                    spatial ? "spatial" : "index", (unsigned long long)fr.visible, (unsigned long long)fr.pairs, fr.launches, si.rounds,
                    si.round1, si.tiles_done, ms, got.c_str());
        REQUIRE(got == (spatial ? frame_sha : frame_sha_index));
    }
    // Frames in flight across the boundary (gs3d::FrameRing): three renderers on three stream
    // priorities take the frames in turn, each lane into its own image; wait(lane) presents them in order.  Every
    // image must be the golden frame.
    buf.set_spatial_order(true);
    {
        FrameRing ring(dev, 3);
        std::vector<Buffer> imgs;
        for (size_t k = 0; k < ring.size(); k++) imgs.emplace_back(dev, (size_t)W * H * 16);
        for (int i = 0; i < 9; i++) {          // each lane: its sizing frame, then two more (the sorts settle)
            const size_t lane = ring.render(buf, *gt, mt, cam, (float *)imgs[i % 3].device_ptr());
            REQUIRE(lane == (size_t)(i % 3));
            if (i < 3) REQUIRE(ring.wait(lane).flags == 0);
        }
        for (size_t k = 0; k < ring.size(); k++) REQUIRE(ring.wait(k).flags == 0);
        const int frames = 21;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < frames; i++) ring.render(buf, *gt, mt, cam, (float *)imgs[i % 3].device_ptr());
        gs_frame_result fr{};
        for (size_t k = 0; k < ring.size(); k++) {
            fr = ring.wait(k);
            REQUIRE(fr.flags == 0 && fr.gaussians == n);
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
        for (size_t k = 0; k < ring.size(); k++) {
            auto px = imgs[k].download<float>(ring.stream(k));
            Sha256 frame;
            frame.update(px.data(), px.size() * 4);
            REQUIRE(frame.hex() == frame_sha);
        }
        std::printf("frame ring: 3 lanes (priorities %d %d %d), launches %u  %.4f ms/frame, every lane's image == golden\n", ring.priority(0),
                    ring.priority(1), ring.priority(2), fr.launches, ms);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 10) {
        std::printf("usage: test_fullsize n sh cov sh_deg width height scene_sha256 frame_sha256 frame_sha256_index_order\n");
        return 2;
    }
    const size_t n = std::strtoull(argv[1], nullptr, 10);
    const int sh = std::atoi(argv[2]), cov = std::atoi(argv[3]);
    const uint32_t sh_deg = (uint32_t)std::atoi(argv[4]), W = (uint32_t)std::atoi(argv[5]), H = (uint32_t)std::atoi(argv[6]);
    int32_t compiled = 0, runtime = 0, driver = 0;
    gs_hip_versions(&compiled, &runtime, &driver);
    std::printf("HIP headers %d, runtime %d, driver %d (no Python, no torch in this process)\n", compiled, runtime, driver);
    std::fflush(stdout);
    Device dev(0);
    Stream s(dev);
    int rc;
    if (sh == GS_SH_NONE && cov == GS_COV3D_ROT_SCALE) rc = run<GaussianPodWithShNoneCov3dRotScaleConfigs>(dev, s, n, sh_deg, W, H, argv[7], argv[8], argv[9]);
    else if (sh == GS_SH_SINGLE && cov == GS_COV3D_ROT_SCALE) rc = run<GaussianPodWithShSingleCov3dRotScaleConfigs>(dev, s, n, sh_deg, W, H, argv[7], argv[8], argv[9]);
    else if (sh == GS_SH_HALF && cov == GS_COV3D_ROT_SCALE) rc = run<GaussianPodWithShHalfCov3dRotScaleConfigs>(dev, s, n, sh_deg, W, H, argv[7], argv[8], argv[9]);
    else { std::printf("unsupported POD configuration %d/%d\n", sh, cov); return 2; }
    if (rc) return rc;
    std::printf("cpp fullsize OK\n");
    return 0;
}
