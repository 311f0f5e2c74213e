// C++ host-mirror test (runs on the GPU box): exercises include/gs3d.hpp the way the reference's
// Rust tests exercise its API — buffer round trips and error variants (tests/buffer/gaussian.rs),
// ComputeBundle array_map_add incl. builder errors (tests/e2e/compute_bundle.rs), one frame.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "gs3d.hpp"

using namespace gs3d;
#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static Gaussian given_gaussian(uint32_t seed) {   // tests/common/given.rs:48-81
    Gaussian g{};
    float b = (float)seed;
    float q[4] = {b + 0.1f, b + 0.2f, b + 0.3f, b + 0.4f};
    float l = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) g.rot[i] = q[i] / l;
    g.pos[0] = b + 1.1f; g.pos[1] = b + 2.2f; g.pos[2] = b + 3.3f;
    for (int i = 0; i < 4; i++) g.color[i] = (uint8_t)std::fmod(b + 11.0f * (i + 1), 256.0f);
    for (int i = 0; i < 15; i++) for (int c = 0; c < 3; c++)
        g.sh[3 * i + c] = std::fmod(b + i * 0.3f + 0.1f * (c + 1), 2.0f) - 1.0f;
    g.scale[0] = b + 0.12f; g.scale[1] = b + 0.34f; g.scale[2] = b + 0.56f;
    return g;
}

// host-only source formats: tests/e2e/ply.rs:68-86 and tests/e2e/spz.rs:82-93,277-293
static int host_codecs() {
    std::vector<Gaussian> gs = {given_gaussian(42), given_gaussian(123)};
    auto ply = PlyGaussians::from_gaussians(gs);
    auto bytes = ply.write_to();
    auto ply2 = PlyGaussians::read_from(bytes.data(), bytes.size());
    REQUIRE(ply2.inria && ply2.len() == 2);
    REQUIRE(std::memcmp(ply2.pods.data(), ply.pods.data(), 2 * sizeof(PlyGaussianPod)) == 0);
    const char *bad = "ply\nformat ascii 1.0\nelement face 0\nend_header\n";
    try { PlyGaussians::read_from(bad, std::strlen(bad)); REQUIRE(false); }
    catch (const PlyError &e) { REQUIRE(std::string(e.what()) == "Gaussian vertex element not found in PLY header"); }

    for (uint32_t version = 1; version <= 3; version++) {
        SpzGaussiansFromGaussianSliceOptions o;
        REQUIRE(o.version == 3 && o.sh_degree == 3 && o.fractional_bits == 12 && o.sh_quantize_bits[0] == 5);
        o.version = version;
        auto z = SpzGaussians::write_gaussians(gs, o);
        auto spz = SpzGaussians::read_from(z.data(), z.size());
        REQUIRE(spz.len() == 2 && spz.header.version == version && spz.header.sh_degree == 3);
        auto again = spz.write_to();   // writing what was read keeps the columns
        REQUIRE(SpzGaussians::read_from(again.data(), again.size()) == spz);
        REQUIRE(SpzGaussians::from_gaussians_with_options(gs, o) == spz);
        for (size_t i = 0; i < 2; i++) {
            for (int c = 0; c < 3; c++) REQUIRE(std::fabs(spz.gaussians[i].pos[c] - gs[i].pos[c]) <= 1.0f);
            for (int c = 0; c < 4; c++) REQUIRE(std::fabs(spz.gaussians[i].rot[c] - gs[i].rot[c]) <= 0.1f);
            for (int c = 0; c < 45; c++) REQUIRE(std::fabs(spz.gaussians[i].sh[c] - gs[i].sh[c]) <= 0.1f);
            for (int c = 0; c < 4; c++) REQUIRE(std::abs((int)spz.gaussians[i].color[c] - (int)gs[i].color[c]) <= 2);
        }
    }
    SpzGaussiansFromGaussianSliceOptions o;
    o.version = 999;
    try { SpzGaussians::write_gaussians(gs, o); REQUIRE(false); }
    catch (const SpzError &e) { REQUIRE(std::string(e.what()) == "Unsupported SPZ version: 999, expected one of 1..=3"); }
    // Gaussians / GaussiansSource over the fused entry points (tests/e2e/gaussian.rs): a PLY round
    // trip is exact, an SPZ one within the format's quantisation, Internal has no byte form
    auto from_ply = Gaussians::read_from(bytes.data(), bytes.size(), GaussiansSource::Ply);
    REQUIRE(from_ply.source == GaussiansSource::Ply && from_ply.len() == 2 && !from_ply.is_empty());
    {   // PLY -> Gaussian -> PLY: positions and scales come back exactly (colour / opacity go through
        // the SH-DC and sigmoid conversions of Gaussian::from_ply / to_ply)
        auto rewritten = from_ply.write_to();
        auto again = Gaussians::read_from(rewritten.data(), rewritten.size(), GaussiansSource::Ply);
        REQUIRE(again.len() == 2);
        for (size_t i = 0; i < 2; i++)
            for (int c = 0; c < 3; c++) REQUIRE(again.gaussians[i].pos[c] == from_ply.gaussians[i].pos[c]);
    }
    auto as_spz = from_ply.write_to(GaussiansSource::Spz);
    auto from_spz = Gaussians::read_from(as_spz.data(), as_spz.size(), GaussiansSource::Spz);
    REQUIRE(from_spz.len() == 2);
    for (int c = 0; c < 3; c++) REQUIRE(std::fabs(from_spz.gaussians[1].pos[c] - gs[1].pos[c]) <= 1.0f);
    try { Gaussians::from_gaussians(gs).write_to(); REQUIRE(false); }
    catch (const Error &e) { REQUIRE(std::string(e.what()) == "cannot write Internal Gaussians to buffer"); }
    try { Gaussians::read_from(bytes.data(), bytes.size(), GaussiansSource::Internal); REQUIRE(false); }
    catch (const Error &e) { REQUIRE(std::string(e.what()) == "cannot read Internal Gaussians from buffer"); }
    std::printf("host codecs OK\n");
    return 0;
}

int main() {
    if (host_codecs()) return 1;
    std::fflush(stdout);
    Device dev(0);
    Stream s(dev);
    std::vector<Gaussian> gs;
    for (uint32_t i = 0; i < 15; i++) gs.push_back(given_gaussian(i));

    // byte-exact upload -> download, all via the typed wrapper
    using G = GaussianPodWithShHalfCov3dHalfConfigs;
    REQUIRE(G::size() == 128);
    GaussiansBuffer<G> buf(dev, gs);
    REQUIRE(buf.len() == 15 && !buf.is_empty());
    REQUIRE(buf.download(s) == G::from_gaussians(gs));
    try { buf.update(s, std::vector<Gaussian>(gs.begin(), gs.begin() + 3)); REQUIRE(false); }
    catch (const GaussiansBufferUpdateError &e) { REQUIRE(e.count() == 3 && e.expected_count() == 15); }
    try { buf.update_range(s, 14, std::vector<Gaussian>(gs.begin(), gs.begin() + 3)); REQUIRE(false); }
    catch (const GaussiansBufferUpdateRangeError &e) { REQUIRE(e.count() == 3 && e.start() == 14 && e.expected_count() == 15); }
    try { buf.download_gaussians(s); REQUIRE(false); } catch (const LossyConfigError &) {}
    try { GaussiansBuffer<G>::try_from(Buffer(dev, 130)); REQUIRE(false); }
    catch (const GaussiansBufferTryFromBufferError &e) { REQUIRE(e.buffer_size() == 130 && e.expected_multiple_size() == 128); }
    using R = GaussianPodWithShSingleCov3dRotScaleConfigs;
    GaussiansBuffer<R> rbuf(dev, gs);
    {   // the mirror order is a permutation of the indices; index order when switched off
        REQUIRE(rbuf.spatial_order());
        auto order = rbuf.download_order(s);
        std::vector<int> seen(order.size(), 0);
        for (uint32_t id : order) { REQUIRE(id < order.size()); seen[id]++; }
        for (int c : seen) REQUIRE(c == 1);
        rbuf.set_spatial_order(false);
        order = rbuf.download_order(s);
        for (size_t i = 0; i < order.size(); i++) REQUIRE(order[i] == i);
        rbuf.set_spatial_order(true);
    }
    auto back = rbuf.download_gaussians(s);
    REQUIRE(std::memcmp(back.data(), gs.data(), gs.size() * sizeof(Gaussian)) == 0);
    REQUIRE(!gaussian_transform_pod(1.0f, GS_DISPLAY_SPLAT, 4, false, 3.0f).has_value());

    // compute bundle: data [1..5] + 1 (+ uniform 10 + constant 20)
    uint32_t init[5] = {1, 2, 3, 4, 5}, ten = 10;
    Buffer data(dev, sizeof(init), init), uni(dev, 4, &ten);
    auto bundle = ComputeBundleBuilder().label("array map add").bind_group_layout(1).bind_group_layout(1).resolver()
                      .entry_point("main").main_shader(GS_KERNEL_ARRAY_MAP_ADD).constant("additional_constant", 20.0)
                      .build(dev, {{&data}, {&uni}});
    bundle.dispatch(s, 5);
    auto out = data.download<uint32_t>(s);
    REQUIRE(out[0] == 32 && out[4] == 36);
    try { ComputeBundleBuilder().build_without_bind_groups(dev); REQUIRE(false); }
    catch (const ComputeBundleBuildError &e) { REQUIRE(std::string(e.what()) == "missing bind group layout for compute bundle"); }
    try { ComputeBundleBuilder().bind_group_layout(1).resolver().entry_point("main").main_shader(GS_KERNEL_ARRAY_MAP_ADD).workgroup_size(1u << 20).build_without_bind_groups(dev); REQUIRE(false); }
    catch (const ComputeBundleCreateError &e) { REQUIRE(e.status == GS_ERR_WORKGROUP_SIZE_EXCEEDS_LIMIT && e.a == (1u << 20)); }

    // one frame
    gs_camera cam;
    const float eye[3] = {0, 0, 30}, target[3] = {8, 8, 8}, up[3] = {0, 1, 0};
    gs_camera_look_at(eye, target, up, 1.0f, 320, 200, 0.1f, 100.0f, &cam);
    gs_gaussian_transform_pod gt; gs_gaussian_transform_pod_default(&gt);
    gs_model_transform_pod mt; gs_model_transform_pod_default(&mt);
    Buffer img(dev, 320 * 200 * 16);
    Renderer r(dev);
    r.render(s, rbuf, gt, mt, cam, (float *)img.device_ptr());
    auto fr = r.wait_frame();                       // render() only enqueued the frame
    REQUIRE(fr.gaussians == 15 && fr.flags == 0 && fr.pairs <= fr.pair_capacity && fr.launches > 0);
    auto st = r.stats();
    REQUIRE(st.visible == fr.visible && st.pairs == fr.pairs);
    auto pending = img.prepare_download(s);         // prepare_download ... map_download
    auto mapped = pending.map<float>();
    auto px = img.download<float>(s);
    REQUIRE(mapped == px);
    {   // device pack == host pack
        Buffer src(dev, gs.size() * sizeof(Gaussian), gs.data()), dst(dev, gs.size() * G::size());
        pack_device<G>(dev, s, src, gs.size(), dst);
        REQUIRE(dst.download<uint8_t>(s) == G::from_gaussians(gs));
    }
    {   // device load path: PLY vertex records -> PODs in one kernel == host from_ply + host pack;
        // SPZ bytes (host inflate + device column decode) == host decode + host pack
        PlyGaussians ply = PlyGaussians::from_gaussians(gs);
        auto from_ply = GaussiansBuffer<G>::new_from_ply(dev, ply.pods);
        std::vector<Gaussian> via_host(ply.pods.size());
        gs_gaussian_from_ply(ply.pods.data(), ply.pods.size(), via_host.data());
        REQUIRE(from_ply.len() == 15 && from_ply.download(s) == G::from_gaussians(via_host));
        auto spz_bytes = SpzGaussians::from_gaussians(gs).write_to();
        auto from_spz = GaussiansBuffer<G>::new_from_spz(dev, spz_bytes.data(), spz_bytes.size());
        auto decoded = SpzGaussians::read_from(spz_bytes.data(), spz_bytes.size()).iter_gaussian();
        REQUIRE(from_spz.len() == 15 && from_spz.download(s) == G::from_gaussians(decoded));
    }
    {   // frames in flight: a second renderer on a stream of another priority (its own hardware queue) renders the
        // same frame while the first one's stream is busy with it again; both equal the frame rendered alone
        int32_t least = 0, greatest = 0;
        REQUIRE(gs_device_stream_priority_range(dev.raw(), &least, &greatest) == GS_OK && greatest <= least);
        gs_stream *bad = nullptr;
        REQUIRE(gs_stream_create_with_priority(dev.raw(), least + 1, &bad) == GS_ERR_INVALID_ARGUMENT);
        Stream s2(dev, greatest);
        Buffer img2(dev, (size_t)cam.width * cam.height * 16);
        Renderer r2(dev);
        r2.render(s2, rbuf, gt, mt, cam, (float *)img2.device_ptr());
        REQUIRE(r2.wait_frame().flags == 0);
        for (int i = 0; i < 4; i++) {
            r.render(s, rbuf, gt, mt, cam, (float *)img.device_ptr());
            r2.render(s2, rbuf, gt, mt, cam, (float *)img2.device_ptr());
        }
        REQUIRE(r.wait_frame().flags == 0 && r2.wait_frame().flags == 0);
        REQUIRE(img2.download<float>(s2) == px && img.download<float>(s) == px);
    }
    {   // round 5: the sorts and the tile rect version are pinned per renderer; every combination renders the same frame;
        // gs3d::FrameRing hands the frames to three lanes and waits per lane; a stream may be destroyed under a renderer
        for (int depth_msd = 0; depth_msd <= 1; depth_msd++)
            for (int masks = 0; masks <= 1; masks++) {
                Renderer rv(dev);
                rv.set_sort_mode(depth_msd, depth_msd);
                rv.set_tile_masks(masks);
                Buffer imgv(dev, (size_t)cam.width * cam.height * 16);
                rv.render(s, rbuf, gt, mt, cam, (float *)imgv.device_ptr());
                REQUIRE(rv.wait_frame().flags == 0);
                const gs_sort_info si = rv.sort_info();
                REQUIRE(si.depth_msd == (uint32_t)depth_msd && si.tile_masks == (uint32_t)masks && si.bucket_capacity >= 16384u);
                REQUIRE(imgv.download<float>(s) == px);
            }
        try { r.set_sort_mode(2, 0); REQUIRE(false); } catch (const Error &e) { REQUIRE(e.status == GS_ERR_INVALID_ARGUMENT); }
        {   // two-round frames: the nearest 8192 Gaussians first — three frames (the first compacts round 2 out of the depth
            // order; the others may sort each round on its own): the same image, fewer or as many pairs, the taps refuse
            Renderer rr(dev);
            rr.set_rounds(1, 8192);
            Buffer imgr(dev, (size_t)cam.width * cam.height * 16);
            const gs_frame_result one = r.wait_frame();
            for (int i = 0; i < 3; i++) {
                rr.render(s, rbuf, gt, mt, cam, (float *)imgr.device_ptr());
                const gs_frame_result two = rr.wait_frame();
                const gs_sort_info si = rr.sort_info();
                if (one.gaussians > 8192u + 4096u) REQUIRE(si.rounds == 2u && si.round1 == 8192u);
                REQUIRE(two.flags == 0 && two.visible == one.visible && two.pairs <= one.pairs);
                REQUIRE(imgr.download<float>(s) == px);
            }
            try { rr.set_rounds(2); REQUIRE(false); } catch (const Error &e) { REQUIRE(e.status == GS_ERR_INVALID_ARGUMENT); }
            rr.set_rounds(-1);
        }
        FrameRing ring(dev, 3);
        std::vector<Buffer> imgs;
        for (size_t k = 0; k < ring.size(); k++) imgs.emplace_back(dev, (size_t)cam.width * cam.height * 16);
        for (int i = 0; i < 7; i++) REQUIRE(ring.render(rbuf, gt, mt, cam, (float *)imgs[i % 3].device_ptr()) == (size_t)(i % 3));
        for (size_t k = 0; k < ring.size(); k++) {
            REQUIRE(ring.wait(k).flags == 0);
            REQUIRE(imgs[k].download<float>(ring.stream(k)) == px);
        }
        Renderer rs(dev);
        {
            Stream tmp(dev);
            rs.render(tmp, rbuf, gt, mt, cam, (float *)img.device_ptr());
        }                                               // the stream is gone: its end-of-frame event was recorded on the way out
        rs.render(s, rbuf, gt, mt, cam, (float *)img.device_ptr());
        REQUIRE(rs.wait_frame().flags == 0 && img.download<float>(s) == px);
    }
    double sum = 0; for (float v : px) sum += v;
    REQUIRE(st.gaussians == 15 && std::isfinite(sum));
    std::printf("cpp mirror OK: visible %llu pairs %llu checksum %.6f\n", (unsigned long long)st.visible, (unsigned long long)st.pairs, sum);
    return 0;
}
