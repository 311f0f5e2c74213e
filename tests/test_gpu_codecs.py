"""GPU tests of the load path end to end through the PRODUCT's own code: file bytes ->
gs_ply_read / gs_spz_decode (csrc/gs_ply.cpp, gs_spz.cpp) -> Gaussian::from_ply / from_spz ->
GaussiansBuffer::new (source records uploaded once, packed on the device by gs_pack_device) ->
one frame, compared with the CPU oracle fed by the ORACLE's own readers of the same reference data
files (examples/model.ply, examples/model.spz; /root/reference/tests/e2e/ply.rs:203-231,
tests/e2e/spz.rs).  Plus: device pack == host pack for all 12 PODs, and the prepare_download /
map_download pair (src/buffer/mod.rs:48-101)."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _oracle_ply(ob, path):
    raw = np.fromfile(path, dtype=np.uint8)
    n = ob.lib().gso_read_inria_ply(raw.ctypes.data, raw.size, None, 0)
    ply = np.zeros(n, dtype=ob.PLY_DTYPE)
    assert ob.lib().gso_read_inria_ply(raw.ctypes.data, raw.size, ply.ctypes.data, n) == n
    g = np.zeros(n, dtype=ob.GAUSSIAN_DTYPE)
    for i in range(n):
        ob.lib().gso_gaussian_from_ply(ply[i:i + 1].ctypes.data, g[i:i + 1].ctypes.data)
    return g


def _frame_pair(gs, ob, device, stream, pod, g_product, g_oracle, W, H, cam_kw, sh_deg=3):
    """product Gaussians through the HIP path, oracle Gaussians through the oracle; same uniforms"""
    ocam = helpers.default_camera(ob, W, H, **cam_kw)
    cam = helpers.copy_camera(ocam, gs.Camera)
    buf = gs.GaussiansBuffer.new(device, pod, g_product)                  # upload + device pack
    opods = ob.pack(pod.sh, pod.cov, g_oracle)
    assert np.array_equal(buf.download(stream), opods), "device-packed PODs differ from the oracle's pack"
    img = gs.Buffer(device, size=W * H * 16)
    r = gs.Renderer(device)
    r.render(stream, buf, gs.gaussian_transform_pod(sh_deg=sh_deg), gs.model_transform_pod(), cam, img.device_ptr())
    rgba = img.download(stream, np.float32).reshape(H, W, 4)
    order = buf.download_order(stream)
    ref, d, vis, _ = ob.render(pod.sh, pod.cov, opods, ob.gaussian_transform(sh_deg=sh_deg), ob.model_transform(), ocam,
                               order=order)
    st = r.stats()
    r.destroy(); img.release(); buf.destroy()
    assert (st.visible, st.pairs) == (vis, d)
    return rgba, ref, st


@pytest.mark.parametrize("pod_idx", [0, 5, 7, 11])
def test_model_ply_file_to_frame(gs, ob, device, stream, pod_idx):
    """BASELINE config #1 input through the product's PLY reader"""
    path = os.path.join(GOLD, "model.ply")
    ply = gs.PlyGaussians.read_from_file(path)
    assert len(ply) == 9
    g = gs.gaussian_from_ply(ply.pods)
    go = _oracle_ply(ob, path)
    for f in ("pos", "color", "sh", "scale", "rot"):
        assert np.array_equal(g[f], go[f]), "product from_ply differs from the oracle's in %s" % f
    pod = gs.ALL_PODS[pod_idx]
    rgba, ref, st = _frame_pair(gs, ob, device, stream, pod, g, go, 640, 480,
                                dict(eye=(4.0, 4.0, 22.0), target=(4.0, 4.0, 4.0)))
    assert st.visible == 9
    assert np.abs(rgba - ref).max() <= 1e-4
    assert np.array_equal(rgba.view(np.uint32), ref.view(np.uint32))
    assert rgba[..., 3].max() > 0.5


def test_model_spz_file_to_frame(gs, ob, device, stream):
    """the reference's examples/model.spz through the product's gunzip + column decode"""
    path = os.path.join(GOLD, "model.spz")
    spz = gs.SpzGaussians.read_from_file(path)
    g = np.ascontiguousarray(spz.iter_gaussian(), dtype=gs.GAUSSIAN_DTYPE)
    import gzip
    go = ob.spz_decode_raw(gzip.decompress(open(path, "rb").read()))
    assert len(g) == len(go) and len(g) > 0
    for f in ("pos", "color", "sh", "scale", "rot"):
        assert np.array_equal(g[f], go[f]), "product from_spz differs from the oracle's in %s" % f
    center = g["pos"].astype(np.float64).mean(axis=0)
    extent = float(np.abs(g["pos"] - center).max()) + 1.0
    cam_kw = dict(eye=(float(center[0]), float(center[1]), float(center[2]) + 3.0 * extent),
                  target=tuple(float(x) for x in center))
    rgba, ref, st = _frame_pair(gs, ob, device, stream, gs.GaussianPod(gs.SH_SINGLE, gs.COV3D_ROT_SCALE), g, go,
                                512, 384, cam_kw)
    assert st.visible >= 1
    assert np.array_equal(rgba.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("sh", [0, 1, 2, 3])
@pytest.mark.parametrize("cov", [0, 1, 2])
def test_device_pack_equals_host_pack(gs, ob, device, stream, sh, cov):
    """gs_pack_device == gs_pack == oracle pack, byte for byte, on random Gaussians that exercise f16
    rounding ties / overflow / subnormals, snorm8 clamping and NaNs (gaussian_config.rs:49-117,219-233)."""
    import synth
    n = 3000          # not a multiple of the pack group: the tail group is partial
    g = synth.scene(n, first=77)
    rng = np.random.default_rng(5)
    g["sh"] = rng.normal(0, 0.6, size=g["sh"].shape).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 65504.0, 65520.0, 1e5, -1e5, 5.96e-8, 2.98e-8, 6.1e-5, 6.0e-5,
                        1.0009765625, 1.00048828125, np.inf, -np.inf, np.nan, 1.0 / 127, 0.999, -1.001, 3.4e38],
                       dtype=np.float32)
    g["sh"].reshape(-1)[:len(special)] = special
    g["scale"][:8] = [[1e-3, 2.0, 300.0]] * 8
    pod = gs.GaussianPod(sh, cov)
    host = pod.from_gaussian(g)
    assert np.array_equal(host, ob.pack(sh, cov, g))
    buf = gs.GaussiansBuffer.new(device, pod, g)
    assert np.array_equal(buf.download(stream), host)
    # update_range through the device pack, and the explicit entry point on caller-owned device memory
    buf.update_range(stream, 100, g[:50])
    exp = host.copy()
    exp[100 * pod.size:150 * pod.size] = host[:50 * pod.size]
    assert np.array_equal(buf.download(stream), exp)
    src = gs.Buffer(device, data=g.view(np.uint8).reshape(-1))
    dst = gs.Buffer(device, size=n * pod.size)
    gs.pack_device(device, stream, pod, src, n, dst)
    assert np.array_equal(dst.download(stream), host)
    src.release(); dst.release(); buf.destroy()


def test_prepare_download_map_download(gs, device, stream):
    """BufferWrapper::prepare_download / map_download: the copy is enqueued, mapped later; the
    mapped bytes are those of the buffer at the time the copy ran on the stream."""
    data = np.arange(1 << 20, dtype=np.uint32)
    buf = gs.Buffer(device, data=data.view(np.uint8))
    d = buf.prepare_download(stream)
    buf.write(stream, 0, np.zeros(16, dtype=np.uint8))       # ordered AFTER the download on the same stream
    got = d.map(np.uint32)
    assert d.ready()
    assert np.array_equal(got, data)
    d.release()
    d2 = buf.prepare_download(stream)
    got2 = d2.map(np.uint32)
    assert not got2[:4].any() and np.array_equal(got2[4:], data[4:])
    d2.release()
    buf.release()


@pytest.mark.parametrize("pod_idx", range(12))
def test_device_from_ply_equals_host_path(gs, ob, device, stream, pod_idx):
    """PLY vertex records -> PODs in ONE device kernel (Gaussian::from_ply fused with G::from_gaussian,
    gs_gaussians_buffer_create_from_ply) must be byte-equal to the host path gs_gaussian_from_ply +
    gs_pack (same gs_convert.h arithmetic on both sides, incl. the specified exp): the reference's
    data file, 300 k synthetic vertices with hostile values (inf / NaN / overflowing and subnormal exp
    results / zero quaternions), and a ranged update."""
    from test_ply import _synthetic_ply
    pod = gs.ALL_PODS[pod_idx]
    for name, ply in (("model.ply", gs.PlyGaussians.read_from_file(os.path.join(GOLD, "model.ply")).pods),
                      ("synthetic", _synthetic_ply(300_000, seed=pod_idx))):
        want = pod.from_gaussian(gs.gaussian_from_ply(ply))
        buf = gs.GaussiansBuffer.new_from_ply(device, pod, ply)
        assert len(buf) == len(ply)
        got = buf.download(stream)
        if not np.array_equal(got, want.reshape(-1)):
            gw, ww = got.view(np.uint32).reshape(len(ply), -1), np.ascontiguousarray(want).view(np.uint32).reshape(len(ply), -1)
            rec, word = np.nonzero(gw != ww)
            msg = ["%s: device from_ply differs from the host path (%s) in %d words of %d records" % (
                name, pod, len(rec), len(np.unique(rec)))]
            for r, w in list(zip(rec, word))[:12]:
                msg.append("record %d word %d: device %08x host %08x | ply scale %s alpha %r rot %s color %s" % (
                    r, w, gw[r, w], ww[r, w], ply["scale"][r], ply["alpha"][r], ply["rot"][r], ply["color"][r]))
            raise AssertionError("\n".join(msg))
        # a ranged update through the same kernel
        if len(ply) > 1000:
            part = _synthetic_ply(777, seed=100 + pod_idx)
            buf.update_range_from_ply(stream, 123, part)
            w2 = want.reshape(len(ply), -1).copy()
            w2[123:123 + 777] = pod.from_gaussian(gs.gaussian_from_ply(part)).reshape(777, -1)
            assert np.array_equal(buf.download(stream), w2.reshape(-1))
            with pytest.raises(gs.GaussiansBufferUpdateRangeError):
                buf.update_range_from_ply(stream, len(ply) - 5, part)
        buf.destroy()
    empty = gs.GaussiansBuffer.new_from_ply(device, pod, np.zeros(0, dtype=gs.PLY_GAUSSIAN_DTYPE))
    assert empty.is_empty()
    empty.destroy()


def test_device_from_ply_frame_and_load_time(gs, ob, device, stream):
    """file -> gs_ply_read -> device from_ply + pack -> frame == the oracle's frame of the oracle's own
    from_ply (libm), on a 1 M-vertex PLY written by the product's writer; also records the load time
    split (host from_ply vs device path) for DESIGN.md §7."""
    import time
    from test_ply import _synthetic_ply
    n = 1_000_000
    ply = _synthetic_ply(n, seed=77)
    ply["pos"][:, 2] = -np.abs(ply["pos"][:, 2]) - 2.0
    ply["scale"] = np.clip(ply["scale"], -7.0, -2.5)
    ply["rot"][25:28] = (1.0, 0.1, 0.2, 0.3)
    data = gs.PlyGaussians(ply).write_to()
    back = gs.PlyGaussians.read_from(data)
    assert np.array_equal(back.pods.view(np.uint32), ply.view(np.uint32))
    pod = gs.GaussianPodWithShHalfCov3dRotScaleConfigs
    t0 = time.perf_counter()
    buf = gs.GaussiansBuffer.new_from_ply(device, pod, back)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    g_host = gs.gaussian_from_ply(back.pods)
    t_host = time.perf_counter() - t0
    go = ob.gaussians_from_ply(ply)
    opods = ob.pack(pod.sh, pod.cov, go)
    assert np.array_equal(buf.download(stream), opods), "device from_ply + pack differs from oracle from_ply + pack"
    assert np.array_equal(pod.from_gaussian(g_host).reshape(-1), opods)
    W, H = 1280, 720
    ocam = helpers.default_camera(ob, W, H)
    cam = helpers.copy_camera(ocam, gs.Camera)
    img = gs.Buffer(device, size=W * H * 16)
    r = gs.Renderer(device)
    r.render(stream, buf, gs.gaussian_transform_pod(sh_deg=3), gs.model_transform_pod(), cam, img.device_ptr())
    rgba = img.download(stream, np.float32).reshape(H, W, 4)
    ref = ob.render(pod.sh, pod.cov, opods, ob.gaussian_transform(sh_deg=3), ob.model_transform(), ocam,
                    order=buf.download_order(stream))[0]
    assert np.array_equal(rgba.view(np.uint32), ref.view(np.uint32))
    print("\nLOAD 1M-vertex PLY: device path (H2D 248 MB + k_from_ply_pods) %.3f s = %.2f GB/s of PLY bytes; "
          "threaded host from_ply alone %.3f s" % (t_dev, n * 248 / t_dev / 1e9, t_host))
    r.destroy(); img.release(); buf.destroy()


@pytest.mark.parametrize("pod_idx", [0, 4, 8, 11])
def test_device_from_spz_equals_host_path(gs, ob, device, stream, pod_idx):
    """SPZ bytes -> PODs with the column decode on the device (host inflate, then Gaussian::from_spz fused
    with G::from_gaussian in k_from_spz_pods) must be byte-equal to the host path gs_spz_decode + gs_pack:
    the reference's data file, and files the product's encoder writes for every version x SH degree x
    fractional-bits combination (f16 / 24-bit positions, first-three / smallest-three quaternions)."""
    import synth
    pod = gs.ALL_PODS[pod_idx]
    cases = [("model.spz", open(os.path.join(GOLD, "model.spz"), "rb").read())]
    g = synth.scene(70_001, first=31)
    g["pos"] *= np.float32(0.05)
    for version in (1, 2, 3):
        for deg in (0, 1, 2, 3):
            for bits in ((12,) if (version, deg) != (2, 3) else (8, 12, 16)):
                opt = gs.spz_options(version=version, sh_degree=deg, fractional_bits=bits)
                cases.append(("v%d-deg%d-fb%d" % (version, deg, bits), gs.SpzGaussians.write_gaussians(g[:70_001 - 7 * deg], opt)))
    for name, data in cases:
        want = pod.from_gaussian(np.ascontiguousarray(gs.SpzGaussians.read_from(data).iter_gaussian(), dtype=gs.GAUSSIAN_DTYPE))
        buf = gs.GaussiansBuffer.new_from_spz(device, pod, data)
        got = buf.download(stream)
        assert len(buf) * pod.size == want.size
        if not np.array_equal(got, want.reshape(-1)):
            gw, ww = got.view(np.uint32).reshape(len(buf), -1), np.ascontiguousarray(want).view(np.uint32).reshape(len(buf), -1)
            rec, word = np.nonzero(gw != ww)
            raise AssertionError("%s (%s): device from_spz differs from the host path in %d words of %d records; first: "
                                 "record %d word %d device %08x host %08x" % (name, pod, len(rec), len(np.unique(rec)), rec[0],
                                                                              word[0], gw[rec[0], word[0]], ww[rec[0], word[0]]))
        buf.destroy()
    # the decompressed entry point and the reference's error for a bad magic number
    import gzip
    raw = gzip.decompress(cases[0][1])
    b2 = gs.GaussiansBuffer.new_from_spz(device, pod, raw, decompressed=True)
    assert len(b2) == b2.spz_header.num_points > 0
    b2.destroy()
    bad = bytearray(raw)
    bad[0] ^= 0xff
    with pytest.raises(gs.SpzError):
        gs.GaussiansBuffer.new_from_spz(device, pod, bytes(bad), decompressed=True)
    with pytest.raises(gs.SpzError):
        gs.GaussiansBuffer.new_from_spz(device, pod, raw[:len(raw) // 2], decompressed=True)
