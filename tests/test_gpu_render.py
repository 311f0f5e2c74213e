"""GPU parity tests of the render hot path (rows x1-x5): HIP kernels through the C ABI vs the CPU
oracle on identical inputs.  Bar: projected records, sort keys / indices and tile ranges bit-exact;
RGBA within 1e-4 per channel (BASELINE.json) — and, because the spec fixes every operation
including exp, we additionally report / require bit-equality of the image.

NOTE the oracle for these stages is the build's own definition ("parity unpinned": the reference
crate has no render stages)."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

RGBA_TOL = 1e-4  # BASELINE.json north_star: <= 1e-4 per channel


def _render_gpu(gs, device, stream, pod, pods, gt, mt, cam, band=None, renderer=None, sort_mode=None):
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    img = gs.Buffer(device, size=cam.height * cam.width * 16)
    r = renderer or gs.Renderer(device)
    if sort_mode is not None:
        r.set_sort_mode(*sort_mode)        # (depth, tile): 1 MSD-first, 0 LSD passes, -1 the renderer chooses
    r.render(stream, buf, gt, mt, cam, img.device_ptr(), band=band)
    stream.synchronize()
    rgba = img.download(stream, np.float32).reshape(cam.height, cam.width, 4)
    return r, buf, img, rgba


def _mirror_order(ob, buf, stream, sh, cov, pods, fresh=True):
    """The buffer's mirror order, which the oracle needs for exact-depth ties.  For
    a freshly created buffer it must equal the oracle's own restatement of the spatial order."""
    order = buf.download_order(stream)
    assert np.array_equal(np.sort(order), np.arange(len(order), dtype=np.uint32)), "order is not a permutation"
    if fresh:
        exp = ob.spatial_order(sh, cov, pods) if buf.spatial_order() and len(order) > 1 else np.arange(
            len(order), dtype=np.uint32)
        assert np.array_equal(order, exp), "mirror order differs from the oracle's spatial order"
    return order


def _oracle_frame(ob, sh, cov, pods, ogt, omt, ocam, band=None, order=None):
    proj, tiles = ob.preprocess(sh, cov, pods, ogt, omt, ocam, band=band)
    tiles_x = (ocam.width + 15) // 16
    tiles_y = (ocam.height + 15) // 16
    keys, idx = ob.build_keys(proj, tiles, tiles_x, order=order)
    skeys, sidx = ob.sort_pairs(keys, idx)
    ranges = ob.tile_ranges(skeys, tiles_x * tiles_y)
    rgba = ob.blend(proj, sidx, ranges, ocam, band=band, gt=ogt)
    return proj, tiles, keys, idx, skeys, sidx, ranges, rgba


def _compare_frame(gs, ob, device, stream, sh, cov, gaussians, W, H, gt_kw=None, mt_kw=None,
                   cam_kw=None, band=None, check_image_exact=True, sort_mode=None, info=None):
    gt_kw, mt_kw, cam_kw = gt_kw or {}, mt_kw or {}, cam_kw or {}
    pod = gs.GaussianPod(sh, cov)
    pods = pod.from_gaussian(gaussians)
    assert np.array_equal(pods, ob.pack(sh, cov, gaussians)), "product pack != oracle pack"
    ogt = ob.gaussian_transform(**gt_kw)
    omt = ob.model_transform(**mt_kw)
    ocam = helpers.default_camera(ob, W, H, **cam_kw)
    gt = gs.gaussian_transform_pod(gt_kw.get("size", 1.0), gt_kw.get("mode", 0), gt_kw.get("sh_deg", 3),
                                   gt_kw.get("no_sh0", False), gt_kw.get("max_std_dev", 3.0))
    mt = gs.model_transform_pod(mt_kw.get("pos", (0, 0, 0)), mt_kw.get("rot", (0, 0, 0, 1)),
                                mt_kw.get("scale", (1, 1, 1)))
    cam = helpers.copy_camera(ocam, gs.Camera)
    assert bytes(gt) == bytes(ogt) and bytes(mt) == bytes(omt)

    r, buf, img, rgba = _render_gpu(gs, device, stream, pod, pods, gt, mt, cam, band, sort_mode=sort_mode)
    if info is not None:
        info.append(r.sort_info())
    order = _mirror_order(ob, buf, stream, sh, cov, pods)
    o_proj, o_tiles, o_keys, o_idx, o_skeys, o_sidx, o_ranges, o_rgba = _oracle_frame(
        ob, sh, cov, pods, ogt, omt, ocam, band, order=order)
    n = len(gaussians)
    st = r.stats()
    g_proj, g_tiles = r.download_projected(n)
    # --- preprocess ---
    bad = np.nonzero(g_tiles != o_tiles)[0]
    assert bad.size == 0, "tiles_touched differs at %d Gaussians, first %s: gpu %s oracle %s" % (
        bad.size, bad[:5], g_tiles[bad[:5]], o_tiles[bad[:5]])
    vis = o_tiles > 0
    gb = g_proj[vis].view(np.uint8).reshape(-1, 48)
    obb = o_proj[vis].view(np.uint8).reshape(-1, 48)
    badrec = np.nonzero((gb != obb).any(axis=1))[0]
    if badrec.size:
        i = badrec[0]
        raise AssertionError("projected record differs for %d of %d visible; first:\n gpu    %s\n oracle %s"
                             % (badrec.size, vis.sum(), g_proj[vis][i], o_proj[vis][i]))
    assert st.visible == int(vis.sum())
    assert st.pairs == len(o_keys)
    # --- keys + sort ---
    g_skeys, g_sidx = r.download_sorted()
    assert np.array_equal(g_skeys, o_skeys), "sorted keys differ"
    assert np.array_equal(g_sidx, o_sidx), "sorted indices differ"
    # --- ranges ---
    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
    g_ranges = r.download_ranges(tiles_x * tiles_y)
    assert np.array_equal(g_ranges, o_ranges), "tile ranges differ"
    # --- image ---
    b0, b1 = band if band is not None else (0, tiles_y)
    y0, y1 = b0 * 16, min(b1 * 16, H)
    diff = np.abs(rgba[y0:y1] - o_rgba[y0:y1])
    assert np.isfinite(rgba[y0:y1]).all()
    assert diff.max(initial=0.0) <= RGBA_TOL, "RGBA L-inf %g > %g (at %s)" % (
        diff.max(), RGBA_TOL, np.unravel_index(diff.argmax(), diff.shape))
    if check_image_exact:
        neq = (rgba[y0:y1].view(np.uint32) != o_rgba[y0:y1].view(np.uint32)).sum()
        assert neq == 0, "%d channel values differ in the last bits (max abs diff %g)" % (neq, diff.max())
    for h in (buf,):
        h.destroy()
    img.release()
    r.destroy()
    return st


def test_synthetic_small_sh0(gs, ob, device, stream):
    import synth
    g = synth.scene(30000)
    st = _compare_frame(gs, ob, device, stream, gs.SH_NONE, gs.COV3D_ROT_SCALE, g, 1920, 1080,
                        gt_kw=dict(sh_deg=0))
    assert st.pairs > 30000


@pytest.mark.parametrize("sh", [0, 1, 2, 3])
@pytest.mark.parametrize("cov", [0, 1, 2])
def test_all_twelve_pods_sh3(gs, ob, device, stream, sh, cov):
    import synth
    g = synth.scene(6000, first=1000)
    _compare_frame(gs, ob, device, stream, sh, cov, g, 640, 360, gt_kw=dict(sh_deg=3))


@pytest.mark.parametrize("deg,no_sh0", [(0, False), (1, False), (2, True), (3, True)])
def test_sh_degrees(gs, ob, device, stream, deg, no_sh0):
    import synth
    g = synth.scene(5000, first=77)
    _compare_frame(gs, ob, device, stream, gs.SH_SINGLE, gs.COV3D_ROT_SCALE, g, 800, 600,
                   gt_kw=dict(sh_deg=deg, no_sh0=no_sh0))


def test_model_transform_size_and_std_dev(gs, ob, device, stream):
    import synth
    g = synth.scene(8000, first=5)
    q = np.array([0.239118, 0.369644, -0.099046, 0.892399], dtype=np.float32)
    _compare_frame(gs, ob, device, stream, gs.SH_HALF, gs.COV3D_ROT_SCALE, g, 1000, 700,
                   gt_kw=dict(sh_deg=2, size=1.7, max_std_dev=2.0),
                   mt_kw=dict(pos=(0.5, -0.25, -3.0), rot=tuple(q / np.linalg.norm(q)), scale=(1.5, 0.75, 1.25)),
                   cam_kw=dict(eye=(1.0, 2.0, 6.0), target=(0.0, 0.0, -10.0), vfov_deg=50.0))


def test_odd_image_size_and_background(gs, ob, device, stream):
    """width/height not multiples of 16; non-black background."""
    import synth
    g = synth.scene(4000, first=123456)
    pod = gs.GaussianPod(gs.SH_NORM8, gs.COV3D_HALF)
    pods = pod.from_gaussian(g)
    ogt, omt = ob.gaussian_transform(sh_deg=1), ob.model_transform()
    ocam = helpers.default_camera(ob, 333, 211)
    ocam.background[:] = [0.25, 0.5, 0.75]
    cam = helpers.copy_camera(ocam, gs.Camera)
    r, buf, img, rgba = _render_gpu(gs, device, stream, pod, pods, gs.gaussian_transform_pod(sh_deg=1),
                                    gs.model_transform_pod(), cam)
    _, _, _, _, _, _, _, o_rgba = _oracle_frame(ob, pod.sh, pod.cov, pods, ogt, omt, ocam,
                                                order=_mirror_order(ob, buf, stream, pod.sh, pod.cov, pods))
    assert np.array_equal(rgba.view(np.uint32), o_rgba.view(np.uint32))


def test_more_than_65536_tiles_uses_wide_tile_keys(gs, ob, device, stream):
    """4112 x 4100 px = 257 x 257 = 66049 tiles: tile ids no longer fit u16, the pair arrays switch
    to u32 tile keys (17 key bits -> 3 tile-sort passes).  Whole frame, all stages, bit-exact; then
    a normal-sized frame on the same renderer (switching back to u16 keys)."""
    import synth
    g = synth.scene(6000, first=4242)
    g["scale"] *= 3.0   # cover more tiles
    st = _compare_frame(gs, ob, device, stream, gs.SH_HALF, gs.COV3D_SINGLE, g, 4112, 4100, gt_kw=dict(sh_deg=2))
    assert st.tiles_x * st.tiles_y == 257 * 257 and st.pairs > 0
    pod = gs.GaussianPod(gs.SH_HALF, gs.COV3D_SINGLE)
    pods = pod.from_gaussian(g)
    r = gs.Renderer(device)
    gt, mt = gs.gaussian_transform_pod(sh_deg=2), gs.model_transform_pod()
    frames = []
    for (w, h) in ((4112, 4100), (640, 480), (4112, 4100)):
        cam = helpers.default_camera(gs, w, h)
        _, buf, img, rgba = _render_gpu(gs, device, stream, pod, pods, gt, mt, cam, renderer=r)
        ocam = helpers.copy_camera(cam, ob.Camera)
        o = ob.render(pod.sh, pod.cov, pods, ob.gaussian_transform(sh_deg=2), ob.model_transform(), ocam,
                      order=_mirror_order(ob, buf, stream, pod.sh, pod.cov, pods))[0]
        assert np.array_equal(rgba.view(np.uint32), o.view(np.uint32)), (w, h)
        buf.destroy(); img.release()
    r.destroy()


def _depth_bits(near, far):
    """bits the depth sort covers: every visible depth lies in (near, far), so the keys
    bits(depth) - bits(near) are below bits(far) - bits(near) — a bound the host knows without
    reading anything back from the device"""
    nb = int(np.float32(max(near, 0.0)).view(np.uint32))
    fb = int(np.float32(far).view(np.uint32)) if far > 0 else 0
    return (fb - nb).bit_length() if fb > nb else 0


@pytest.mark.parametrize("msd", [0, 1])
@pytest.mark.parametrize("planes", [(0.001, 100.0), (5.0, 20.0), (0.0, 1000.0), (9.99, 10.02), (0.1, 100.0)])
@pytest.mark.parametrize("case", ["same_depth", "two_depths", "wide_range", "narrow_range"])
def test_depth_key_ranges(gs, ob, device, stream, case, planes, msd):
    """The depth sort's pass count comes from the camera's near / far planes alone (no read-back of
    the scene's depth range inside a frame): 9-bit digits when they save a pass (19-27 bits), 8-bit
    digits otherwise; scenes with all-equal, two-valued, wide and narrow depth distributions must
    come out in exactly the oracle's order under every plan."""
    import synth
    near, far = planes
    g = synth.scene(3000, first=999)
    rng = np.random.default_rng(11)
    n0 = max(near, 1e-3)
    span = min(far, 95.0) - n0
    lo, hi = n0 + 0.02 * span, n0 + 0.98 * span
    mid = np.float32(0.5 * (lo + hi))
    if case == "same_depth":
        g["pos"][:, 2] = -mid
    elif case == "two_depths":
        g["pos"][:, 2] = np.where(rng.random(len(g)) < 0.5, -mid, -np.nextafter(mid, np.float32(np.inf))).astype(np.float32)
    elif case == "wide_range":
        glo = n0 * 1.02 if n0 * 1.02 < hi else lo     # log-uniform from just behind the near plane where that fits
        g["pos"][:, 2] = -np.exp(rng.uniform(np.log(glo), np.log(hi), len(g))).astype(np.float32)
        g["pos"][:, :2] *= (-g["pos"][:, 2:3] / 14.0)
    else:
        g["pos"][:, 2] = (-mid * (1.0 - rng.random(len(g)) * 1e-4)).astype(np.float32)
        g["pos"][:, :2] *= mid / 14.0
    info = []
    st = _compare_frame(gs, ob, device, stream, gs.SH_NONE, gs.COV3D_ROT_SCALE, g, 640, 360,
                        gt_kw=dict(sh_deg=0), cam_kw=dict(near=near, far=far), sort_mode=(msd, 0), info=info)
    assert st.visible > 100
    tile_passes = 2   # 40 x 23 = 920 tiles -> 10 bits -> 5 + 5
    bits = _depth_bits(near, far)
    p8, p9 = -(-bits // 8), -(-bits // 9)
    # MSD-first (round 5: a scatter on the top 10 bits + one workgroup per bucket for the low bits, at most 2 x 9)
    # exists for 11..28 key bits; pinned it is taken, and in either mode the frame reports its largest top-digit bucket
    is_msd = 1 if msd and 10 < bits <= 28 else 0
    assert info[0].depth_msd == is_msd, (bits, info[0].depth_msd)
    want = 1 - (-(bits - 10) // 9) if is_msd else min(p8, p9)
    assert st.sort_passes - tile_passes == want, (st.sort_passes, bits)
    if bits > 10:
        assert 0 < info[0].depth_bucket_max <= st.visible
        if case == "same_depth":
            assert info[0].depth_bucket_max == st.visible       # every key in one bucket


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_block_culling_is_conservative_over_random_views(gs, ob, device, stream, seed, extreme=False):
    """Whole 1024-slot blocks are skipped when their bounds prove them invisible.  Random cameras
    (inside / outside the scene, narrow and wide FOV, near planes cutting through blocks), model
    transforms with rotation and anisotropic scale, `size` up to 3 and splats up to 8x the usual
    size: visible count, pair count and every pixel must equal the oracle's, which culls only per
    Gaussian."""
    import synth
    rng = np.random.default_rng(seed)
    n = 150_000
    g = synth.scene(n, first=seed * 1000)
    g["scale"] *= rng.choice([1.0, 3.0, 8.0], size=(n, 1)).astype(np.float32)
    sh, cov = [(0, 0), (1, 2), (2, 1)][(seed - 1) % 3]
    pod = gs.GaussianPod(sh, cov)
    pods = pod.from_gaussian(g)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    order = _mirror_order(ob, buf, stream, sh, cov, pods)
    W, H = 480, 272
    img = gs.Buffer(device, size=W * H * 16)
    r = gs.Renderer(device)
    culled_some = 0
    for view in range(14):
        eye = rng.normal(0, 1, 3) * rng.choice([0.5, 12.0, 40.0])
        eye[2] -= 14.0 if view % 3 else 0.0
        target = np.array([rng.uniform(-14, 14), rng.uniform(-8, 8), -rng.uniform(2, 26)])
        near = float(rng.choice([0.01, 0.5, 3.0]))
        far = float(rng.choice([15.0, 100.0]))
        fov = float(rng.uniform(20, 100))
        if extreme:   # tools/soak_block_cull.py: cameras inside the scene, extreme planes and lenses
            eye = np.array([rng.uniform(-14, 14), rng.uniform(-8, 8), -rng.uniform(2, 26)])
            near = float(rng.choice([1e-3, 0.05, 8.0]))
            far = float(rng.choice([near * 1.5 + 0.5, 1000.0]))
            fov = float(rng.choice([3.0, 60.0, 150.0]))
        ocam = ob.camera_look_at(tuple(eye), tuple(target), (0, 1, 0), float(np.deg2rad(fov)), W, H, near, far)
        cam = helpers.copy_camera(ocam, gs.Camera)
        size = float(rng.choice([0.5, 1.0, 3.0]))
        q = rng.normal(0, 1, 4)
        q /= np.linalg.norm(q)
        mt_kw = dict(pos=tuple(rng.normal(0, 2, 3)), rot=tuple(q), scale=tuple(rng.uniform(0.3, 2.5, 3))) if view % 2 else {}
        if extreme and view % 2:
            mt_kw["scale"] = tuple(float(x) for x in rng.choice([0.01, 1.0, 40.0], 3))
        gt = gs.gaussian_transform_pod(size, 0, 3, False, float(rng.choice([3.0, 1.5])))
        ogt = ob.GaussianTransform.from_buffer_copy(bytes(gt))      # bit-identical uniforms on both sides
        mt = gs.model_transform_pod(**mt_kw) if mt_kw else gs.model_transform_pod()
        omt = ob.ModelTransform.from_buffer_copy(bytes(mt))
        band = (3, 9) if view % 4 == 3 else None
        r.render(stream, buf, gt, mt, cam, img.device_ptr(), band=band)
        got = img.download(stream, np.float32).reshape(H, W, 4)
        st = r.stats()
        exp, d, vis, _ = ob.render(sh, cov, pods, ogt, omt, ocam, band=band, order=order)
        assert (st.visible, st.pairs) == (vis, d), (view, st.visible, vis, st.pairs, d)
        y0, y1 = (band[0] * 16, min(band[1] * 16, H)) if band else (0, H)
        assert np.array_equal(got[y0:y1].view(np.uint32), exp[y0:y1].view(np.uint32)), view
        culled_some += vis < n // 2
    assert extreme or culled_some >= 3, "the views were meant to cull large parts of the scene"
    buf.destroy(); img.release(); r.destroy()


def test_block_culling_with_cancelling_translations(gs, ob, device, stream):
    """Positions stored far from the origin (+3000) and brought back by the model transform: the
    matrix products then round at the 1e-4 level, which the block test's slack must cover — views
    with the near plane and the screen edges cutting through the scene."""
    import synth
    rng = np.random.default_rng(9)
    n = 120_000
    g = synth.scene(n, first=555)
    g["pos"] += np.float32(3000.0)
    sh, cov = 1, 0
    pod = gs.GaussianPod(sh, cov)
    pods = pod.from_gaussian(g)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    order = _mirror_order(ob, buf, stream, sh, cov, pods)
    W, H = 400, 240
    img = gs.Buffer(device, size=W * H * 16)
    r = gs.Renderer(device)
    mt = gs.model_transform_pod(pos=(-3000.0, -3000.0, -3000.0))
    omt = ob.ModelTransform.from_buffer_copy(bytes(mt))
    gt = gs.gaussian_transform_pod(sh_deg=3)
    ogt = ob.GaussianTransform.from_buffer_copy(bytes(gt))
    for view in range(8):
        eye = (rng.uniform(-10, 10), rng.uniform(-6, 6), -rng.uniform(0, 20))
        target = (rng.uniform(-14, 14), rng.uniform(-8, 8), -rng.uniform(2, 26))
        ocam = ob.camera_look_at(eye, target, (0, 1, 0), float(np.deg2rad(rng.uniform(25, 70))), W, H,
                                 float(rng.choice([0.2, 2.0, 6.0])), 60.0)
        cam = helpers.copy_camera(ocam, gs.Camera)
        r.render(stream, buf, gt, mt, cam, img.device_ptr())
        got = img.download(stream, np.float32).reshape(H, W, 4)
        st = r.stats()
        exp, d, vis, _ = ob.render(sh, cov, pods, ogt, omt, ocam, order=order)
        assert (st.visible, st.pairs) == (vis, d), (view, st.visible, vis)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), view
    buf.destroy(); img.release(); r.destroy()


def test_block_list_mode_follows_the_view(gs, ob, device, stream):
    """The block test runs either inside the preprocess kernel or ahead of it (k_block_cull + block list,
    DESIGN.md §4.2), chosen per frame from what the newest finished frame saw.  One renderer, views that
    see everything / a corner / nothing in an order that takes every transition (test in the kernel ->
    list -> list -> in the kernel -> list after a gap, an empty list): every frame equals the oracle's,
    and the extra launch shows up exactly where the rule says."""
    import synth
    if os.environ.get("GS3D_BLOCK_LIST") or os.environ.get("GS3D_BLOCK_CULL") == "0" or os.environ.get("GS3D_SPATIAL_ORDER") == "0":
        pytest.skip("the rule under test is overridden by a debug switch")
    n = 90_000
    g = synth.scene(n, first=4242)
    sh, cov = 1, 0
    pod = gs.GaussianPod(sh, cov)
    pods = pod.from_gaussian(g)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    order = _mirror_order(ob, buf, stream, sh, cov, pods)
    W, H = 320, 192
    img = gs.Buffer(device, size=W * H * 16)
    r = gs.Renderer(device)
    gt, mt = gs.gaussian_transform_pod(sh_deg=3), gs.model_transform_pod()
    ogt, omt = ob.GaussianTransform.from_buffer_copy(bytes(gt)), ob.ModelTransform.from_buffer_copy(bytes(mt))
    views = {"all": ((0, 0, 12), (0, 0, -14), 70.0), "corner": ((0, 0, 0), (13, 7, -3), 25.0), "none": ((0, 0, 0), (0, 0, 5), 60.0)}
    seq = ["all", "corner", "corner", "all", "all", "none", "none", "all", "all", "corner", "corner"]
    launches, vis_seen = [], []
    for k, name in enumerate(seq):
        eye, target, fov = views[name]
        ocam = ob.camera_look_at(eye, target, (0, 1, 0), float(np.deg2rad(fov)), W, H, 0.1, 100.0)
        cam = helpers.copy_camera(ocam, gs.Camera)
        r.render(stream, buf, gt, mt, cam, img.device_ptr())
        got = img.download(stream, np.float32).reshape(H, W, 4)
        st = r.stats()
        exp, d, vis, _ = ob.render(sh, cov, pods, ogt, omt, ocam, order=order)
        assert (st.visible, st.pairs) == (vis, d), (k, name, st.visible, vis, st.pairs, d)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (k, name)
        # the taps of EVERY frame — list frames keep their per-slot arrays in list space, the
        # translation back to Gaussian indices goes through the block list and the mirror order
        o_proj, o_tiles, _, _, o_skeys, o_sidx, o_ranges, _ = _oracle_frame(ob, sh, cov, pods, ogt, omt, ocam, order=order)
        g_proj, g_tiles = r.download_projected(n)
        assert np.array_equal(g_tiles, o_tiles), (k, name)
        seen = o_tiles > 0
        assert np.array_equal(g_proj[seen].view(np.uint8), o_proj[seen].view(np.uint8)), (k, name)
        g_skeys, g_sidx = r.download_sorted()
        assert np.array_equal(g_skeys, o_skeys) and np.array_equal(g_sidx, o_sidx), (k, name)
        assert np.array_equal(r.download_ranges((W // 16) * (H // 16)), o_ranges), (k, name)
        launches.append(r.wait_frame().launches)
        vis_seen.append(vis)
    assert vis_seen[0] > n // 2 and vis_seen[1] < n // 2 and vis_seen[5] == 0, vis_seen
    # frame k takes the list when frame k - 1 (finished: the test downloads every image) saw < n / 2
    for k in range(1, len(seq)):
        took_list = vis_seen[k - 1] < n // 2
        same_view_other_mode = [j for j in range(1, len(seq)) if seq[j] == seq[k] and (vis_seen[j - 1] < n // 2) != took_list]
        for j in same_view_other_mode:
            assert launches[k] - launches[j] == (1 if took_list else -1), (k, j, launches)
    assert any(vis_seen[k - 1] < n // 2 for k in range(1, len(seq))) and any(vis_seen[k - 1] >= n // 2 for k in range(2, len(seq)))
    buf.destroy(); img.release(); r.destroy()


@pytest.mark.parametrize("n", [1, 1023, 1024, 1025, 70_001, 300_123])
def test_list_frames_with_a_partial_last_block(gs, ob, device, stream, n):
    """A list frame writes its outputs in list space, [0, blocks * 1024): the lanes of the buffer's last,
    partial block past N sit inside that range and must read as culled; more than 256 blocks take more than
    one group of k_block_cull (the look-back across groups); a band keeps only part of the blocks.  Image,
    counts and every tap equal the oracle's, for the full frame forced onto the list and for two bands."""
    import synth
    if os.environ.get("GS3D_BLOCK_CULL") == "0" or os.environ.get("GS3D_BLOCK_LIST") == "0":
        pytest.skip("the list is switched off")
    g = synth.scene(n, first=77)
    sh, cov = 2, 0
    pod = gs.GaussianPod(sh, cov)
    pods = pod.from_gaussian(g)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    order = _mirror_order(ob, buf, stream, sh, cov, pods)
    W, H = 640, 368
    tiles_y = H // 16
    img = gs.Buffer(device, size=W * H * 16)
    gt, mt = gs.gaussian_transform_pod(sh_deg=2), gs.model_transform_pod()
    ogt, omt = ob.GaussianTransform.from_buffer_copy(bytes(gt)), ob.ModelTransform.from_buffer_copy(bytes(mt))
    for band in ((0, tiles_y), (3, 9), (tiles_y - 2, tiles_y)):
        # a band's first frame takes the list; a whole frame takes it once the previous frame saw less than
        # half of the Gaussians: the whole-frame case looks through a narrow field of view
        ocam = helpers.default_camera(ob, W, H, vfov_deg=22.0 if band == (0, tiles_y) else 60.0)
        cam = helpers.copy_camera(ocam, gs.Camera)
        r = gs.Renderer(device)
        for rep in range(2):             # the sizing frame and a steady-state frame
            fr = r.render(stream, buf, gt, mt, cam, img.device_ptr(), band=band)
        if n >= 1024:
            assert fr.launches == r.render(stream, buf, gt, mt, cam, img.device_ptr(), band=band).launches
        o_proj, o_tiles, _, _, o_skeys, o_sidx, o_ranges, o_rgba = _oracle_frame(ob, sh, cov, pods, ogt, omt, ocam, band, order=order)
        st = r.stats()
        assert (st.visible, st.pairs) == (int((o_tiles > 0).sum()), len(o_skeys)), (n, band)
        g_proj, g_tiles = r.download_projected(n)
        assert np.array_equal(g_tiles, o_tiles), (n, band)
        seen = o_tiles > 0
        assert np.array_equal(g_proj[seen].view(np.uint8), o_proj[seen].view(np.uint8)), (n, band)
        g_skeys, g_sidx = r.download_sorted()
        assert np.array_equal(g_skeys, o_skeys) and np.array_equal(g_sidx, o_sidx), (n, band)
        assert np.array_equal(r.download_ranges((W // 16) * tiles_y), o_ranges), (n, band)
        got = img.download(stream, np.float32).reshape(H, W, 4)
        y0, y1 = band[0] * 16, min(band[1] * 16, H)
        assert np.array_equal(got[y0:y1].view(np.uint32), o_rgba[y0:y1].view(np.uint32)), (n, band)
        r.destroy()
    buf.destroy(); img.release()


def test_pathological_gaussians_do_not_derail_the_frame(gs, ob, device, stream):
    """NaN / inf positions, zero, denormal, huge and NaN scales, non-unit and NaN quaternions mixed
    into a normal scene: every one of them must be culled or rendered exactly as the oracle does
    (same order, same counts, same pixels), with the mirror order on and off."""
    import synth
    g = synth.scene(30000, first=31337)
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    bad = [
        dict(pos=(nan, 0, -5)), dict(pos=(0, nan, nan)), dict(pos=(inf, 0, -5)), dict(pos=(0, 0, -inf)),
        dict(pos=(-inf, inf, -5)), dict(scale=(0, 0, 0)), dict(scale=(1e-42, 1e-42, 1e-42)),
        dict(scale=(1e20, 1e20, 1e20)), dict(scale=(nan, 1, 1)), dict(scale=(inf, 0.1, 0.1)),
        dict(rot=(0, 0, 0, 0)), dict(rot=(nan, 0, 0, 1)), dict(rot=(10, -3, 2, 7)), dict(pos=(0, 0, -0.1)),
        dict(pos=(0, 0, -100.0)), dict(pos=(0, 0, 0)), dict(scale=(30, 1e-6, 1e-6)),
    ]
    for k, b in enumerate(bad):
        for f, v in b.items():
            g[f][100 + 97 * k] = v
    for spatial in (True, False):
        for sh, cov in ((0, 0), (1, 2), (3, 1)):
            pod = gs.GaussianPod(sh, cov)
            pods = pod.from_gaussian(g)
            assert np.array_equal(pods, ob.pack(sh, cov, g))
            buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
            buf.set_spatial_order(spatial)
            cam = helpers.default_camera(gs, 512, 288)
            ocam = helpers.copy_camera(cam, ob.Camera)
            img = gs.Buffer(device, size=cam.height * cam.width * 16)
            r = gs.Renderer(device)
            gt, mt = gs.gaussian_transform_pod(sh_deg=3), gs.model_transform_pod()
            r.render(stream, buf, gt, mt, cam, img.device_ptr())
            got = img.download(stream, np.float32).reshape(cam.height, cam.width, 4)
            order = _mirror_order(ob, buf, stream, sh, cov, pods)
            exp, d, vis, _ = ob.render(sh, cov, pods, ob.gaussian_transform(sh_deg=3), ob.model_transform(), ocam,
                                       order=order)
            st = r.stats()
            assert (st.visible, st.pairs) == (vis, d), (spatial, sh, cov)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (spatial, sh, cov)
            assert np.isfinite(got).all()
            buf.destroy(); img.release(); r.destroy()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("std", [3.0, 1.5])
def test_display_modes_ellipse_and_point(gs, ob, device, stream, mode, std):
    """GaussianDisplayMode Ellipse / Point: same keys, sort and ranges as Splat; the
    blend uses the flat k-sigma ellipse / the fixed 1.5-px dot.  All stages bit-exact against the oracle."""
    import synth
    g = synth.scene(15000, first=77 + mode)
    g["scale"] *= 2.5
    st = _compare_frame(gs, ob, device, stream, gs.SH_NORM8, gs.COV3D_ROT_SCALE, g, 512, 300,
                        gt_kw=dict(mode=mode, sh_deg=2, max_std_dev=std))
    assert st.pairs > 0
    # and the three modes really differ
    pod = gs.GaussianPod(gs.SH_NORM8, gs.COV3D_ROT_SCALE)
    pods = pod.from_gaussian(g)
    cam = helpers.default_camera(gs, 512, 300)
    frames = [_render_gpu(gs, device, stream, pod, pods, gs.gaussian_transform_pod(1.0, m, 2, False, std),
                          gs.model_transform_pod(), cam)[3] for m in (0, mode)]
    assert not np.array_equal(frames[0], frames[1])


def test_edge_cases(gs, ob, device, stream):
    """empty buffer, single Gaussian, everything culled, one splat covering the whole screen."""
    import synth
    pod = gs.GaussianPod(gs.SH_SINGLE, gs.COV3D_ROT_SCALE)
    # empty
    g0 = np.zeros(0, dtype=gs.GAUSSIAN_DTYPE)
    st = _compare_frame(gs, ob, device, stream, pod.sh, pod.cov, g0, 320, 200)
    assert st.pairs == 0 and st.visible == 0
    # single
    g1 = synth.scene(1, first=3)
    g1["pos"][0] = (0.1, -0.2, -5.0)
    _compare_frame(gs, ob, device, stream, pod.sh, pod.cov, g1, 320, 200)
    # all behind the camera
    gb = synth.scene(3000)
    gb["pos"][:, 2] *= -1.0
    st = _compare_frame(gs, ob, device, stream, pod.sh, pod.cov, gb, 320, 200)
    assert st.pairs == 0
    # huge splat close to the camera + many small ones (rect = whole screen; deep tile lists)
    gh = synth.scene(2000, first=9)
    gh["pos"][0] = (0.0, 0.0, -0.5)
    gh["scale"][0] = (0.6, 0.5, 0.4)
    gh["color"][0] = (10, 200, 90, 40)
    st = _compare_frame(gs, ob, device, stream, pod.sh, pod.cov, gh, 640, 480)
    assert st.pairs >= 40 * 30


@pytest.mark.parametrize("case", ["one_tile", "screen_fillers", "single_tile_splats", "ragged_counts"])
def test_pair_emission_regimes(gs, ob, device, stream, case):
    """k_pairs_emit cuts the pairs by output slots: a wave owns 2048 consecutive pairs wherever the
    Gaussian boundaries fall.  Regimes: an image of one tile (0 tile-key bits, the sort still has to
    run its generating pass), splats whose rect is the whole 1080p screen (8160 pairs each: one
    Gaussian spans several waves and workgroups), splats of one tile each (2048 Gaussians per wave),
    and visible counts that are not multiples of the 256-Gaussian batch."""
    import synth
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    if case == "one_tile":
        g = synth.scene(500, first=77)
        st = _compare_frame(gs, ob, device, stream, pod.sh, pod.cov, g, 16, 16, gt_kw=dict(sh_deg=0))
        assert st.tiles_x * st.tiles_y == 1 and st.pairs == st.visible > 0
    elif case == "screen_fillers":
        g = synth.scene(300, first=5)
        for k in range(5):
            g["pos"][7 * k] = (0.05 * k, -0.03 * k, -0.4 - 0.1 * k)
            g["scale"][7 * k] = (2.0, 1.5, 1.0)
            g["color"][7 * k, 3] = 30
        st = _compare_frame(gs, ob, device, stream, pod.sh, pod.cov, g, 1920, 1080, gt_kw=dict(sh_deg=0))
        assert st.pairs >= 5 * 120 * 68
    elif case == "single_tile_splats":
        g = synth.scene(30000, first=1234)
        g["scale"] *= 0.02
        st = _compare_frame(gs, ob, device, stream, pod.sh, pod.cov, g, 640, 480, gt_kw=dict(sh_deg=0))
        assert st.pairs < 2 * st.visible
    else:
        for n in (255, 257, 1021, 4099):
            g = synth.scene(n, first=n)
            g["pos"][:, 2] = -np.abs(g["pos"][:, 2]) - 3.0       # all in front: V is close to n
            _compare_frame(gs, ob, device, stream, pod.sh, pod.cov, g, 333, 211, gt_kw=dict(sh_deg=0))


def test_two_renderers_keep_frames_in_flight_on_one_buffer(gs, ob, device, stream):
    """One Gaussian buffer, two renderers on two streams taking frames alternately with nothing but
    the final synchronisation between them: the mirror is built on the first frame's stream and the
    other stream must wait for it; every frame equals the oracle's."""
    import synth
    g = synth.scene(30000, first=99)
    pod = gs.GaussianPod(gs.SH_HALF, gs.COV3D_ROT_SCALE)
    pods = pod.from_gaussian(g)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    cams = [helpers.default_camera(gs, 640, 480), helpers.default_camera(gs, 640, 480, eye=(0.5, 0.2, 1.0), target=(0.4, 0.1, -1.0))]
    gt, mt = gs.gaussian_transform_pod(sh_deg=2), gs.model_transform_pod()
    streams = [device.create_stream(), device.create_stream()]
    rs = [gs.Renderer(device), gs.Renderer(device)]
    imgs = [gs.Buffer(device, size=640 * 480 * 16) for _ in range(2)]
    for i in range(6):                    # first frames are sizing frames of each renderer, then pipelined ones
        k = i & 1
        rs[k].render(streams[k], buf, gt, mt, cams[k], imgs[k].device_ptr(), check=False)
    for s in streams:
        s.synchronize()
    order = _mirror_order(ob, buf, stream, pod.sh, pod.cov, pods)
    for k in range(2):
        rgba = imgs[k].download(streams[k], np.float32).reshape(480, 640, 4)
        ocam = helpers.copy_camera(cams[k], ob.Camera)
        o = ob.render(pod.sh, pod.cov, pods, ob.gaussian_transform(sh_deg=2), ob.model_transform(), ocam, order=order)[0]
        assert np.array_equal(rgba.view(np.uint32), o.view(np.uint32)), k
        rs[k].destroy(); imgs[k].release()
    buf.destroy()


def test_one_renderer_alternating_streams_is_ordered(gs, ob, device, stream):
    """ONE renderer handed frames on two streams in turn with no host synchronisation in between: the
    frames share the renderer's scratch buffers, so gs_render_frame orders each frame behind the
    previous one's end-of-frame event when the stream changes (round-2 advisor finding: a silent data
    race before).  Frames of two cameras and an empty buffer (whose result block is published in
    stream order) must all equal the oracle's; the empty frame in between must report V = D = 0."""
    import synth
    g = synth.scene(60000, first=7)
    pod = gs.GaussianPod(gs.SH_HALF, gs.COV3D_ROT_SCALE)
    pods = pod.from_gaussian(g)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    empty = gs.GaussiansBuffer.new_with_pods(device, pod, pods[:0])
    W, H = 800, 600
    cams = [helpers.default_camera(gs, W, H), helpers.default_camera(gs, W, H, eye=(0.5, 0.2, 1.0), target=(0.4, 0.1, -1.0))]
    gt, mt = gs.gaussian_transform_pod(sh_deg=2), gs.model_transform_pod()
    streams = [device.create_stream(), device.create_stream()]
    r = gs.Renderer(device)
    for k in range(2):       # sizing frames (blocking), so that the loop below only enqueues
        tmp = gs.Buffer(device, size=W * H * 16)
        r.render(streams[k], buf, gt, mt, cams[k], tmp.device_ptr())
        tmp.release()
    imgs = [gs.Buffer(device, size=W * H * 16) for _ in range(8)]
    for i in range(8):
        r.render(streams[i & 1], buf, gt, mt, cams[(i >> 1) & 1], imgs[i].device_ptr(), check=False)
    fr = r.wait_frame()
    assert fr.pairs > 0
    # an empty frame between two full ones, on alternating streams
    e_img = gs.Buffer(device, size=W * H * 16)
    r.render(streams[0], buf, gt, mt, cams[0], imgs[0].device_ptr(), check=False)
    r.render(streams[1], empty, gt, mt, cams[0], e_img.device_ptr(), check=False)
    fe = r.wait_frame()
    assert (fe.visible, fe.pairs, fe.flags) == (0, 0, 0)
    r.render(streams[0], buf, gt, mt, cams[1], imgs[1].device_ptr(), check=False)
    r.wait_frame()
    for s in streams:
        s.synchronize()
    order = _mirror_order(ob, buf, stream, pod.sh, pod.cov, pods)
    want = [ob.render(pod.sh, pod.cov, pods, ob.gaussian_transform(sh_deg=2), ob.model_transform(),
                      helpers.copy_camera(c, ob.Camera), order=order)[0] for c in cams]
    for i in range(8):
        k = (i >> 1) & 1 if i > 1 else i      # imgs[0] / imgs[1] were re-rendered with cams[0] / cams[1] above
        rgba = imgs[i].download(stream, np.float32).reshape(H, W, 4)
        assert np.array_equal(rgba.view(np.uint32), want[k].view(np.uint32)), i
        imgs[i].release()
    bg = e_img.download(stream, np.float32).reshape(H, W, 4)
    assert not bg.any()          # background 0, alpha 0
    e_img.release()
    r.destroy()
    buf.destroy()
    empty.destroy()


def test_image_size_limits_are_rejected(gs, device, stream):
    """more than 2^22 tiles, or more than 65535 tiles along one axis (tile rects are packed as 16-bit
    coordinates): GS_ERR_INVALID_ARGUMENT before anything is launched"""
    import synth
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pod.from_gaussian(synth.scene(10)))
    img = gs.Buffer(device, size=1024)
    r = gs.Renderer(device)
    gt, mt = gs.gaussian_transform_pod(sh_deg=0), gs.model_transform_pod()
    for (w, h) in ((16 * 70000, 16), (16, 16 * 70000), (16 * 4096, 16 * 2048)):
        cam = helpers.default_camera(gs, w, h)
        with pytest.raises(gs.InvalidArgumentError):
            r.render(stream, buf, gt, mt, cam, img.device_ptr())
    buf.destroy(); img.release(); r.destroy()


def test_pair_overflow_is_reported_not_wrapped(gs, ob, device, stream):
    """150k screen-filling splats at 4K need 150k x 32400 > 2^32 pairs: the frame must fail with
    GS_ERR_PAIR_OVERFLOW instead of wrapping the 32-bit pair count (and must stay usable after)."""
    n = 150_000
    g = np.zeros(n, dtype=gs.GAUSSIAN_DTYPE)
    g["rot"] = [0, 0, 0, 1]
    g["pos"] = [0, 0, -5]
    g["scale"] = [50, 50, 50]
    g["color"] = [255, 255, 255, 255]
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pod.from_gaussian(g))
    cam = helpers.default_camera(gs, 3840, 2160)
    img = gs.Buffer(device, size=cam.height * cam.width * 16)
    r = gs.Renderer(device)
    gt, mt = gs.gaussian_transform_pod(sh_deg=0), gs.model_transform_pod()
    with pytest.raises(gs.PairOverflowError):
        r.render(stream, buf, gt, mt, cam, img.device_ptr())
    small = helpers.default_camera(gs, 64, 64)      # 16 tiles x 150k pairs fits
    r.render(stream, buf, gt, mt, small, img.device_ptr())
    stream.synchronize()
    assert r.stats().pairs == 16 * n
    buf.destroy(); img.release(); r.destroy()


def test_deep_tiles_early_termination(gs, ob, device, stream):
    """Many opaque splats stacked in few tiles: exercises multi-batch staging and the per-pixel /
    per-tile early-out (T < 1e-4)."""
    import synth
    g = synth.scene(40000, first=42)
    g["pos"][:, 0] *= 0.02
    g["pos"][:, 1] *= 0.02
    g["color"][:, 3] = 250
    _compare_frame(gs, ob, device, stream, gs.SH_NONE, gs.COV3D_ROT_SCALE, g, 256, 256,
                   gt_kw=dict(sh_deg=0))


def test_model_ply_config(gs, ob, device, stream):
    """BASELINE config #1 input: examples/model.ply (reference data fixture), rendered on both paths."""
    raw = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "model.ply"), dtype=np.uint8)
    n = ob.lib().gso_read_inria_ply(raw.ctypes.data, raw.size, None, 0)
    assert n == 9
    ply = np.zeros(n, dtype=ob.PLY_DTYPE)
    assert ob.lib().gso_read_inria_ply(raw.ctypes.data, raw.size, ply.ctypes.data, n) == 9
    g = np.zeros(n, dtype=ob.GAUSSIAN_DTYPE)
    for i in range(n):
        ob.lib().gso_gaussian_from_ply(ply[i:i + 1].ctypes.data, g[i:i + 1].ctypes.data)
    st = _compare_frame(gs, ob, device, stream, gs.SH_HALF, gs.COV3D_HALF, g, 640, 480,
                        cam_kw=dict(eye=(4.0, 4.0, 22.0), target=(4.0, 4.0, 4.0)))
    assert st.visible == 9


def test_tile_row_bands_stitch_bit_exact(gs, ob, device, stream):
    """Row-band sharding (multi-GPU decomposition): rendering bands separately and
    stitching equals the full frame bit for bit."""
    import synth
    g = synth.scene(20000, first=31)
    pod = gs.GaussianPod(gs.SH_HALF, gs.COV3D_ROT_SCALE)
    pods = pod.from_gaussian(g)
    gt, mt = gs.gaussian_transform_pod(sh_deg=3), gs.model_transform_pod()
    cam = helpers.default_camera(gs, 1280, 720)
    r, buf, img, full = _render_gpu(gs, device, stream, pod, pods, gt, mt, cam)
    tiles_y = (720 + 15) // 16
    stitched = np.zeros_like(full)
    bands = [(0, 11), (11, 23), (23, 40), (40, tiles_y)]
    for b in bands:
        r2, buf2, img2, part = _render_gpu(gs, device, stream, pod, pods, gt, mt, cam, band=b)
        y0, y1 = b[0] * 16, min(b[1] * 16, 720)
        stitched[y0:y1] = part[y0:y1]
        # per-band parity against the oracle too
        ocam = helpers.copy_camera(cam, ob.Camera)
        o = _oracle_frame(ob, pod.sh, pod.cov, pods, ob.gaussian_transform(sh_deg=3),
                          ob.model_transform(), ocam, band=b,
                          order=_mirror_order(ob, buf2, stream, pod.sh, pod.cov, pods))
        assert np.array_equal(part[y0:y1].view(np.uint32), o[-1][y0:y1].view(np.uint32))
        buf2.destroy(); img2.release(); r2.destroy()
    assert np.array_equal(stitched.view(np.uint32), full.view(np.uint32))


def test_update_then_rerender_uses_new_data(gs, ob, device, stream):
    """update_range must invalidate the planar mirror (a stale mirror would re-render old data)."""
    import synth
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    g = synth.scene(5000)
    cam = helpers.default_camera(gs, 512, 512)
    gt, mt = gs.gaussian_transform_pod(sh_deg=0), gs.model_transform_pod()
    buf = gs.GaussiansBuffer.new(device, pod, g)
    img = gs.Buffer(device, size=512 * 512 * 16)
    r = gs.Renderer(device)
    r.render(stream, buf, gt, mt, cam, img.device_ptr())
    a = img.download(stream, np.float32).copy()
    g2 = g.copy()
    g2["color"][1000:3000, :3] = 255 - g2["color"][1000:3000, :3]
    buf.update_range(stream, 1000, g2[1000:3000])
    r.render(stream, buf, gt, mt, cam, img.device_ptr())
    b = img.download(stream, np.float32).copy()
    assert not np.array_equal(a, b)
    ocam = helpers.copy_camera(cam, ob.Camera)
    # update_range keeps the order computed at creation (from the OLD positions, which are unchanged here)
    o = ob.render(pod.sh, pod.cov, ob.pack(pod.sh, pod.cov, g2), ob.gaussian_transform(sh_deg=0),
                  ob.model_transform(), ocam, order=_mirror_order(ob, buf, stream, pod.sh, pod.cov,
                                                                  ob.pack(pod.sh, pod.cov, g)))[0]
    assert np.array_equal(b.view(np.uint32).reshape(-1), o.view(np.uint32).reshape(-1))


@pytest.mark.parametrize("sh,cov", [(0, 0), (1, 2), (2, 1), (3, 0)])
def test_partial_updates_remirror_only_their_range_correctly(gs, ob, device, stream, sh, cov):
    """Several update_range calls with awkward starts / counts (crossing 128-Gaussian repack groups
    and 1024-Gaussian mirror blocks, a single Gaussian, the tail) between frames: every frame must
    equal the oracle's frame of the current scene bit for bit."""
    import synth
    pod = gs.GaussianPod(sh, cov)
    g = synth.scene(4500, first=77)
    cam = helpers.default_camera(gs, 400, 304)
    gt, mt = gs.gaussian_transform_pod(sh_deg=3), gs.model_transform_pod()
    ocam = helpers.copy_camera(cam, ob.Camera)
    buf = gs.GaussiansBuffer.new(device, pod, g)
    img = gs.Buffer(device, size=cam.height * cam.width * 16)
    r = gs.Renderer(device)
    rng = np.random.default_rng(5)
    for start, count in [(0, 0), (1000, 300), (127, 2), (4499, 1), (1023, 1026), (0, 4500), (3000, 1500)]:
        if count:
            g[start:start + count]["pos"][:, :2] += rng.normal(0, 0.3, (count, 2)).astype(np.float32)
            g["color"][start:start + count, :3] = 255 - g["color"][start:start + count, :3]
            buf.update_range(stream, start, g[start:start + count])
        r.render(stream, buf, gt, mt, cam, img.device_ptr())
        got = img.download(stream, np.float32)
        # positions move between updates: a whole-buffer update re-sorts, a partial one keeps the slots
        order = _mirror_order(ob, buf, stream, sh, cov, None, fresh=False)
        if (start, count) == (0, 4500) and buf.spatial_order():
            assert np.array_equal(order, ob.spatial_order(sh, cov, ob.pack(sh, cov, g)))
        exp = ob.render(sh, cov, ob.pack(sh, cov, g), ob.gaussian_transform(sh_deg=3), ob.model_transform(), ocam,
                        order=order)[0]
        assert np.array_equal(got.view(np.uint32).reshape(-1), exp.view(np.uint32).reshape(-1)), (start, count)
    buf.destroy(); img.release(); r.destroy()


# ---- stand-alone primitives -------------------------------------------------------------------

@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 100000, 1 << 20])
def test_exclusive_scan(gs, device, stream, n):
    rng = np.random.default_rng(n)
    v = rng.integers(0, 50, size=n, dtype=np.uint32)
    out, total = gs.exclusive_scan_u32(device, stream, v)
    exp = np.concatenate([[0], np.cumsum(v, dtype=np.uint64)[:-1]]).astype(np.uint32) if n else v
    assert np.array_equal(out, exp)
    assert total == int(v.sum(dtype=np.uint64))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2047, 2048, 2049, 4096, 10007, 300000])
@pytest.mark.parametrize("end_bit", [8, 24, 45, 64])
def test_radix_sort_matches_stable_sort(gs, ob, device, stream, n, end_bit):
    rng = np.random.default_rng(n * 131 + end_bit)
    keys = (rng.integers(0, 1 << 32, size=n, dtype=np.uint64) << np.uint64(32)) | \
        rng.integers(0, 1 << 32, size=n, dtype=np.uint64)
    if end_bit < 64:
        keys &= np.uint64((1 << end_bit) - 1)
    if n > 10:  # force many ties to exercise stability
        keys[rng.integers(0, n, size=n // 3)] = keys[0]
    vals = np.arange(n, dtype=np.uint32)
    k, v = gs.sort_pairs_u64(device, stream, keys, vals, end_bit)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order])
    assert np.array_equal(v, vals[order]), "sort is not stable"
    ok, ov = ob.sort_pairs(keys, vals)
    assert np.array_equal(k, ok) and np.array_equal(v, ov)


def test_radix_sort_skewed_digits(gs, device, stream):
    """All keys share most digits (like tile ids): long digit runs, single-bin passes."""
    n = 50000
    keys = (np.arange(n, dtype=np.uint64) % np.uint64(7)) << np.uint64(32)
    keys |= np.uint64(0x3F800000)
    vals = np.arange(n, dtype=np.uint32)[::-1].copy()
    k, v = gs.sort_pairs_u64(device, stream, keys, vals, 45)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])


def test_pair_capacity_overflow_is_flagged_and_recovered(gs, ob, device, stream):
    """Steady-state frames size the pair buffers from earlier frames (no read-back inside a frame).
    When the camera jumps so that a frame needs far more pairs than that, the device notices before the
    blend and the frame is SKIPPED: the image keeps what it held (never a frame without its farthest
    splats), the result carries FRAME_FLAG_PAIR_OVERFLOW | FRAME_FLAG_SKIPPED and wait_frame raises
    PairCapacityError with the true D; the NEXT pipelined frame already has the larger buffers and is
    exact.  render(check=True) does the wait + re-render itself."""
    import synth
    g = synth.scene(60000, first=5)
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    pods = pod.from_gaussian(g)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    W, H = 960, 540
    img = gs.Buffer(device, size=W * H * 16)
    r = gs.Renderer(device)
    gt, mt = gs.gaussian_transform_pod(sh_deg=0), gs.model_transform_pod()
    gt_big = gs.gaussian_transform_pod(size=4.0, sh_deg=0)      # 4x larger splats: ~10x the pairs, same "shape"
    far_cam = helpers.default_camera(gs, W, H, eye=(0.0, 0.0, 60.0), target=(0.0, 0.0, 0.0))   # tiny splats: few pairs
    near_cam = helpers.default_camera(gs, W, H)                                                 # inside the scene
    order = buf.download_order(stream)
    ogt, omt = ob.gaussian_transform(size=4.0, sh_deg=0), ob.model_transform()
    ref, d, vis, _ = ob.render(pod.sh, pod.cov, pods, ogt, omt, helpers.copy_camera(near_cam, ob.Camera), order=order)

    fr_far = r.render(stream, buf, gt, mt, far_cam, img.device_ptr())        # sizing frame
    assert fr_far.flags == 0
    far_img = img.download(stream, np.float32).copy()
    # same shape (N, image size, band), very different view: pipelined frame with the old capacity
    r.render(stream, buf, gt_big, mt, near_cam, img.device_ptr(), check=False)
    with pytest.raises(gs.PairCapacityError) as e:
        r.wait_frame()
    assert e.value.pairs > e.value.capacity >= fr_far.pairs
    d_true = e.value.pairs
    assert d_true == d
    # the skipped frame wrote NOTHING: the target still holds the previous frame, bit for bit
    assert np.array_equal(img.download(stream, np.float32).view(np.uint32), far_img.view(np.uint32)), \
        "a frame that lost pairs must not reach the image"
    # a viewer's loop: the next pipelined frame (check=False) already has the grown buffers and is exact
    r.render(stream, buf, gt_big, mt, near_cam, img.device_ptr(), check=False)
    fr = r.wait_frame()
    assert fr.flags == 0 and fr.pairs == d_true and fr.pair_capacity >= d_true and fr.visible == vis
    rgba = img.download(stream, np.float32).reshape(H, W, 4)
    assert np.array_equal(rgba.view(np.uint32), ref.view(np.uint32))
    # check=True on a fresh renderer: grows and re-renders by itself
    r2 = gs.Renderer(device)
    r2.render(stream, buf, gt, mt, far_cam, img.device_ptr())
    fr = r2.render(stream, buf, gt_big, mt, near_cam, img.device_ptr())
    assert fr.flags == 0 and fr.pairs == d_true
    assert np.array_equal(img.download(stream, np.float32).reshape(H, W, 4).view(np.uint32), ref.view(np.uint32))
    # and back: the larger buffers stay, nothing is flagged
    fr2 = r.render(stream, buf, gt, mt, far_cam, img.device_ptr(), check=False)
    assert fr2 is None
    assert r.wait_frame().flags == 0
    r.destroy(); r2.destroy(); img.release(); buf.destroy()


def test_steady_zoom_never_reaches_the_skip_path(gs, ob, device, stream):
    """A view whose D grows 12 -> 28 % per frame (accelerating; 4.3x over 9 frames), enqueued two
    frames at a time with nothing but the pair's last wait in between (so a frame is sized from results
    up to two frames old): the trend-aware head room (last step extrapolated three frames ahead, then
    25 %) must keep every frame inside its capacity — 25 % over the last D alone would not (D grows by
    up to 1.64x over two frames).  Every frame has its own target pre-filled with NaN: a skipped frame
    would leave its target untouched.  The last frame is compared with the oracle."""
    import synth
    g = synth.scene(80000, first=11)
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    pods = pod.from_gaussian(g)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    W, H = 960, 540
    nframes = 9
    sentinel = np.full(W * H * 4, np.nan, dtype=np.float32)
    imgs = [gs.Buffer(device, data=sentinel) for _ in range(nframes)]
    r = gs.Renderer(device)
    mt = gs.model_transform_pod()
    sizes = [1.2 ** i for i in range(nframes)]            # oracle: D = 119 k, 134 k, 153 k, ... 516 k
    gts = [gs.gaussian_transform_pod(size=float(sz), sh_deg=0) for sz in sizes]
    cam = helpers.default_camera(gs, W, H)
    fr0 = r.render(stream, buf, gts[0], mt, cam, imgs[0].device_ptr())    # sizing frame
    pairs = [fr0.pairs]
    for i in range(1, nframes):
        r.render(stream, buf, gts[i], mt, cam, imgs[i].device_ptr(), check=False)
        if i % 2 == 0 or i == nframes - 1:
            try:
                pairs.append(r.wait_frame().pairs)
            except gs.PairCapacityError as e:
                raise AssertionError("frame %d was skipped: D %d > capacity %d (history %s)" % (i, e.pairs, e.capacity, pairs))
    assert pairs[-1] > 4 * pairs[0], pairs          # the sequence really grew
    for i in range(nframes):
        a = imgs[i].download(stream, np.float32)
        assert not np.isnan(a).any(), "frame %d left its target untouched: it was skipped" % i
    order = buf.download_order(stream)
    ref = ob.render(pod.sh, pod.cov, pods, ob.gaussian_transform(size=float(sizes[-1]), sh_deg=0), ob.model_transform(),
                    helpers.copy_camera(cam, ob.Camera), order=order)[0]
    rgba = imgs[-1].download(stream, np.float32).reshape(H, W, 4)
    assert np.array_equal(rgba.view(np.uint32), ref.view(np.uint32))
    r.destroy(); buf.destroy()
    for im in imgs:
        im.release()


@pytest.mark.parametrize("case", [(3840, 2160, 400.0, 45.0, 255, (1927.9, 1082.1)), (7680, 4320, 800.0, 30.0, 255, (3840.25, 2160.25)),
                                  (3840, 2160, 250.0, 30.0, 2, (1913.3, 1077.8)), (3840, 2160, 60.0, 45.0, 255, (1920.25, 1080.25)),
                                  (3840, 2160, 120.0, 83.0, 128, (16.02, 2143.98))],
                         ids=lambda c: "%dx%d-s%g-t%g-k%d" % c[:5])
def test_adversarial_needles_match_the_unclipped_frame(gs, ob, device, stream, case):
    """DESIGN.md §3.3, version 3 of the tile rect: splats hundreds of pixels long and thinner than a
    pixel (cond(cov2d) up to 2e6) with their tips on screen — where the round-2 clip lost threshold-level
    pixels (tests/test_rect_versions.py) — render, on the HIP path with the default (clipping) rect,
    bit-identically to the oracle with the same rect AND to the oracle with the unclipped version-1
    rect; short needles are still clipped (fewer pairs than version 1)."""
    W, H, sigma, theta, op, c = case
    g, ocam = helpers.needle_gaussian(ob, theta, sigma, op, c, W, H)
    pod = gs.GaussianPod(0, 0)
    pods = pod.from_gaussian(g)
    cam = helpers.copy_camera(ocam, gs.Camera)
    gt, mt = gs.gaussian_transform_pod(sh_deg=0), gs.model_transform_pod()
    r, buf, img, rgba = _render_gpu(gs, device, stream, pod, pods, gt, mt, cam)
    st = r.stats()
    old = ob.rect_version()
    try:
        same = ob.render(0, 0, pods, ob.gaussian_transform(sh_deg=0), ob.model_transform(), ocam)
        ob.set_rect_version(1)
        v1 = ob.render(0, 0, pods, ob.gaussian_transform(sh_deg=0), ob.model_transform(), ocam)
    finally:
        ob.set_rect_version(old)
    assert st.pairs == same[1]
    assert np.array_equal(rgba.view(np.uint32), same[0].view(np.uint32)), "HIP frame differs from the oracle's (same rect version)"
    assert np.array_equal(rgba.view(np.uint32), v1[0].view(np.uint32)), "the clipped rect changed the image"
    if old != 1:
        assert (st.pairs < v1[1]) == (sigma <= 60.0), (st.pairs, v1[1])      # the guard keeps the square of the long thin ones
    r.destroy(); img.release(); buf.destroy()


def test_two_frames_in_flight_on_priority_streams(gs, ob, device):
    """gs_stream_create_with_priority: two renderers on two streams of different priority (= two hardware queues) take
    frames of two cameras in turn without waiting for each other; every frame equals the one rendered alone."""
    import synth
    least, greatest = device.stream_priority_range()
    assert greatest <= least
    with pytest.raises(gs.GsError):
        device.create_stream(priority=least + 1)
    g = synth.scene(40000, first=99)
    pod = gs.GaussianPod(gs.SH_HALF, gs.COV3D_ROT_SCALE)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pod.from_gaussian(g))
    gt, mt = gs.gaussian_transform_pod(sh_deg=2), gs.model_transform_pod()
    W, H = 1280, 720
    cams = [helpers.copy_camera(helpers.default_camera(ob, W, H, eye=e), gs.Camera) for e in ((0, 0, 0), (0.5, 0.25, 1.0))]
    s0 = device.create_stream()
    alone = []
    r0 = gs.Renderer(device)
    img0 = gs.Buffer(device, size=W * H * 16)
    for cam in cams:
        r0.render(s0, buf, gt, mt, cam, img0.device_ptr())
        alone.append(img0.download(s0, np.float32).view(np.uint32).copy())
    lanes = []
    for k, prio in enumerate((greatest, least)):
        st = device.create_stream(priority=prio)
        r = gs.Renderer(device)
        img = gs.Buffer(device, size=W * H * 16)
        r.render(st, buf, gt, mt, cams[k], img.device_ptr())        # sizing frame
        lanes.append((st, r, img))
    for i in range(12):                                             # only enqueued: the two queues run side by side
        st, r, img = lanes[i & 1]
        r.render(st, buf, gt, mt, cams[i & 1], img.device_ptr(), check=False)
    for k, (st, r, img) in enumerate(lanes):
        assert r.wait_frame().flags == 0
        assert np.array_equal(img.download(st, np.float32).view(np.uint32), alone[k]), "lane %d differs" % k
    for st, r, img in lanes:
        r.destroy()
        img.release()
        st.close()
    # the same through FrameRing: three lanes, frames of the two cameras in turn, one target per lane
    ring = gs.FrameRing(device, 3)
    assert len(ring) == 3 and len(set(ring.priorities)) == (3 if least - greatest >= 2 else len(set(ring.priorities)))
    imgs = [gs.Buffer(device, size=W * H * 16) for _ in range(3)]
    last_cam = [None] * 3
    for i in range(15):
        lane = ring.render(buf, gt, mt, cams[i & 1], imgs[i % 3].device_ptr())
        assert lane == i % 3
        last_cam[lane] = i & 1
    assert all(fr.flags == 0 for fr in ring.wait())
    for k in range(3):
        assert np.array_equal(imgs[k].download(ring.streams[k], np.float32).view(np.uint32), alone[last_cam[k]]), "ring lane %d" % k
    ring.close()
    for b in imgs:
        b.release()
    r0.destroy()
    img0.release()
    s0.close()
    buf.destroy()


def test_renderer_survives_the_stream_of_its_last_frame(gs, ob, device):
    """the end-of-frame event is recorded lazily, on the PREVIOUS stream, when a renderer moves to another
    stream — which failed for good once the caller had destroyed that stream.  gs_stream_destroy now records the event
    on its way out: frames on a new stream, wait_frame and the taps keep working, and the frame is the oracle's."""
    import synth
    g = synth.scene(20000, first=321)
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    pods = pod.from_gaussian(g)
    gt, mt = gs.gaussian_transform_pod(sh_deg=0), gs.model_transform_pod()
    cam = helpers.default_camera(gs, 640, 360)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    img = gs.Buffer(device, size=cam.height * cam.width * 16)
    r = gs.Renderer(device)
    a = device.create_stream()
    r.render(a, buf, gt, mt, cam, img.device_ptr(), check=False)      # only enqueued
    a.close()                                                          # the stream goes away under the renderer
    assert r.wait_frame().flags == 0                                   # ... which waits on the event instead
    assert r.stats().visible > 0
    b = device.create_stream()
    for _ in range(3):
        assert r.render(b, buf, gt, mt, cam, img.device_ptr()).flags == 0
    c = device.create_stream()
    r.render(c, buf, gt, mt, cam, img.device_ptr(), check=False)
    c.close()
    r.render(b, buf, gt, mt, cam, img.device_ptr(), check=False)       # ordered behind the frame of the dead stream
    b.synchronize()
    rgba = img.download(b, np.float32).reshape(cam.height, cam.width, 4)
    o = ob.render(pod.sh, pod.cov, pods, ob.gaussian_transform(sh_deg=0), ob.model_transform(), helpers.copy_camera(cam, ob.Camera),
                  order=_mirror_order(ob, buf, b, pod.sh, pod.cov, pods))[0]
    assert np.array_equal(rgba.view(np.uint32), o.view(np.uint32))
    r.destroy()
    buf.destroy()
    img.release()
    b.close()
