"""The render stages x1-x5 against an INDEPENDENT float64 witness and closed-form cases.

tests/golden/witness_v2.npz is produced by tests/golden/make_witness.py: a numpy/float64
restatement of DESIGN.md §3 written from the text and from Kerbl et al. 2023, not from
oracle/gs_oracle.c (different loop structure, matrix algebra instead of expanded terms, np.exp,
np.linalg.inv / eigvalsh).  These tests narrow the common-mode risk that the oracle and the HIP
kernels share a convention error; the rows stay *parity unpinned* against the reference, which has
no implementation of them.

Tolerances: projected records relative 1e-5 (absolute floors where a quantity passes through 0;
the conic is compared with its conditioning, see below), image <= 1e-4 per channel (the north
star's bar).  Decisions that are discontinuous in the inputs (ceil of the radius, floor of the tile
rect, depth culls) are compared exactly only where the witness says the float32 result cannot
flip (margins stored in the fixture).
"""
import os

import numpy as np
import pytest

import helpers

HERE = os.path.dirname(os.path.abspath(__file__))
WIT = np.load(os.path.join(HERE, "golden", "witness_v2.npz"))
VIEWS = [str(v) for v in WIT["views"]]
IMAGE_TOL = 1e-4


def _gaussians(dtype):
    g = np.zeros(len(WIT["pos"]), dtype=dtype)
    for f in ("rot", "pos", "color", "sh", "scale"):
        g[f] = WIT[f]
    return g


def _uniforms(mod, name):
    """mod = oracle binding or product package: both expose the same helper names"""
    w = lambda k: WIT["%s/%s" % (name, k)]
    W, H = int(w("width")), int(w("height"))
    cam = mod.camera_look_at(tuple(w("eye")), tuple(w("target")), tuple(w("up")), 1.0, W, H,
                             float(w("near")), float(w("far")))
    # intrinsics and background are explicit inputs of the witness
    cam.fx, cam.fy, cam.cx, cam.cy = float(w("fx")), float(w("fy")), float(w("cx")), float(w("cy"))
    cam.background[:] = [float(x) for x in w("background")]
    # the view matrix both sides use must be the look-at of these vectors (f32 rounding only)
    assert np.abs(np.array(list(cam.view)).reshape(4, 4).T - w("view")).max() < 1e-6
    return cam, W, H


def _oracle_inputs(ob, name):
    w = lambda k: WIT["%s/%s" % (name, k)]
    cam, W, H = _uniforms(ob, name)
    gt = ob.gaussian_transform(size=float(w("size")), mode=0, sh_deg=int(w("sh_deg")), no_sh0=bool(w("no_sh0")),
                               max_std_dev=float(w("max_std_dev")))
    assert gt.flags[3] == int(w("max_std_dev_u8"))
    mt = ob.model_transform(tuple(w("model_pos")), tuple(w("model_rot")), tuple(w("model_scale")))
    return cam, gt, mt, W, H


def _compare_projected(name, proj, tiles):
    """proj: structured array with the 48-byte gs_projected fields; tiles: tiles touched"""
    w = lambda k: WIT["%s/%s" % (name, k)]
    vis_w = w("visible")
    vis = tiles > 0
    # visibility may only differ where a cull decision is within float32 noise of flipping
    safe = (w("m_depth") > 1e-4) & (w("m_radius") > 1e-3) & (w("m_rect") > 2e-2)
    assert np.array_equal(vis[safe], vis_w[safe]), "visibility differs from the witness"
    both = vis & vis_w
    assert both.sum() > 1000
    p = proj[both]
    # mean: f32 pixel coordinates up to ~200 px -> 1e-5 relative of the image extent
    assert np.abs(p["mx"] - w("mx")[both]).max() <= 2e-3
    assert np.abs(p["my"] - w("my")[both]).max() <= 2e-3
    assert np.abs(p["depth"] / w("depth")[both] - 1).max() <= 1e-5
    assert np.abs(p["opacity"] - w("opacity")[both]).max() <= 1e-7
    rgb = np.stack([p["r"], p["g"], p["b"]], 1)
    assert np.abs(rgb - w("rgb")[both]).max() <= 1e-5 * max(1.0, np.abs(w("rgb")[both]).max())
    # the records store the conic pre-scaled: (ca, cb, cc) = (-A/2, -B, -C/2).
    # inverse of a 2x2 in float32: relative error ~ eps * (a c / det), so the tolerance carries the
    # witness's conditioning a*c/det = A*C*det.
    A, B, C = w("conic")[both].T
    kappa = A * C * w("det")[both]
    tol = 1e-5 * np.maximum(1.0, kappa) * np.maximum(np.maximum(np.abs(A), np.abs(C)), np.abs(B))
    assert (np.abs(-2.0 * p["ca"] - A) <= tol).all()
    assert (np.abs(-p["cb"] - B) <= tol).all()
    assert (np.abs(-2.0 * p["cc"] - C) <= tol).all()
    # tile rect and pair count: exact where no floor/ceil argument is within noise of an integer
    exact = both & safe
    rect = np.stack([proj["tx0"], proj["ty0"], proj["tx1"], proj["ty1"]], 1).astype(np.int64)
    assert np.array_equal(rect[exact], w("rect")[exact])
    return int(exact.sum()), int(both.sum())


@pytest.mark.parametrize("name", VIEWS)
def test_oracle_projection_matches_witness(ob, name):
    cam, gt, mt, W, H = _oracle_inputs(ob, name)
    pods = ob.pack(ob.SH_SINGLE, ob.COV_ROT_SCALE, _gaussians(ob.GAUSSIAN_DTYPE))
    proj, tiles = ob.preprocess(ob.SH_SINGLE, ob.COV_ROT_SCALE, pods, gt, mt, cam)
    exact, both = _compare_projected(name, proj, tiles)
    assert exact >= 0.97 * both          # the margins exclude only a handful of borderline splats


@pytest.mark.parametrize("name", VIEWS)
def test_oracle_image_matches_witness(ob, name):
    cam, gt, mt, W, H = _oracle_inputs(ob, name)
    pods = ob.pack(ob.SH_SINGLE, ob.COV_ROT_SCALE, _gaussians(ob.GAUSSIAN_DTYPE))
    rgba, d, vis, _ = ob.render(ob.SH_SINGLE, ob.COV_ROT_SCALE, pods, gt, mt, cam)
    ref = WIT["%s/image" % name]
    err = np.abs(rgba.astype(np.float64) - ref)
    assert err.max() <= IMAGE_TOL, "oracle frame differs from the float64 witness by %g" % err.max()
    # the witness counts the pairs of the CLIPPED rect (version 3); version 4 then drops unreachable corner tiles of small
    # rects, which the witness — a renderer in float64, not a pair counter — does not restate: same image, fewer pairs
    old = ob.rect_version()
    try:
        ob.set_rect_version(3 if old >= 3 else old)
        rgba3, d3, _, _ = ob.render(ob.SH_SINGLE, ob.COV_ROT_SCALE, pods, gt, mt, cam)
    finally:
        ob.set_rect_version(old)
    assert abs(d3 - int(WIT["%s/pairs" % name])) <= 0.01 * d3   # pair count up to borderline rects
    assert d <= d3 and np.array_equal(rgba3.view(np.uint32), rgba.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", VIEWS)
def test_hip_frame_matches_witness(gs, device, stream, name):
    """the HIP path straight against the witness (no oracle in between)"""
    w = lambda k: WIT["%s/%s" % (name, k)]
    cam, W, H = _uniforms(gs, name)
    gt = gs.gaussian_transform_pod(size=float(w("size")), display_mode=0, sh_deg=int(w("sh_deg")), no_sh0=bool(w("no_sh0")),
                                   max_std_dev=float(w("max_std_dev")))
    mt = gs.model_transform_pod(tuple(w("model_pos")), tuple(w("model_rot")), tuple(w("model_scale")))
    pod = gs.GaussianPodWithShSingleCov3dRotScaleConfigs
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pod.from_gaussian(_gaussians(gs.GAUSSIAN_DTYPE)))
    img = gs.Buffer(device, size=W * H * 16)
    r = gs.Renderer(device)
    r.render(stream, buf, gt, mt, cam, img.device_ptr())
    rgba = img.download(stream, np.float32).reshape(H, W, 4)
    err = np.abs(rgba.astype(np.float64) - w("image"))
    assert err.max() <= IMAGE_TOL, "HIP frame differs from the float64 witness by %g" % err.max()
    proj, tiles = r.download_projected(len(WIT["pos"]))
    _compare_projected(name, proj, tiles)
    r.destroy()
    img.release()
    buf.destroy()


# ------------------------------------------------------------------------------------------------
# closed-form cases (no witness code involved: the expected values are written out by hand)
# ------------------------------------------------------------------------------------------------

def _one(ob, pos, scale=(0.05, 0.05, 0.05), rot=(0, 0, 0, 1), color=(128, 128, 128, 255), sh=None):
    g = np.zeros(1, dtype=ob.GAUSSIAN_DTYPE)
    g["pos"], g["scale"], g["rot"], g["color"] = pos, scale, rot, color
    if sh is not None:
        g["sh"] = sh
    return ob.pack(ob.SH_SINGLE, ob.COV_ROT_SCALE, g)


def test_closed_form_axis_aligned_gaussian_at_image_centre(ob):
    """Camera at the origin looking down -Z, one axis-aligned Gaussian on the optical axis at
    distance z: the Jacobian is diag(f/z, f/z) there, so Sigma' = diag((f sx / z)^2, (f sy / z)^2)
    + 0.3 I, the conic is its inverse, and radius = ceil(3 sqrt(lambda_max))."""
    W, H = 640, 480
    cam = helpers.default_camera(ob, W, H, vfov_deg=60.0)
    f = 0.5 * H / np.tan(np.deg2rad(30.0))
    z, sx, sy, sz = 5.0, 0.2, 0.05, 0.4
    pods = _one(ob, (0, 0, -z), (sx, sy, sz))
    proj, tiles = ob.preprocess(ob.SH_SINGLE, ob.COV_ROT_SCALE, pods, ob.gaussian_transform(sh_deg=0),
                                ob.model_transform(), cam)
    a, c = (f * sx / z) ** 2 + 0.3, (f * sy / z) ** 2 + 0.3
    p = proj[0]
    assert abs(p["mx"] - W / 2) < 1e-4 and abs(p["my"] - H / 2) < 1e-4
    assert abs(p["depth"] - z) < 1e-6
    assert abs(-2 * p["ca"] - 1 / a) <= 1e-6 / a and abs(-2 * p["cc"] - 1 / c) <= 1e-6 / c
    assert abs(p["cb"]) <= 1e-9
    # rect: the radius square, clipped to the box the splat can colour: with the axes
    # aligned the region {alpha >= 1/255} (opacity byte 255) extends sqrt(2 (ln 255 + 0.1) Sigma'_xx) along
    # x and sqrt(2 (ln 255 + 0.1) Sigma'_yy) along y; tile t holds the pixel centres 16 t + 0.5 ... 16 t + 15.5
    r = np.ceil(3.0 * np.sqrt(max(a, c)))
    ex, ey = np.sqrt(2 * (np.log(255.0) + 0.1) * a), np.sqrt(2 * (np.log(255.0) + 0.1) * c)
    assert ey < r < ex + 1      # this Gaussian is 4:1: the box is clearly narrower than the square along y
    assert p["tx0"] == max(int((W / 2 - r) // 16), int(np.floor((W / 2 - ex - 15.5) / 16)) + 1)
    assert p["tx1"] == min(int((W / 2 + r) // 16) + 1, int(np.floor((W / 2 + ex - 0.5) / 16)) + 1)
    assert p["ty0"] == max(int((H / 2 - r) // 16), int(np.floor((H / 2 - ey - 15.5) / 16)) + 1)
    assert p["ty1"] == min(int((H / 2 + r) // 16) + 1, int(np.floor((H / 2 + ey - 0.5) / 16)) + 1)
    assert (p["ty1"] - p["ty0"]) < (int((H / 2 + r) // 16) + 1 - int((H / 2 - r) // 16))   # and it does drop tile rows
    # size scales the covariance by size^2 before the 0.3 dilation
    proj2, _ = ob.preprocess(ob.SH_SINGLE, ob.COV_ROT_SCALE, pods, ob.gaussian_transform(size=2.0, sh_deg=0),
                             ob.model_transform(), cam)
    assert abs(-2 * proj2[0]["ca"] - 1 / (4 * (a - 0.3) + 0.3)) <= 1e-6
    # the blended centre pixel: alpha = min(0.99, opacity * exp(-0.5 d^T conic d)) with d = (0.5, 0.5) px
    rgba = ob.render(ob.SH_SINGLE, ob.COV_ROT_SCALE, pods, ob.gaussian_transform(sh_deg=0), ob.model_transform(), cam)[0]
    alpha = min(0.99, 1.0 * np.exp(-0.5 * (0.25 / a + 0.25 / c)))
    assert abs(rgba[H // 2, W // 2, 3] - alpha) <= 1e-6
    assert np.abs(rgba[H // 2, W // 2, :3] - alpha * 128 / 255).max() <= 1e-6


def test_closed_form_y_flip_and_handedness(ob):
    """+Y in the world is up: a Gaussian above the optical axis lands in the upper image half
    (smaller row index); +X is right; looking down -Z, depth = -z_view > 0."""
    W, H = 640, 480
    cam = helpers.default_camera(ob, W, H)
    f = 0.5 * H / np.tan(np.deg2rad(30.0))
    for pos, exp in (((0, 1, -4), (W / 2, H / 2 - f / 4)), ((1, 0, -4), (W / 2 + f / 4, H / 2)),
                     ((-0.5, -0.5, -2), (W / 2 - f / 4, H / 2 + f / 4))):
        proj, _ = ob.preprocess(ob.SH_SINGLE, ob.COV_ROT_SCALE, _one(ob, pos), ob.gaussian_transform(sh_deg=0),
                                ob.model_transform(), cam)
        assert abs(proj[0]["mx"] - exp[0]) < 1e-3 and abs(proj[0]["my"] - exp[1]) < 1e-3
        assert abs(proj[0]["depth"] + pos[2]) < 1e-6
    # behind the camera / beyond far: culled
    for pos in ((0, 0, 1), (0, 0, -0.05), (0, 0, -150)):
        _, tiles = ob.preprocess(ob.SH_SINGLE, ob.COV_ROT_SCALE, _one(ob, pos), ob.gaussian_transform(sh_deg=0),
                                 ob.model_transform(), cam)
        assert tiles[0] == 0


def test_closed_form_degree1_sh_sign_per_axis_and_no_sh0(ob):
    """Degree-1 real SH in the 3DGS convention: colour = base - C1 y sh[0] + C1 z sh[1] - C1 x sh[2]
    with d the unit vector from the camera to the Gaussian; rest coefficient k = 0 is the first
    degree-1 coefficient (CHANGELOG.md:34,41).  One axis at a time, camera placed so that d is a
    coordinate axis."""
    C1 = 0.4886025119029199
    base = np.array([128, 64, 32]) / 255.0
    sh = np.zeros(45, dtype=np.float32)
    sh[0:3] = (0.10, 0.20, 0.30)      # k = 0
    sh[3:6] = (-0.05, 0.15, 0.25)     # k = 1
    sh[6:9] = (0.30, -0.10, 0.05)     # k = 2
    pods = _one(ob, (0, 0, 0), color=(128, 64, 32, 255), sh=sh)
    cases = {  # camera position -> direction camera->Gaussian
        (0, 0, 5): (0, 0, -1), (0, 0, -5): (0, 0, 1), (5, 0, 0): (-1, 0, 0), (-5, 0, 0): (1, 0, 0),
        (0, 5, 0): (0, -1, 0), (0, -5, 0): (0, 1, 0)}
    for eye, d in cases.items():
        up = (0, 1, 0) if d[1] == 0 else (0, 0, 1)
        cam = ob.camera_look_at(eye, (0, 0, 0), up, float(np.deg2rad(60.0)), 320, 240, 0.1, 100.0)
        x, y, z = d
        exp = base - C1 * y * sh[0:3] + C1 * z * sh[3:6] - C1 * x * sh[6:9]
        proj, tiles = ob.preprocess(ob.SH_SINGLE, ob.COV_ROT_SCALE, pods, ob.gaussian_transform(sh_deg=1),
                                    ob.model_transform(), cam)
        assert tiles[0] > 0
        got = np.array([proj[0]["r"], proj[0]["g"], proj[0]["b"]])
        assert np.abs(got - np.maximum(exp, 0)).max() <= 2e-6, (eye, got, exp)
        # no_sh0 drops the baked DC colour and keeps the rest
        proj0, _ = ob.preprocess(ob.SH_SINGLE, ob.COV_ROT_SCALE, pods, ob.gaussian_transform(sh_deg=1, no_sh0=True),
                                 ob.model_transform(), cam)
        got0 = np.array([proj0[0]["r"], proj0[0]["g"], proj0[0]["b"]])
        assert np.abs(got0 - np.maximum(exp - base, 0)).max() <= 2e-6
        # sh_deg = 0: the baked colour alone
        projd, _ = ob.preprocess(ob.SH_SINGLE, ob.COV_ROT_SCALE, pods, ob.gaussian_transform(sh_deg=0),
                                 ob.model_transform(), cam)
        assert np.abs(np.array([projd[0]["r"], projd[0]["g"], projd[0]["b"]]) - base).max() <= 1e-7


def test_closed_form_view_direction_goes_through_inverse_model_rotation(ob):
    """The SH lobe is attached to the MODEL: rotating the model by R and moving the camera with it
    (eye' = R eye) leaves the colour unchanged; a non-uniform model scale S changes the direction to
    normalize(S^-1 R^T d_world) (model_transform.wesl:85-101)."""
    C1 = 0.4886025119029199
    sh = np.zeros(45, dtype=np.float32)
    sh[0:3], sh[3:6], sh[6:9] = 0.2, -0.1, 0.3
    pods = _one(ob, (0, 0, 0), color=(100, 100, 100, 255), sh=sh)
    gt = ob.gaussian_transform(sh_deg=1)
    # Rz(90 deg): model x -> world y
    q = (0.0, 0.0, np.sin(np.pi / 4), np.cos(np.pi / 4))
    scale = (2.0, 1.0, 0.5)
    mt = ob.model_transform((0, 0, 0), q, scale)
    eye = np.array([3.0, 4.0, 12.0])
    cam = ob.camera_look_at(tuple(eye), (0, 0, 0), (0, 1, 0), float(np.deg2rad(60.0)), 320, 240, 0.1, 100.0)
    dw = -eye / np.linalg.norm(eye)
    R = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    dm = (R.T @ dw) / np.array(scale)
    dm /= np.linalg.norm(dm)
    exp = 100 / 255.0 - C1 * dm[1] * 0.2 + C1 * dm[2] * (-0.1) - C1 * dm[0] * 0.3
    proj, tiles = ob.preprocess(ob.SH_SINGLE, ob.COV_ROT_SCALE, pods, gt, mt, cam)
    assert tiles[0] > 0
    assert abs(proj[0]["r"] - exp) <= 2e-6
