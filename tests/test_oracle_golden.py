"""CPU tests pinning the oracle (oracle/gs_oracle.c) to (a) the golden vectors of
tests/golden/golden_v1.npz — an independent numpy restatement of the reference's formulas — and
(b) the known-answer values and tolerances of the reference's own tests.  Rows a1-a12."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SH, COV = ["single", "half", "norm8", "none"], ["rot_scale", "single", "half"]
SIZES = {(0, 0): 224, (0, 1): 224, (0, 2): 208, (1, 0): 144, (1, 1): 144, (1, 2): 128,
         (2, 0): 96, (2, 1): 96, (2, 2): 80, (3, 0): 48, (3, 1): 48, (3, 2): 32}


def _gaussians(ob, golden):
    g = np.zeros(len(golden["seeds"]), dtype=ob.GAUSSIAN_DTYPE)
    for f in ("rot", "pos", "color", "sh", "scale"):
        g[f] = golden[f]
    return g


def test_pod_sizes_and_features(ob):
    """src/buffer/gaussian.rs:373-384 sizes; :270-286 feature vector"""
    for (sh, cov), size in SIZES.items():
        assert ob.pod_size(sh, cov) == size
        f = (np.ctypeslib.ctypes.c_int * 7)()
        ob.lib().gso_pod_features(sh, cov, f)
        assert list(f) == [int(i == sh) for i in range(4)] + [int(i == cov) for i in range(3)]


def test_given_fixture_matches_numpy_twin(ob, golden):
    """tests/common/given.rs:48-81"""
    seeds = [int(s) for s in golden["seeds"] if s >= 0]
    g = ob.given_gaussians(seeds)
    for f in ("pos", "color", "sh", "scale"):
        assert np.array_equal(g[f], golden[f][: len(seeds)]), f
    assert np.abs(g["rot"] - golden["rot"][: len(seeds)]).max() <= 6e-8
    i42 = seeds.index(42)
    assert list(g["color"][i42]) == [53, 64, 75, 86]                       # SURVEY §8c pins
    assert np.allclose(g["sh"][i42][:3], [-0.9, -0.8, -0.7], atol=1e-5)
    assert np.allclose(g["sh"][i42][9:12], [0.0, 0.1, 0.2], atol=1e-5)


@pytest.mark.parametrize("sh", range(4))
@pytest.mark.parametrize("cov", range(3))
def test_pack_bytes_match_golden(ob, golden, sh, cov):
    """G::from_gaussian byte layout, all 12 PODs, seeds 0..14, 42, 123 + the unit-test Gaussian"""
    got = ob.pack(sh, cov, _gaussians(ob, golden))
    exp = golden["pod_%s_%s" % (SH[sh], COV[cov])]
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("sh", range(4))
@pytest.mark.parametrize("cov", range(3))
def test_unpack_functions_match_golden_and_reference_tolerances(ob, golden, sh, cov):
    """gaussian.wesl:24-149 vs numpy twin (exact) and vs the un-quantised inputs with the
    tolerances of tests/shader/gaussian.rs:160-367"""
    g = _gaussians(ob, golden)
    pods = ob.pack(sh, cov, g)
    size = ob.pod_size(sh, cov)
    for i in range(len(g)):
        out = ob.shader_test_gaussian(sh, cov, pods[i * size:(i + 1) * size])
        assert np.array_equal(out[:4], golden["unpack_color"][i])
        assert np.array_equal(out[4:49], golden["unpack_sh_" + SH[sh]][i])
        assert np.array_equal(out[49:55], golden["unpack_cov_" + COV[cov]][i])
        if golden["seeds"][i] == 42:
            assert np.abs(out[:4] - g["color"][i] / 255.0).max() < 1e-4
            if sh != 3:
                assert np.abs(out[4:49] - g["sh"][i]).max() < (1e-2, 1e-1, 1e-1)[sh]
            assert np.abs(out[49:55] - golden["cov6_f64"][i]).max() < (1e-2, 1e-2, 1.0)[cov]


def test_unit_test_gaussian_fields(ob, golden):
    """src/buffer/gaussian.rs:386-527 (fixed Gaussian): field equality + lossy inversions"""
    g = _gaussians(ob, golden)[-1:]
    assert list(g["color"][0]) == [255, 128, 64, 32]
    for sh in range(4):
        for cov in range(3):
            pods = ob.pack(sh, cov, g)
            assert np.array_equal(pods[:12].view(np.float32), [1, 2, 3])
            assert list(pods[12:16]) == [255, 128, 64, 32]
            rc, back = ob.unpack_to_gaussian(sh, cov, pods)
            if sh == 3 or cov != 0:
                assert rc == -1          # reference: should_panic
            else:
                assert rc == 0
                assert np.array_equal(back["pos"], g["pos"]) and np.array_equal(back["rot"], g["rot"])
                assert np.array_equal(ob.pack(sh, cov, back), pods)   # pod.sh == from_sh(gaussian.sh)
    # identity rotation, scale (1,2,3): Sigma = diag(1,4,9)
    c = ob.shader_test_gaussian(0, 1, ob.pack(0, 1, g))[49:55]
    assert list(c) == [1.0, 0.0, 0.0, 4.0, 0.0, 9.0]


def test_norm8_truncates_toward_zero(ob):
    """gaussian_config.rs:92-100: `as i8` truncates (0.999*127 = 126.87 -> 126, not 127)"""
    g = np.zeros(1, dtype=ob.GAUSSIAN_DTYPE)
    g["sh"][0, :4] = [0.999, -0.999, 2.0, -2.0]
    b = ob.pack(2, 0, g)[16:20].view(np.int8)
    assert list(b) == [126, -126, 127, -127]


def test_transform_flags(ob, golden):
    """gaussian_transform.rs:63-77,178-194; gaussian_transform.wesl:14-31;
    tests/shader/gaussian_transform.rs:87-150"""
    L = ob.lib()
    for mode, deg, no_sh0, std, u8, flags, dec in golden["flags_table"]:
        gt = ob.gaussian_transform(1.0, int(mode), int(deg), bool(no_sh0), float(std))
        assert gt.flags_u32 == int(flags) and gt.flags[3] == int(u8)
        assert L.gso_transform_display_mode(int(flags)) == int(mode)
        assert L.gso_transform_sh_deg(int(flags)) == int(deg)
        assert L.gso_transform_no_sh0(int(flags)) == int(no_sh0)
        assert L.gso_transform_max_std_dev(int(flags)) == np.float32(dec)
    gt = ob.gaussian_transform(1.0, 1, 2, True, 3.0)
    assert abs(L.gso_transform_max_std_dev(gt.flags_u32) - 3.0) < 1e-6
    enc = np.zeros(1, np.uint8)
    for v, e in ((3.0, 255), (2.0, 170), (1.5, 127), (0.0, 0)):
        assert L.gso_max_std_dev_encode(v, enc.ctypes.data) == 0 and enc[0] == e
    assert L.gso_max_std_dev_encode(3.5, enc.ctypes.data) == -1
    with pytest.raises(ValueError):
        ob.gaussian_transform(sh_deg=4)


def test_model_matrices(ob, golden):
    """model_transform.wesl:13-143 vs float64 (tests/shader/model_transform.rs:100-201, tol 1e-6
    absolute between two f32 results; vs a float64 expectation we allow one f32 ulp on top)"""
    L = ob.lib()
    for case in golden["model_cases"]:
        pos, rot, scale, p = case[0:3], case[3:7], case[7:10], case[10:13]
        mt = ob.model_transform(pos, rot, scale)
        mat, sr, inv, w = (np.zeros(16, np.float32), np.zeros(9, np.float32),
                           np.zeros(9, np.float32), np.zeros(4, np.float32))
        L.gso_model_transform_mat(ob.C.byref(mt), mat.ctypes.data)
        L.gso_model_scale_rot_mat(ob.C.byref(mt), sr.ctypes.data)
        L.gso_model_transform_inv_sr_mat(ob.C.byref(mt), inv.ctypes.data)
        L.gso_model_to_world(ob.C.byref(mt), np.asarray(p, np.float32).ctypes.data, w.ctypes.data)
        for got, exp in ((mat, case[13:29]), (sr, case[29:38]), (inv, case[38:47]), (w, case[47:51])):
            assert np.all(np.abs(got - exp) <= 1e-6 + 1.2e-7 * np.abs(exp))
    # SURVEY §8c known answers for the first case
    c = golden["model_cases"][0]
    assert np.allclose(c[47:51], [15.884003, 9.196152, 23.055576, 1.0], atol=2e-5)


def test_model_ply_fast_path(ob, golden):
    """examples/model.ply through the Inria fast path (ply.rs:292-384) + Gaussian::from_ply"""
    raw = np.fromfile(os.path.join(HERE, "golden", "model.ply"), dtype=np.uint8)
    n = ob.lib().gso_read_inria_ply(raw.ctypes.data, raw.size, None, 0)
    assert n == 9 and int(golden["ply_header_bytes"][0]) == 1526 and raw.size == 1526 + 9 * 248
    ply = np.zeros(n, dtype=ob.PLY_DTYPE)
    assert ob.lib().gso_read_inria_ply(raw.ctypes.data, raw.size, ply.ctypes.data, n) == 9
    assert np.array_equal(ply.view(np.float32).reshape(9, 62), golden["ply_body"])
    g = np.zeros(n, dtype=ob.GAUSSIAN_DTYPE)
    back = np.zeros(n, dtype=ob.PLY_DTYPE)
    for i in range(n):
        ob.lib().gso_gaussian_from_ply(ply[i:i + 1].ctypes.data, g[i:i + 1].ctypes.data)
        ob.lib().gso_gaussian_to_ply(g[i:i + 1].ctypes.data, back[i:i + 1].ctypes.data)
    assert np.abs(g["pos"] - golden["ply_pos"]).max() < 1e-4          # tests/common/assert.rs:4
    assert np.abs(g["rot"] - golden["ply_rot_xyzw"]).max() < 1e-4
    assert np.abs(g["scale"] - golden["ply_scale"]).max() < 1e-4
    assert np.abs(g["color"].astype(int) - np.floor(golden["ply_color_f64"]).astype(int)).max() <= 1
    assert np.array_equal(g["sh"], golden["ply_sh"].astype(np.float32))
    assert list(g["color"][:, 3]) == [255] * 9                        # opacity = +inf
    # f_dc = -0.00695086: (dc*0.2820948 + 0.5)*255 = 126.99999 -> `as u8` truncates to 126
    assert g["color"][7, 0] == 126
    # to_ply(from_ply(x)) round trip of the invertible fields
    assert np.abs(back["pos"] - ply["pos"]).max() < 1e-4 and np.abs(back["scale"] - ply["scale"]).max() < 1e-4
    # truncated body
    assert ob.lib().gso_read_inria_ply(raw.ctypes.data, raw.size - 10, ply.ctypes.data, n) == -3


def test_launch_arithmetic(ob):
    """compute_bundle.rs:131 count.div_ceil(workgroup_size)"""
    L = ob.lib()
    for count, wg, exp in ((5, 1, 5), (5, 1024, 1), (1024, 256, 4), (1025, 256, 5), (0, 64, 0)):
        assert L.gso_dispatch_workgroups(count, wg) == exp


def test_exp_accuracy_and_exactness(ob):
    """gso_exp is within 1e-6 relative of exp on [-5.6, 0] and exact at 0"""
    L = ob.lib()
    xs = np.linspace(-5.6, 0.0, 20001, dtype=np.float32)
    got = np.array([L.gso_exp(float(x)) for x in xs], dtype=np.float64)
    assert np.abs(got / np.exp(xs.astype(np.float64)) - 1.0).max() < 1e-6
    assert L.gso_exp(0.0) == 1.0 and L.gso_exp(-100.0) == 0.0
    assert L.gso_exp(-5.6) < 1.0 / 255.0   # justifies the kernels' early-out at power < -5.6


def test_oracle_sort_and_ranges_against_numpy(ob):
    rng = np.random.default_rng(5)
    n = 20000
    keys = (rng.integers(0, 200, n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 50, n, dtype=np.uint64)
    idx = np.arange(n, dtype=np.uint32)
    k, v = ob.sort_pairs(keys, idx)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(v, idx[order])
    r = ob.tile_ranges(k, 256)
    tiles = (k >> np.uint64(32)).astype(np.int64)
    for t in range(256):
        where = np.nonzero(tiles == t)[0]
        if where.size:
            assert (r[t, 0], r[t, 1]) == (where[0], where[-1] + 1)
        else:
            assert (r[t, 0], r[t, 1]) == (0, 0)


def test_oracle_render_invariants(ob):
    """Whole-frame properties of the definition itself: alpha in [0,1], band rendering = slices of
    the full frame, result independent of the OpenMP thread count."""
    import synth
    g = synth.scene(3000)
    pods = ob.pack(1, 0, g)
    gt, mt = ob.gaussian_transform(sh_deg=3), ob.model_transform()
    cam = ob.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60)), 320, 200)
    full, d, vis, _ = ob.render(1, 0, pods, gt, mt, cam)
    assert 0 < vis <= 3000 and d >= vis
    assert full[..., 3].min() >= 0.0 and full[..., 3].max() <= 1.0 and np.isfinite(full).all()
    ob.lib().gso_set_threads(1)
    one, d1, _, _ = ob.render(1, 0, pods, gt, mt, cam)
    ob.lib().gso_set_threads(ob.lib().gso_get_max_threads())
    assert d1 == d and np.array_equal(one, full)
    stitched = np.zeros_like(full)
    for b in ((0, 5), (5, 9), (9, 13)):
        part = ob.render(1, 0, pods, gt, mt, cam, band=b)[0]
        stitched[b[0] * 16:min(b[1] * 16, 200)] = part[b[0] * 16:min(b[1] * 16, 200)]
    assert np.array_equal(stitched, full)


def test_synth_generator_twin():
    import synth
    a, b = synth.scene(2000, first=12345), synth.scene_numpy(2000, first=12345)
    for f in a.dtype.names:
        assert np.abs(a[f].astype(np.float64) - b[f].astype(np.float64)).max() <= 1e-6, f
    assert np.array_equal(synth.scene(10, first=5), synth.scene(100)[5:15])   # counter based
