"""The frame's depth sort, MSD-first (round 5): one compacting scatter on the top 9 bits of the depth
key, then one workgroup per bucket finishes the low bits on its CU (gs::k_bucket_sort).  Against the oracle and
against the LSD passes, on the register path (buckets up to `bucket_capacity`), on the chunked in-kernel fallback
(larger buckets: depths that collapse into a few top digits) and through the renderer's own choice between the two
sorts, which follows the bucket sizes the frames report."""
import numpy as np
import pytest

import helpers
from test_gpu_render import _compare_frame, _mirror_order, _oracle_frame

pytestmark = pytest.mark.gpu


def _scene(n, depth, first=31337):
    """depth: 'spread' (the generator's 2..26), 'wall' (all within 1e-4 relative of z = 9: one or two top digits),
    'plane' (bit-identical depths)"""
    import synth
    g = synth.scene(n, first=first)
    rng = np.random.default_rng(5)
    if depth == "wall":
        g["pos"][:, 2] = (-9.0 * (1.0 - rng.random(n) * 1e-4)).astype(np.float32)
        g["pos"][:, :2] *= 9.0 / 14.0
    elif depth == "plane":
        g["pos"][:, 2] = np.float32(-9.0)
        g["pos"][:, :2] *= 9.0 / 14.0
    return g


@pytest.mark.parametrize("depth", ["spread", "wall", "plane"])
@pytest.mark.parametrize("msd", [0, 1])
def test_depth_sort_modes_match_the_oracle(gs, ob, device, stream, depth, msd):
    """60 000 Gaussians, every stage against the oracle, both sorts pinned.  'wall' and 'plane' put > 30 720 keys into
    one bucket: the MSD-first sort must take its chunked path and still be exact."""
    g = _scene(60000, depth)
    g["scale"] *= 0.5          # keeps the pair count of the walls moderate
    info = []
    st = _compare_frame(gs, ob, device, stream, gs.SH_NONE, gs.COV3D_ROT_SCALE, g, 1280, 720, gt_kw=dict(sh_deg=0),
                        sort_mode=(msd, 0), info=info)
    si = info[0]
    assert si.depth_msd == msd and si.bucket_capacity >= 16384
    if depth == "spread":
        assert 0 < si.depth_bucket_max <= si.bucket_capacity
    else:
        assert si.depth_bucket_max > si.bucket_capacity, (si.depth_bucket_max, st.visible)     # the chunked path ran


def test_bucket_sizes_across_the_register_path_variants(gs, ob, device, stream):
    """The register path is instantiated for 4 / 8 / 16 / 30 keys per lane: bucket sizes on both sides of every
    boundary (and of the capacity), all in ONE frame — slabs of bit-different depths inside distinct top digits."""
    import synth
    sizes = [1, 63, 64, 65, 1000, 4095, 4096, 4097, 8192, 8193, 16384, 16385, 30719, 30720, 30721, 40000]
    n = sum(sizes)
    g = synth.scene(n, first=99)
    rng = np.random.default_rng(17)
    # top digit = (bits(z) - bits(near)) >> 18 with 27 key bits (near 0.1, far 100): slab i fills digit 110 + i (z in
    # 1.03 .. 1.6) with random low 18 bits — the camera sits at the origin and looks down -z, so the view depth IS z
    near_bits = int(np.float32(0.1).view(np.uint32))
    zb = np.empty(n, dtype=np.uint32)
    o = 0
    for i, m in enumerate(sizes):
        zb[o:o + m] = near_bits + ((110 + i) << 18) + rng.integers(0, 1 << 18, m, dtype=np.uint32)
        o += m
    z = zb.view(np.float32)
    assert z.min() > 1.0 and z.max() < 2.0
    perm = rng.permutation(n)
    g["pos"][:, 2] = -z[perm]
    g["pos"][:, :2] *= (z[perm, None] / 14.0)
    g["scale"] *= 0.2
    info = []
    _compare_frame(gs, ob, device, stream, gs.SH_NONE, gs.COV3D_ROT_SCALE, g, 1280, 720, gt_kw=dict(sh_deg=0),
                   sort_mode=(1, 0), info=info)
    assert info[0].depth_msd == 1 and info[0].bucket_capacity < info[0].depth_bucket_max <= 40000


def test_renderer_chooses_from_the_reported_buckets(gs, ob, device, stream):
    """Unpinned: a small scene starts MSD-first; a wall (one bucket larger than the register path holds) is reported
    by the frames and the renderer goes to the LSD passes within a few frames; back on a spread-out view of the same
    size it returns to MSD-first.  Every frame of every phase is the oracle's image."""
    n = 80000
    gt, mt = gs.gaussian_transform_pod(1.0, 0, 0, False, 3.0), gs.model_transform_pod()
    ogt, omt = ob.gaussian_transform(sh_deg=0), ob.model_transform()
    cam = helpers.default_camera(gs, 1280, 720)
    ocam = helpers.copy_camera(cam, ob.Camera)
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    r = gs.Renderer(device)
    img = gs.Buffer(device, size=cam.height * cam.width * 16)
    modes = {}
    for phase in ("spread", "wall", "spread"):
        g = _scene(n, phase)
        g["scale"] *= 0.5
        pods = pod.from_gaussian(g)
        buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
        seq = []
        for i in range(6):
            r.render(stream, buf, gt, mt, cam, img.device_ptr())
            seq.append(r.sort_info().depth_msd)
        rgba = img.download(stream, np.float32).reshape(cam.height, cam.width, 4)
        o_rgba = _oracle_frame(ob, pod.sh, pod.cov, pods, ogt, omt, ocam, order=_mirror_order(ob, buf, stream, pod.sh, pod.cov, pods))[-1]
        assert np.array_equal(rgba.view(np.uint32), o_rgba.view(np.uint32)), phase
        modes.setdefault(phase, []).append(seq)
        buf.destroy()
    assert modes["spread"][0][0] == 1 and modes["spread"][0][-1] == 1, modes       # small scene: MSD-first from the start
    assert modes["wall"][0][-1] == 0, modes                                          # the wall was noticed
    assert modes["spread"][1][-1] == 1, modes                                        # and the way back
    r.destroy()
    img.release()
