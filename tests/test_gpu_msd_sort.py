"""The frame's depth sort, MSD-first (round 5): one compacting scatter on the top 10 bits of the depth
key, then one workgroup per bucket finishes the low bits on its CU (gs::k_bucket_sort).  Against the oracle and
against the LSD passes, on the register path (buckets up to `bucket_capacity`), on the chunked in-kernel fallback
(larger buckets: depths that collapse into a few top digits) and through the renderer's own choice between the two
sorts, which follows the bucket sizes the frames report."""
import numpy as np
import pytest

import helpers
from test_gpu_render import _compare_frame, _mirror_order, _oracle_frame

pytestmark = pytest.mark.gpu

import os as _os
ROOT_DIR = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))


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
    # top digit = (bits(z) - bits(near)) >> 17 with 27 key bits (near 0.1, far 100): slab i fills digit 220 + i (z in
    # 1.03 .. 1.3) with random low 17 bits — the camera sits at the origin and looks down -z, so the view depth IS z
    near_bits = int(np.float32(0.1).view(np.uint32))
    zb = np.empty(n, dtype=np.uint32)
    o = 0
    for i, m in enumerate(sizes):
        zb[o:o + m] = near_bits + ((220 + i) << 17) + rng.integers(0, 1 << 17, m, dtype=np.uint32)
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


# ---- the tile sort, MSD-first: k_pairs_emit counts the top 10 bits of the tile id, one scatter, k_bucket_sort finishes
# ---- the 2^(bits - 10) tiles of every bucket and writes the tile ranges ----

@pytest.mark.parametrize("size,expect", [((1280, 720), 1), ((1920, 1080), 1), ((4000, 2200), 1), ((640, 360), 0), ((4112, 4100), 0)])
@pytest.mark.parametrize("tile_msd", [0, 1])
def test_tile_sort_modes_match_the_oracle(gs, ob, device, stream, size, expect, tile_msd):
    """12 / 13 / 16 tile-id bits take the MSD-first tile sort when pinned; 10 bits (nothing left below the top digit) and
    more than 65536 tiles (u32 tile ids) stay with the LSD passes whatever is pinned.  Keys, indices, RANGES (written by
    the bucket kernel) and the image against the oracle."""
    import synth
    g = synth.scene(20000, first=808)
    g["scale"] *= 2.0
    info = []
    st = _compare_frame(gs, ob, device, stream, gs.SH_NONE, gs.COV3D_ROT_SCALE, g, size[0], size[1], gt_kw=dict(sh_deg=0),
                        sort_mode=(-1, tile_msd), info=info)
    assert info[0].tile_msd == (tile_msd and expect)
    if info[0].tile_msd:
        assert 0 < info[0].tile_bucket_max <= st.pairs


def test_tile_buckets_beyond_the_register_path(gs, ob, device, stream):
    """50 000 large splats on a small part of the screen: single 8-tile buckets hold far more pairs than the register
    path takes (30 720) — the bucket kernel's chunked path, and the ranges it writes, must match the oracle."""
    import synth
    g = synth.scene(50000, first=4711)
    g["pos"][:, 0] *= 0.03
    g["pos"][:, 1] *= 0.03
    info = []
    st = _compare_frame(gs, ob, device, stream, gs.SH_NONE, gs.COV3D_ROT_SCALE, g, 1920, 1080, gt_kw=dict(sh_deg=0),
                        sort_mode=(-1, 1), info=info)
    assert info[0].tile_msd == 1 and info[0].tile_bucket_max > info[0].bucket_capacity, (info[0].tile_bucket_max, st.pairs)


def test_renderer_leaves_the_msd_tile_sort_when_buckets_overflow(gs, ob, device, stream):
    """With the renderer allowed to choose the MSD-first tile sort (GS3D_TILE_MSD_AUTO=1; off by default: at 1 M it is
    5 us slower than the LSD passes): the dense scene above starts MSD-first (few pairs in all), reports its oversized
    bucket one frame late, and the renderer continues with the LSD passes; every frame is the oracle's image."""
    import os
    import synth
    if os.environ.get("GS3D_TILE_MSD_AUTO") != "1":
        pytest.skip("the renderer does not choose the MSD-first tile sort by itself (GS3D_TILE_MSD_AUTO=1 enables)")
    g = synth.scene(50000, first=4711)
    g["pos"][:, 0] *= 0.03
    g["pos"][:, 1] *= 0.03
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    pods = pod.from_gaussian(g)
    gt, mt = gs.gaussian_transform_pod(1.0, 0, 0, False, 3.0), gs.model_transform_pod()
    cam = helpers.default_camera(gs, 1920, 1080)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    img = gs.Buffer(device, size=cam.height * cam.width * 16)
    r = gs.Renderer(device)
    o_rgba = _oracle_frame(ob, pod.sh, pod.cov, pods, ob.gaussian_transform(sh_deg=0), ob.model_transform(),
                           helpers.copy_camera(cam, ob.Camera), order=_mirror_order(ob, buf, stream, pod.sh, pod.cov, pods))[-1]
    seq = []
    for i in range(6):
        r.render(stream, buf, gt, mt, cam, img.device_ptr())
        seq.append(r.sort_info().tile_msd)
        rgba = img.download(stream, np.float32).reshape(cam.height, cam.width, 4)
        assert np.array_equal(rgba.view(np.uint32), o_rgba.view(np.uint32)), (i, seq)
    assert seq[0] == 1 and seq[-1] == 0 and seq[-2] == 0, seq
    for h in (r, buf):
        h.destroy()
    img.release()


def test_tile_masks_follow_the_scene_size_unless_pinned(gs, device, stream):
    """gs_renderer_set_tile_masks: unpinned (the tests' process pins it through GS3D_TILE_MASKS, so this runs in a child
    without the variable), a scene whose records fit the Infinity Cache renders with rect version 3 — its preprocess
    kernel is instruction-bound and the tile test would cost more than the pairs it saves — and a scene beyond 512 MB of
    records with version 4; both pins override."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, synth, wgpu_3dgs_core_amd as gs\n"
        "dev = gs.Device(0); st = dev.create_stream()\n"
        "cam = gs.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), 1.0, 640, 360, 0.1, 100.0)\n"
        "gt, mt = gs.gaussian_transform_pod(sh_deg=0), gs.model_transform_pod()\n"
        "img = gs.Buffer(dev, size=640 * 360 * 16)\n"
        "out = []\n"
        "for n, pod in ((20000, gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)), (2500000, gs.GaussianPod(gs.SH_SINGLE, gs.COV3D_ROT_SCALE))):\n"
        "    buf = gs.GaussiansBuffer.new_empty(dev, pod, n)\n"
        "    for first in range(0, n, 500000):\n"
        "        buf.update_range_with_pod(st, first, pod.from_gaussian(synth.scene(min(500000, n - first), first=first)))\n"
        "    for mode in (-1, 0, 1):\n"
        "        r = gs.Renderer(dev); r.set_tile_masks(mode)\n"
        "        r.render(st, buf, gt, mt, cam, img.device_ptr())\n"
        "        out.append((n, mode, r.sort_info().tile_masks, r.stats().pairs)); r.destroy()\n"
        "    buf.destroy()\n"
        "print('RESULT', out)\n" % (ROOT_DIR, os.path.join(ROOT_DIR, 'tools')))
    env = dict(os.environ)
    env.pop("GS3D_TILE_MASKS", None)
    res = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0 and "RESULT" in res.stdout, res.stdout[-3000:]
    out = eval(res.stdout.split("RESULT", 1)[1].strip())
    by = {(n, mode): (masks, pairs) for n, mode, masks, pairs in out}
    assert by[(20000, -1)][0] == 0 and by[(2500000, -1)][0] == 1, by          # the renderer's own choice
    for n in (20000, 2500000):
        assert by[(n, 0)][0] == 0 and by[(n, 1)][0] == 1, by                  # the pins
        assert by[(n, 1)][1] < by[(n, 0)][1], by                              # version 4 emits fewer pairs
        assert by[(n, -1)][1] == by[(n, by[(n, -1)][0])][1], by
