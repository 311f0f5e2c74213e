"""Two-round frames (round 5; DESIGN.md §4.2 "rounds", gs_renderer_set_rounds): the nearest K visible Gaussians are
rendered first, their blend marks the finished tiles and leaves the pixel state of the others in the image; the second
round drops the Gaussians whose small rect lies in finished tiles and resumes.  The image must be the single round's —
and the oracle's — bit for bit, whatever K is."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _deep_scene(n, first=4242, opacity=250, scale=3.5):
    """layers of nearly opaque splats: most tiles are finished long before their lists end"""
    import synth
    g = synth.scene(n, first=first)
    g["color"][:, 3] = opacity
    g["scale"] *= np.float32(scale)
    return g


def _setup(gs, ob, g, W, H, sh, cov, mode=0, **cam_kw):
    pod = gs.GaussianPod(sh, cov)
    pods = pod.from_gaussian(g)
    ogt = ob.gaussian_transform(sh_deg=0, mode=mode)
    omt = ob.model_transform()
    ocam = helpers.default_camera(ob, W, H, **cam_kw)
    gt = gs.gaussian_transform_pod(1.0, mode, 0, False, 3.0)
    mt = gs.model_transform_pod((0, 0, 0), (0, 0, 0, 1), (1, 1, 1))
    cam = helpers.copy_camera(ocam, gs.Camera)
    return pod, pods, ogt, omt, ocam, gt, mt, cam


def _render(gs, device, stream, buf, gt, mt, cam, rounds, band=None, masks=None, renderer=None, frames=1):
    # poison: whatever a round leaves behind must be overwritten by the frame
    img = gs.Buffer(device, data=np.full(cam.height * cam.width * 4, np.float32(-7.0)))
    r = renderer or gs.Renderer(device)
    if rounds is None:
        r.set_rounds(0)
    else:
        r.set_rounds(1, rounds)
    if masks is not None:
        r.set_tile_masks(masks)
    # The first frame of a renderer sizes its pair buffers and compacts round 2 out of the full depth order; from the
    # second frame on a two-round frame is PARTITIONED (each round sorts only its side of a depth threshold): every frame
    # must produce the same image
    first = None
    for i in range(max(frames, 2)):
        r.render(stream, buf, gt, mt, cam, img.device_ptr(), band=band)
        stream.synchronize()
        rgba = img.download(stream, np.float32).reshape(cam.height, cam.width, 4)
        if first is None:
            first = rgba
        else:
            assert np.array_equal(first.view(np.uint32), rgba.view(np.uint32)), "frame %d differs from the renderer's first" % i
    fr = r.wait_frame()
    return r, fr, rgba


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_two_rounds_equal_one_round_and_the_oracle(gs, ob, device, stream, mode):
    g = _deep_scene(120000)
    W, H = 640, 360
    pod, pods, ogt, omt, ocam, gt, mt, cam = _setup(gs, ob, g, W, H, gs.SH_NONE, gs.COV3D_ROT_SCALE, mode=mode)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    order = buf.download_order(stream)
    want = ob.render(gs.SH_NONE, gs.COV3D_ROT_SCALE, pods, ogt, omt, ocam, order=order)[0]
    r1, fr1, one = _render(gs, device, stream, buf, gt, mt, cam, None)
    assert r1.sort_info().rounds == 1
    assert np.array_equal(one.view(np.uint32), want.view(np.uint32))
    v = fr1.visible
    assert v > 20000
    seen_fewer = False
    for k in (2048, 4096, v // 4, v // 2, (v // 2048) * 2048, v + 5000):
        r2, fr2, two = _render(gs, device, stream, buf, gt, mt, cam, k)
        si = r2.sort_info()
        if (k + 2047) // 2048 * 2048 >= len(g):
            assert si.rounds == 1                    # nothing left for a second round: one round
        else:
            assert si.rounds == 2 and si.round1 == (k + 2047) // 2048 * 2048
            assert fr2.pairs <= fr1.pairs and fr2.visible == v
            seen_fewer = seen_fewer or fr2.pairs < fr1.pairs
            with pytest.raises(gs.GsError):
                r2.download_ranges(4)
        assert np.array_equal(two.view(np.uint32), one.view(np.uint32)), "K = %d" % k
    if mode == 0:
        assert seen_fewer, "no Gaussian was dropped: the scene finishes no tile"


@pytest.mark.parametrize("masks", [0, 1])
@pytest.mark.parametrize("sh,cov", [("SH_SINGLE", "COV3D_SINGLE"), ("SH_NONE", "COV3D_HALF")])
def test_two_rounds_with_and_without_tile_masks(gs, ob, device, stream, masks, sh, cov):
    sh, cov = getattr(gs, sh), getattr(gs, cov)
    g = _deep_scene(100000, first=77)
    W, H = 500, 300                                   # ragged: 32 x 19 tiles, the last column / row partial
    pod, pods, ogt, omt, ocam, gt, mt, cam = _setup(gs, ob, g, W, H, sh, cov)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    _, fr1, one = _render(gs, device, stream, buf, gt, mt, cam, None, masks=masks)
    for k in (16384, 40000):
        r2, fr2, two = _render(gs, device, stream, buf, gt, mt, cam, k, masks=masks)
        assert r2.sort_info().rounds == 2 and r2.sort_info().tile_masks == masks
        assert np.array_equal(two.view(np.uint32), one.view(np.uint32))
        assert fr2.pairs < fr1.pairs


@pytest.mark.parametrize("W,H", [(240, 160), (387, 144), (16, 16)])
def test_two_rounds_on_small_images(gs, ob, device, stream, W, H):
    """150 / 225 / 1 tiles: a tile sort of ONE pass (or none) leaves the sorted pairs on the other side of its ping-pong
    buffers than the two passes of the larger tests (found by tools/soak_rounds.py: round 1's blend read the side the
    PREVIOUS frame had left its pairs on)."""
    g = _deep_scene(60000, first=123, scale=1.5)
    pod, pods, ogt, omt, ocam, gt, mt, cam = _setup(gs, ob, g, W, H, gs.SH_NONE, gs.COV3D_ROT_SCALE)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    want = ob.render(gs.SH_NONE, gs.COV3D_ROT_SCALE, pods, ogt, omt, ocam, order=buf.download_order(stream))[0]
    for k in (2048, 16384):
        r2, fr2, two = _render(gs, device, stream, buf, gt, mt, cam, k, frames=3)
        assert r2.sort_info().rounds == 2
        assert np.array_equal(two.view(np.uint32), want.view(np.uint32)), "K = %d" % k


def test_round_2_is_skipped_when_round_1_finishes_every_tile(gs, ob, device, stream):
    """k_round2_gate: a view the scene covers with opaque splats — after a long enough round 1 all 920 tiles are finished,
    round 2's kernels return at once (the frame's pair count is round 1's) and the image is still the oracle's; one tile
    short of that, round 2 runs."""
    g = _deep_scene(150000, first=31, opacity=255, scale=5.0)
    W, H = 640, 360
    pod, pods, ogt, omt, ocam, gt, mt, cam = _setup(gs, ob, g, W, H, gs.SH_NONE, gs.COV3D_ROT_SCALE)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    want = ob.render(gs.SH_NONE, gs.COV3D_ROT_SCALE, pods, ogt, omt, ocam, order=buf.download_order(stream))[0]
    seen = set()
    for k in (8192, 32768, 65536):
        r2, fr2, two = _render(gs, device, stream, buf, gt, mt, cam, k, frames=3)
        si = r2.sort_info()
        assert si.rounds == 2
        assert np.array_equal(two.view(np.uint32), want.view(np.uint32)), "K = %d" % k
        seen.add(si.tiles_done == 40 * 23)
        if si.tiles_done == 40 * 23:
            # everything round 2 could have emitted was dropped
            one = gs.Renderer(device)
            one.set_rounds(0)
            img = gs.Buffer(device, size=W * H * 16)
            one.render(stream, buf, gt, mt, cam, img.device_ptr())
            assert fr2.pairs < one.wait_frame().pairs
    assert seen == {False, True}, "the scene must finish every tile for the longest round 1 only: %s" % seen


@pytest.mark.parametrize("W,H", [(4112, 4100), (4112, 640)])
def test_two_rounds_with_wide_tile_keys_and_unpacked_rects(gs, ob, device, stream, W, H):
    """4112 x 4100 px = 66 049 tiles: u32 tile keys, 8-byte rects (the packed format stops at 256 tiles per axis), and the
    box table read from global memory instead of LDS; 4112 x 640 = 257 x 40 tiles: 8-byte rects with u16 keys and the table
    in LDS — the two-round kernels' other instantiations."""
    g = _deep_scene(30000, first=5, opacity=255, scale=12.0)
    pod, pods, ogt, omt, ocam, gt, mt, cam = _setup(gs, ob, g, W, H, gs.SH_NONE, gs.COV3D_ROT_SCALE)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    want = ob.render(gs.SH_NONE, gs.COV3D_ROT_SCALE, pods, ogt, omt, ocam, order=buf.download_order(stream))[0]
    r2, fr2, two = _render(gs, device, stream, buf, gt, mt, cam, 8192, frames=3)
    si = r2.sort_info()
    assert si.rounds == 2 and r2.stats().tiles_x == 257
    assert np.array_equal(two.view(np.uint32), want.view(np.uint32))


def test_two_rounds_in_a_band_and_over_several_frames(gs, ob, device, stream):
    """a rank's band (tile rows 5..14) and the steady state: frames 2.. size their first round from the previous frame's
    visible count (first_round = 0)"""
    g = _deep_scene(120000, first=9)
    W, H = 640, 368
    pod, pods, ogt, omt, ocam, gt, mt, cam = _setup(gs, ob, g, W, H, gs.SH_NONE, gs.COV3D_ROT_SCALE)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    band = (5, 14)
    _, fr1, one = _render(gs, device, stream, buf, gt, mt, cam, None, band=band)
    r2, fr2, two = _render(gs, device, stream, buf, gt, mt, cam, 0, band=band, frames=4)
    si = r2.sort_info()
    assert si.rounds == 2 and si.round1 == (fr1.visible // 4 + 2047) // 2048 * 2048
    rows = slice(band[0] * 16, band[1] * 16)
    assert np.array_equal(two[rows].view(np.uint32), one[rows].view(np.uint32))
    assert np.all(two[:rows.start] == np.float32(-7.0)) and np.all(two[rows.stop:] == np.float32(-7.0))     # rows outside the band untouched


def test_an_empty_and_a_tiny_scene_stay_single_round(gs, ob, device, stream):
    g = _deep_scene(3000)
    pod, pods, ogt, omt, ocam, gt, mt, cam = _setup(gs, ob, g, 320, 200, gs.SH_NONE, gs.COV3D_ROT_SCALE)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    r, fr, img = _render(gs, device, stream, buf, gt, mt, cam, 2048)
    assert r.sort_info().rounds == 1                 # fewer than 4096 Gaussians
    want = ob.render(gs.SH_NONE, gs.COV3D_ROT_SCALE, pods, ogt, omt, ocam, order=buf.download_order(stream))[0]
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
    with pytest.raises(gs.GsError):
        r.set_rounds(2)


def test_two_rounds_at_full_size_match_the_golden_frame(gs, device, stream):
    """BASELINE config 2 (10 M, SH 3, 1080p): three frames with two rounds pinned — the first with the default first round
    of its sizing frame, the others with ~400 pairs per tile — must hash to the oracle's frame (tests/golden/
    fullsize_v3.json), with most tiles finished by round 1 and well under the oracle's pair count emitted."""
    import hashlib
    import json
    import os
    from test_gpu_fullsize import _upload
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = json.load(open(os.path.join(root, "tests", "golden", "fullsize_v3.json")))["10m"]
    pod, buf, scene_sha = _upload(gs, device, stream, g)
    assert scene_sha == g["scene_sha256"]
    cam = gs.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60)), g["width"], g["height"], 0.1, 100.0)
    gt, mt = gs.gaussian_transform_pod(sh_deg=g["sh_deg"]), gs.model_transform_pod()
    img = gs.Buffer(device, size=g["width"] * g["height"] * 16)
    r = gs.Renderer(device)
    r.set_rounds(1)
    for i in range(3):
        r.render(stream, buf, gt, mt, cam, img.device_ptr())
        fr = r.wait_frame()
        si = r.sort_info()
        rgba = img.download(stream, np.float32)
        assert hashlib.sha256(rgba.tobytes()).hexdigest() == g["frame_sha256"], "frame %d" % i
        assert si.rounds == 2 and fr.visible == g["visible"] and fr.pairs < g["pairs"]
    tiles = ((g["width"] + 15) // 16) * ((g["height"] + 15) // 16)
    assert si.tiles_done > 0.9 * tiles and fr.pairs < 0.5 * g["pairs"], (si.tiles_done, tiles, fr.pairs, g["pairs"])
    assert 500_000 <= si.round1 <= 2_000_000, si.round1
    buf.destroy()


def test_renderer_chooses_two_rounds_for_deep_scenes_only():
    """Unpinned (a child process without the tests' GS3D_ROUNDS pin): the 1 M scene of the headline stays with one round
    (296 pairs per tile: its tiles finish at the end of their lists), the 10 M scene (2 970 per tile) takes two from its
    second frame on — and a scene that finishes no tile (sparse, translucent) goes back to one round after the feedback
    has lengthened round 1 three times — and tries again 512 frames later.  All frames equal their single-round frame."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, synth, hashlib, wgpu_3dgs_core_amd as gs\n"
        "dev = gs.Device(0); st = dev.create_stream()\n"
        "cam = gs.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60)), 1920, 1080, 0.1, 100.0)\n"
        "gt, mt = gs.gaussian_transform_pod(sh_deg=0), gs.model_transform_pod()\n"
        "img = gs.Buffer(dev, size=1920 * 1080 * 16)\n"
        "out = []\n"
        "pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)\n"
        "for name, n, alpha in (('1m', 1000000, None), ('10m', 10000000, None), ('thin', 10000000, 6)):\n"
        "    buf = gs.GaussiansBuffer.new_empty(dev, pod, n)\n"
        "    for first in range(0, n, 1000000):\n"
        "        g = synth.scene(min(1000000, n - first), first=first)\n"
        "        if alpha is not None: g['color'][:, 3] = alpha; g['scale'] *= np.float32(1.3)\n"
        "        buf.update_range_with_pod(st, first, pod.from_gaussian(g))\n"
        "    ref = gs.Renderer(dev); ref.set_rounds(0); ref.render(st, buf, gt, mt, cam, img.device_ptr())\n"
        "    want = hashlib.sha256(img.download(st, np.float32).tobytes()).hexdigest(); ref.destroy()\n"
        "    r = gs.Renderer(dev); rounds = []\n"
        "    for i in range(12):\n"
        "        r.render(st, buf, gt, mt, cam, img.device_ptr())\n"
        "        rounds.append(int(r.sort_info().rounds))\n"
        "        assert hashlib.sha256(img.download(st, np.float32).tobytes()).hexdigest() == want, (name, i)\n"
        "    if name == 'thin':\n"
        "        for i in range(530):\n"
        "            r.render(st, buf, gt, mt, cam, img.device_ptr(), check=False)\n"
        "            rounds.append(int(r.sort_info().rounds))\n"
        "        assert hashlib.sha256(img.download(st, np.float32).tobytes()).hexdigest() == want, name\n"
        "    out.append((name, rounds)); print(name, r.stats().pairs, r.stats().visible, rounds[:12], flush=True); r.destroy(); buf.destroy()\n"
        "print('RESULT', out)\n" % (root, os.path.join(root, 'tools')))
    env = dict(os.environ)
    env.pop("GS3D_ROUNDS", None)
    env.pop("GS3D_ROUND1", None)
    env.pop("GS3D_ROUND_PARTITION", None)
    res = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0 and "RESULT" in res.stdout, res.stdout[-3000:]
    by = dict(eval(res.stdout.split("RESULT", 1)[1].strip()))
    assert by["1m"] == [1] * 12, by["1m"]
    assert by["10m"][0] == 1 and by["10m"][1:] == [2] * 11, by["10m"]
    thin = by["thin"]
    assert thin[0] == 1 and thin[1] == 2 and thin[11] == 1 and thin[-1] == 1, thin[:12]
    assert thin[1:12].count(2) <= 4, thin[:12]       # 1.0, 1.5, 2.25, 3.375 x — then off ...
    assert thin[12:500].count(2) == 0 and 1 <= thin[500:].count(2) <= 4, thin[500:]      # ... for 512 frames, then another try
