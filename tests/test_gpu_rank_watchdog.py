"""The radix sorts rank with returning LDS atomics when the order probe of gs_device_create passes
(DESIGN.md §4.4).  The probe is a bet on an undocumented property, so every frame re-checks it: one workgroup
of every radix pass (and one bucket of the bucket sort) — a different one every frame — compares the ranks the atomics
handed out in one of its rounds with the ballot-based ones.  A mismatch
reaches the host as gs_frame_result.flags bit 2 / GS_ERR_RANK_ORDER, the device falls back to the
ballot-based rank and the frame is rendered again.

The hardware does not fail on demand: GS3D_TEST_RANK_FAULT=1 (read once per process) shifts the
watchdog's expectation by one, so that it fires in the first frame of a renderer.  Each case therefore
runs in a child process — one at a time."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARMED = os.environ.get("GS3D_TEST_RANK_FAULT") == "1"


def _scene(gs, ob, n=20000, W=800, H=600):
    import synth
    g = synth.scene(n, first=4242)
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    pods = pod.from_gaussian(g)
    ogt, omt = ob.gaussian_transform(sh_deg=0), ob.model_transform()
    ocam = helpers.default_camera(ob, W, H)
    gt = gs.gaussian_transform_pod(1.0, 0, 0, False, 3.0)
    mt = gs.model_transform_pod((0, 0, 0), (0, 0, 0, 1), (1, 1, 1))
    cam = helpers.copy_camera(ocam, gs.Camera)
    return pod, pods, (ogt, omt, ocam), (gt, mt, cam)


def _oracle_image(gs, ob, buf, stream, pods, o):
    from test_gpu_render import _oracle_frame
    order = buf.download_order(stream)
    return _oracle_frame(ob, gs.SH_NONE, gs.COV3D_ROT_SCALE, pods, o[0], o[1], o[2], None, order=order)[-1]


@pytest.mark.skipif(not ARMED, reason="runs in the child process of test_watchdog_children (GS3D_TEST_RANK_FAULT=1)")
def test_child_explicit_wait_reports_and_switches(gs, ob, device, stream):
    if os.environ.get("GS3D_DISABLE_FAST_RANK"):
        pytest.skip("the LDS-atomic rank is switched off: nothing to watch")
    assert device.fast_rank(), "the order probe failed on this device: the watchdog has nothing to watch"
    pod, pods, o, p = _scene(gs, ob)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    img = gs.Buffer(device, size=p[2].height * p[2].width * 16)
    r = gs.Renderer(device)
    r.render(stream, buf, p[0], p[1], p[2], img.device_ptr(), check=False)
    with pytest.raises(gs.RankOrderError):
        r.wait_frame()
    assert not device.fast_rank()                       # the device has been switched
    # the next frame: ballot-based rank, no flag, exact image
    fr = r.render(stream, buf, p[0], p[1], p[2], img.device_ptr())
    assert fr.flags == 0
    rgba = img.download(stream, np.float32).reshape(p[2].height, p[2].width, 4)
    o_rgba = _oracle_image(gs, ob, buf, stream, pods, o)
    assert np.array_equal(rgba.view(np.uint32), o_rgba.view(np.uint32))
    # a second renderer of the same device starts on the ballot-based rank: nothing fires
    r2 = gs.Renderer(device)
    assert r2.render(stream, buf, p[0], p[1], p[2], img.device_ptr()).flags == 0
    for h in (r, r2, buf):
        h.destroy()
    img.release()


@pytest.mark.skipif(not ARMED, reason="runs in the child process of test_watchdog_children (GS3D_TEST_RANK_FAULT=1)")
def test_child_checked_render_retries_by_itself(gs, ob, device, stream):
    """render(check=True) — the validated use — renders the frame again on its own; a fresh device,
    because the one of the fixture was switched by the test above"""
    if os.environ.get("GS3D_DISABLE_FAST_RANK"):
        pytest.skip("the LDS-atomic rank is switched off: nothing to watch")
    dev = gs.Device(device.ordinal)
    st = dev.create_stream()
    assert dev.fast_rank()
    pod, pods, o, p = _scene(gs, ob, n=12000, W=640, H=360)
    buf = gs.GaussiansBuffer.new_with_pods(dev, pod, pods)
    img = gs.Buffer(dev, size=p[2].height * p[2].width * 16)
    r = gs.Renderer(dev)
    fr = r.render(st, buf, p[0], p[1], p[2], img.device_ptr())
    assert fr.flags == 0 and not dev.fast_rank()
    rgba = img.download(st, np.float32).reshape(p[2].height, p[2].width, 4)
    assert np.array_equal(rgba.view(np.uint32), _oracle_image(gs, ob, buf, st, pods, o).view(np.uint32))
    for h in (r, buf):
        h.destroy()
    img.release()


@pytest.mark.skipif(not ARMED, reason="runs in the child process of test_watchdog_children (GS3D_TEST_RANK_FAULT=1)")
def test_child_pipelined_frames_switch_without_a_wait(gs, ob, device, stream):
    """frames that are only enqueued (a viewer's loop): the flag reaches the host through the result
    history, the switch happens at a later render call, and the frames after it are exact"""
    if os.environ.get("GS3D_DISABLE_FAST_RANK"):
        pytest.skip("the LDS-atomic rank is switched off: nothing to watch")
    dev = gs.Device(device.ordinal)
    st = dev.create_stream()
    pod, pods, o, p = _scene(gs, ob, n=12000, W=640, H=360)
    buf = gs.GaussiansBuffer.new_with_pods(dev, pod, pods)
    img = gs.Buffer(dev, size=p[2].height * p[2].width * 16)
    r = gs.Renderer(dev)
    flags_word = gs.Buffer(dev, size=4)
    r.set_frame_flags_target(flags_word.device_ptr())
    seen = []
    for i in range(6):
        r.render(st, buf, p[0], p[1], p[2], img.device_ptr(), check=False)
        st.synchronize()                                  # so that the history holds the frame at the next call
        seen.append(int(flags_word.download(st, np.uint32)[0]))
    assert seen[0] & gs.FRAME_FLAG_RANK_FAULT              # the device word carries the bit too
    assert not dev.fast_rank() and seen[-1] == 0 and seen[-2] == 0
    rgba = img.download(st, np.float32).reshape(p[2].height, p[2].width, 4)
    assert np.array_equal(rgba.view(np.uint32), _oracle_image(gs, ob, buf, st, pods, o).view(np.uint32))
    for h in (r, buf):
        h.destroy()
    img.release()
    flags_word.release()


@pytest.mark.skipif(not ARMED, reason="runs in the child process of test_watchdog_children (GS3D_TEST_RANK_FAULT=1)")
def test_child_ring_of_three_recovers_on_every_lane(gs, ob, device, stream):
    """Several renderers on one device (FrameRing, parallel.lanes) with a faulted frame in flight on EACH: the first
    one to report switches the device; the others must still clear their own watchdog word (round 4 left it set for
    good: every later frame of those lanes carried the flag and wait_frame raised forever — ADVICE r04)."""
    if os.environ.get("GS3D_DISABLE_FAST_RANK"):
        pytest.skip("the LDS-atomic rank is switched off: nothing to watch")
    dev = gs.Device(device.ordinal)
    assert dev.fast_rank()
    pod, pods, o, p = _scene(gs, ob, n=12000, W=640, H=360)
    buf = gs.GaussiansBuffer.new_with_pods(dev, pod, pods)
    ring = gs.FrameRing(dev, 3)
    imgs = [gs.Buffer(dev, size=p[2].height * p[2].width * 16) for _ in range(3)]
    for k in range(3):                       # one armed frame in flight per lane before anybody looks
        ring.render(buf, p[0], p[1], p[2], imgs[k].device_ptr(), check=False)
    faults = 0
    for r in ring.renderers:
        try:
            r.wait_frame()
        except gs.RankOrderError:
            faults += 1
    assert faults == 3 and not dev.fast_rank()
    for rnd in range(3):
        for k in range(3):
            ring.render(buf, p[0], p[1], p[2], imgs[k].device_ptr(), check=False)
        res = ring.wait()                    # raises if any lane still carries the flag
        assert all(fr.flags == 0 for fr in res), (rnd, [fr.flags for fr in res])
    want = _oracle_image(gs, ob, buf, ring.streams[0], pods, o).view(np.uint32)
    for k in range(3):
        rgba = imgs[k].download(ring.streams[k], np.float32).reshape(p[2].height, p[2].width, 4)
        assert np.array_equal(rgba.view(np.uint32), want), k
    ring.close()
    buf.destroy()
    for im in imgs:
        im.release()


@pytest.mark.skipif(ARMED, reason="this is the child")
@pytest.mark.parametrize("extra", [{}, {"GS3D_TEST_RANK_WATCH": "1234567"}, {"GS3D_DEPTH_MSD": "0", "GS3D_TEST_RANK_WATCH": "3"}],
                         ids=["watch=generation", "watch=1234567", "lsd-watch=3"])
def test_watchdog_children(extra):
    """The armed cases, each set in a child process: with the watchdog's sample following the frame generation (frame 1:
    tile 1 of every pass, never tile 0), pinned to an arbitrary tile / round (GS3D_TEST_RANK_WATCH: tile = n mod live
    tiles, round = (n / live tiles) mod keys per lane — round 4 only ever looked at tile 0, round 0), and with the depth
    sort's LSD passes instead of the MSD-first sort (whose bucket kernel carries its own check)."""
    env = dict(os.environ, GS3D_TEST_RANK_FAULT="1", **extra)
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu",
                          "-k", "child", "-p", "no:cacheprovider"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "4 passed" in res.stdout, res.stdout[-1000:]       # (a child that skipped its cases is a failure)


@pytest.mark.skipif(ARMED, reason="this is the child")
def test_watchdog_is_quiet_on_this_device(gs, ob, device, stream):
    """not armed: frames of every size class of the sorts leave the flag clear and the device on the fast rank"""
    if not device.fast_rank():
        pytest.skip("the order probe failed / GS3D_DISABLE_FAST_RANK: the ballot-based rank is in use")
    pod, pods, o, p = _scene(gs, ob, n=60000, W=1920, H=1080)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    img = gs.Buffer(device, size=p[2].height * p[2].width * 16)
    r = gs.Renderer(device)
    for _ in range(3):
        assert r.render(stream, buf, p[0], p[1], p[2], img.device_ptr()).flags == 0
    assert device.fast_rank()
    for h in (r, buf):
        h.destroy()
    img.release()
