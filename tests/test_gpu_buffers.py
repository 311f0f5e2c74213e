"""GPU tests of the data seam, mirroring the reference's tests/buffer/{gaussian,gaussian_transform,
model_transform,mod}.rs: sizes, len, update / update_range incl. error variants, byte-exact
upload->download round trips for all 12 PODs, lossy download panics -> LossyConfigError,
TryFrom size checks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEEDS = list(range(15))


def _pods(gs, ob, pod, seeds=SEEDS):
    g = ob.given_gaussians(seeds)
    return g, pod.from_gaussian(g)


@pytest.mark.parametrize("sh", [0, 1, 2, 3])
@pytest.mark.parametrize("cov", [0, 1, 2])
def test_new_len_size_and_roundtrip(gs, ob, device, stream, sh, cov):
    """tests/buffer/gaussian.rs:24-45,142-170: downloaded bytes == packed bytes."""
    pod = gs.GaussianPod(sh, cov)
    g, pods = _pods(gs, ob, pod)
    buf = gs.GaussiansBuffer.new(device, pod, g)
    assert buf.len() == len(g) and not buf.is_empty()
    assert buf.buffer().size() == len(g) * pod.size
    assert np.array_equal(buf.download(stream), pods)
    buf2 = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    assert np.array_equal(buf2.download(stream), pods)
    empty = gs.GaussiansBuffer.new_empty(device, pod, 7)
    assert empty.len() == 7 and not empty.download(stream).any()
    zero = gs.GaussiansBuffer.new_empty(device, pod, 0)
    assert zero.is_empty()


@pytest.mark.parametrize("pod_idx", [0, 5, 11])
def test_update_and_errors(gs, ob, device, stream, pod_idx):
    """tests/buffer/gaussian.rs:142-280"""
    pod = gs.ALL_PODS[pod_idx]
    g, pods = _pods(gs, ob, pod)
    buf = gs.GaussiansBuffer.new_empty(device, pod, len(g))
    buf.update(stream, g)
    assert np.array_equal(buf.download(stream), pods)
    buf.update_with_pod(stream, pods[::-1].copy()[::-1])
    with pytest.raises(gs.GaussiansBufferUpdateError) as e:
        buf.update(stream, g[:3])
    assert (e.value.count, e.value.expected_count) == (3, len(g))
    with pytest.raises(gs.GaussiansBufferUpdateError):
        buf.update_with_pod(stream, pods[: 2 * pod.size])
    # update_range in the middle
    g2 = ob.given_gaussians([100, 101, 102])
    buf.update_range(stream, 4, g2)
    exp = pods.copy()
    exp[4 * pod.size: 7 * pod.size] = pod.from_gaussian(g2)
    assert np.array_equal(buf.download(stream), exp)
    buf.update_range_with_pod(stream, len(g) - 3, pod.from_gaussian(g2))
    with pytest.raises(gs.GaussiansBufferUpdateRangeError) as e:
        buf.update_range(stream, len(g) - 2, g2)
    assert (e.value.count, e.value.start, e.value.expected_count) == (3, len(g) - 2, len(g))


@pytest.mark.parametrize("sh", [0, 1, 2, 3])
@pytest.mark.parametrize("cov", [0, 1, 2])
def test_download_gaussians(gs, ob, device, stream, sh, cov):
    """tests/buffer/gaussian.rs:282-402: invertible configs round-trip (within the config's
    quantisation), lossy ones panic in the reference -> LossyConfigError."""
    pod = gs.GaussianPod(sh, cov)
    g, pods = _pods(gs, ob, pod)
    buf = gs.GaussiansBuffer.new(device, pod, g)
    if sh == gs.SH_NONE or cov != gs.COV3D_ROT_SCALE:
        with pytest.raises(gs.LossyConfigError):
            buf.download_gaussians(stream)
        return
    out = buf.download_gaussians(stream)
    rc, exp = ob.unpack_to_gaussian(sh, cov, pods)
    assert rc == 0
    assert out.tobytes() == exp.tobytes()
    tol = {gs.SH_SINGLE: 0.0, gs.SH_HALF: 1e-3, gs.SH_NORM8: 1e-2}[sh]
    assert np.abs(out["sh"] - g["sh"]).max() <= tol
    assert np.array_equal(out["rot"], g["rot"]) and np.array_equal(out["scale"], g["scale"])


def test_try_from_buffer(gs, device, stream):
    """tests/buffer/gaussian.rs:404-468"""
    pod = gs.GaussianPodWithShSingleCov3dRotScaleConfigs
    raw = gs.Buffer(device, size=pod.size * 5)
    assert gs.GaussiansBuffer.try_from(raw, pod).len() == 5
    bad = gs.Buffer(device, size=pod.size * 5 + 16)
    with pytest.raises(gs.GaussiansBufferTryFromBufferError) as e:
        gs.GaussiansBuffer.try_from(bad, pod)
    assert (e.value.buffer_size, e.value.expected_multiple_size) == (pod.size * 5 + 16, pod.size)


def test_transform_buffers(gs, device, stream):
    """tests/buffer/gaussian_transform.rs, model_transform.rs, mod.rs:97-134"""
    t = gs.GaussianTransformBuffer(device)
    assert t.size() == 8
    assert bytes(t.download(stream)) == bytes(gs.gaussian_transform_pod())
    t.update(stream, 2.5, gs.DISPLAY_ELLIPSE, 2, True, 1.5)
    raw = t.download(stream)
    assert raw[:4].view(np.float32)[0] == 2.5 and list(raw[4:]) == [1, 2, 1, 127]
    m = gs.ModelTransformBuffer(device)
    assert m.size() == 48
    assert bytes(m.download(stream)) == bytes(gs.model_transform_pod())
    m.update(stream, (1, 2, 3), (0, 0, 0.7071068, 0.7071068), (2, 2, 2))
    f = m.download(stream, np.float32)
    assert list(f[:3]) == [1, 2, 3] and list(f[8:11]) == [2, 2, 2]
    with pytest.raises(gs.FixedSizeBufferWrapperError) as e:
        gs.GaussianTransformBuffer.try_from(gs.Buffer(device, size=16))
    assert (e.value.buffer_size, e.value.expected_size) == (16, 8)
    with pytest.raises(gs.FixedSizeBufferWrapperError):
        gs.ModelTransformBuffer.try_from(gs.Buffer(device, size=8))
    assert gs.ModelTransformBuffer.try_from(gs.Buffer(device, size=48)).size() == 48


def test_generic_buffer_download_and_clone(gs, device, stream):
    data = np.arange(1000, dtype=np.uint32)
    b = gs.Buffer(device, data=data)
    c = b.clone()
    b.release()
    assert np.array_equal(c.download(stream, np.uint32), data)
    c.write(stream, 40, np.array([7, 8, 9], dtype=np.uint32))
    out = c.download(stream, np.uint32)
    assert list(out[10:13]) == [7, 8, 9] and out[13] == 13
    with pytest.raises(gs.InvalidArgumentError):
        c.write(stream, 3999, np.zeros(2, np.uint8))


def test_large_upload_roundtrip_property(gs, device, stream):
    """Full-size property (config 2 scale): 1 M x 48 B upload -> download is the identity."""
    import synth
    pod = gs.GaussianPodWithShNoneCov3dRotScaleConfigs
    g = synth.scene(1_000_000)
    pods = pod.from_gaussian(g)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    assert buf.len() == 1_000_000
    assert np.array_equal(buf.download(stream), pods)


def test_misaligned_device_pointers_are_rejected(gs, device, stream):
    """Adopted raw pointers and the frame pointer are moved with 16-byte vector accesses: a
    misaligned one must be refused up front instead of faulting on the device."""
    import helpers
    pod = gs.GaussianPodWithShNoneCov3dHalfConfigs     # 32-byte records
    big = gs.Buffer(device, size=32 * 8 + 16)
    raw = gs.Buffer.from_raw(device, big.device_ptr() + 4, 32 * 8)
    with pytest.raises(gs.InvalidArgumentError):
        gs.GaussiansBuffer.try_from(raw, pod)
    ok = gs.GaussiansBuffer.try_from(gs.Buffer.from_raw(device, big.device_ptr() + 16, 32 * 8), pod)
    assert len(ok) == 8
    img = gs.Buffer(device, size=64 * 64 * 16 + 16)
    r = gs.Renderer(device)
    cam = helpers.default_camera(gs, 64, 64)
    with pytest.raises(gs.InvalidArgumentError):
        r.render(stream, ok, gs.gaussian_transform_pod(sh_deg=0), gs.model_transform_pod(), cam, img.device_ptr() + 4)
    r.destroy()
