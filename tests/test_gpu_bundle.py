"""GPU tests of the kernel-library and launch seams, mirroring tests/shader/*.rs and
tests/e2e/compute_bundle.rs of the reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dispatch_test_gaussian(gs, device, stream, pod, g, wg=None):
    buf = gs.GaussiansBuffer.new(device, pod, g)
    out = gs.Buffer(device, size=56 * 4)
    b = (gs.ComputeBundleBuilder().bind_group_layout(2).resolver(gs.KernelRegistry())
         .wesl_compile_options(pod).main_shader("test_gaussian").entry_point("main"))
    if wg:
        b = b.workgroup_size(wg)
    bundle = b.build(device, [[buf.buffer(), out]])
    bundle.dispatch(stream, 1)
    return out.download(stream, np.float32)


@pytest.mark.parametrize("sh", [0, 1, 2, 3])
@pytest.mark.parametrize("cov", [0, 1, 2])
def test_unpack_against_oracle_and_reference_tolerances(gs, ob, golden, device, stream, sh, cov):
    """tests/shader/gaussian.rs:160-367 with the reference's inputs (seed 42) and tolerances, plus
    bit-equality with the oracle's restatement of the WESL functions."""
    pod = gs.GaussianPod(sh, cov)
    g = ob.given_gaussians([42])
    out = _dispatch_test_gaussian(gs, device, stream, pod, g)
    exp = ob.shader_test_gaussian(sh, cov, pod.from_gaussian(g))
    assert out.tobytes() == exp.tobytes()
    assert np.abs(out[:4] - g["color"][0] / 255.0).max() < 1e-4
    if sh != gs.SH_NONE:
        tol = {gs.SH_SINGLE: 1e-2, gs.SH_HALF: 1e-1, gs.SH_NORM8: 1e-1}[sh]
        assert np.abs(out[4:49] - g["sh"][0]).max() < tol
    else:
        assert not out[4:49].any()
    i42 = list(golden["seeds"]).index(42)
    tol = {gs.COV3D_ROT_SCALE: 1e-2, gs.COV3D_SINGLE: 1e-2, gs.COV3D_HALF: 1.0}[cov]
    assert np.abs(out[49:55] - golden["cov6_f64"][i42]).max() < tol


@pytest.mark.parametrize("sh", [0, 1, 2, 3])
@pytest.mark.parametrize("cov", [0, 1, 2])
def test_unpack_soa_many(gs, ob, device, stream, sh, cov):
    """AoS POD -> SoA f32 for many Gaussians, bit-exact vs the oracle."""
    import synth
    pod = gs.GaussianPod(sh, cov)
    n = 3000
    g = synth.scene(n, first=17)
    pods = pod.from_gaussian(g)
    buf = gs.GaussiansBuffer.new_with_pods(device, pod, pods)
    out = gs.Buffer(device, size=55 * n * 4)
    bundle = gs.ComputeBundle.new("unpack", device, [2], [[buf.buffer(), out]], gs.KERNEL_UNPACK_SOA,
                                  pod, workgroup_size=256)
    bundle.dispatch(stream, n)
    got = out.download(stream, np.float32).reshape(55, n)
    exp = np.stack([ob.shader_test_gaussian(sh, cov, pods[i * pod.size:(i + 1) * pod.size])[:55]
                    for i in range(n)], axis=1)
    assert got.tobytes() == exp.tobytes()


def test_gaussian_transform_flags(gs, golden, device, stream):
    """tests/shader/gaussian_transform.rs:87-150 + the full flag table."""
    tb = gs.GaussianTransformBuffer(device)
    out = gs.Buffer(device, size=16)
    bundle = (gs.ComputeBundleBuilder().bind_group_layout(2).resolver(gs.KernelRegistry())
              .main_shader("test_gaussian_transform").entry_point("main").build(device, [[tb, out]]))
    tb.update(stream, 1.0, gs.DISPLAY_ELLIPSE, 2, True, 3.0)
    bundle.dispatch(stream, 1)
    o = out.download(stream, np.uint32)
    assert list(o[:3]) == [1, 2, 1] and abs(o[3:4].view(np.float32)[0] - 3.0) < 1e-6
    for mode, deg, no_sh0, std, u8, flags, dec in golden["flags_table"]:
        tb.update(stream, 1.0, int(mode), int(deg), bool(no_sh0), float(std))
        raw = tb.download(stream)
        assert int.from_bytes(bytes(raw[4:]), "little") == int(flags)
        bundle.dispatch(stream, 1)
        o = out.download(stream, np.uint32)
        assert list(o[:3]) == [int(mode), int(deg), int(no_sh0)]
        assert o[3:4].view(np.float32)[0] == np.float32(dec)


def test_model_transform_matrices(gs, ob, golden, device, stream):
    """tests/shader/model_transform.rs:100-201"""
    for case in golden["model_cases"]:
        pos, rot, scale, p = case[0:3], case[3:7], case[7:10], case[10:13]
        mat, sr, inv, world = case[13:29], case[29:38], case[38:47], case[47:51]
        mb = gs.ModelTransformBuffer(device)
        mb.update(stream, pos, rot, scale)
        tp = gs.Buffer(device, data=np.array(list(p) + [0.0], dtype=np.float32))
        out = gs.Buffer(device, size=44 * 4)
        bundle = (gs.ComputeBundleBuilder().bind_group_layout(3).resolver(gs.KernelRegistry())
                  .main_shader("test_model_transform").entry_point("main").build(device, [[mb, tp, out]]))
        bundle.dispatch(stream, 1)
        o = out.download(stream, np.float32).astype(np.float64)

        def close(a, b):  # reference tolerance 1e-6 (absolute, f32 vs f32); vs an f64 expectation
            return np.all(np.abs(a - b) <= 1e-6 + 1.2e-7 * np.abs(b))  # allow 1 ulp of f32
        assert close(o[0:4], world)
        assert close(o[4:20], mat)
        assert close(o[20:32].reshape(3, 4)[:, :3].reshape(-1), inv)
        assert close(o[32:44].reshape(3, 4)[:, :3].reshape(-1), sr)
        # and bit-equal to the oracle restatement
        omt = ob.model_transform(pos, rot, scale)
        w = np.zeros(4, np.float32)
        ob.lib().gso_model_to_world(ob.C.byref(omt), np.asarray(p, np.float32).ctypes.data, w.ctypes.data)
        assert out.download(stream, np.float32)[:4].tobytes() == w.tobytes()


# ---- tests/e2e/compute_bundle.rs ----------------------------------------------------------------

def _data(gs, device):
    return gs.Buffer(device, data=np.array([1, 2, 3, 4, 5], dtype=np.uint32))


@pytest.mark.parametrize("wg", [1, None])
def test_array_map_add_managed(gs, device, stream, wg):
    """compute_bundle.rs:10-48: wg = 1 and wg = device limit"""
    data = _data(gs, device)
    b = (gs.ComputeBundleBuilder().label("array map add").bind_group_layout(1)
         .resolver(gs.KernelRegistry()).main_shader("array_map_add").entry_point("main"))
    if wg:
        b = b.workgroup_size(wg)
    bundle = b.build(device, [[data]])
    limit = min(device.limits().max_compute_workgroup_size_x,
                device.limits().max_compute_invocations_per_workgroup)
    assert bundle.workgroup_size() == (wg or limit)
    assert bundle.label() == "array map add" and bundle.bind_groups() == 1
    bundle.dispatch(stream, 5)
    assert bundle.last_workgroup_count() == -(-5 // bundle.workgroup_size())
    assert list(data.download(stream, np.uint32)) == [2, 3, 4, 5, 6]


def test_array_map_add_unmanaged_and_rebinding(gs, device, stream):
    """compute_bundle.rs:50-79"""
    d1, d2 = _data(gs, device), _data(gs, device)
    bundle = (gs.ComputeBundleBuilder().bind_group_layout(1).resolver(gs.KernelRegistry())
              .main_shader("array_map_add").entry_point("main").workgroup_size(64)
              .build_without_bind_groups(device))
    assert bundle.bind_groups() == 0 and bundle.bind_group_layouts() == 1
    bundle.dispatch(stream, 5, [[d1]])
    bundle.dispatch(stream, 5, [[d2]])
    bundle.dispatch(stream, 5, [[d2]])
    assert list(d1.download(stream, np.uint32)) == [2, 3, 4, 5, 6]
    assert list(d2.download(stream, np.uint32)) == [3, 4, 5, 6, 7]
    managed = (gs.ComputeBundleBuilder().bind_group_layout(1).resolver(gs.KernelRegistry())
               .main_shader("array_map_add").entry_point("main").build(device, [[d1]]))
    assert managed.update_bind_group_with_binding_resources(0, [d2]) is True
    assert managed.update_bind_group_with_binding_resources(1, [d2]) is False
    managed.dispatch(stream, 5)
    assert list(d2.download(stream, np.uint32)) == [4, 5, 6, 7, 8]


def test_two_groups_constant_and_features(gs, device, stream):
    """compute_bundle.rs:113-240: uniform 10 + constant 20 -> +31 per pass"""
    data = _data(gs, device)
    uni = gs.Buffer(device, data=np.array([10], dtype=np.uint32))
    bundle = (gs.ComputeBundleBuilder().bind_group_layouts([1, 1]).resolver(gs.KernelRegistry())
              .pipeline_compile_options({"additional_constant": 20.0})
              .main_shader("array_map_add").entry_point("main").build(device, [[data], [uni]]))
    bundle.dispatch(stream, 5)
    assert list(data.download(stream, np.uint32)) == [32, 33, 34, 35, 36]
    # dispatching fewer invocations than elements touches only the covered workgroups
    big = gs.Buffer(device, data=np.zeros(1000, dtype=np.uint32))
    b2 = gs.ComputeBundle.new(None, device, [1], [[big]], gs.KERNEL_ARRAY_MAP_ADD, workgroup_size=64)
    b2.dispatch(stream, 100)
    assert b2.last_workgroup_count() == 2
    o = big.download(stream, np.uint32)
    assert o[:128].tolist() == [1] * 128 and not o[128:].any()


def test_builder_and_creation_errors(gs, device):
    """compute_bundle.rs:81-111, 242-378: the six error variants, in the reference's order."""
    B = gs.ComputeBundleBuilder
    with pytest.raises(gs.MissingBindGroupLayout):
        B().build_without_bind_groups(device)
    with pytest.raises(gs.MissingResolver):
        B().bind_group_layout(1).build_without_bind_groups(device)
    with pytest.raises(gs.MissingEntryPoint):
        B().bind_group_layout(1).resolver(gs.KernelRegistry()).build_without_bind_groups(device)
    with pytest.raises(gs.MissingMainShader):
        B().bind_group_layout(1).resolver(gs.KernelRegistry()).entry_point("main").build_without_bind_groups(device)
    with pytest.raises(gs.KernelResolveError):
        (B().bind_group_layout(1).resolver(gs.KernelRegistry()).entry_point("main")
         .main_shader("does_not_exist").build_without_bind_groups(device))
    limit = min(device.limits().max_compute_workgroup_size_x,
                device.limits().max_compute_invocations_per_workgroup)
    with pytest.raises(gs.WorkgroupSizeExceedsDeviceLimit) as e:
        (B().bind_group_layout(1).resolver(gs.KernelRegistry()).entry_point("main")
         .main_shader("array_map_add").workgroup_size(limit + 1).build_without_bind_groups(device))
    assert (e.value.workgroup_size, e.value.device_limit) == (limit + 1, limit)
    data = gs.Buffer(device, data=np.zeros(4, np.uint32))
    with pytest.raises(gs.ResourceCountMismatch) as e:
        (B().bind_group_layouts([1, 1]).resolver(gs.KernelRegistry()).entry_point("main")
         .main_shader("array_map_add").build(device, [[data]]))
    assert (e.value.resource_count, e.value.bind_group_layout_count) == (1, 2)


def test_bindings_too_small_are_rejected_on_the_host(gs, device, stream):
    pod = gs.GaussianPodWithShSingleCov3dRotScaleConfigs
    small = gs.Buffer(device, size=64)
    out = gs.Buffer(device, size=56 * 4)
    bundle = gs.ComputeBundle.new(None, device, [2], [[small, out]], gs.KERNEL_TEST_GAUSSIAN, pod)
    with pytest.raises(gs.InvalidArgumentError):
        bundle.dispatch(stream, 1)


# ---- custom shaders compiled at run time (the ComputeBundleBuilder + WESL path of the reference) ----

ARRAY_MAP_ADD_SRC = r"""
// tests/common/shader/array_map_add.wesl restated in HIP C++
#include <wgpu_3dgs_core.h>
extern "C" __global__ void main(gs::BundleArgs a, uint32_t count) {
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t *data = (uint32_t *)a.ptr[0];
    if (index >= (uint32_t)(a.size[0] / 4)) return;
    data[index] = data[index] + 1;
#ifdef second_group
    data[index] = data[index] + *(const uint32_t *)a.ptr[1];
#endif
#ifdef additional_constant
    data[index] = data[index] + additional_constant;
#endif
}
"""

TEST_GAUSSIAN_SRC = r"""
// tests/shader/gaussian.rs:14-59: a caller's module importing the device library
#include <wgpu_3dgs_core.h>
extern "C" __global__ void main(gs::BundleArgs a, uint32_t count) {
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= 1) return;
    const uint32_t *g = (const uint32_t *)a.ptr[0];
    float *out = (float *)a.ptr[1];
    gs::vec4 c = gs::gaussian_unpack_color(g);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
    for (uint32_t i = 0; i < 15; i++) {
        gs::vec3 s = gs::gaussian_unpack_sh<GS_SH>(g, i);
        out[4 + 3 * i] = s.x; out[5 + 3 * i] = s.y; out[6 + 3 * i] = s.z;
    }
    float cov[6];
    gs::gaussian_unpack_cov3d<GS_SH, GS_COV>(g, cov);
    for (int k = 0; k < 6; k++) out[49 + k] = cov[k];
    out[55] = 0.0f;
}
"""


def test_custom_shader_array_map_add_with_features_and_constant(gs, device, stream):
    """compute_bundle.rs:113-240 with a runtime-compiled module: WESL features -> macros,
    override constants -> macros, wg = 1 and wg = device limit"""
    for wg in (1, None):
        data = _data(gs, device)
        uni = gs.Buffer(device, data=np.array([10], dtype=np.uint32))
        b = (gs.ComputeBundleBuilder().label("custom").bind_group_layouts([1, 1])
             .resolver(gs.SourceResolver({"array_map_add": ARRAY_MAP_ADD_SRC}))
             .wesl_compile_options(None, defines=["second_group"])
             .pipeline_compile_options({"additional_constant": 20})
             .main_shader("array_map_add").entry_point("main"))
        if wg:
            b = b.workgroup_size(wg)
        bundle = b.build(device, [[data], [uni]])
        bundle.dispatch(stream, 5)
        assert list(data.download(stream, np.uint32)) == [32, 33, 34, 35, 36]
    plain = (gs.ComputeBundleBuilder().bind_group_layout(1)
             .resolver(gs.SourceResolver({"m": ARRAY_MAP_ADD_SRC})).main_shader("m").entry_point("main")
             .build_without_bind_groups(device))
    d2 = _data(gs, device)
    plain.dispatch(stream, 5, [[d2]])
    assert list(d2.download(stream, np.uint32)) == [2, 3, 4, 5, 6]


@pytest.mark.parametrize("pod_idx", [0, 4, 8, 11])
def test_custom_shader_imports_device_library(gs, ob, device, stream, pod_idx):
    """A caller's module importing gaussian_unpack_* (tests/shader/gaussian.rs) is bit-equal to the
    built-in kernel and to the oracle."""
    pod = gs.ALL_PODS[pod_idx]
    g = ob.given_gaussians([42])
    buf = gs.GaussiansBuffer.new(device, pod, g)
    out = gs.Buffer(device, size=56 * 4)
    bundle = (gs.ComputeBundleBuilder().bind_group_layout(2)
              .resolver(gs.SourceResolver({"test_gaussian": TEST_GAUSSIAN_SRC}))
              .wesl_compile_options(pod).main_shader("test_gaussian").entry_point("main")
              .build(device, [[buf.buffer(), out]]))
    bundle.dispatch(stream, 1)
    got = out.download(stream, np.float32)
    assert got.tobytes() == ob.shader_test_gaussian(pod.sh, pod.cov, pod.from_gaussian(g)).tobytes()


def test_custom_shader_errors(gs, device):
    """Wesl compile error analogue, unknown module, missing entry point, resource count mismatch"""
    B = gs.ComputeBundleBuilder
    with pytest.raises(gs.KernelResolveError) as e:
        (B().bind_group_layout(1).resolver(gs.SourceResolver({"bad": "this is not C++"}))
         .main_shader("bad").entry_point("main").build_without_bind_groups(device))
    assert "error" in str(e.value)
    with pytest.raises(gs.KernelResolveError):
        (B().bind_group_layout(1).resolver(gs.SourceResolver({})).main_shader("nope").entry_point("main")
         .build_without_bind_groups(device))
    with pytest.raises(gs.GsError):
        (B().bind_group_layout(1).resolver(gs.SourceResolver({"m": ARRAY_MAP_ADD_SRC})).main_shader("m")
         .entry_point("not_there").build_without_bind_groups(device))
    data = gs.Buffer(device, data=np.zeros(4, np.uint32))
    with pytest.raises(gs.ResourceCountMismatch):
        (B().bind_group_layouts([1, 1]).resolver(gs.SourceResolver({"m": ARRAY_MAP_ADD_SRC})).main_shader("m")
         .entry_point("main").build(device, [[data]]))
