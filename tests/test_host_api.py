"""CPU tests of the product's host side (no GPU, no compute calls): the C-ABI library loads and
exports every symbol include/gs3d.h declares; host packing matches the golden bytes; transform
PODs; error behaviour without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SH, COV = ["single", "half", "norm8", "none"], ["rot_scale", "single", "half"]


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gs3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(gs):
    names = _declared_functions()
    assert len(names) > 60
    lib = C.CDLL(gs._capi.library_path())
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the ctypes table covers exactly the header
    assert sorted(gs._capi.SIGNATURES) == names
    assert lib.gs_abi_version() == 1


def test_no_oracle_in_product():
    """The product path must not import, link or call the oracle."""
    pkg = os.path.join(ROOT, "wgpu-3dgs-core_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("gs_oracle", "gso_", "import oracle", "from oracle", "oracle/"):
                    assert needle not in src, (f, needle)
    out = os.popen("ldd %s" % os.path.join(pkg, "lib", "libgs3d_hip.so")).read()
    assert "oracle" not in out


def test_device_creation_fails_loudly_without_gpu(gs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(gs.NoDeviceError):
        gs.Device(0)


@pytest.mark.parametrize("sh", range(4))
@pytest.mark.parametrize("cov", range(3))
def test_host_pack_matches_golden_and_oracle(gs, ob, golden, sh, cov):
    g = np.zeros(len(golden["seeds"]), dtype=gs.GAUSSIAN_DTYPE)
    for f in ("rot", "pos", "color", "sh", "scale"):
        g[f] = golden[f]
    pod = gs.GaussianPod(sh, cov)
    got = pod.from_gaussian(g)
    assert np.array_equal(got, golden["pod_%s_%s" % (SH[sh], COV[cov])])
    import synth
    big = synth.scene(70000, first=999)        # > threading threshold of the host packer
    assert np.array_equal(pod.from_gaussian(big), ob.pack(sh, cov, big))
    assert pod.features() == [(n, i == sh or i == 4 + cov) for i, n in enumerate(gs.FEATURE_NAMES)]
    if sh == 3 or cov != 0:
        with pytest.raises(gs.LossyConfigError):
            pod.into_gaussian(got)
    else:
        back = pod.into_gaussian(got)
        rc, exp = ob.unpack_to_gaussian(sh, cov, got)
        assert rc == 0 and back.tobytes() == exp.tobytes()


def test_f16_rounding_edge_cases(gs, ob):
    """host f32->f16 must be RNE incl. ties, subnormals, overflow (vs numpy float16)"""
    vals = np.array([0.0, -0.0, 1.0, 1.0009765625, 1.00048828125, 1.00146484375, 65504.0, 65519.9, 65520.0,
                     1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, -3.3e-5, 1e5, -1e5, 0.1, 0.3333333],
                    dtype=np.float32)
    g = np.zeros(1, dtype=gs.GAUSSIAN_DTYPE)
    g["sh"][0, :len(vals)] = vals
    got = gs.GaussianPodWithShHalfCov3dRotScaleConfigs.from_gaussian(g)[16:16 + 2 * len(vals)].view(np.float16)
    with np.errstate(over="ignore"):
        exp = vals.astype(np.float16)
    assert np.array_equal(got.view(np.uint16), exp.view(np.uint16))
    assert np.array_equal(ob.pack(1, 0, g)[16:16 + 2 * len(vals)].view(np.uint16), exp.view(np.uint16))


def test_transform_pods(gs, golden):
    for mode, deg, no_sh0, std, u8, flags, dec in golden["flags_table"]:
        pod = gs.gaussian_transform_pod(1.0, int(mode), int(deg), bool(no_sh0), float(std))
        assert int.from_bytes(bytes(pod.flags), "little") == int(flags)
    d = gs.gaussian_transform_pod()
    assert d.size == 1.0 and list(d.flags) == [0, 3, 0, 255]            # Default impl
    assert gs.GaussianShDegree.new(4) is None and gs.GaussianShDegree.new(3).get() == 3
    assert gs.GaussianMaxStdDev.new(3.1) is None and gs.GaussianMaxStdDev.new(-0.1) is None
    assert gs.GaussianMaxStdDev.new(1.5).as_u8() == 127
    assert abs(gs.GaussianMaxStdDev.new(1.5).get() - 1.494117) < 1e-5
    with pytest.raises(gs.InvalidArgumentError):
        gs.gaussian_transform_pod(sh_deg=4)
    with pytest.raises(gs.InvalidArgumentError):
        gs.gaussian_transform_pod(max_std_dev=3.01)
    m = gs.model_transform_pod()
    assert bytes(m) == np.array([0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 0], dtype=np.float32).tobytes()
    assert C.sizeof(gs.GaussianTransformPod) == 8 and C.sizeof(gs.ModelTransformPod) == 48


def test_camera_helper_matches_oracle(gs, ob):
    a = gs.camera_look_at((1, 2, 3), (0, -1, -9), (0, 1, 0), 0.9, 1920, 1080, 0.2, 50.0)
    b = ob.camera_look_at((1, 2, 3), (0, -1, -9), (0, 1, 0), 0.9, 1920, 1080, 0.2, 50.0)
    assert bytes(a) == bytes(b)
    assert abs(gs.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60)), 1920, 1080).fx - 935.307) < 1e-2


def test_builder_errors_need_no_device(gs):
    B = gs.ComputeBundleBuilder
    with pytest.raises(gs.MissingBindGroupLayout):
        B().build_without_bind_groups(None)
    with pytest.raises(gs.MissingResolver):
        B().bind_group_layout(1).build(None, [])
    with pytest.raises(gs.MissingEntryPoint):
        B().bind_group_layout(1).resolver(gs.KernelRegistry()).build(None, [])
    with pytest.raises(gs.MissingMainShader):
        B().bind_group_layout(1).resolver(gs.KernelRegistry()).entry_point("main").build(None, [])


def test_generated_rust_ffi_is_in_sync_with_the_header():
    """bindings/rust/gs3d_sys.rs is generated from include/gs3d.h: it must be regenerated whenever the
    header changes, and declare exactly the functions the ctypes table binds."""
    import re
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    rs = open(os.path.join(ROOT, "bindings", "rust", "gs3d_sys.rs")).read()
    from wgpu_3dgs_core_amd import _capi
    assert sorted(re.findall(r"pub fn (gs_\w+)\(", rs)) == sorted(_capi.SIGNATURES)
    for struct, size in (("gs_gaussian", 236), ("gs_camera", 116), ("gs_projected", 48), ("gs_spz_header", 16)):
        assert "pub struct %s {" % struct in rs
