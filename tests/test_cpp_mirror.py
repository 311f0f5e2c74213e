"""The C++ host mirror (include/gs3d.hpp): compiles against the C ABI on CPU; on the GPU box the
compiled test mirrors the reference's buffer / compute-bundle tests end to end."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    import wgpu_3dgs_core_amd  # noqa: F401  (builds the library if needed)
    return ge.build_cpp_mirror_test()


def test_cpp_mirror_compiles_and_fails_loudly_without_gpu():
    import torch
    exe = _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert "host codecs OK" in res.stdout, res.stdout
    assert res.returncode != 0 and "no HIP device" in res.stdout


@pytest.mark.gpu
def test_cpp_mirror_on_gpu():
    exe = os.path.join(ROOT, "build", "test_mirror")
    if not os.path.exists(exe):
        exe = _build()
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert res.returncode == 0, res.stdout
    assert "cpp mirror OK" in res.stdout


def test_cpp_fullsize_compiles_and_fails_loudly_without_gpu():
    import torch
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    exe = ge.build_cpp_fullsize_test()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = subprocess.run([exe] + ["1000", "3", "0", "0", "64", "64", "x", "y", "z"], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    assert res.returncode != 0 and "HIP headers" in res.stdout and "no HIP device" in res.stdout, res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["1m", "10m"])
def test_cpp_fullsize_on_the_runtime_the_library_is_built_for(name):
    """under Python the product runs on the HIP runtime bundled with the torch
    wheel (ROCm 7.0.x) while it is compiled against /opt/rocm's 7.2 headers.  This test renders the
    BASELINE workloads WITHOUT Python or torch in the process — the runtime is the one the library's
    RUNPATH names — and requires the oracle's frame hashes (tests/golden/fullsize_v3.json), for the
    spatial and the index mirror order."""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_v3.json")))[name]
    exe = os.path.join(ROOT, "build", "test_fullsize")
    if not os.path.exists(exe):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        exe = ge.build_cpp_fullsize_test()
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    env.pop("GS3D_ROUNDS", None)          # the renderer's own choice: the 10 M frames take two rounds from the second frame on
    res = subprocess.run([exe, str(g["n"]), str(g["sh"]), str(g["cov"]), str(g["sh_deg"]), str(g["width"]), str(g["height"]),
                          g["scene_sha256"], g["frame_sha256"], g["frame_sha256_index_order"]],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    print(res.stdout)
    assert res.returncode == 0 and "cpp fullsize OK" in res.stdout, res.stdout[-3000:]
    assert ("rounds 2" in res.stdout) == (name == "10m"), res.stdout[-3000:]
    # the process ran on the system runtime, not on torch's bundled copy
    first = res.stdout.splitlines()[0]
    assert first.startswith("HIP headers"), first
    compiled, runtime = [int(x.split()[-1]) for x in first.split("(")[0].split(",")[:2]]
    assert compiled // 100000 == runtime // 100000 or compiled // 10000000 == runtime // 10000000, first
