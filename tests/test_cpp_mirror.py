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
