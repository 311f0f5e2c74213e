"""SURVEY §8e on hardware: the real RCCL collective on ONE GPU, in a fresh child process per import
order (tests/rccl_child.py).  A `nccl` process group of world size 1 runs the actual
`all_gather_into_tensor` (in place, async_op=True) with its stream semantics — what gloo tests
cannot show — and the product library and torch must share ONE HIP runtime whichever is imported
first (wgpu-3dgs-core_amd/_hiprt.py; round 2 died with "No HIP GPUs are available" in the
product-first order).  One child at a time; a child never re-execs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(order, env=None, timeout=600):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), order], cwd=ROOT, env=e,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["product-first", "torch-first"])
def test_rccl_world1_frame_pipeline_on_dedicated_stream(order):
    res = _child(order)
    assert res.returncode == 0, res.stdout[-4000:]
    line = [x for x in res.stdout.splitlines() if x.startswith("{")][-1]
    j = json.loads(line)
    assert j["ok"] and j["backend"] == "nccl" and j["world_size"] == 1
    assert len(j["libamdhip64"]) == 1 and len(j["libhsa"]) == 1
    # both orders end on the same runtime: torch's bundled copy
    assert os.sep + "torch" + os.sep in j["libamdhip64"][0], j


def _torch_with_bundled_runtime():
    """these CPU tests need the built product library (hipcc or a prebuilt .so) AND a ROCm torch wheel
    that bundles its own libamdhip64 — without the latter there is only one runtime to begin with"""
    import importlib.util
    if importlib.util.find_spec("torch") is None:
        return False
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from importlib import import_module
    return import_module("wgpu_3dgs_core_amd._hiprt").torch_lib_dir() is not None


needs_bundled_torch = pytest.mark.skipif(not _torch_with_bundled_runtime(),
                                         reason="needs a ROCm torch wheel with a bundled libamdhip64.so")


@needs_bundled_torch
def test_import_orders_share_one_hip_runtime_cpu():
    """No GPU needed: after `import product; import torch` and after `import torch; import product`
    exactly one libamdhip64 / libhsa-runtime64 / libhiprtc is mapped, and it is the same file."""
    code = ("import sys, json; sys.path.insert(0, %r)\n"
            "%s\n"
            "from importlib import import_module\n"
            "print(json.dumps(import_module('wgpu_3dgs_core_amd._hiprt').check()))\n")
    seen = []
    for imports in ("import wgpu_3dgs_core_amd\nimport torch", "import torch\nimport wgpu_3dgs_core_amd"):
        res = subprocess.run([sys.executable, "-c", code % (ROOT, imports)], stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-3000:]
        m = json.loads(res.stdout.strip().splitlines()[-1])
        assert all(len(v) == 1 for v in m.values()), m
        seen.append(m)
    assert seen[0] == seen[1]


@needs_bundled_torch
def test_forced_system_runtime_then_torch_is_refused_cpu():
    """GS3D_HIP_RUNTIME=system + torch imported afterwards = two runtimes: refused with a message,
    not left to fail later as 'No HIP GPUs are available'."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import wgpu_3dgs_core_amd, torch\n"
            "from importlib import import_module\n"
            "import_module('wgpu_3dgs_core_amd._hiprt').check('torch')\n") % ROOT
    env = dict(os.environ, GS3D_HIP_RUNTIME="system")
    res = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=600)
    assert res.returncode != 0 and "two HIP runtimes" in res.stdout, res.stdout[-2000:]
    assert "GS3D_HIP_RUNTIME=system" in res.stdout           # the message names the ways out


@needs_bundled_torch
def test_forced_system_runtime_without_torch_imports_fine_cpu():
    """the package must stay importable on the system runtime when torch is merely installed."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import wgpu_3dgs_core_amd as gs\n"
            "from importlib import import_module\n"
            "h = import_module('wgpu_3dgs_core_amd._hiprt')\n"
            "m = h.check()\n"
            "assert h.info()['source'] == 'system' and len(m['libamdhip64']) == 1 and 'torch' not in m['libamdhip64'][0]\n"
            "c, r, d = gs.hip_versions(); assert c > 0 and r > 0\n"
            "print('ok', c, r)\n") % ROOT
    env = dict(os.environ, GS3D_HIP_RUNTIME="system")
    res = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-2000:]


@needs_bundled_torch
def test_soname_mismatch_falls_back_to_the_system_runtime_cpu():
    """A torch wheel whose bundled libamdhip64 has another SONAME than libgs3d_hip.so NEEDs cannot serve
    the product: auto mode must not preload it (the product then runs on /opt/rocm's runtime and the
    package imports), and GS3D_HIP_RUNTIME=torch must say why it cannot be honoured."""
    code = ("import importlib.util, os\n"       # the module alone: importing the package would already load the library
            "spec = importlib.util.spec_from_file_location('_hiprt_alone', os.path.join(%r, 'wgpu-3dgs-core_amd', '_hiprt.py'))\n"
            "h = importlib.util.module_from_spec(spec); spec.loader.exec_module(h)\n"
            "soname, needed = h.elf_dynamic(h.torch_lib_dir() + '/libamdhip64.so')\n"
            "assert soname == h.product_needs() and soname.startswith('libamdhip64.so.'), (soname, h.product_needs())\n"
            "h.product_needs = lambda family='libamdhip64': 'libamdhip64.so.99'\n"
            "import os\n"
            "if os.environ.get('GS3D_HIP_RUNTIME') == 'torch':\n"
            "    try:\n"
            "        h.prepare(); print('not refused')\n"
            "    except ImportError as e:\n"
            "        print('refused:', e)\n"
            "else:\n"
            "    assert h.prepare() == 'system' and not h.info()['preloaded'] and 'libamdhip64.so.99' in h.info()['note']\n"
            "    assert not h.mapped()['libamdhip64']\n"
            "    print('fell back')\n") % ROOT
    res = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0 and "fell back" in res.stdout, res.stdout[-2000:]
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GS3D_HIP_RUNTIME="torch"), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0 and "refused:" in res.stdout and "libamdhip64.so.99" in res.stdout, res.stdout[-2000:]
