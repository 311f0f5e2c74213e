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
