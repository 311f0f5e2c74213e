import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Tile rect version 4 is the oracle's default; the product turns it on by itself only for scenes large
# enough that its test is free (gs_renderer_set_tile_masks), so the parity tests pin it for the whole process — product and
# oracle binding read the same variable (GS3D_TILE_MASKS=0 runs both on version 3).
os.environ.setdefault("GS3D_TILE_MASKS", "1")
# Two-round frames (gs_renderer_set_rounds) emit fewer pairs than the oracle counts and refuse the pair / range taps: the
# parity tests pin one round; tests/test_gpu_rounds.py pins two through the API and checks the renderer's own choice in
# a child process without the variable.
os.environ.setdefault("GS3D_ROUNDS", "0")
# ... and two-round frames are PARTITIONED from a renderer's second frame on (by default only from 32 M Gaussians): the
# first frame of every renderer in tests/test_gpu_rounds.py compacts round 2 out of the full depth order, the following
# ones sort each round's side of the depth threshold.
os.environ.setdefault("GS3D_ROUND_PARTITION", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle (test infrastructure)."""
    from oracle import binding
    binding.build()
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def gs():
    import wgpu_3dgs_core_amd
    return wgpu_3dgs_core_amd


@pytest.fixture(scope="session")
def device(gs):
    """A real HIP device.  GPU tests must fail (not skip) when the extension cannot reach one."""
    dev = gs.Device(0)
    yield dev
    dev.close()


@pytest.fixture()
def stream(device):
    s = device.create_stream()
    yield s
    s.synchronize()
    s.close()
