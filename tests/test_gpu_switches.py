"""The debug switches of DESIGN.md §4.4 are read once per process, so the alternative code paths they
select (the pair-emission start table at small sizes, the ballot-based rank of the radix scatter,
versions 1 and 3 of the tile rect — for which the oracle binding switches its version as well —, the MSD-first /
LSD sorts, the unpacked rect format, the store policies) are
exercised by running a subset of the parity tests in a child process per setting — one child at a
time."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUBSET = "synthetic_small or emission or capacity_overflow or edge_cases or wide_tile_keys or radix_sort_skewed or adversarial_needles or all_twelve_pods"


@pytest.mark.parametrize("env", [{"GS3D_CURSOR_KERNEL": "1", "GS3D_RANGES_IN_BLEND": "1"}, {"GS3D_DISABLE_FAST_RANK": "1"},
                                 {"GS3D_BLEND_GROUPS": "1", "GS3D_RANGES_IN_BLEND": "1"}, {"GS3D_RECT_V1": "1"}, {"GS3D_SPATIAL_ORDER": "0"},
                                 {"GS3D_XCD_REMAP": "0", "GS3D_EVENT_FENCE": "1", "GS3D_FRAME_EVENT": "1", "GS3D_NT_LOADS": "1", "GS3D_BLOCK_LIST": "1", "GS3D_NT_SCATTER": "1",
                                  "GS3D_RANGES_SEARCH": "1", "GS3D_RANGES_IN_BLEND": "0", "GS3D_DEPTH_SORT_LARGE": "1"},
                                 {"GS3D_TILE_MASKS": "0", "GS3D_DEPTH_MSD": "0", "GS3D_WT_STORES": "0"},
                                 {"GS3D_DEPTH_MSD": "1", "GS3D_TILE_MSD": "1", "GS3D_WT_RECORDS": "1", "GS3D_DISABLE_FAST_RANK": "1"},
                                 {"GS3D_RECT32": "0", "GS3D_TILE_MSD": "1"},
                                 {"GS3D_PRE_PIPELINE": "0", "GS3D_SCAN_ROWS_SMALL": "0", "GS3D_NT_LOADS": "0", "GS3D_BLOCK_LIST": "0",
                                  "GS3D_NT_SCATTER": "0", "GS3D_RANGES_SEARCH": "0", "GS3D_RANGES_IN_BLEND": "0", "GS3D_CHUNK_HIST": "0", "GS3D_EXPAND_XCD": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_parity_subset_under_switch(env):
    if any(os.environ.get(k) == v for k, v in env.items()):
        pytest.skip("already running under this switch")
    child_env = dict(os.environ)
    child_env.update(env)
    # the block-culling views only where the switch set moves the block test (GS3D_BLOCK_LIST)
    subset = SUBSET + (" or block_culling" if "GS3D_BLOCK_LIST" in env else "")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_render.py"), "-x", "-q",
                          "-m", "gpu", "-k", subset, "-p", "no:cacheprovider"],
                         cwd=ROOT, env=child_env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert " passed" in res.stdout and "deselected" in res.stdout, res.stdout[-1000:]
