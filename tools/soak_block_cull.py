#!/usr/bin/env python3
"""One-off soak of the block-culling bound: runs tests/test_gpu_render.py's randomized conservativeness
test for many more seeds than the suite does.  usage (GPU box): python tools/soak_block_cull.py 100 160 [extreme]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import wgpu_3dgs_core_amd as gs  # noqa: E402
from oracle import binding as ob  # noqa: E402
import test_gpu_render as t  # noqa: E402


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    ob.build()
    dev = gs.Device(0)
    stream = dev.create_stream()
    for seed in range(lo, hi):
        # the test indexes a 3-entry config table with seed - 1: fold the seed, keep the rng seed
        class _S(int):
            pass
        fn = t.test_block_culling_is_conservative_over_random_views.__wrapped__ if hasattr(
            t.test_block_culling_is_conservative_over_random_views, "__wrapped__") else \
            t.test_block_culling_is_conservative_over_random_views
        try:
            fn(gs, ob, dev, stream, seed, extreme=len(sys.argv) > 3)
        except IndexError:
            raise
        print("seed", seed, "ok", flush=True)


if __name__ == "__main__":
    main()
