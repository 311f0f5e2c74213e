#!/usr/bin/env python3
"""One-off soak of the whole frame against the oracle on randomly drawn scenes: Gaussian count, image
size, splat scale (from sub-pixel to screen-filling), depth range, POD layout, SH degree, band.
Every stage is compared bit for bit (tests/test_gpu_render.py::_compare_frame).
usage (GPU box): python tools/soak_frames.py <first seed> <last seed> [big]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import synth  # noqa: E402
import wgpu_3dgs_core_amd as gs  # noqa: E402
from oracle import binding as ob  # noqa: E402
import test_gpu_render as t  # noqa: E402


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    ob.build()
    dev = gs.Device(0)
    stream = dev.create_stream()
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 7, 63, 64, 65, 255, 256, 257, 1000, 1023, 1025, 4097, 20000, 60000]))
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        if rng.random() < 0.2:
            w, h = int(rng.choice([16, 1920, 33])), int(rng.choice([16, 1080, 31]))
        if big:                                     # several sort tiles per pass, several emit workgroups
            n = int(rng.choice([150_000, 300_000, 500_000]))
            w, h = int(rng.choice([1920, 1280, 3840])), int(rng.choice([1080, 720, 2160]))
        g = synth.scene(n, first=int(rng.integers(0, 1 << 30)))
        g["scale"] *= np.float32(rng.choice([0.02, 0.3, 1.0, 3.0, 12.0]))
        if rng.random() < 0.3:                      # a few screen-fillers right in front of the camera
            k = min(n, int(rng.integers(1, 6)))
            g["pos"][:k, 2] = -rng.uniform(0.3, 1.0, k).astype(np.float32)
            g["pos"][:k, :2] *= 0.01
            g["scale"][:k] = rng.uniform(0.5, 2.5, (k, 3)).astype(np.float32)
        if rng.random() < 0.5:
            g["pos"][:, :2] *= np.float32(0.15)     # most of them inside the frustum
        if rng.random() < 0.25:
            g["pos"][:, 2] = np.float32(-5.0)       # one depth: the order is all ties
        sh, cov = int(rng.integers(0, 4)), int(rng.integers(0, 3))
        deg = int(rng.integers(0, 4))
        tiles_y = (h + 15) // 16
        band = None
        if rng.random() < 0.3 and tiles_y > 1:
            a = int(rng.integers(0, tiles_y))
            band = (a, int(rng.integers(a + 1, tiles_y + 1)))
        gt_kw = dict(sh_deg=deg)
        mt_kw = {}
        if rng.random() < 0.5:                      # transform uniforms: size, max_std_dev, display mode, model transform
            gt_kw.update(size=float(rng.choice([0.5, 1.0, 2.0])), max_std_dev=float(rng.choice([1.0, 1.5, 2.5, 3.0])),
                         mode=int(rng.choice([0, 0, 1, 2])), no_sh0=bool(rng.random() < 0.3))
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            mt_kw = dict(pos=tuple(float(x) for x in rng.uniform(-1, 1, 3)), rot=tuple(float(x) for x in q),
                         scale=tuple(float(x) for x in rng.uniform(0.5, 2.0, 3)))
        st = t._compare_frame(gs, ob, dev, stream, sh, cov, g, w, h, gt_kw=gt_kw, mt_kw=mt_kw, band=band)
        print("seed %d ok: n %d %dx%d sh %d cov %d deg %d band %s -> V %d D %d" % (
            seed, n, w, h, sh, cov, deg, band, st.visible, st.pairs), flush=True)


if __name__ == "__main__":
    main()
