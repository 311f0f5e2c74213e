#!/usr/bin/env python3
"""Full-size golden results: the CPU oracle (oracle/gs_oracle.c) run once, offline, on the
BASELINE.json configurations — the synthetic scenes of SURVEY §8(d) at their full N and resolution.
Per workload it records N, the visible count V, the pair count D, the sha256 of the packed scene
(so a differing generator is told apart from a differing renderer), of the spatial mirror order and
of the f32 RGBA frame (in that order, and in plain index order for buffers with the spatial order
switched off).  The HIP path is bit-exact against the oracle, so tests/test_gpu_fullsize.py compares hashes.
Output: tests/golden/fullsize_v2.json.   Run:  python tests/golden/make_golden_fullsize.py [workload ...]
(1m takes seconds, 10m a few minutes on 8 cores, 50m needs ~25 GB of RAM.)

fullsize_v1.json (round 1 / 2, kept) was made with version 1 of the tile rect (the
radius square).  Version 2 clips the rect to the splat's visible box: fewer pairs, fewer visible
Gaussians, THE SAME IMAGE.  This generator renders every workload with both versions, refuses to
write anything unless the two frames are bit-identical, and records the counts of both."""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import synth  # noqa: E402
from bench import WORKLOADS  # noqa: E402
from oracle import binding as ob  # noqa: E402


def run(name):
    wl = WORKLOADS[name]
    pods_hash = hashlib.sha256()
    pods = np.empty(wl["n"] * ob.pod_size(wl["sh"], wl["cov"]), dtype=np.uint8)
    step, psz = 1_000_000, ob.pod_size(wl["sh"], wl["cov"])
    for first in range(0, wl["n"], step):
        cnt = min(step, wl["n"] - first)
        p = ob.pack(wl["sh"], wl["cov"], synth.scene(cnt, first=first))
        pods[first * psz:(first + cnt) * psz] = p.reshape(-1)
        pods_hash.update(p.tobytes())
    cam = ob.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60.0)), wl["width"], wl["height"],
                            0.1, 100.0)
    gt, mt = ob.gaussian_transform(sh_deg=wl["sh_deg"]), ob.model_transform()
    t0 = time.time()
    # the buffer's default mirror order: pairs of bit-identical depth follow it
    order = ob.spatial_order(wl["sh"], wl["cov"], pods)
    ob.set_rect_version(1)
    rgba1, d1, v1, _ = ob.render(wl["sh"], wl["cov"], pods, gt, mt, cam, want_image=True, order=order)
    ob.set_rect_version(3)      # the default since round 3 (version 2 + the rounding guard): identical tile counts on
    rgba, d, v, _ = ob.render(  # these workloads (tests/test_rect_versions.py), so fullsize_v2.json standswl["sh"], wl["cov"], pods, gt, mt, cam, want_image=True, order=order)
    dt = time.time() - t0
    if not np.array_equal(rgba.view(np.uint32), rgba1.view(np.uint32)):
        raise SystemExit("%s: rect versions 1 and 2 give different frames" % name)
    del rgba1
    rgba_index_order = ob.render(wl["sh"], wl["cov"], pods, gt, mt, cam, want_image=True)[0]
    alpha = rgba[..., 3]
    return dict(n=wl["n"], sh=wl["sh"], cov=wl["cov"], sh_deg=wl["sh_deg"], width=wl["width"], height=wl["height"],
                visible=v, pairs=d, visible_rect_v1=v1, pairs_rect_v1=d1, scene_sha256=pods_hash.hexdigest(),
                frame_sha256=hashlib.sha256(rgba.tobytes()).hexdigest(),
                order_sha256=hashlib.sha256(order.tobytes()).hexdigest(),
                frame_sha256_index_order=hashlib.sha256(rgba_index_order.tobytes()).hexdigest(),
                pixels_differing_from_index_order=int((rgba.view(np.uint32) != rgba_index_order.view(np.uint32)).any(axis=2).sum()),
                frame_sum=float(rgba.astype(np.float64).sum()), alpha_max=float(alpha.max()),
                covered_pixels=int((alpha > 0).sum()), oracle_seconds=round(dt, 2))


def main():
    names = sys.argv[1:] or ["100k", "1m", "10m", "10m-4k"]
    path = os.path.join(HERE, "fullsize_v2.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    ob.build()
    for name in names:
        out[name] = run(name)
        print(name, out[name], flush=True)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
