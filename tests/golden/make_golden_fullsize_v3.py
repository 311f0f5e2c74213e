#!/usr/bin/env python3
"""fullsize_v3.json: the full-size goldens under version 4 of the tile rect (the exact tile test for
rects of at most 3 x 3 tiles, round 5).  Version 4 only drops (tile, Gaussian) pairs that colour no pixel, so THE
FRAMES DO NOT CHANGE: per workload this script renders the scene with version 3 — and requires the visible count, the
pair count and both frame hashes of fullsize_v2.json — and with version 4 — and requires the same frame hashes again.
What changes, and is recorded, are V and D (the old ones stay as visible_rect_v3 / pairs_rect_v3).
Run:  python tests/golden/make_golden_fullsize_v3.py [workload ...]      (50m needs ~25 GB of RAM)"""
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


def run(name, old):
    wl = WORKLOADS[name]
    psz = ob.pod_size(wl["sh"], wl["cov"])
    pods = np.empty(wl["n"] * psz, dtype=np.uint8)
    h = hashlib.sha256()
    step = 1_000_000
    for first in range(0, wl["n"], step):
        cnt = min(step, wl["n"] - first)
        p = ob.pack(wl["sh"], wl["cov"], synth.scene(cnt, first=first))
        pods[first * psz:(first + cnt) * psz] = p.reshape(-1)
        h.update(p.tobytes())
    assert h.hexdigest() == old["scene_sha256"], "the scene generator changed"
    cam = ob.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60.0)), wl["width"], wl["height"], 0.1, 100.0)
    gt, mt = ob.gaussian_transform(sh_deg=wl["sh_deg"]), ob.model_transform()
    order = ob.spatial_order(wl["sh"], wl["cov"], pods)
    assert hashlib.sha256(order.tobytes()).hexdigest() == old["order_sha256"]
    t0 = time.time()
    ob.set_rect_version(3)
    rgba, d3, v3, _ = ob.render(wl["sh"], wl["cov"], pods, gt, mt, cam, want_image=True, order=order)
    assert (v3, d3, hashlib.sha256(rgba.tobytes()).hexdigest()) == (old["visible"], old["pairs"], old["frame_sha256"]), \
        "version 3 no longer reproduces fullsize_v2.json"
    del rgba
    ob.set_rect_version(4)
    rgba, d4, v4, _ = ob.render(wl["sh"], wl["cov"], pods, gt, mt, cam, want_image=True, order=order)
    if hashlib.sha256(rgba.tobytes()).hexdigest() != old["frame_sha256"]:
        raise SystemExit("%s: rect version 4 changes the frame" % name)
    del rgba
    rgba_i, d4i, v4i, _ = ob.render(wl["sh"], wl["cov"], pods, gt, mt, cam, want_image=True)
    if hashlib.sha256(rgba_i.tobytes()).hexdigest() != old["frame_sha256_index_order"] or (d4i, v4i) != (d4, v4):
        raise SystemExit("%s: rect version 4 changes the index-order frame" % name)
    new = dict(old)
    new.update(visible=v4, pairs=d4, visible_rect_v3=v3, pairs_rect_v3=d3, oracle_seconds=round(time.time() - t0, 2))
    return new


def main():
    names = sys.argv[1:] or ["100k", "1m", "10m", "10m-4k"]
    v2 = json.load(open(os.path.join(HERE, "fullsize_v2.json")))
    path = os.path.join(HERE, "fullsize_v3.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    ob.build()
    for name in names:
        out[name] = run(name, v2[name])
        print(name, {k: out[name][k] for k in ("visible", "pairs", "visible_rect_v3", "pairs_rect_v3", "oracle_seconds")}, flush=True)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
