#!/usr/bin/env python3
"""Per-rank cost of a tile-row band on ONE GPU (what one rank of an N-GPU frame would do, without
the all-gather): python tools/band_bench.py --workload 10m --ranks 8 [--which 3]
Prints the frame and stage times of the chosen band; run with GS3D_FORCE_BANDED=0/1 to compare the
single-phase and the two-phase (band-culled SH loads) preprocess kernels."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="10m")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--which", type=int, default=-1, help="rank whose band is rendered (-1 = middle)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--morton", action="store_true", help="experiment: upload the scene in Morton order of the positions")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="also time the band with this many renderers on priority streams taking the frames in turn (FrameRing)")
    args = ap.parse_args()
    import torch  # noqa: F401  (device memory for the frame, as in bench.py)
    import synth
    import wgpu_3dgs_core_amd as gs
    from bench import WORKLOADS, upload_scene
    from wgpu_3dgs_core_amd import parallel as par
    wl = WORKLOADS[args.workload]
    dev = gs.Device(0)
    stream = dev.create_stream()
    if args.morton:
        g = synth.scene(wl["n"])
        q = g["pos"].astype(np.float64)
        q = ((q - q.min(0)) / (q.max(0) - q.min(0)) * 1023.0).astype(np.uint64)
        code = np.zeros(len(g), dtype=np.uint64)
        for b in range(10):
            for a in range(3):
                code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
        g = g[np.argsort(code, kind="stable")]
        pod = gs.GaussianPod(wl["sh"], wl["cov"])
        buf = gs.GaussiansBuffer.new_empty(dev, pod, wl["n"])
        for first in range(0, wl["n"], 1_000_000):
            buf.update_range_with_pod(stream, first, pod.from_gaussian(g[first:first + 1_000_000]))
        del g, q, code
    else:
        pod, buf = upload_scene(gs, synth, dev, stream, wl)
    W, H = wl["width"], wl["height"]
    cam = gs.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60.0)), W, H, 0.1, 100.0)
    gt, mt = gs.gaussian_transform_pod(sh_deg=wl["sh_deg"]), gs.model_transform_pod()
    bands, padded = par.band_plan(H, args.ranks)
    band = bands[args.which if args.which >= 0 else args.ranks // 2]
    img = gs.Buffer(dev, size=max(padded, H) * W * 16)
    r = gs.Renderer(dev)
    for _ in range(3):
        r.render(stream, buf, gt, mt, cam, img.device_ptr(), band=band)
    stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.render(stream, buf, gt, mt, cam, img.device_ptr(), band=band, check=False)
    stream.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    ms_ring = None
    if args.frames_in_flight > 1:
        ring = gs.FrameRing(dev, args.frames_in_flight)
        imgs = [gs.Buffer(dev, size=max(padded, H) * W * 16) for _ in range(len(ring))]
        for k in range(len(ring)):
            ring.render(buf, gt, mt, cam, imgs[k].device_ptr(), band=band, check=True)
        for i in range(100):
            ring.render(buf, gt, mt, cam, imgs[i % len(ring)].device_ptr(), band=band)
        ring.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            ring.render(buf, gt, mt, cam, imgs[i % len(ring)].device_ptr(), band=band)
        ring.synchronize()
        ms_ring = (time.perf_counter() - t0) * 1e3 / 200
        assert all(fr.flags == 0 for fr in ring.wait())
        ring.close()
    r.set_timing(True)
    r.reset_stats()
    for _ in range(args.steps):
        r.render(stream, buf, gt, mt, cam, img.device_ptr(), band=band)
    st = r.stats()
    stages = {n: round(st.stage_ms[i] / max(st.timed_frames, 1), 4) for i, n in enumerate(gs.STAGE_NAMES)}
    print(json.dumps(dict(workload=args.workload, ranks=args.ranks, band=band, ms_per_frame=round(ms, 4),
                          ms_per_frame_in_flight=None if ms_ring is None else round(ms_ring, 4),
                          frames_in_flight=args.frames_in_flight,
                          visible=int(st.visible), pairs=int(st.pairs), stages_ms=stages,
                          forced=os.environ.get("GS3D_FORCE_BANDED"))))


if __name__ == "__main__":
    main()
