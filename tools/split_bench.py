#!/usr/bin/env python3
"""One frame as K tile-row bands on K streams of ONE GPU (fork / join with events every frame): the
latency-bound sort chain of one band overlaps the VALU-bound blend of another.
python tools/split_bench.py --workload 1m --bands 2 [--steps 200]"""
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
    ap.add_argument("--workload", default="1m")
    ap.add_argument("--bands", type=int, default=2)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--cut", type=str, default="", help="comma-separated tile rows where the bands are cut (default: equal rows)")
    args = ap.parse_args()
    import torch
    import synth
    import wgpu_3dgs_core_amd as gs
    from bench import WORKLOADS, upload_scene
    from wgpu_3dgs_core_amd import parallel as par
    wl = WORKLOADS[args.workload]
    dev = gs.Device(0)
    K = args.bands
    tstreams = [torch.cuda.Stream() for _ in range(K)]
    streams = [dev.wrap_stream(s.cuda_stream) for s in tstreams]
    pod, buf = upload_scene(gs, synth, dev, streams[0], wl)
    W, H = wl["width"], wl["height"]
    cam = gs.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60.0)), W, H, 0.1, 100.0)
    gt, mt = gs.gaussian_transform_pod(sh_deg=wl["sh_deg"]), gs.model_transform_pod()
    rows = (H + 15) // 16
    if args.cut:
        cuts = [0] + [int(x) for x in args.cut.split(",")] + [rows]
        bands = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
        K = len(bands)
    else:
        bands, _ = par.band_plan(H, K)
    img = gs.Buffer(dev, size=H * W * 16)
    ref = gs.Buffer(dev, size=H * W * 16)
    rs = [gs.Renderer(dev) for _ in range(K)]
    r0 = gs.Renderer(dev)
    r0.render(streams[0], buf, gt, mt, cam, ref.device_ptr())
    streams[0].synchronize()
    fork, joins = torch.cuda.Event(), [torch.cuda.Event() for _ in range(K)]

    def frame(check):
        fork.record(tstreams[0])
        for k in range(1, K):
            tstreams[k].wait_event(fork)
        for k in range(K):
            rs[k].render(streams[k], buf, gt, mt, cam, img.device_ptr(), band=tuple(bands[k]), check=check)
        for k in range(1, K):
            joins[k].record(tstreams[k])
            tstreams[0].wait_event(joins[k])

    for _ in range(5):
        frame(True)
    torch.cuda.synchronize()
    a = img.download(streams[0], np.uint32)
    b = ref.download(streams[0], np.uint32)
    same = bool(np.array_equal(a, b))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame(False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r0.render(streams[0], buf, gt, mt, cam, ref.device_ptr(), check=False)
    torch.cuda.synchronize()
    ms1 = (time.perf_counter() - t0) * 1e3 / args.steps
    print(json.dumps(dict(workload=args.workload, bands=[list(x) for x in bands], split_ms=round(ms, 4), single_ms=round(ms1, 4),
                          bit_identical=same)))


if __name__ == "__main__":
    main()
