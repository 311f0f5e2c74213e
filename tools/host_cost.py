#!/usr/bin/env python3
"""Host cost of enqueueing one frame (gs_render_frame returns after its launches are queued):
python tools/host_cost.py --workload 1m [--ranks 8]   -> microseconds per call while the GPU is still busy."""
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
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--frames", type=int, default=200)
    args = ap.parse_args()
    import torch  # noqa: F401
    import synth
    import wgpu_3dgs_core_amd as gs
    from bench import WORKLOADS, upload_scene
    from wgpu_3dgs_core_amd import parallel as par
    wl = WORKLOADS[args.workload]
    dev = gs.Device(0)
    stream = dev.create_stream()
    pod, buf = upload_scene(gs, synth, dev, stream, wl)
    W, H = wl["width"], wl["height"]
    cam = gs.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60.0)), W, H, 0.1, 100.0)
    gt, mt = gs.gaussian_transform_pod(sh_deg=wl["sh_deg"]), gs.model_transform_pod()
    bands, padded = par.band_plan(H, args.ranks)
    band = bands[args.ranks // 2] if args.ranks > 1 else None
    img = gs.Buffer(dev, size=max(padded, H) * W * 16)
    r = gs.Renderer(dev)
    for _ in range(5):
        r.render(stream, buf, gt, mt, cam, img.device_ptr(), band=band)
    stream.synchronize()
    calls = []
    t0 = time.perf_counter()
    for _ in range(args.frames):
        a = time.perf_counter()
        r.render(stream, buf, gt, mt, cam, img.device_ptr(), band=band, check=False)
        calls.append(time.perf_counter() - a)
    t_enq = time.perf_counter() - t0
    stream.synchronize()
    t_all = time.perf_counter() - t0
    calls.sort()
    print(json.dumps(dict(workload=args.workload, ranks=args.ranks, frames=args.frames,
                          enqueue_us_per_frame=round(t_enq * 1e6 / args.frames, 1),
                          call_us_median=round(calls[len(calls) // 2] * 1e6, 1), call_us_p95=round(calls[int(len(calls) * 0.95)] * 1e6, 1),
                          gpu_us_per_frame=round(t_all * 1e6 / args.frames, 1), launches=int(r.wait_frame().launches))))


if __name__ == "__main__":
    main()
