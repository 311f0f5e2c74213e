#!/usr/bin/env python3
"""Throughput with TWO frames in flight: two renderers (each with its own scratch buffers) on two
streams take the frames alternately, so the latency-bound sort chain of frame i + 1 fills the gaps of
frame i's VALU-bound blend.  Latency per frame is unchanged; this measures what a viewer that
double-buffers its frames gets.  usage (GPU box): python tools/two_in_flight.py [workload] [frames]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import bench  # noqa: E402
import synth  # noqa: E402
import wgpu_3dgs_core_amd as gs  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "1m"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    wl = bench.WORKLOADS[name]
    dev = gs.Device(0)
    up = dev.create_stream()
    pod, buf = bench.upload_scene(gs, synth, dev, up, wl)
    up.synchronize()
    cam = bench._camera(gs, wl)
    gt = gs.gaussian_transform_pod(sh_deg=wl["sh_deg"])
    mt = gs.model_transform_pod()
    W, H = wl["width"], wl["height"]
    for lanes in (1, 2, 3):
        streams = [dev.create_stream() for _ in range(lanes)]
        rs = [gs.Renderer(dev) for _ in range(lanes)]
        imgs = [gs.Buffer(dev, size=W * H * 16) for _ in range(lanes)]
        for k in range(lanes):                      # sizing frames (blocking once each)
            rs[k].render(streams[k], buf, gt, mt, cam, imgs[k].device_ptr(), check=True)
        for i in range(20):
            k = i % lanes
            rs[k].render(streams[k], buf, gt, mt, cam, imgs[k].device_ptr(), check=False)
        for s in streams:
            s.synchronize()
        t0 = time.perf_counter()
        for i in range(frames):
            k = i % lanes
            rs[k].render(streams[k], buf, gt, mt, cam, imgs[k].device_ptr(), check=False)
        t_enq = time.perf_counter() - t0
        for s in streams:
            s.synchronize()
        dt = time.perf_counter() - t0
        sums = [float(imgs[k].download(streams[k], np.float32).astype(np.float64).sum()) for k in range(lanes)]
        print("%s: %d frame(s) in flight: %.4f ms per frame (host enqueue %.4f ms per frame), %.0f Msplats/s, image sums %s" % (
            name, lanes, dt * 1e3 / frames, t_enq * 1e3 / frames, wl["n"] / (dt / frames) / 1e6, sums), flush=True)
        for r in rs:
            r.destroy()
        for im in imgs:
            im.release()


if __name__ == "__main__":
    main()
