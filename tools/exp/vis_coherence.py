"""How coherent is visibility in mirror (Morton) order?  Renders the 10m workload once, takes the
per-Gaussian tile counts, orders them as the mirror does and histograms visible-per-group."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, synth
import wgpu_3dgs_core_amd as gs

name = sys.argv[1] if len(sys.argv) > 1 else "10m"
wl = bench.WORKLOADS[name]
dev = gs.Device(0)
stream = dev.create_stream()
pod, buf = bench.upload_scene(gs, synth, dev, stream, wl)
n = wl["n"]
pos = np.empty((n, 3), np.float32)
for first in range(0, n, 1_000_000):
    cnt = min(1_000_000, n - first)
    g = synth.scene(cnt, first=first)
    pos[first:first + cnt] = g["pos"]
lo = pos.min(0); hi = pos.max(0)
q = np.clip(((pos - lo) / (hi - lo) * np.float32(1024.0)).astype(np.int64), 0, 1023).astype(np.uint32)
code = np.zeros(n, np.uint32)
for a in range(3):
    for b in range(10):
        code |= ((q[:, a] >> b) & 1) << (3 * b + a)
order = np.argsort(code, kind="stable")
cam = bench._camera(gs, wl)
r = bench.make_renderer(gs, dev, stream, pod, buf, wl, cam) if hasattr(bench, "make_renderer") else None
print("have make_renderer", r is not None)
