#!/usr/bin/env python3
"""A scene several times BASELINE's largest on ONE GPU, one round against the renderer's own choice (two rounds): ms per
frame of both, and the two images must be identical.  (The opt-in test tests/test_gpu_fullsize.py::
test_largest_scene_one_gpu checks V and D of the single-round frame against the oracle's preprocess.)
usage (GPU box): python tools/big_scene.py [millions=200]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth  # noqa: E402
import wgpu_3dgs_core_amd as gs  # noqa: E402


def main():
    n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 200_000_000
    dev = gs.Device(0)
    st = dev.create_stream()
    pod = gs.GaussianPod(gs.SH_HALF, gs.COV3D_ROT_SCALE)
    buf = gs.GaussiansBuffer.new_empty(dev, pod, n)
    t0 = time.perf_counter()
    for first in range(0, n, 1_000_000):
        buf.update_range_with_pod(st, first, pod.from_gaussian(synth.scene(min(1_000_000, n - first), first=first)))
        if first % 20_000_000 == 0:
            print("uploaded %d M (%.0f s)" % (first // 1_000_000, time.perf_counter() - t0), flush=True)
    st.synchronize()
    W, H = 1920, 1080
    cam = gs.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60)), W, H, 0.1, 100.0)
    gt, mt = gs.gaussian_transform_pod(sh_deg=3), gs.model_transform_pod()
    img = gs.Buffer(dev, size=W * H * 16)
    out = {}
    for name, mode in (("one round", 0), ("the renderer's choice", -1)):
        r = gs.Renderer(dev)
        r.set_rounds(mode)
        for _ in range(8):                                   # sizing frame + the frames the choices settle in
            fr = r.render(st, buf, gt, mt, cam, img.device_ptr())
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            r.render(st, buf, gt, mt, cam, img.device_ptr(), check=False)
        st.synchronize()
        ms = (time.perf_counter() - t0) * 1e2
        fr = r.wait_frame()
        si = r.sort_info()
        sha = hashlib.sha256(img.download(st, np.float32).tobytes()).hexdigest()
        out[name] = sha
        print("%d M Gaussians (%d B records), %s: %.3f ms per frame = %.0f Msplats/s; rounds %d partitioned %d round1 %d tiles_done %d; "
              "V %d pairs %d launches %d; sha256 %s" % (n // 1_000_000, pod.size, name, ms, n / ms / 1e3, si.rounds, si.partitioned,
                                                      si.round1, si.tiles_done, fr.visible, fr.pairs, fr.launches, sha[:16]), flush=True)
        r.destroy()
    assert len(set(out.values())) == 1, "the images differ"
    print("big scene OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
