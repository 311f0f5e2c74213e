#!/usr/bin/env python3
"""Load-path timing on the GPU box: an N-vertex PLY / SPZ scene from host memory to a
renderable GaussiansBuffer, (a) host conversion + host pack + upload of the PODs (the reference's
way, gs_gaussian_from_ply + gs_pack: on one thread and on all threads), (b) the device path
(gs_gaussians_buffer_create_from_ply / _from_spz: records uploaded as they are, one kernel).
One process per measurement of (a) with GS3D_HOST_THREADS, so the thread count is really applied.

    python tools/load_bench.py [--n 10000000]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def child(n, what):
    import numpy as np
    import wgpu_3dgs_core_amd as gs
    from test_ply import _synthetic_ply
    ply = _synthetic_ply(n, seed=3)
    ply["rot"][:40] = (1.0, 0.1, 0.2, 0.3)
    pod = gs.GaussianPodWithShHalfCov3dRotScaleConfigs
    dev = gs.Device(0)
    st = dev.create_stream()
    warm = gs.GaussiansBuffer.new_from_ply(dev, pod, ply[:1000])     # first-touch costs (module load) out of the way
    warm.destroy()
    out = {"n": n, "what": what, "ply_bytes": n * 248}
    if what == "host":
        t0 = time.perf_counter()
        g = gs.gaussian_from_ply(ply)
        t1 = time.perf_counter()
        pods = pod.from_gaussian(g)
        t2 = time.perf_counter()
        buf = gs.GaussiansBuffer.new_with_pods(dev, pod, pods)
        st.synchronize()
        t3 = time.perf_counter()
        out.update(from_ply_s=t1 - t0, pack_s=t2 - t1, upload_pods_s=t3 - t2, total_s=t3 - t0,
                   threads=os.environ.get("GS3D_HOST_THREADS", "default (<= 32)"))
    elif what == "device":
        t0 = time.perf_counter()
        buf = gs.GaussiansBuffer.new_from_ply(dev, pod, ply)
        t1 = time.perf_counter()
        out.update(total_s=t1 - t0, gbs_of_ply_bytes=n * 248 / (t1 - t0) / 1e9)
    elif what == "spz":
        import synth
        g = synth.scene(n)
        data = gs.SpzGaussians.write_gaussians(g)
        t0 = time.perf_counter()
        g2 = np.ascontiguousarray(gs.SpzGaussians.read_from(data).iter_gaussian(), dtype=gs.GAUSSIAN_DTYPE)
        t1 = time.perf_counter()
        b0 = gs.GaussiansBuffer.new(dev, pod, g2)
        t2 = time.perf_counter()
        b0.destroy()
        buf = gs.GaussiansBuffer.new_from_spz(dev, pod, data)
        t3 = time.perf_counter()
        out.update(spz_bytes=len(data), host_decode_s=t1 - t0, host_path_total_s=t2 - t0, device_path_total_s=t3 - t2)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--child", default="")
    args = ap.parse_args()
    if args.child:
        return child(args.n, args.child)
    for what, env in (("host", {"GS3D_HOST_THREADS": "1"}), ("host", {}), ("device", {}), ("spz", {})):
        e = dict(os.environ)
        e.update(env)
        n = args.n if what != "spz" else min(args.n, 10_000_000)
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--n", str(n), "--child", what], env=e,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        print(lines[-1] if lines else "FAILED %s: %s" % (what, res.stderr[-500:]), flush=True)


if __name__ == "__main__":
    main()
