#!/usr/bin/env python3
"""One-off soak of two-round frames (DESIGN.md §4.2 "rounds") against single-round frames of the same renderer
inputs, bit for bit, on randomly drawn scenes: Gaussian count, image size, splat scale and opacity (from scenes that
finish no tile to scenes that finish all of them early), POD layout, display mode, band, length of round 1 — and a
camera that moves from frame to frame while one renderer keeps rendering (its history then belongs to other views).
Both ways of making round 2 run: the first frame of a renderer compacts it out of the full depth order, the following
ones sort each round's side of a depth threshold (GS3D_ROUND_PARTITION=1, set here unless the caller set it).
usage (GPU box): python tools/soak_rounds.py <first seed> <last seed> | orbit"""
import os
import sys

import numpy as np

os.environ.setdefault("GS3D_ROUND_PARTITION", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import synth  # noqa: E402
import wgpu_3dgs_core_amd as gs  # noqa: E402


def orbit(dev, stream, n=10_000_000, frames=60):
    """The renderer's OWN choice under a camera that keeps moving (a slow orbit with a zoom: the visible count and the pair
    count change from frame to frame, frames are enqueued without waiting, like a viewer's loop): every frame must equal
    the single-round frame of the same camera, and none may be skipped."""
    pod = gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE)
    buf = gs.GaussiansBuffer.new_empty(dev, pod, n)
    for first in range(0, n, 1_000_000):
        buf.update_range_with_pod(stream, first, pod.from_gaussian(synth.scene(min(1_000_000, n - first), first=first)))
    gt, mt = gs.gaussian_transform_pod(sh_deg=0), gs.model_transform_pod()
    one, auto = gs.Renderer(dev), gs.Renderer(dev)
    one.set_rounds(0)
    w, h = 1920, 1080
    img1, img2 = gs.Buffer(dev, size=w * h * 16), gs.Buffer(dev, size=w * h * 16)
    rounds, skipped = [], 0
    for f in range(frames):
        ang = 0.02 * f
        eye = (float(3.0 * np.sin(ang)), 0.3 * float(np.sin(0.5 * ang)), float(3.0 - 3.0 * np.cos(ang)) - 0.05 * f)
        cam = gs.camera_look_at(eye, (0.0, 0.0, -8.0), (0, 1, 0), float(np.deg2rad(60)), w, h, 0.1, 100.0)
        one.render(stream, buf, gt, mt, cam, img1.device_ptr())
        auto.render(stream, buf, gt, mt, cam, img2.device_ptr(), check=False)      # a viewer's loop: enqueue only
        a = img1.download(stream, np.uint32)
        b = img2.download(stream, np.uint32)
        si = auto.sort_info()
        rounds.append(int(si.rounds))
        st = auto.stats()
        if not np.array_equal(a, b):
            # a frame flagged as skipped leaves the image undefined: count it, render it again the validated way
            auto.render(stream, buf, gt, mt, cam, img2.device_ptr())
            b = img2.download(stream, np.uint32)
            skipped += 1
            if not np.array_equal(a, b):
                print("MISMATCH orbit frame %d rounds %d round1 %d tiles_done %d" % (f, si.rounds, si.round1, si.tiles_done))
                return 1
        if f % 10 == 0:
            print("orbit frame %d: rounds %d round1 %d tiles_done %d visible %d pairs %d" % (f, si.rounds, si.round1, si.tiles_done,
                                                                                         st.visible, st.pairs), flush=True)
    print("orbit OK: %d frames, rounds %s, %d rendered again after a skip" % (frames, "".join(str(x) for x in rounds), skipped))
    return 0 if skipped <= frames // 10 else 1


def main():
    dev = gs.Device(0)
    stream = dev.create_stream()
    if sys.argv[1] == "orbit":
        return orbit(dev, stream)
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    dropped = frames = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([5000, 20000, 60000, 150000, 400000]))
        w, h = int(rng.integers(64, 1300)), int(rng.integers(64, 800))
        if rng.random() < 0.25:
            w, h = 1920, 1080
        g = synth.scene(n, first=int(rng.integers(0, 1 << 30)))
        g["scale"] *= np.float32(rng.choice([0.3, 1.0, 2.0, 3.5, 6.0]))
        if rng.random() < 0.7:
            g["color"][:, 3] = int(rng.choice([3, 40, 200, 250, 255]))
        if rng.random() < 0.4:
            g["pos"][:, :2] *= np.float32(rng.choice([0.15, 0.5]))     # most of them inside the frustum / half of the image empty
        if rng.random() < 0.15:
            g["pos"][:, 2] = np.float32(-5.0)                            # one depth: the order is all ties
        sh, cov = int(rng.integers(0, 4)), int(rng.integers(0, 3))
        mode = int(rng.choice([0, 0, 0, 1, 2]))
        pod = gs.GaussianPod(sh, cov)
        buf = gs.GaussiansBuffer.new_with_pods(dev, pod, pod.from_gaussian(g))
        gt = gs.gaussian_transform_pod(float(rng.choice([0.5, 1.0, 2.0])), mode, int(rng.integers(0, 4)), False, 3.0)
        mt = gs.model_transform_pod()
        tiles_y = (h + 15) // 16
        band = None
        if rng.random() < 0.3 and tiles_y > 2:
            a = int(rng.integers(0, tiles_y - 1))
            band = (a, int(rng.integers(a + 1, tiles_y + 1)))
        k = int(rng.choice([0, 2048, 4096, n // 16, n // 4, n // 2, n]))
        one, two = gs.Renderer(dev), gs.Renderer(dev)
        one.set_rounds(0)
        two.set_rounds(1, k)
        img1 = gs.Buffer(dev, size=w * h * 16)
        img2 = gs.Buffer(dev, data=np.full(w * h * 4, np.float32(-3.0)))
        moving = rng.random() < 0.5
        for f in range(5):
            eye = (0.0, 0.0, 0.0)
            target = (0.0, 0.0, -1.0)
            if moving:
                eye = tuple(float(x) for x in rng.normal(scale=1.5, size=3))
                target = tuple(float(x) for x in (rng.normal(scale=0.4, size=3) + np.array([0, 0, -4.0])))
            cam = gs.camera_look_at(eye, target, (0, 1, 0), float(np.deg2rad(rng.choice([40, 60, 90]))) if f == 0 or moving else
                                    float(np.deg2rad(60)), w, h, 0.1, 100.0)
            one.render(stream, buf, gt, mt, cam, img1.device_ptr(), band=band)
            two.render(stream, buf, gt, mt, cam, img2.device_ptr(), band=band)
            a = img1.download(stream, np.uint32).reshape(h, w, 4)
            b = img2.download(stream, np.uint32).reshape(h, w, 4)
            rows = slice(0, h) if band is None else slice(band[0] * 16, min(band[1] * 16, h))
            f1, f2 = one.wait_frame(), two.wait_frame()
            si = two.sort_info()
            if not np.array_equal(a[rows], b[rows]) or f1.visible != f2.visible or f2.pairs > f1.pairs:
                print("MISMATCH seed %d frame %d: n %d %dx%d band %s mode %d pod %d/%d K %d rounds %d round1 %d tiles_done %d V %d/%d D %d/%d" % (
                    seed, f, n, w, h, band, mode, sh, cov, k, si.rounds, si.round1, si.tiles_done, f1.visible, f2.visible, f1.pairs, f2.pairs))
                return 1
            frames += 1
            dropped += f1.pairs - f2.pairs
        print("seed %d ok: n %d %dx%d band %s mode %d K %d rounds %d tiles_done %d pairs %d -> %d%s" % (
            seed, n, w, h, band, mode, k, si.rounds, si.tiles_done, f1.pairs, f2.pairs, " moving" if moving else ""), flush=True)
        for x in (one, two):
            x.destroy()
        buf.destroy()
    print("soak_rounds OK: %d frames, %d pairs dropped in all" % (frames, dropped))
    return 0


if __name__ == "__main__":
    sys.exit(main())
