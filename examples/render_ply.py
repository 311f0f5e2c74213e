#!/usr/bin/env python3
"""Render an Inria-style .ply (or an .spz) with the MI355X path and write a PPM — the counterpart of the
reference's examples/read_ply.rs / read_spz.rs followed by one frame of the viewer.
usage: python examples/render_ply.py tests/golden/model.ply out.ppm [--size 960x540] [--eye 0,0,4]
       [--mode splat|ellipse|point] [--pod ShHalf/Cov3dHalf]          (needs a GPU: there is no CPU fallback)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_3dgs_core_amd as gs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("out")
    ap.add_argument("--size", default="960x540")
    ap.add_argument("--eye", default="0,0,4")
    ap.add_argument("--target", default="0,0,0")
    ap.add_argument("--mode", default="splat", choices=["splat", "ellipse", "point"])
    ap.add_argument("--pod", default="ShSingle/Cov3dRotScale")
    args = ap.parse_args()
    sh, cov = args.pod.split("/")
    pod = getattr(gs, "GaussianPodWith%s%sConfigs" % (sh, cov))
    W, H = (int(v) for v in args.size.split("x"))
    dev = gs.Device(0)
    stream = dev.create_stream()
    # the device-side load path: the file's records cross PCIe as they are; from_ply / from_spz and the
    # pack to the POD layout run in one kernel (gs_gaussians_buffer_create_from_ply / _from_spz)
    if args.scene.endswith(".spz"):
        buf = gs.GaussiansBuffer.new_from_spz(dev, pod, open(args.scene, "rb").read())
    else:
        buf = gs.GaussiansBuffer.new_from_ply(dev, pod, gs.PlyGaussians.read_from_file(args.scene))
    img = gs.Buffer(dev, size=W * H * 16)
    cam = gs.camera_look_at(tuple(float(v) for v in args.eye.split(",")), tuple(float(v) for v in args.target.split(",")),
                            (0, 1, 0), float(np.deg2rad(60.0)), W, H)
    mode = {"splat": gs.DISPLAY_SPLAT, "ellipse": gs.DISPLAY_ELLIPSE, "point": gs.DISPLAY_POINT}[args.mode]
    r = gs.Renderer(dev)
    r.render(stream, buf, gs.gaussian_transform_pod(1.0, mode, 3, False, 3.0), gs.model_transform_pod(), cam,
             img.device_ptr())
    rgba = img.download(stream, np.float32).reshape(H, W, 4)
    st = r.stats()
    rgb8 = (np.clip(rgba[..., :3], 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    with open(args.out, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (W, H))
        f.write(rgb8.tobytes())
    print("%d Gaussians, %d visible, %d (tile, Gaussian) pairs -> %s" % (len(buf), st.visible, st.pairs, args.out))


if __name__ == "__main__":
    main()
