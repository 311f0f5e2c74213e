"""Shared helpers for the parity tests (scene construction identical for oracle and HIP path)."""
import numpy as np

import synth


def default_camera(mod, width=1920, height=1080, eye=(0, 0, 0), target=(0, 0, -1), vfov_deg=60.0,
                   near=0.1, far=100.0):
    """mod is either the oracle binding or the product package (same look-at helper on both)."""
    return mod.camera_look_at(eye, target, (0, 1, 0), float(np.deg2rad(vfov_deg)), width, height,
                              near, far)


def copy_camera(src, dst_cls):
    """Copy a camera struct field by field between the oracle's and the product's ctypes types so
    that both paths see bit-identical uniforms."""
    dst = dst_cls()
    for name, _ in src._fields_:
        v = getattr(src, name)
        if hasattr(v, "__len__"):
            getattr(dst, name)[:] = list(v)
        else:
            setattr(dst, name, v)
    return dst


def synth_pods(ob, sh, cov, n, first=0):
    g = synth.scene(n, first=first)
    return g, ob.pack(sh, cov, g)


def needle_gaussian(mod, theta_deg, sigma_px, opacity_byte, center_px, width, height, z=4.0, thin=1e-4):
    """One splat that projects to a needle: standard deviation `sigma_px` pixels along a direction
    `theta_deg` from the image x axis, far thinner than a pixel across (the +0.3 dilation of DESIGN.md
    §3.3 then sets the width), centred on `center_px` — cond(cov2d) = sigma_px^2 / 0.3.  Returns
    (gaussians[1], camera of `mod`)."""
    cam = default_camera(mod, width, height)
    g = synth.scene(1)
    g["sh"][:] = 0
    g["color"][0] = (200, 120, 40, opacity_byte)
    mx, my = center_px
    g["pos"][0] = ((mx - cam.cx) / cam.fx * z, -(my - cam.cy) / cam.fy * z, -z)
    g["scale"][0] = (sigma_px * z / cam.fx, thin, thin)
    t = np.deg2rad(theta_deg) / 2
    g["rot"][0] = (0.0, 0.0, np.sin(t), np.cos(t))
    return g, cam
