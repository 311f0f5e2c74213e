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
