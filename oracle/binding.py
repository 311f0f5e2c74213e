"""ctypes binding of the CPU oracle (oracle/_build/libgs_oracle.so).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgs_oracle.so")

SH_SINGLE, SH_HALF, SH_NORM8, SH_NONE = 0, 1, 2, 3
COV_ROT_SCALE, COV_SINGLE, COV_HALF = 0, 1, 2
SH_NAMES = ["single", "half", "norm8", "none"]
COV_NAMES = ["rot_scale", "single", "half"]

GAUSSIAN_DTYPE = np.dtype([("rot", "<f4", 4), ("pos", "<f4", 3), ("color", "u1", 4),
                           ("sh", "<f4", 45), ("scale", "<f4", 3)])
assert GAUSSIAN_DTYPE.itemsize == 224
PROJECTED_DTYPE = np.dtype([("mx", "<f4"), ("my", "<f4"), ("ca", "<f4"), ("cb", "<f4"),
                            ("cc", "<f4"), ("opacity", "<f4"), ("r", "<f4"), ("g", "<f4"),
                            ("b", "<f4"), ("depth", "<f4"), ("tx0", "<u2"), ("ty0", "<u2"),
                            ("tx1", "<u2"), ("ty1", "<u2")])
assert PROJECTED_DTYPE.itemsize == 48
PLY_DTYPE = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3), ("color", "<f4", 3),
                      ("sh", "<f4", 45), ("alpha", "<f4"), ("scale", "<f4", 3), ("rot", "<f4", 4)])
assert PLY_DTYPE.itemsize == 248


class GaussianTransform(C.Structure):
    _fields_ = [("size", C.c_float), ("flags", C.c_uint8 * 4)]

    @property
    def flags_u32(self):
        return int.from_bytes(bytes(self.flags), "little")


class ModelTransform(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("_pad0", C.c_float), ("rot", C.c_float * 4),
                ("scale", C.c_float * 3), ("_pad1", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("pos", C.c_float * 3), ("fx", C.c_float),
                ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("near_plane", C.c_float), ("far_plane", C.c_float), ("width", C.c_uint32),
                ("height", C.c_uint32), ("background", C.c_float * 3)]


def _stale():
    return not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                          for f in ("gs_oracle.c", "gs_oracle.h")))


def build(force=False):
    """make the oracle library if needed (serialised across processes with a file lock)"""
    if force or _stale():
        import fcntl
        os.makedirs(os.path.dirname(_LIB_PATH), exist_ok=True)
        with open(os.path.join(os.path.dirname(_LIB_PATH), ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or _stale():
                subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                               stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, sz, u32, i32, f32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_float
    L.gso_pod_size.restype = sz
    L.gso_pod_size.argtypes = [i32, i32]
    L.gso_sh_bytes.restype = sz
    L.gso_cov_bytes.restype = sz
    L.gso_pod_features.argtypes = [i32, i32, vp]
    L.gso_pack.argtypes = [i32, i32, vp, sz, vp]
    L.gso_unpack_to_gaussian.argtypes = [i32, i32, vp, sz, vp]
    L.gso_unpack_color.argtypes = [vp, vp]
    L.gso_unpack_sh.argtypes = [i32, vp, u32, vp]
    L.gso_unpack_cov3d.argtypes = [i32, i32, vp, vp]
    L.gso_shader_test_gaussian.argtypes = [i32, i32, vp, vp]
    L.gso_max_std_dev_encode.argtypes = [f32, vp]
    L.gso_gaussian_transform_new.argtypes = [f32, u32, u32, i32, f32, vp]
    for n in ("gso_transform_display_mode", "gso_transform_sh_deg", "gso_transform_no_sh0"):
        getattr(L, n).restype = u32
        getattr(L, n).argtypes = [u32]
    L.gso_transform_max_std_dev.restype = f32
    L.gso_transform_max_std_dev.argtypes = [u32]
    L.gso_model_transform_new.argtypes = [vp, vp, vp, vp]
    for n in ("gso_model_transform_mat", "gso_model_transform_inv_sr_mat", "gso_model_scale_rot_mat"):
        getattr(L, n).argtypes = [vp, vp]
    L.gso_model_to_world.argtypes = [vp, vp, vp]
    L.gso_given_gaussian_with_seed.argtypes = [u32, vp]
    L.gso_gaussian_from_ply.argtypes = [vp, vp]
    L.gso_gaussian_to_ply.argtypes = [vp, vp]
    L.gso_gaussians_from_ply.argtypes = [vp, sz, vp]
    L.gso_read_inria_ply.restype = C.c_long
    L.gso_read_inria_ply.argtypes = [vp, sz, vp, sz]
    L.gso_spz_decode_raw.restype = C.c_long
    L.gso_spz_decode_raw.argtypes = [vp, sz, vp, sz]
    L.gso_spz_encode_raw.restype = C.c_long
    L.gso_spz_encode_raw.argtypes = [vp, sz, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp, sz]
    L.gso_dispatch_workgroups.restype = u32
    L.gso_dispatch_workgroups.argtypes = [u32, u32]
    L.gso_exp.restype = f32
    L.gso_exp.argtypes = [f32]
    L.gso_camera_look_at.argtypes = [vp, vp, vp, f32, u32, u32, f32, f32, vp]
    L.gso_preprocess.argtypes = [i32, i32, vp, sz, vp, vp, vp, u32, u32, vp, vp, vp]
    L.gso_build_keys.restype = C.c_uint64
    L.gso_build_keys.argtypes = [vp, vp, vp, sz, u32, vp, vp]
    L.gso_sort_pairs.argtypes = [vp, vp, C.c_uint64]
    L.gso_tile_ranges.argtypes = [vp, C.c_uint64, u32, vp]
    L.gso_blend.argtypes = [vp, vp, vp, vp, u32, u32, vp]
    L.gso_render.restype = C.c_uint64
    L.gso_render.argtypes = [i32, i32, vp, sz, vp, vp, vp, u32, u32, vp, vp]
    L.gso_last_stage_seconds.argtypes = [vp]
    L.gso_set_threads.argtypes = [i32]
    L.gso_get_max_threads.restype = i32
    L.gso_effective_threads.restype = i32
    L.gso_set_rect_version.argtypes = [i32]
    L.gso_rect_version.restype = i32
    # the product's A/B switch GS3D_RECT_V1=1 (spec version 1 of the tile rect) selects the matching oracle
    # the product's switches (tests run oracle and product under the same environment): GS3D_RECT_V1=1 keeps the
    # radius square, GS3D_TILE_MASKS=0 stops at version 3 (no exact tile test for small rects)
    L.gso_set_rect_version(1 if os.environ.get("GS3D_RECT_V1") == "1" else 3 if os.environ.get("GS3D_TILE_MASKS") == "0" else 4)
    _lib = L
    return L


def set_rect_version(v):
    """1 = radius square, 2 = clipped to the splat's visible box (default)"""
    lib().gso_set_rect_version(int(v))


def rect_version():
    return int(lib().gso_rect_version())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pod_size(sh, cov):
    return lib().gso_pod_size(sh, cov)


def pack(sh, cov, gaussians):
    g = np.ascontiguousarray(gaussians, dtype=GAUSSIAN_DTYPE)
    out = np.zeros(len(g) * pod_size(sh, cov), dtype=np.uint8)
    lib().gso_pack(sh, cov, _p(g), len(g), _p(out))
    return out


def unpack_to_gaussian(sh, cov, pods):
    pods = np.ascontiguousarray(pods, dtype=np.uint8)
    n = len(pods) // pod_size(sh, cov)
    out = np.zeros(n, dtype=GAUSSIAN_DTYPE)
    rc = lib().gso_unpack_to_gaussian(sh, cov, _p(pods), n, _p(out))
    return rc, out


def gaussians_from_ply(ply):
    """Gaussian::from_ply over an array of PlyGaussianPod (libm expf: the reference's f32::exp on Linux)"""
    p = np.ascontiguousarray(np.atleast_1d(ply), dtype=PLY_DTYPE)
    out = np.zeros(len(p), dtype=GAUSSIAN_DTYPE)
    lib().gso_gaussians_from_ply(_p(p), len(p), _p(out))
    return out


def shader_test_gaussian(sh, cov, pod_bytes):
    pod = np.ascontiguousarray(pod_bytes, dtype=np.uint8)
    out = np.zeros(56, dtype=np.float32)
    lib().gso_shader_test_gaussian(sh, cov, _p(pod), _p(out))
    return out


def given_gaussians(seeds):
    out = np.zeros(len(seeds), dtype=GAUSSIAN_DTYPE)
    for i, s in enumerate(seeds):
        lib().gso_given_gaussian_with_seed(int(s), C.c_void_p(out[i:i + 1].ctypes.data))
    return out


def gaussian_transform(size=1.0, mode=0, sh_deg=3, no_sh0=False, max_std_dev=3.0):
    gt = GaussianTransform()
    rc = lib().gso_gaussian_transform_new(size, mode, sh_deg, int(no_sh0), max_std_dev, C.byref(gt))
    if rc:
        raise ValueError("invalid gaussian transform")
    return gt


def model_transform(pos=(0, 0, 0), rot=(0, 0, 0, 1), scale=(1, 1, 1)):
    mt = ModelTransform()
    lib().gso_model_transform_new(_p(np.asarray(pos, np.float32)), _p(np.asarray(rot, np.float32)),
                                  _p(np.asarray(scale, np.float32)), C.byref(mt))
    return mt


def camera_look_at(eye, target, up, vfov_rad, width, height, near=0.1, far=100.0):
    cam = Camera()
    lib().gso_camera_look_at(_p(np.asarray(eye, np.float32)), _p(np.asarray(target, np.float32)),
                             _p(np.asarray(up, np.float32)), vfov_rad, width, height, near, far,
                             C.byref(cam))
    return cam


class TilesTouched(np.ndarray):
    """tiles touched per Gaussian (uint32) + `.rows`: the tile row codes of rect version 4 (uint16 per Gaussian, 0 =
    the whole rect), which build_keys needs to enumerate the tiles that remain of a small rect"""

    def __array_finalize__(self, obj):
        self.rows = getattr(obj, "rows", None)


def preprocess(sh, cov, pods, gt, mt, cam, band=None):
    pods = np.ascontiguousarray(pods, dtype=np.uint8)
    n = len(pods) // pod_size(sh, cov)
    tiles_y = (cam.height + 15) // 16
    b0, b1 = band if band is not None else (0, tiles_y)
    proj = np.zeros(n, dtype=PROJECTED_DTYPE)
    tiles = np.zeros(n, dtype=np.uint32).view(TilesTouched)
    tiles.rows = np.zeros(n, dtype=np.uint16)
    lib().gso_preprocess(sh, cov, _p(pods), n, C.byref(gt), C.byref(mt), C.byref(cam), b0, b1,
                         _p(proj), _p(tiles), _p(tiles.rows))
    return proj, tiles


def spatial_order(sh, cov, pods):
    """slot -> Gaussian index of the spatially ordered mirror"""
    pods = np.ascontiguousarray(pods, dtype=np.uint8)
    n = len(pods.reshape(-1)) // pod_size(sh, cov)
    order = np.zeros(n, dtype=np.uint32)
    lib().gso_spatial_order.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    lib().gso_spatial_order.restype = None
    lib().gso_spatial_order(_p(pods), n, pod_size(sh, cov), _p(order))
    return order


def build_keys(proj, tiles, tiles_x, order=None):
    d = int(tiles.astype(np.uint64).sum())
    keys = np.zeros(max(d, 1), dtype=np.uint64)
    idx = np.zeros(max(d, 1), dtype=np.uint32)
    fn = lib().gso_build_keys_ordered
    fn.restype = C.c_uint64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
        assert len(order) == len(proj)
    rows = getattr(tiles, "rows", None)
    if rows is not None:
        rows = np.ascontiguousarray(rows, dtype=np.uint16)
        assert len(rows) == len(proj)
    tiles = np.ascontiguousarray(tiles, dtype=np.uint32)
    d2 = fn(_p(proj), _p(tiles), _p(rows) if rows is not None else None, len(proj), tiles_x, _p(keys), _p(idx),
            _p(order) if order is not None else None)
    assert d2 == d
    return keys[:d], idx[:d]


def sort_pairs(keys, idx):
    k = np.ascontiguousarray(keys.copy())
    i = np.ascontiguousarray(idx.copy())
    lib().gso_sort_pairs(_p(k), _p(i), len(k))
    return k, i


def tile_ranges(keys, num_tiles):
    r = np.zeros((num_tiles, 2), dtype=np.uint32)
    k = np.ascontiguousarray(keys)
    lib().gso_tile_ranges(_p(k), len(k), num_tiles, _p(r))
    return r


def blend(proj, idx, ranges, cam, band=None, gt=None):
    """gt: the GaussianTransform whose display mode / max_std_dev apply (None = splat, 3 sigma)"""
    tiles_y = (cam.height + 15) // 16
    b0, b1 = band if band is not None else (0, tiles_y)
    rgba = np.zeros((cam.height, cam.width, 4), dtype=np.float32)
    idx = np.ascontiguousarray(idx)
    mode, k = 0, 3.0
    if gt is not None:
        mode = int(gt.flags[0])
        lib().gso_transform_max_std_dev.restype = C.c_float
        lib().gso_transform_max_std_dev.argtypes = [C.c_uint32]
        k = lib().gso_transform_max_std_dev(gt.flags_u32)
    fn = lib().gso_blend_mode
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                   C.c_float]
    fn(_p(proj), _p(idx), _p(ranges), C.byref(cam), b0, b1, _p(rgba), mode, k)
    return rgba


def render(sh, cov, pods, gt, mt, cam, band=None, want_image=True, order=None):
    pods = np.ascontiguousarray(pods, dtype=np.uint8)
    n = len(pods) // pod_size(sh, cov)
    tiles_y = (cam.height + 15) // 16
    b0, b1 = band if band is not None else (0, tiles_y)
    rgba = np.zeros((cam.height, cam.width, 4), dtype=np.float32) if want_image else None
    vis = C.c_uint64(0)
    fn = lib().gso_render_ordered
    fn.restype = C.c_uint64
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                   C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
        assert len(order) == n
    d = fn(sh, cov, _p(pods), n, C.byref(gt), C.byref(mt), C.byref(cam), b0, b1,
           _p(rgba) if want_image else None, C.byref(vis), _p(order) if order is not None else None)
    st = (C.c_double * 5)()
    lib().gso_last_stage_seconds(st)
    return rgba, int(d), int(vis.value), list(st)


def spz_decode_raw(data):
    """decompressed SPZ payload -> Gaussians (negative return codes raise ValueError)"""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    n = lib().gso_spz_decode_raw(_p(buf), buf.size, None, 0)
    if n < 0:
        raise ValueError(n)
    out = np.zeros(n, dtype=GAUSSIAN_DTYPE)
    rc = lib().gso_spz_decode_raw(_p(buf), buf.size, _p(out), n)
    if rc < 0:
        raise ValueError(rc)
    return out


def spz_encode_raw(gaussians, version=3, sh_degree=3, fractional_bits=12, antialiased=False,
                   sh_quantize_bits=(5, 4, 4)):
    g = np.ascontiguousarray(gaussians, dtype=GAUSSIAN_DTYPE)
    bits = np.array(sh_quantize_bits, dtype=np.uint32)
    n = lib().gso_spz_encode_raw(_p(g), len(g), version, sh_degree, fractional_bits, int(antialiased),
                                 _p(bits), None, 0)
    if n < 0:
        raise ValueError(n)
    out = np.zeros(n, dtype=np.uint8)
    rc = lib().gso_spz_encode_raw(_p(g), len(g), version, sh_degree, fractional_bits, int(antialiased),
                                  _p(bits), _p(out), n)
    if rc < 0:
        raise ValueError(rc)
    return out.tobytes()
