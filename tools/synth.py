"""Python front-end of tools/gs_synth.c (deterministic synthetic scenes)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libgs_synth.so")
SEED = 0x3D650001

GAUSSIAN_DTYPE = np.dtype([("rot", "<f4", 4), ("pos", "<f4", 3), ("color", "u1", 4),
                           ("sh", "<f4", 45), ("scale", "<f4", 3)])


def build(force=False):
    """make the generator library if needed (serialised across processes with a file lock)"""
    src = os.path.join(_HERE, "gs_synth.c")

    def stale():
        return not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src)

    if force or stale():
        import fcntl
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        with open(os.path.join(os.path.dirname(_LIB), ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or stale():
                subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB


_lib = None
_threads = None


def set_threads(n):
    """OpenMP threads of the generator (torchrun exports OMP_NUM_THREADS=1 to every rank)."""
    global _threads
    _threads = int(n)
    if _lib is not None:
        _lib.gs_synth_set_threads(_threads)


def scene(count, first=0, seed=SEED):
    """Gaussians [first, first+count) of the synthetic scene as a structured array."""
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.gs_synth_scene.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        _lib.gs_synth_set_threads.argtypes = [C.c_int]
        if _threads is not None:
            _lib.gs_synth_set_threads(_threads)
    out = np.zeros(count, dtype=GAUSSIAN_DTYPE)
    _lib.gs_synth_scene(seed, first, count, out.ctypes.data_as(C.c_void_p))
    return out


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def scene_numpy(count, first=0, seed=SEED):
    """numpy twin of gs_synth_scene (cross-check; may differ by 1 ulp where libm differs)."""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + count, dtype=np.uint64)

        def uni(k):
            h = _splitmix64(np.uint64(seed) ^ (i * np.uint64(64) + np.uint64(k)))
            return (h >> np.uint64(40)).astype(np.float64) / 16777216.0
        out = np.zeros(count, dtype=GAUSSIAN_DTYPE)
        out["pos"][:, 0] = -14.0 + 28.0 * uni(0)
        out["pos"][:, 1] = -8.0 + 16.0 * uni(1)
        out["pos"][:, 2] = -(2.0 + 24.0 * uni(2))
        z = np.zeros((count, 8))
        for p in range(4):
            u1 = 1.0 - uni(3 + 2 * p)
            u2 = uni(4 + 2 * p)
            r = np.sqrt(-2.0 * np.log(u1))
            z[:, 2 * p] = r * np.cos(6.283185307179586 * u2)
            z[:, 2 * p + 1] = r * np.sin(6.283185307179586 * u2)
        ln = np.sqrt((z[:, :4] ** 2).sum(axis=1))
        out["rot"] = (z[:, :4] / ln[:, None])
        out["scale"] = np.exp(-3.6 + 0.5 * z[:, 4:7])
        for c in range(3):
            out["color"][:, c] = (256.0 * uni(11 + c)).astype(np.uint8)
        out["color"][:, 3] = (32.0 + np.floor(224.0 * uni(14))).astype(np.uint8)
        for c in range(45):
            out["sh"][:, c] = -0.25 + 0.5 * uni(16 + c)
    return out
