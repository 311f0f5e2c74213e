"""One HIP runtime per process, whatever the import order.

The cause of round 2's `RuntimeError: No HIP GPUs are available` (DESIGN.md §5 "One HIP runtime"):
libgs3d_hip.so NEEDs `libamdhip64.so.7` / `libhiprtc.so.7` (RUNPATH /opt/rocm-*/lib), while the
PyTorch-ROCm wheel ships its OWN copy of the runtime in torch/lib (`libamdhip64.so`,
`libhsa-runtime64.so`, `libhiprtc.so`, `libamd_comgr.so`, `librccl.so`; SONAMEs `libamdhip64.so.7`
...) and its libraries NEED the unversioned names with RPATH $ORIGIN.

* torch first: the product's request for `libamdhip64.so.7` is satisfied by the SONAME of torch's
  already-mapped copy -> one runtime (torch's).
* product first: /opt/rocm's runtime is mapped; torch's request for `libamdhip64.so` matches neither
  the path nor the SONAME of anything loaded, RPATH finds torch/lib/libamdhip64.so, a different
  file -> a SECOND HIP + HSA runtime in the process, and the second one finds no GPU.

torch cannot be pointed at the system runtime (its RPATH wins over every search path), so the two
can only meet on torch's copy.  `prepare()` therefore maps torch's runtime FIRST whenever torch is
installed in this interpreter (without importing torch), so that both import orders end on the same
runtime; `check()` refuses to go on when two runtimes are mapped anyway.  Without torch (a C++ or
Rust host) the product runs on the runtime its RUNPATH names.

GS3D_HIP_RUNTIME = auto (default) | system (never preload torch's copy) | torch (require it).
"""
import ctypes
import importlib.util
import os

_FAMILIES = ("libamdhip64", "libhsa-runtime64", "libhiprtc")
_state = {"source": None, "preloaded": []}


def mapped():
    """{family: sorted real paths of the copies of that library mapped into this process}"""
    found = {f: set() for f in _FAMILIES}
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                parts = line.split(None, 5)
                if len(parts) < 6:
                    continue
                path = parts[5].strip()
                base = os.path.basename(path)
                for f in _FAMILIES:
                    if base.startswith(f + ".so"):
                        found[f].add(os.path.realpath(path))
    except OSError:
        pass
    return {f: sorted(v) for f, v in found.items()}


def torch_lib_dir():
    """torch/lib of the torch installed in this interpreter, or None; does NOT import torch."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.submodule_search_locations:
        return None
    d = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    return d if os.path.exists(os.path.join(d, "libamdhip64.so")) else None


def prepare():
    """Call BEFORE dlopen(libgs3d_hip.so).  Returns the runtime the product will bind to:
    'already-mapped' | 'torch' | 'system'."""
    mode = os.environ.get("GS3D_HIP_RUNTIME", "auto").lower()
    if mode not in ("auto", "system", "torch"):
        raise ImportError("GS3D_HIP_RUNTIME must be auto, system or torch (got %r)" % mode)
    if mapped()["libamdhip64"]:
        _state["source"] = "already-mapped"          # the loader reuses it through its SONAME
        return _state["source"]
    d = torch_lib_dir() if mode != "system" else None
    if d is None:
        if mode == "torch":
            raise ImportError("GS3D_HIP_RUNTIME=torch, but no torch with a bundled libamdhip64.so is installed")
        _state["source"] = "system"
        return _state["source"]
    # libamdhip64 pulls libhsa-runtime64 / libamd_comgr from the same directory (RPATH $ORIGIN);
    # RTLD_GLOBAL so that later NEEDED entries (SONAME libamdhip64.so.7, libhiprtc.so.7) bind to these
    for name in ("libamdhip64.so", "libhiprtc.so"):
        p = os.path.join(d, name)
        if os.path.exists(p):
            ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
            _state["preloaded"].append(p)
    _state["source"] = "torch"
    return _state["source"]


def check(where="libgs3d_hip.so"):
    """Raise ImportError when more than one HIP runtime (libamdhip64) is mapped: the second one finds
    no GPU and every call through it fails in confusing ways.  A second libhsa-runtime64 / libhiprtc
    alone is reported by info() but tolerated: rocprofv3's tool library links /opt/rocm's HSA runtime
    next to torch's, and only the copy the (single) HIP runtime initialises ever talks to the GPU."""
    m = mapped()
    dup = {f: v for f, v in m.items() if len(v) > 1 and f == "libamdhip64"}
    if dup:
        raise ImportError(
            "two HIP runtimes are mapped into this process after loading %s: %s.  libgs3d_hip.so and "
            "PyTorch must share one runtime: import wgpu_3dgs_core_amd with GS3D_HIP_RUNTIME=auto (the "
            "default) or import torch first (wgpu-3dgs-core_amd/_hiprt.py)." % (where, dup))
    return m


def info():
    """What the product is bound to (for bench.py / DESIGN): source + the mapped paths."""
    return {"source": _state["source"], "preloaded": list(_state["preloaded"]), "mapped": mapped()}
