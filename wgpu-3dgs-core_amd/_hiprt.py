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

Before preloading, the SONAME of torch's copy is compared with what libgs3d_hip.so NEEDs: a wheel
whose bundled runtime has another SONAME major could not satisfy the product's request anyway (the
loader would map /opt/rocm's copy NEXT to it), so auto mode then leaves torch's copy alone and the
product runs on the system runtime — importable and usable by itself; only a process that ALSO
imports torch is refused (by check(), with the way out in the message).

GS3D_HIP_RUNTIME = auto (default) | system (never preload torch's copy) | torch (require it).
"""
import ctypes
import importlib.util
import os
import struct
import sys

_FAMILIES = ("libamdhip64", "libhsa-runtime64", "libhiprtc")
_state = {"source": None, "preloaded": [], "note": None}


def elf_dynamic(path):
    """(SONAME or None, [NEEDED ...]) of a 64-bit little-endian ELF shared object; (None, []) when the
    file cannot be parsed.  Pure Python: no readelf on the GPU box is assumed."""
    try:
        with open(path, "rb") as fh:
            data = fh.read()
        if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
            return None, []
        e_phoff, = struct.unpack_from("<Q", data, 0x20)
        e_phentsize, e_phnum = struct.unpack_from("<HH", data, 0x36)
        loads, dyn = [], None
        for i in range(e_phnum):
            p_type, _flags, p_offset, p_vaddr, _pa, p_filesz = struct.unpack_from("<IIQQQQ", data, e_phoff + i * e_phentsize)
            if p_type == 1:
                loads.append((p_vaddr, p_offset, p_filesz))
            elif p_type == 2:
                dyn = (p_offset, p_filesz)
        if dyn is None:
            return None, []

        def file_off(vaddr):
            for va, off, size in loads:
                if va <= vaddr < va + size:
                    return off + (vaddr - va)
            return None

        entries = [struct.unpack_from("<qQ", data, dyn[0] + k) for k in range(0, dyn[1], 16)]
        strtab = next((file_off(v) for t, v in entries if t == 5), None)
        if strtab is None:
            return None, []

        def cstr(o):
            end = data.index(b"\0", strtab + o)
            return data[strtab + o:end].decode("ascii", "replace")

        soname = next((cstr(v) for t, v in entries if t == 14), None)
        return soname, [cstr(v) for t, v in entries if t == 1]
    except (OSError, ValueError, struct.error, IndexError):
        return None, []


def product_needs(family="libamdhip64"):
    """the NEEDED entry of libgs3d_hip.so for `family` (e.g. 'libamdhip64.so.7'), or None"""
    _, needed = elf_dynamic(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgs3d_hip.so"))
    return next((n for n in needed if n.startswith(family + ".so")), None)


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
    if d is not None:
        # can torch's copy satisfy the product's request at all?  (same SONAME, e.g. libamdhip64.so.7)
        soname, _ = elf_dynamic(os.path.join(d, "libamdhip64.so"))
        want = product_needs()
        if soname and want and soname != want:
            _state["note"] = ("torch bundles %s but libgs3d_hip.so needs %s: torch's runtime is left alone and the "
                              "product runs on the system runtime (do not import torch in this process)" % (soname, want))
            if mode == "torch":
                raise ImportError("GS3D_HIP_RUNTIME=torch: " + _state["note"])
            d = None
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
    if dup and "torch" not in sys.modules and not where.startswith("torch"):
        # a second copy is mapped but nothing has initialised it (torch is not imported): the product is
        # bound to ONE runtime and works; only a process that goes on to use torch must be stopped
        _state["note"] = "two libamdhip64 mapped, torch not imported: %s" % (dup,)
        return m
    if dup:
        raise ImportError(
            "two HIP runtimes are mapped into this process after loading %s: %s.  libgs3d_hip.so and "
            "PyTorch must share one runtime.  Ways out: (a) default GS3D_HIP_RUNTIME=auto and a torch wheel "
            "whose bundled libamdhip64 has the SONAME libgs3d_hip.so needs (%s) — then either import order "
            "works; (b) a process that does not need torch: GS3D_HIP_RUNTIME=system and no `import torch`; "
            "(c) rebuild libgs3d_hip.so against the ROCm of the torch wheel (wgpu-3dgs-core_amd/_hiprt.py)."
            % (where, dup, product_needs()))
    return m


def info():
    """What the product is bound to (for bench.py / DESIGN): source, the mapped paths, the SONAME the
    product asks for, and a note when auto mode had to step aside.  (The numeric versions — HIP_VERSION
    of the headers, hipRuntimeGetVersion of the runtime in use — come from the library itself:
    wgpu_3dgs_core_amd.hip_versions().)"""
    return {"source": _state["source"], "preloaded": list(_state["preloaded"]), "mapped": mapped(),
            "product_needs": product_needs(), "note": _state["note"]}
