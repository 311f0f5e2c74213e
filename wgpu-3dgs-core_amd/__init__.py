"""wgpu_3dgs_core_amd — host-side mirror of LioQing/wgpu-3dgs-core's API over the MI355X C ABI.

The product is libgs3d_hip.so (include/gs3d.h); this package is the thin Python host layer that
tests, bench.py and torch.distributed plumbing use.  Names, argument meaning and error behaviour
follow the reference (citations relative to the reference repository):

    Gaussian / GaussianPod configs ...... src/gaussian.rs:53-60, src/buffer/gaussian.rs:239-384
    GaussiansBuffer ..................... src/buffer/gaussian.rs:17-229
    GaussianTransformBuffer / Pod ....... src/buffer/gaussian_transform.rs
    ModelTransformBuffer / Pod .......... src/buffer/model_transform.rs
    BufferWrapper.download .............. src/buffer/mod.rs:17-102
    ComputeBundle / ComputeBundleBuilder  src/compute_bundle.rs
    error enums ......................... src/error.rs:55-143

There is no CPU implementation behind any device call: constructing a Device without a HIP GPU
raises NoDeviceError, and importing the package without libgs3d_hip.so raises ImportError.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import Camera, FrameResult, FrameStats, GaussianTransformPod, Limits, ModelTransformPod, SortInfo

_L = _capi.load()

# ------------------------------------------------------------------------------------------------
# data model
# ------------------------------------------------------------------------------------------------

#: struct Gaussian — src/gaussian.rs:53-60
GAUSSIAN_DTYPE = np.dtype([("rot", "<f4", 4), ("pos", "<f4", 3), ("color", "u1", 4),
                           ("sh", "<f4", 45), ("scale", "<f4", 3)])
#: projected splat record (include/gs3d.h gs_projected)
PROJECTED_DTYPE = np.dtype([("mx", "<f4"), ("my", "<f4"), ("ca", "<f4"), ("cb", "<f4"),
                            ("cc", "<f4"), ("opacity", "<f4"), ("r", "<f4"), ("g", "<f4"),
                            ("b", "<f4"), ("depth", "<f4"), ("tx0", "<u2"), ("ty0", "<u2"),
                            ("tx1", "<u2"), ("ty1", "<u2")])

SH_SINGLE, SH_HALF, SH_NORM8, SH_NONE = 0, 1, 2, 3
COV3D_ROT_SCALE, COV3D_SINGLE, COV3D_HALF = 0, 1, 2
SH_CONFIG_NAMES = ["Single", "Half", "Norm8", "None"]
COV3D_CONFIG_NAMES = ["RotScale", "Single", "Half"]
FEATURE_NAMES = [_L.gs_feature_name(i).decode() for i in range(7)]

DISPLAY_SPLAT, DISPLAY_ELLIPSE, DISPLAY_POINT = 0, 1, 2

KERNEL_ARRAY_MAP_ADD = 0
KERNEL_TEST_GAUSSIAN = 1
KERNEL_TEST_GAUSSIAN_TRANSFORM = 2
KERNEL_TEST_MODEL_TRANSFORM = 3
KERNEL_UNPACK_SOA = 4


class GsError(Exception):
    """Base of every error raised by the C ABI; carries the variant fields of src/error.rs."""
    code = None

    def __init__(self, info):
        self.status = info.code
        self.a, self.b, self.c = info.a, info.b, info.c
        super().__init__(info.message.decode(errors="replace"))


class InvalidArgumentError(GsError): code = -1
class NoDeviceError(GsError): code = -2
class HipError(GsError): code = -3
class OutOfMemoryError(GsError): code = -4


class GaussiansBufferUpdateError(GsError):
    """CountMismatch{count, expected_count} — src/error.rs:66-70"""
    code = -10
    count = property(lambda s: s.a)
    expected_count = property(lambda s: s.b)


class GaussiansBufferUpdateRangeError(GsError):
    """CountMismatch{count, start, expected_count} — src/error.rs:73-81"""
    code = -11
    count = property(lambda s: s.a)
    start = property(lambda s: s.b)
    expected_count = property(lambda s: s.c)


class GaussiansBufferTryFromBufferError(GsError):
    """BufferSizeNotMultiple{buffer_size, expected_multiple_size} — src/error.rs:85-94"""
    code = -12
    buffer_size = property(lambda s: s.a)
    expected_multiple_size = property(lambda s: s.b)


class FixedSizeBufferWrapperError(GsError):
    """BufferSizeMismatched{buffer_size, expected_size} — src/error.rs:97-104"""
    code = -13
    buffer_size = property(lambda s: s.a)
    expected_size = property(lambda s: s.b)


class ComputeBundleCreateError(GsError):
    """ResourceCountMismatch / WorkgroupSizeExceedsDeviceLimit — src/error.rs:107-126"""


class ResourceCountMismatch(ComputeBundleCreateError):
    code = -14
    resource_count = property(lambda s: s.a)
    bind_group_layout_count = property(lambda s: s.b)


class WorkgroupSizeExceedsDeviceLimit(ComputeBundleCreateError):
    code = -15
    workgroup_size = property(lambda s: s.a)
    device_limit = property(lambda s: s.b)


class ComputeBundleBuildError(Exception):
    """src/error.rs:129-143; raised by ComputeBundleBuilder.build in the reference's order."""


class MissingBindGroupLayout(ComputeBundleBuildError): pass
class MissingResolver(ComputeBundleBuildError): pass
class MissingEntryPoint(ComputeBundleBuildError): pass
class MissingMainShader(ComputeBundleBuildError): pass
class KernelResolveError(ComputeBundleBuildError): """ComputeBundleBuildError::Wesl analogue"""


class LossyConfigError(GsError): code = -21
class SpzError(GsError):
    """std::io::Error of the SPZ reader / header validation"""
    code = -25


class PlyError(GsError):
    """std::io::Error of PlyGaussians::read_from (message = the reference's message)"""
    code = -24
class DownloadBufferError(GsError): code = -22
class PairOverflowError(GsError):
    """a frame needs more (tile, Gaussian) pairs than 32-bit pair indices can address"""
    code = -23


class PairCapacityError(GsError):
    """gs_renderer_wait_frame: the frame produced more pairs than the renderer's buffers hold and was
    SKIPPED (its image was not written).  The next frame grows the buffers: render again."""
    code = -26

    def __init__(self, info):
        super().__init__(info)
        self.pairs, self.capacity = info.a, info.b


class RankOrderError(GsError):
    """gs_renderer_wait_frame: the watchdog of the radix sort's LDS-atomic rank fired in this frame (its blend
    order may be wrong).  The device has been switched to the ballot-based rank: render again."""
    code = -27


_ERRORS = {c.code: c for c in (InvalidArgumentError, NoDeviceError, HipError, OutOfMemoryError,
                               GaussiansBufferUpdateError, GaussiansBufferUpdateRangeError,
                               GaussiansBufferTryFromBufferError, FixedSizeBufferWrapperError,
                               ResourceCountMismatch, WorkgroupSizeExceedsDeviceLimit,
                               LossyConfigError, DownloadBufferError, PlyError, SpzError, PairOverflowError,
                               PairCapacityError, RankOrderError)}


def _check(status):
    if status == 0:
        return
    info = _capi.ErrorInfo()
    _L.gs_last_error(C.byref(info))
    if info.code != status:
        info.code = status
        info.message = _L.gs_status_string(status)
    raise _ERRORS.get(status, GsError)(info)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class GaussianPod:
    """One of the 12 GaussianPodWithSh{S}Cov3d{C}Configs layouts (src/buffer/gaussian.rs:373-384)."""

    def __init__(self, sh, cov):
        self.sh, self.cov = int(sh), int(cov)
        self.name = "GaussianPodWithSh%sCov3d%sConfigs" % (SH_CONFIG_NAMES[sh], COV3D_CONFIG_NAMES[cov])

    @property
    def size(self):
        return _L.gs_pod_size(self.sh, self.cov)

    def features(self):
        """GaussianPod::features() — src/buffer/gaussian.rs:270-286"""
        out = (C.c_uint8 * 7)()
        _check(_L.gs_pod_features(self.sh, self.cov, out))
        return [(FEATURE_NAMES[i], bool(out[i])) for i in range(7)]

    def from_gaussian(self, gaussians):
        """G::from_gaussian over an array of Gaussians -> packed bytes (host)."""
        g = np.ascontiguousarray(np.atleast_1d(gaussians), dtype=GAUSSIAN_DTYPE)
        out = np.zeros(len(g) * self.size, dtype=np.uint8)
        _check(_L.gs_pack(self.sh, self.cov, _ptr(g), len(g), _ptr(out)))
        return out

    def into_gaussian(self, pods):
        """Into<Gaussian>; raises LossyConfigError where the reference panics."""
        pods = np.ascontiguousarray(pods, dtype=np.uint8)
        n = len(pods) // self.size
        out = np.zeros(n, dtype=GAUSSIAN_DTYPE)
        _check(_L.gs_unpack_to_gaussian(self.sh, self.cov, _ptr(pods), n, _ptr(out)))
        return out

    def __repr__(self):
        return self.name

    def __eq__(self, o):
        return isinstance(o, GaussianPod) and (self.sh, self.cov) == (o.sh, o.cov)

    def __hash__(self):
        return hash((self.sh, self.cov))


ALL_PODS = [GaussianPod(s, c) for s in range(4) for c in range(3)]
for _p in ALL_PODS:
    globals()[_p.name] = _p


class GaussianShDegree:
    """src/buffer/gaussian_transform.rs:17-52"""

    def __init__(self, v):
        self.v = int(v)

    @staticmethod
    def new(sh_deg):
        return GaussianShDegree(sh_deg) if 0 <= sh_deg <= 3 else None

    def get(self):
        return self.v


class GaussianMaxStdDev:
    """src/buffer/gaussian_transform.rs:55-98"""

    def __init__(self, u8):
        self.u8 = int(u8)

    @staticmethod
    def new(max_std_dev):
        out = C.c_uint8()
        if _L.gs_max_std_dev_encode(max_std_dev, C.byref(out)) != 0:
            return None
        return GaussianMaxStdDev(out.value)

    def get(self):
        return _L.gs_max_std_dev_decode(self.u8)

    def as_u8(self):
        return self.u8


def gaussian_transform_pod(size=1.0, display_mode=DISPLAY_SPLAT, sh_deg=3, no_sh0=False,
                           max_std_dev=3.0):
    """GaussianTransformPod::new — raises InvalidArgumentError where the Rust newtypes return None."""
    pod = GaussianTransformPod()
    _check(_L.gs_gaussian_transform_pod_new(size, display_mode, sh_deg, int(bool(no_sh0)),
                                            max_std_dev, C.byref(pod)))
    return pod


def model_transform_pod(pos=(0.0, 0.0, 0.0), rot=(0.0, 0.0, 0.0, 1.0), scale=(1.0, 1.0, 1.0)):
    """ModelTransformPod::new — src/buffer/model_transform.rs:68-77"""
    pod = ModelTransformPod()
    _L.gs_model_transform_pod_new(_ptr(np.asarray(pos, np.float32)), _ptr(np.asarray(rot, np.float32)),
                                  _ptr(np.asarray(scale, np.float32)), C.byref(pod))
    return pod


def camera_look_at(eye, target, up, vfov_radians, width, height, near=0.1, far=100.0,
                   background=(0.0, 0.0, 0.0)):
    cam = Camera()
    _L.gs_camera_look_at(_ptr(np.asarray(eye, np.float32)), _ptr(np.asarray(target, np.float32)),
                         _ptr(np.asarray(up, np.float32)), vfov_radians, width, height, near, far,
                         C.byref(cam))
    cam.background[:] = background
    return cam


# ------------------------------------------------------------------------------------------------
# PLY source format — src/source_format/ply.rs
# ------------------------------------------------------------------------------------------------

#: PlyGaussianPod — src/source_format/ply.rs:11-21
PLY_GAUSSIAN_DTYPE = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3), ("color", "<f4", 3),
                               ("sh", "<f4", 45), ("alpha", "<f4"), ("scale", "<f4", 3), ("rot", "<f4", 4)])
PLY_PROPERTIES = [_L.gs_ply_property_name(i).decode() for i in range(62)]


def gaussian_from_ply(ply):
    """Gaussian::from_ply over an array of PlyGaussianPod"""
    p = np.ascontiguousarray(np.atleast_1d(ply), dtype=PLY_GAUSSIAN_DTYPE)
    out = np.zeros(len(p), dtype=GAUSSIAN_DTYPE)
    _L.gs_gaussian_from_ply(_ptr(p), len(p), _ptr(out))
    return out


def expf(x):
    """gs_expf: the exp of Gaussian::from_ply on host and device (csrc/gs_convert.h)"""
    return float(_L.gs_expf(float(x)))


def gaussian_to_ply(gaussians):
    """Gaussian::to_ply"""
    g = np.ascontiguousarray(np.atleast_1d(gaussians), dtype=GAUSSIAN_DTYPE)
    out = np.zeros(len(g), dtype=PLY_GAUSSIAN_DTYPE)
    _L.gs_gaussian_to_ply(_ptr(g), len(g), _ptr(out))
    return out


class PlyGaussians:
    """PlyGaussians — src/source_format/ply.rs:200-449 (a Vec<PlyGaussianPod> with I/O)."""

    def __init__(self, pods, inria=None):
        self.pods = np.ascontiguousarray(pods, dtype=PLY_GAUSSIAN_DTYPE)
        self.inria = inria

    def __len__(self):
        return len(self.pods)

    def is_empty(self):
        return len(self.pods) == 0

    @staticmethod
    def read_from(data):
        """ReadIterGaussian::read_from on a bytes-like object"""
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        n, inria = C.c_size_t(), C.c_int32()
        _check(_L.gs_ply_read(_ptr(buf), buf.size, None, 0, C.byref(n), C.byref(inria)))
        pods = np.zeros(n.value, dtype=PLY_GAUSSIAN_DTYPE)
        _check(_L.gs_ply_read(_ptr(buf), buf.size, _ptr(pods), n.value, C.byref(n), C.byref(inria)))
        return PlyGaussians(pods, bool(inria.value))

    @staticmethod
    def read_from_file(path):
        with open(path, "rb") as f:
            return PlyGaussians.read_from(f.read())

    @staticmethod
    def from_gaussians(gaussians):
        """FromIterator<Gaussian>"""
        return PlyGaussians(gaussian_to_ply(gaussians))

    def write_to(self):
        """WriteIterGaussian::write_to -> bytes"""
        n = C.c_size_t()
        _check(_L.gs_ply_write(_ptr(self.pods), len(self.pods), None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        _check(_L.gs_ply_write(_ptr(self.pods), len(self.pods), _ptr(out), out.size, C.byref(n)))
        return out.tobytes()

    def write_to_file(self, path):
        with open(path, "wb") as f:
            f.write(self.write_to())

    def iter_gaussian(self):
        """IterGaussian: the Gaussians (Gaussian::from_ply of every pod)"""
        return gaussian_from_ply(self.pods)

    def __eq__(self, o):
        return isinstance(o, PlyGaussians) and self.pods.tobytes() == o.pods.tobytes()


# ------------------------------------------------------------------------------------------------
# SPZ source format — src/source_format/spz.rs
# ------------------------------------------------------------------------------------------------

def spz_options(version=3, sh_degree=3, fractional_bits=12, antialiased=False, sh_quantize_bits=(5, 4, 4)):
    """SpzGaussiansFromGaussianSliceOptions (defaults as the reference's)"""
    o = _capi.SpzOptions()
    _L.gs_spz_options_default(C.byref(o))
    o.version, o.sh_degree, o.fractional_bits = version, sh_degree, fractional_bits
    o.antialiased = int(bool(antialiased))
    o.sh_quantize_bits[:] = list(sh_quantize_bits)
    return o


def _sized_call(fn, *head):
    """two-call pattern of the C ABI: size query with out == NULL, then fill"""
    n = C.c_size_t()
    _check(fn(*head, None, 0, C.byref(n)))
    out = np.zeros(max(n.value, 1), dtype=np.uint8)
    _check(fn(*head, _ptr(out), n.value, C.byref(n)))
    return out[:n.value].tobytes()


class SpzGaussians:
    """SpzGaussians — spz.rs:514-959.  Holds the decompressed payload (header + columns) exactly as
    read or encoded, so write_to reproduces the same columns; `gaussians` is Gaussian::from_spz of
    every point (what iter_gaussian yields)."""

    def __init__(self, payload):
        self.payload = bytes(payload)
        buf = np.frombuffer(self.payload, dtype=np.uint8)
        hdr, n = _capi.SpzHeader(), C.c_size_t()
        _check(_L.gs_spz_decode_decompressed(_ptr(buf), buf.size, C.byref(hdr), None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=GAUSSIAN_DTYPE)
        _check(_L.gs_spz_decode_decompressed(_ptr(buf), buf.size, C.byref(hdr), _ptr(out), n.value, C.byref(n)))
        self.header, self.gaussians = hdr, out

    def __len__(self):
        return len(self.gaussians)

    def is_empty(self):
        return len(self.gaussians) == 0

    def __eq__(self, o):
        return isinstance(o, SpzGaussians) and self.payload == o.payload

    # ---- reading
    @staticmethod
    def read_from(data):
        """ReadIterGaussian::read_from (gzip'd .spz bytes)"""
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        return SpzGaussians(_sized_call(_L.gs_spz_decompress, _ptr(buf), buf.size))

    @staticmethod
    def read_decompressed(data):
        return SpzGaussians(data)

    @staticmethod
    def read_from_file(path):
        with open(path, "rb") as f:
            return SpzGaussians.read_from(f.read())

    # ---- from Gaussians
    @staticmethod
    def from_gaussians_with_options(gaussians, options=None):
        """from_gaussians_with_options (spz.rs:806-835); from_gaussians / FromIterator use the defaults"""
        return SpzGaussians(SpzGaussians.write_gaussians_decompressed(gaussians, options))

    from_gaussians = from_gaussians_with_options

    def iter_gaussian(self):
        return self.gaussians

    # ---- writing
    def write_decompressed(self):
        return self.payload

    def write_to(self):
        """WriteIterGaussian::write_to -> gzip'd bytes"""
        buf = np.frombuffer(self.payload, dtype=np.uint8)
        return _sized_call(_L.gs_spz_compress, _ptr(buf), buf.size)

    def write_to_file(self, path):
        with open(path, "wb") as f:
            f.write(self.write_to())

    # ---- one-shot helpers over the C ABI's fused entry points
    @staticmethod
    def _encode(fn, gaussians, options):
        g = np.ascontiguousarray(np.atleast_1d(gaussians), dtype=GAUSSIAN_DTYPE)
        options = options or spz_options()
        return _sized_call(fn, _ptr(g), len(g), C.byref(options))

    @staticmethod
    def write_gaussians(gaussians, options=None):
        """from_gaussians_with_options + write_to -> gzip'd bytes (gs_spz_encode)"""
        return SpzGaussians._encode(_L.gs_spz_encode, gaussians, options)

    @staticmethod
    def write_gaussians_decompressed(gaussians, options=None):
        return SpzGaussians._encode(_L.gs_spz_encode_decompressed, gaussians, options)

    @staticmethod
    def decode(data):
        """gs_spz_decode: gzip'd bytes -> (header, Gaussians) in one call"""
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        hdr, n = _capi.SpzHeader(), C.c_size_t()
        _check(_L.gs_spz_decode(_ptr(buf), buf.size, C.byref(hdr), None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=GAUSSIAN_DTYPE)
        _check(_L.gs_spz_decode(_ptr(buf), buf.size, C.byref(hdr), _ptr(out), n.value, C.byref(n)))
        return hdr, out


class GaussiansSource:
    """GaussiansSource — src/gaussian.rs:394-416"""
    Internal, Ply, Spz = "Internal", "Ply", "Spz"


class Gaussians:
    """Gaussians — the unified representation (src/gaussian.rs:412-548): Internal holds an array of
    Gaussian, Ply a PlyGaussians, Spz a SpzGaussians."""

    def __init__(self, inner):
        if isinstance(inner, Gaussians):
            inner = inner.inner
        if not isinstance(inner, (PlyGaussians, SpzGaussians)):
            inner = np.ascontiguousarray(np.atleast_1d(inner), dtype=GAUSSIAN_DTYPE)   # From<Vec<Gaussian>>
        self.inner = inner

    @staticmethod
    def from_gaussians_iter(gaussians, source):
        g = np.ascontiguousarray(np.atleast_1d(gaussians), dtype=GAUSSIAN_DTYPE)
        if source == GaussiansSource.Internal:
            return Gaussians(g)
        if source == GaussiansSource.Ply:
            return Gaussians(PlyGaussians.from_gaussians(g))
        if source == GaussiansSource.Spz:
            return Gaussians(SpzGaussians.from_gaussians(g))
        raise ValueError("unknown GaussiansSource %r" % (source,))

    def source(self):
        if isinstance(self.inner, PlyGaussians):
            return GaussiansSource.Ply
        if isinstance(self.inner, SpzGaussians):
            return GaussiansSource.Spz
        return GaussiansSource.Internal

    def __len__(self):
        return len(self.inner)

    def is_empty(self):
        return len(self.inner) == 0

    @staticmethod
    def read_from(data, source):
        if source == GaussiansSource.Internal:
            raise ValueError("cannot read Internal Gaussians from buffer")      # gaussian.rs:480-483
        cls = PlyGaussians if source == GaussiansSource.Ply else SpzGaussians
        return Gaussians(cls.read_from(data))

    @staticmethod
    def read_from_file(path, source):
        if source == GaussiansSource.Internal:
            raise ValueError("cannot read Internal Gaussians from file")        # gaussian.rs:461-464
        cls = PlyGaussians if source == GaussiansSource.Ply else SpzGaussians
        return Gaussians(cls.read_from_file(path))

    def write_to(self):
        if self.source() == GaussiansSource.Internal:
            raise ValueError("cannot write Internal Gaussians to buffer")       # gaussian.rs:510-513
        return self.inner.write_to()

    def write_to_file(self, path):
        if self.source() == GaussiansSource.Internal:
            raise ValueError("cannot write Internal Gaussians to file")         # gaussian.rs:498-501
        self.inner.write_to_file(path)

    def iter_gaussian(self):
        """IterGaussian — the Gaussians in the internal format, whatever the source"""
        return self.inner if self.source() == GaussiansSource.Internal else self.inner.iter_gaussian()

    def __eq__(self, o):
        if not isinstance(o, Gaussians) or self.source() != o.source():
            return False
        if self.source() == GaussiansSource.Internal:
            return self.inner.tobytes() == o.inner.tobytes()
        return self.inner == o.inner


# ------------------------------------------------------------------------------------------------
# device / stream / buffers
# ------------------------------------------------------------------------------------------------

def hip_versions():
    """(compiled, runtime, driver): HIP_VERSION of the headers libgs3d_hip.so was built with and the
    versions of the runtime / driver it runs on (gs_hip_versions; wgpu AdapterInfo::driver_info)."""
    a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _L.gs_hip_versions(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


class Device:
    """wgpu::Device + wgpu::Queue."""

    def __init__(self, ordinal=0):
        h = C.c_void_p()
        _check(_L.gs_device_create(ordinal, C.byref(h)))
        self._h = h
        self.ordinal = ordinal

    def limits(self):
        lim = Limits()
        _check(_L.gs_device_limits(self._h, C.byref(lim)))
        return lim

    def synchronize(self):
        _check(_L.gs_device_synchronize(self._h))

    def fast_rank(self):
        """True while this device's radix sorts rank with returning LDS atomics (gs_device_fast_rank)"""
        return bool(_L.gs_device_fast_rank(self._h))

    def create_stream(self, priority=None):
        """priority: see gs_stream_create_with_priority — streams meant to overlap on the device (frames in flight)
        need different priorities to be sure of different hardware queues"""
        return Stream(self, priority=priority)

    def stream_priority_range(self):
        """(least, greatest) of hipDeviceGetStreamPriorityRange; numerically lower = higher priority"""
        lo, hi = C.c_int32(0), C.c_int32(0)
        _check(_L.gs_device_stream_priority_range(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def wrap_stream(self, native_handle):
        return Stream(self, native=native_handle)

    def close(self):
        if self._h:
            _L.gs_device_destroy(self._h)
            self._h = None


class Stream:
    """CommandEncoder + queue.submit: an ordered HIP stream."""

    def __init__(self, device, native=None, priority=None):
        h = C.c_void_p()
        if native is None and priority is not None:
            _check(_L.gs_stream_create_with_priority(device._h, int(priority), C.byref(h)))
        elif native is None:
            _check(_L.gs_stream_create(device._h, C.byref(h)))
        else:
            _check(_L.gs_stream_wrap(device._h, C.c_void_p(native), C.byref(h)))
        self._h = h
        self.device = device

    def synchronize(self):
        _check(_L.gs_stream_synchronize(self._h))

    def native(self):
        return _L.gs_stream_native(self._h)

    def close(self):
        if self._h:
            _L.gs_stream_destroy(self._h)
            self._h = None


class Buffer:
    """wgpu::Buffer with the BufferWrapper download contract (src/buffer/mod.rs:17-102)."""

    def __init__(self, device, size=None, data=None, _handle=None):
        self.device = device
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        if data is not None:
            data = np.ascontiguousarray(data)
            size = data.nbytes
        _check(_L.gs_buffer_create(device._h, size, _ptr(data) if data is not None else None,
                                   C.byref(h)))
        self._h = h

    @staticmethod
    def from_raw(device, device_ptr, size):
        h = C.c_void_p()
        _check(_L.gs_buffer_from_raw(device._h, C.c_void_p(device_ptr), size, C.byref(h)))
        return Buffer(device, _handle=h)

    def clone(self):
        return Buffer(self.device, _handle=C.c_void_p(_L.gs_buffer_retain(self._h)))

    def size(self):
        return _L.gs_buffer_size(self._h)

    def device_ptr(self):
        return _L.gs_buffer_device_ptr(self._h)

    def write(self, stream, offset, data):
        data = np.ascontiguousarray(data)
        _check(_L.gs_buffer_write(self._h, stream._h, offset, _ptr(data), data.nbytes))

    def download(self, stream, dtype=np.uint8):
        """BufferWrapper::download::<T> — blocking."""
        out = np.zeros(self.size(), dtype=np.uint8)
        _check(_L.gs_buffer_download(self._h, stream._h, _ptr(out), out.nbytes))
        return out.view(dtype)

    def prepare_download(self, stream):
        """BufferWrapper::prepare_download: enqueue the copy into a staging buffer; returns a Download"""
        h = C.c_void_p()
        _check(_L.gs_buffer_prepare_download(self._h, stream._h if stream else None, C.byref(h)))
        return Download(h)

    def release(self):
        if self._h:
            _L.gs_buffer_release(self._h)
            self._h = None


class Download:
    """A pending device-to-host copy (gs_download): map() = BufferWrapper::map_download"""

    def __init__(self, handle):
        self._h = handle

    def ready(self):
        return bool(_L.gs_download_ready(self._h))

    def map(self, dtype=np.uint8):
        """blocks until the copy has landed; returns a numpy COPY of the staged bytes"""
        data, n = C.c_void_p(), C.c_size_t()
        _check(_L.gs_download_map(self._h, C.byref(data), C.byref(n)))
        raw = (C.c_uint8 * n.value).from_address(data.value) if n.value else b""
        return np.frombuffer(bytes(raw), dtype=np.uint8).view(dtype).copy()

    def release(self):
        if self._h:
            _L.gs_download_release(self._h)
            self._h = None


def pack_device(device, stream, pod, gaussians_buffer, count, pods_buffer):
    """gs_pack_device: Gaussians (struct Gaussian records in `gaussians_buffer`) -> PODs in `pods_buffer`"""
    _check(_L.gs_pack_device(device._h, stream._h if stream else None, pod.sh, pod.cov,
                             C.c_void_p(gaussians_buffer.device_ptr()), count, C.c_void_p(pods_buffer.device_ptr())))


class GaussiansBuffer:
    """GaussiansBuffer<G> — src/buffer/gaussian.rs:17-229."""

    def __init__(self, device, pod, _handle):
        self.device, self.pod, self._h = device, pod, _handle

    @staticmethod
    def new(device, pod, gaussians):
        g = np.ascontiguousarray(np.atleast_1d(gaussians), dtype=GAUSSIAN_DTYPE)
        h = C.c_void_p()
        _check(_L.gs_gaussians_buffer_create_from_gaussians(device._h, pod.sh, pod.cov, _ptr(g),
                                                            len(g), C.byref(h)))
        return GaussiansBuffer(device, pod, h)

    @staticmethod
    def new_from_ply(device, pod, ply):
        """PlyGaussians (or an array of PlyGaussianPod) -> buffer: the vertex records are uploaded as
        they are and ONE device kernel does Gaussian::from_ply + G::from_gaussian
        (gs_gaussians_buffer_create_from_ply) — bit-equal to new(device, pod, gaussian_from_ply(ply))."""
        p = np.ascontiguousarray(np.atleast_1d(getattr(ply, "pods", ply)), dtype=PLY_GAUSSIAN_DTYPE)
        h = C.c_void_p()
        _check(_L.gs_gaussians_buffer_create_from_ply(device._h, pod.sh, pod.cov, _ptr(p), len(p), C.byref(h)))
        return GaussiansBuffer(device, pod, h)

    @staticmethod
    def new_from_spz(device, pod, data, decompressed=False):
        """SPZ file bytes (gzip'd, or the decompressed payload) -> buffer: inflate on the host, then ONE
        device kernel does Gaussian::from_spz + G::from_gaussian (gs_gaussians_buffer_create_from_spz) —
        bit-equal to new(device, pod, SpzGaussians.read_from(data).iter_gaussian())."""
        raw = np.frombuffer(bytes(data), dtype=np.uint8)
        h, hdr = C.c_void_p(), _capi.SpzHeader()
        fn = _L.gs_gaussians_buffer_create_from_spz_decompressed if decompressed else _L.gs_gaussians_buffer_create_from_spz
        _check(fn(device._h, pod.sh, pod.cov, _ptr(raw), raw.size, C.byref(hdr), C.byref(h)))
        b = GaussiansBuffer(device, pod, h)
        b.spz_header = hdr
        return b

    @staticmethod
    def new_with_pods(device, pod, pods):
        pods = np.ascontiguousarray(pods, dtype=np.uint8)
        assert pods.nbytes % pod.size == 0
        h = C.c_void_p()
        _check(_L.gs_gaussians_buffer_create(device._h, pod.sh, pod.cov, _ptr(pods),
                                             pods.nbytes // pod.size, C.byref(h)))
        return GaussiansBuffer(device, pod, h)

    @staticmethod
    def new_empty(device, pod, length):
        h = C.c_void_p()
        _check(_L.gs_gaussians_buffer_create(device._h, pod.sh, pod.cov, None, length, C.byref(h)))
        return GaussiansBuffer(device, pod, h)

    @staticmethod
    def try_from(buffer, pod):
        """TryFrom<wgpu::Buffer> — raises GaussiansBufferTryFromBufferError."""
        h = C.c_void_p()
        _check(_L.gs_gaussians_buffer_from_buffer(buffer._h, pod.sh, pod.cov, C.byref(h)))
        return GaussiansBuffer(buffer.device, pod, h)

    def len(self):
        return _L.gs_gaussians_buffer_len(self._h)

    __len__ = len

    def is_empty(self):
        return self.len() == 0

    def buffer(self):
        """BufferWrapper::buffer() — a new handle to the same device allocation."""
        return Buffer(self.device, _handle=C.c_void_p(
            _L.gs_buffer_retain(_L.gs_gaussians_buffer_buffer(self._h))))

    def update(self, stream, gaussians):
        g = np.ascontiguousarray(np.atleast_1d(gaussians), dtype=GAUSSIAN_DTYPE)
        _check(_L.gs_gaussians_buffer_update_gaussians(self._h, stream._h, _ptr(g), len(g)))

    def update_with_pod(self, stream, pods):
        pods = np.ascontiguousarray(pods, dtype=np.uint8)
        _check(_L.gs_gaussians_buffer_update(self._h, stream._h, _ptr(pods),
                                             pods.nbytes // self.pod.size))

    def update_range(self, stream, start, gaussians):
        g = np.ascontiguousarray(np.atleast_1d(gaussians), dtype=GAUSSIAN_DTYPE)
        _check(_L.gs_gaussians_buffer_update_range_gaussians(self._h, stream._h, start, _ptr(g), len(g)))

    def update_range_from_ply(self, stream, start, ply):
        p = np.ascontiguousarray(np.atleast_1d(getattr(ply, "pods", ply)), dtype=PLY_GAUSSIAN_DTYPE)
        _check(_L.gs_gaussians_buffer_update_range_ply(self._h, stream._h, start, _ptr(p), len(p)))

    def update_range_with_pod(self, stream, start, pods):
        pods = np.ascontiguousarray(pods, dtype=np.uint8)
        _check(_L.gs_gaussians_buffer_update_range(self._h, stream._h, start, _ptr(pods),
                                                   pods.nbytes // self.pod.size))

    def download(self, stream):
        out = np.zeros(self.len() * self.pod.size, dtype=np.uint8)
        _check(_L.gs_gaussians_buffer_download(self._h, stream._h, _ptr(out), self.len()))
        return out

    def download_gaussians(self, stream):
        out = np.zeros(self.len(), dtype=GAUSSIAN_DTYPE)
        _check(_L.gs_gaussians_buffer_download_gaussians(self._h, stream._h, _ptr(out), self.len()))
        return out

    def mark_dirty(self):
        _L.gs_gaussians_buffer_mark_dirty(self._h)

    def set_spatial_order(self, enabled):
        """mirror slots in spatial (Morton) order (default) or in index order — DESIGN.md §3.4a"""
        _check(_L.gs_gaussians_buffer_set_spatial_order(self._h, int(bool(enabled))))

    def spatial_order(self):
        return bool(_L.gs_gaussians_buffer_spatial_order(self._h))

    def download_order(self, stream=None):
        """order[slot] = Gaussian index of the renderer's mirror (the identity in index order)"""
        out = np.zeros(self.len(), dtype=np.uint32)
        _check(_L.gs_gaussians_buffer_download_order(self._h, stream._h if stream is not None else None,
                                                      _ptr(out), self.len()))
        return out

    def destroy(self):
        if self._h:
            _L.gs_gaussians_buffer_destroy(self._h)
            self._h = None


class GaussianTransformBuffer(Buffer):
    """src/buffer/gaussian_transform.rs:104-163"""

    def __init__(self, device, _handle=None):
        if _handle is None:
            _handle = C.c_void_p()
            _check(_L.gs_gaussian_transform_buffer_create(device._h, C.byref(_handle)))
        super().__init__(device, _handle=_handle)

    @staticmethod
    def try_from(buffer):
        _check(_L.gs_gaussian_transform_buffer_from_buffer(buffer._h))
        return GaussianTransformBuffer(buffer.device, _handle=C.c_void_p(_L.gs_buffer_retain(buffer._h)))

    def update(self, stream, size, display_mode, sh_deg, no_sh0, max_std_dev):
        self.update_with_pod(stream, gaussian_transform_pod(size, display_mode, sh_deg, no_sh0, max_std_dev))

    def update_with_pod(self, stream, pod):
        _check(_L.gs_gaussian_transform_buffer_update(self._h, stream._h, C.byref(pod)))


class ModelTransformBuffer(Buffer):
    """src/buffer/model_transform.rs:10-58"""

    def __init__(self, device, _handle=None):
        if _handle is None:
            _handle = C.c_void_p()
            _check(_L.gs_model_transform_buffer_create(device._h, C.byref(_handle)))
        super().__init__(device, _handle=_handle)

    @staticmethod
    def try_from(buffer):
        _check(_L.gs_model_transform_buffer_from_buffer(buffer._h))
        return ModelTransformBuffer(buffer.device, _handle=C.c_void_p(_L.gs_buffer_retain(buffer._h)))

    def update(self, stream, pos, rot, scale):
        self.update_with_pod(stream, model_transform_pod(pos, rot, scale))

    def update_with_pod(self, stream, pod):
        _check(_L.gs_model_transform_buffer_update(self._h, stream._h, C.byref(pod)))


# ------------------------------------------------------------------------------------------------
# ComputeBundle
# ------------------------------------------------------------------------------------------------

class KernelRegistry:
    """The resolver: maps a module path to a kernel of the built-in library (the HIP analogue of
    wesl::PkgResolver over shader::PACKAGE, src/shader.rs:8-56)."""
    MODULES = {"array_map_add": KERNEL_ARRAY_MAP_ADD, "test_gaussian": KERNEL_TEST_GAUSSIAN,
               "test_gaussian_transform": KERNEL_TEST_GAUSSIAN_TRANSFORM,
               "test_model_transform": KERNEL_TEST_MODEL_TRANSFORM, "unpack_soa": KERNEL_UNPACK_SOA}

    def resolve(self, path):
        key = path.split("::")[-1]
        if key not in self.MODULES:
            raise KernelResolveError("module not found: %s" % path)
        return self.MODULES[key]


class SourceResolver:
    """Resolver over caller-supplied HIP C++ modules (the analogue of a wesl::PkgResolver holding
    the caller's own package next to shader::PACKAGE).  A module's text may
    `#include <wgpu_3dgs_core.h>` to import the device library."""

    def __init__(self, modules=None):
        self.modules = dict(modules or {})

    def add_module(self, path, source):
        self.modules[path] = source
        return self

    def resolve(self, path):
        if path not in self.modules:
            raise KernelResolveError("module not found: %s" % path)
        return self.modules[path]


def _buffer_array(buffers):
    arr = (C.c_void_p * len(buffers))(*[b._h for b in buffers])
    return arr


class ComputeBundle:
    """src/compute_bundle.rs:49-351.  `layouts` = bindings per bind group."""

    def __init__(self, device, handle, managed):
        self.device, self._h, self._managed = device, handle, managed

    @staticmethod
    def _desc(label, kernel, pod, layouts, workgroup_size, constants, keep):
        d = _capi.BundleDesc()
        d.label = label.encode() if label else None
        d.kernel = kernel
        d.sh, d.cov = (pod.sh, pod.cov) if pod is not None else (0, 0)
        d.bind_group_count = len(layouts)
        arr = (C.c_uint32 * max(len(layouts), 1))(*layouts)
        d.bindings_per_group = arr
        d.workgroup_size = workgroup_size or 0
        names = (C.c_char_p * max(len(constants), 1))(*[k.encode() for k in constants])
        vals = (C.c_double * max(len(constants), 1))(*[float(v) for v in constants.values()])
        d.constant_names, d.constant_values, d.constant_count = names, vals, len(constants)
        keep.extend([arr, names, vals])
        return d

    @staticmethod
    def new(label, device, layouts, resources, kernel, pod=None, workgroup_size=None, constants=None):
        """ComputeBundle::new — :141-188"""
        keep = []
        d = ComputeBundle._desc(label, kernel, pod, list(layouts), workgroup_size, constants or {}, keep)
        resources = [list(r) for r in resources]
        groups = [_buffer_array(r) for r in resources]
        gp = (C.c_void_p * max(len(groups), 1))(*[C.cast(g, C.c_void_p) for g in groups])
        counts = (C.c_uint32 * max(len(groups), 1))(*[len(r) for r in resources])
        h = C.c_void_p()
        _check(_L.gs_bundle_create_with_bind_groups(device._h, C.byref(d), gp, counts, len(groups),
                                                    C.byref(h)))
        return ComputeBundle(device, h, True)

    @staticmethod
    def new_without_bind_groups(label, device, layouts, kernel, pod=None, workgroup_size=None,
                                constants=None):
        """ComputeBundle::new_without_bind_groups — :260-341"""
        keep = []
        d = ComputeBundle._desc(label, kernel, pod, list(layouts), workgroup_size, constants or {}, keep)
        h = C.c_void_p()
        _check(_L.gs_bundle_create(device._h, C.byref(d), C.byref(h)))
        return ComputeBundle(device, h, False)

    @staticmethod
    def new_from_source(label, device, layouts, source, entry_point, pod=None, workgroup_size=None,
                        constants=None, defines=(), resources=None):
        """Compile `source` (HIP C++) with hiprtc: ComputeBundleBuilder::build for a custom shader."""
        constants = constants or {}
        d = _capi.BundleSourceDesc()
        d.label = label.encode() if label else None
        d.source = source.encode()
        d.entry_point = entry_point.encode()
        d.sh, d.cov = (pod.sh, pod.cov) if pod is not None else (0, 0)
        layouts = list(layouts)
        arr = (C.c_uint32 * max(len(layouts), 1))(*layouts)
        d.bind_group_count, d.bindings_per_group = len(layouts), arr
        d.workgroup_size = workgroup_size or 0
        names = (C.c_char_p * max(len(constants), 1))(*[k.encode() for k in constants])
        vals = (C.c_double * max(len(constants), 1))(*[float(v) for v in constants.values()])
        d.constant_names, d.constant_values, d.constant_count = names, vals, len(constants)
        defs = (C.c_char_p * max(len(defines), 1))(*[x.encode() for x in defines])
        d.defines, d.define_count = defs, len(defines)
        h = C.c_void_p()
        status = _L.gs_bundle_create_from_source(device._h, C.byref(d), C.byref(h))
        if status == -20:
            info = _capi.ErrorInfo()
            _L.gs_last_error(C.byref(info))
            raise KernelResolveError(info.message.decode(errors="replace"))
        _check(status)
        bundle = ComputeBundle(device, h, False)
        if resources is not None:
            resources = [list(r) for r in resources]
            groups = [_buffer_array(r) for r in resources]
            gp = (C.c_void_p * max(len(groups), 1))(*[C.cast(g, C.c_void_p) for g in groups])
            counts = (C.c_uint32 * max(len(groups), 1))(*[len(r) for r in resources])
            try:
                _check(_L.gs_bundle_attach_bind_groups(h, gp, counts, len(groups)))
            except GsError:
                bundle.destroy()
                raise
            bundle._managed = True
        return bundle

    def workgroup_size(self):
        return _L.gs_bundle_workgroup_size(self._h)

    def label(self):
        v = _L.gs_bundle_label(self._h)
        return v.decode() if v else None

    def bind_group_layouts(self):
        return _L.gs_bundle_bind_group_layout_count(self._h)

    def bind_groups(self):
        return _L.gs_bundle_bind_group_count(self._h)

    def update_bind_group_with_binding_resources(self, index, resources):
        """Returns False when index is out of bounds (Rust: None) — :222-231"""
        if index >= self.bind_groups():
            return False
        resources = list(resources)
        _check(_L.gs_bundle_set_bind_group(self._h, index, _buffer_array(resources), len(resources)))
        return True

    def dispatch(self, stream, count, bind_groups=None):
        """dispatch(encoder, count) / ComputeBundle<()>::dispatch(encoder, count, bind_groups)"""
        if bind_groups is None:
            _check(_L.gs_bundle_dispatch(self._h, stream._h, count))
            return
        bind_groups = [list(g) for g in bind_groups]
        groups = [_buffer_array(g) for g in bind_groups]
        gp = (C.c_void_p * max(len(groups), 1))(*[C.cast(g, C.c_void_p) for g in groups])
        counts = (C.c_uint32 * max(len(groups), 1))(*[len(g) for g in bind_groups])
        _check(_L.gs_bundle_dispatch_with_bind_groups(self._h, stream._h, count, gp, counts, len(groups)))

    def last_workgroup_count(self):
        return _L.gs_bundle_last_workgroup_count(self._h)

    def destroy(self):
        if self._h:
            _L.gs_bundle_destroy(self._h)
            self._h = None


class ComputeBundleBuilder:
    """src/compute_bundle.rs:364-593: same required fields, checked in the same order."""

    def __init__(self):
        self._label = None
        self._layouts = []
        self._constants = {}
        self._entry_point = None
        self._main_shader = None
        self._pod = None
        self._resolver = None
        self._workgroup_size = None
        self._defines = ()

    def label(self, label):
        self._label = label
        return self

    def bind_group_layout(self, bindings):
        """`bindings` = number of entries of the BindGroupLayoutDescriptor"""
        self._layouts.append(int(bindings))
        return self

    def bind_group_layouts(self, layouts):
        self._layouts.extend(int(b) for b in layouts)
        return self

    def pipeline_compile_options(self, constants):
        self._constants = dict(constants)
        return self

    def entry_point(self, name):
        self._entry_point = name
        return self

    def main_shader(self, module_path):
        self._main_shader = module_path
        return self

    def wesl_compile_options(self, features, defines=()):
        """`features` = a GaussianPod (its features() select the kernel instantiation / the
        GS_SH, GS_COV and feature-name macros); `defines` = further feature flags"""
        self._pod = features
        self._defines = tuple(defines)
        return self

    def resolver(self, resolver):
        self._resolver = resolver
        return self

    def workgroup_size(self, n):
        self._workgroup_size = n
        return self

    def _resolve(self):
        if not self._layouts:
            raise MissingBindGroupLayout("missing bind group layout for compute bundle")
        if self._resolver is None:
            raise MissingResolver("missing resolver for compute bundle")
        if self._entry_point is None:
            raise MissingEntryPoint("missing entry point for compute bundle")
        if self._main_shader is None:
            raise MissingMainShader("missing main shader for compute bundle")
        return self._resolver.resolve(self._main_shader)

    def build(self, device, resources):
        kernel = self._resolve()
        if isinstance(kernel, str):   # a module of a SourceResolver: compile it
            return ComputeBundle.new_from_source(self._label, device, self._layouts, kernel,
                                                 self._entry_point, self._pod, self._workgroup_size,
                                                 self._constants, self._defines, resources)
        return ComputeBundle.new(self._label, device, self._layouts, resources, kernel, self._pod,
                                 self._workgroup_size, self._constants)

    def build_without_bind_groups(self, device):
        kernel = self._resolve()
        if isinstance(kernel, str):
            return ComputeBundle.new_from_source(self._label, device, self._layouts, kernel,
                                                 self._entry_point, self._pod, self._workgroup_size,
                                                 self._constants, self._defines, None)
        return ComputeBundle.new_without_bind_groups(self._label, device, self._layouts, kernel,
                                                     self._pod, self._workgroup_size, self._constants)


# ------------------------------------------------------------------------------------------------
# renderer
# ------------------------------------------------------------------------------------------------

FRAME_FLAG_PAIR_OVERFLOW, FRAME_FLAG_SKIPPED, FRAME_FLAG_RANK_FAULT = 1, 2, 4     # gs_frame_result.flags
STAGE_NAMES = ["repack", "preprocess", "scan", "depth_sort", "expand", "tile_sort", "ranges", "blend", "frame"]


class Renderer:
    """One render context (scratch buffers + stats) on a device: gs_render_frame."""

    def __init__(self, device):
        h = C.c_void_p()
        _check(_L.gs_renderer_create(device._h, C.byref(h)))
        self._h, self.device = h, device

    def set_timing(self, enabled):
        _check(_L.gs_renderer_set_timing(self._h, int(bool(enabled))))

    def reset_stats(self):
        _check(_L.gs_renderer_reset_stats(self._h))

    def set_frame_flags_target(self, device_word_ptr):
        """gs_renderer_set_frame_flags_target: device word (or None) that receives every following
        frame's flags in stream order (FRAME_FLAG_*; 0 = rendered)."""
        _check(_L.gs_renderer_set_frame_flags_target(self._h, C.c_void_p(device_word_ptr or 0)))

    def stats(self):
        st = FrameStats()
        _check(_L.gs_renderer_stats(self._h, C.byref(st)))
        return st

    def sort_info(self):
        """gs_renderer_sort_info: how the last frame sorted (MSD-first or LSD passes, largest bucket)."""
        si = SortInfo()
        _check(_L.gs_renderer_sort_info(self._h, C.byref(si)))
        return si

    def set_sort_mode(self, depth_msd=-1, tile_msd=-1):
        """gs_renderer_set_sort_mode: 1 MSD-first, 0 LSD passes, -1 the renderer chooses (default)."""
        _check(_L.gs_renderer_set_sort_mode(self._h, int(depth_msd), int(tile_msd)))

    def set_tile_masks(self, mode=-1):
        """gs_renderer_set_tile_masks: 1 / 0 pin tile rect version 4 / 3, -1 the renderer chooses (by scene size)."""
        _check(_L.gs_renderer_set_tile_masks(self._h, int(mode)))

    def set_rounds(self, mode=-1, first_round=0):
        """gs_renderer_set_rounds: 1 two-round frames (the nearest `first_round` visible Gaussians first, the rest without
        what lies in finished tiles), 0 one round, -1 the renderer chooses."""
        _check(_L.gs_renderer_set_rounds(self._h, int(mode), int(first_round)))

    def render(self, stream, gaussians, gaussian_transform, model_transform, camera,
               rgba_device_ptr, band=None, check=True):
        """gs_render_frame.  check=True (the validated use: tests, one-off renders) waits for the
        frame and, if it exceeded the pair capacity sized from earlier frames, renders it again
        with the grown buffers; check=False only enqueues (the pipelined use: a viewer's frame
        loop, bench.py) — call wait_frame() / synchronise the stream before reading the image."""
        b0, b1 = band if band is not None else (0, 0xFFFFFFFF)
        for attempt in range(4):
            _check(_L.gs_render_frame(self._h, stream._h, gaussians._h, C.byref(gaussian_transform),
                                      C.byref(model_transform), C.byref(camera), b0, b1,
                                      C.c_void_p(rgba_device_ptr)))
            if not check:
                return None
            try:
                return self.wait_frame()
            except (PairCapacityError, RankOrderError):
                if attempt == 3:
                    raise

    def wait_frame(self):
        """blocks until the last frame has completed; raises PairCapacityError / PairOverflowError"""
        fr = FrameResult()
        _check(_L.gs_renderer_wait_frame(self._h, C.byref(fr)))
        return fr

    def download_projected(self, n):
        proj = np.zeros(n, dtype=PROJECTED_DTYPE)
        tiles = np.zeros(n, dtype=np.uint32)
        _check(_L.gs_renderer_download_projected(self._h, _ptr(proj), _ptr(tiles), n))
        return proj, tiles

    def _pairs(self, fn):
        d = C.c_uint64()
        _check(fn(self._h, None, None, 0, C.byref(d)))
        keys = np.zeros(max(d.value, 1), dtype=np.uint64)
        idx = np.zeros(max(d.value, 1), dtype=np.uint32)
        _check(fn(self._h, _ptr(keys), _ptr(idx), d.value, C.byref(d)))
        return keys[:d.value], idx[:d.value]

    def download_sorted(self):
        return self._pairs(_L.gs_renderer_download_sorted)

    def download_ranges(self, num_tiles):
        r = np.zeros((num_tiles, 2), dtype=np.uint32)
        _check(_L.gs_renderer_download_ranges(self._h, _ptr(r), num_tiles))
        return r

    def destroy(self):
        if self._h:
            _L.gs_renderer_destroy(self._h)
            self._h = None


class FrameRing:
    """Frames in flight: `frames` renderers, each on a stream of its own PRIORITY, take the frames in turn
    (what a viewer with double / triple buffering does).  A frame is a chain of dependent kernels — latency-bound
    in its sorts, VALU-bound in its blend — so the frames of different renderers overlap on the device, provided
    their streams sit on different hardware queues: HIP keeps separate queues per priority level, streams of one
    priority may share a queue (gs_stream_create_with_priority).  Every frame is the frame its renderer would
    have rendered alone; only the order in which the device works on them changes.  No reference item (the
    viewer owns the frame loop)."""

    def __init__(self, device, frames=3):
        least, greatest = device.stream_priority_range()
        cycle = [greatest, least, 0] if least != greatest else [0]
        self.priorities = [cycle[k % len(cycle)] for k in range(max(1, int(frames)))]
        self.streams = [device.create_stream(priority=p) for p in self.priorities]
        self.renderers = [Renderer(device) for _ in self.streams]
        self._next = 0

    def __len__(self):
        return len(self.renderers)

    def render(self, gaussians, gaussian_transform, model_transform, camera, rgba_device_ptr, band=None, check=False):
        """enqueues one frame on the next lane; returns the lane's index (its stream: `streams[lane]`)"""
        lane = self._next % len(self.renderers)
        self._next += 1
        self.renderers[lane].render(self.streams[lane], gaussians, gaussian_transform, model_transform, camera,
                                    rgba_device_ptr, band=band, check=check)
        return lane

    def wait(self):
        """blocks until every lane's last frame has completed; their FrameResults (raises as wait_frame does)"""
        return [r.wait_frame() for r in self.renderers]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        for r in self.renderers:
            r.destroy()
        for s in self.streams:
            s.close()
        self.renderers, self.streams = [], []


def sort_pairs_u64(device, stream, keys, values, end_bit=64):
    """Device radix sort of host arrays (stable LSD on key bits [0, end_bit)); returns copies."""
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    v = np.ascontiguousarray(values, dtype=np.uint32).copy()
    _check(_L.gs_sort_pairs_u64(device._h, stream._h if stream else None, _ptr(k), _ptr(v), len(k), end_bit))
    return k, v


def exclusive_scan_u32(device, stream, values):
    a = np.ascontiguousarray(values, dtype=np.uint32)
    out = np.zeros_like(a)
    total = C.c_uint64()
    _check(_L.gs_exclusive_scan_u32(device._h, stream._h if stream else None, _ptr(a), _ptr(out),
                                    len(a), C.byref(total)))
    return out, total.value
