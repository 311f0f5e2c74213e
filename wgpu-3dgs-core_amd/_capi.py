"""ctypes declarations of include/gs3d.h (libgs3d_hip.so).  Loading fails loudly when the HIP
library is missing or cannot be built: there is no pure-Python / CPU path behind this package."""
import ctypes as C
import os

from . import _hiprt
from . import build as _build

vp, sz, u8, u32, i32, u64, f32 = (C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint32, C.c_int32,
                                  C.c_uint64, C.c_float)


class ErrorInfo(C.Structure):
    _fields_ = [("code", i32), ("a", u64), ("b", u64), ("c", u64), ("message", C.c_char * 256)]


class GaussianTransformPod(C.Structure):
    """GaussianTransformPod — src/buffer/gaussian_transform.rs:166-174"""
    _fields_ = [("size", f32), ("flags", u8 * 4)]


class ModelTransformPod(C.Structure):
    """ModelTransformPod — src/buffer/model_transform.rs:61-66"""
    _fields_ = [("pos", f32 * 3), ("_pad0", f32), ("rot", f32 * 4), ("scale", f32 * 3), ("_pad1", f32)]


class Limits(C.Structure):
    _fields_ = [("max_compute_workgroup_size_x", u32), ("max_compute_invocations_per_workgroup", u32),
                ("compute_units", u32), ("wavefront_size", u32), ("total_memory_bytes", u64),
                ("arch_name", C.c_char * 64)]


class Camera(C.Structure):
    _fields_ = [("view", f32 * 16), ("pos", f32 * 3), ("fx", f32), ("fy", f32), ("cx", f32),
                ("cy", f32), ("near_plane", f32), ("far_plane", f32), ("width", u32),
                ("height", u32), ("background", f32 * 3)]


class FrameStats(C.Structure):
    _fields_ = [("gaussians", u64), ("visible", u64), ("pairs", u64), ("tiles_x", u32),
                ("tiles_y", u32), ("sort_passes", u32), ("timed_frames", u32),
                ("stage_ms", C.c_double * 12)]


class FrameResult(C.Structure):
    _fields_ = [("gaussians", u64), ("visible", u64), ("pairs", u64), ("pair_capacity", u64),
                ("flags", u32), ("launches", u32)]


class SortInfo(C.Structure):
    _fields_ = [("depth_msd", u32), ("depth_bucket_max", u32), ("bucket_capacity", u32), ("tile_msd", u32),
                ("tile_bucket_max", u32), ("tile_masks", u32), ("rounds", u32), ("round1", u32), ("tiles_done", u32), ("partitioned", u32)]


class BundleDesc(C.Structure):
    _fields_ = [("label", C.c_char_p), ("kernel", i32), ("sh", i32), ("cov", i32),
                ("bind_group_count", u32), ("bindings_per_group", C.POINTER(u32)),
                ("workgroup_size", u32), ("constant_names", C.POINTER(C.c_char_p)),
                ("constant_values", C.POINTER(C.c_double)), ("constant_count", u32)]


class SpzHeader(C.Structure):
    _fields_ = [("magic", u32), ("version", u32), ("num_points", u32), ("sh_degree", u8),
                ("fractional_bits", u8), ("flags", u8), ("reserved", u8)]


class SpzOptions(C.Structure):
    _fields_ = [("version", u32), ("sh_degree", u8), ("fractional_bits", u8), ("antialiased", u8),
                ("_pad", u8), ("sh_quantize_bits", u32 * 3)]


class BundleSourceDesc(C.Structure):
    _fields_ = [("label", C.c_char_p), ("source", C.c_char_p), ("entry_point", C.c_char_p),
                ("sh", i32), ("cov", i32), ("bind_group_count", u32),
                ("bindings_per_group", C.POINTER(u32)), ("workgroup_size", u32),
                ("constant_names", C.POINTER(C.c_char_p)), ("constant_values", C.POINTER(C.c_double)),
                ("constant_count", u32), ("defines", C.POINTER(C.c_char_p)), ("define_count", u32)]


# name -> (restype, argtypes); must list every function declared in include/gs3d.h
SIGNATURES = {
    "gs_last_error": (None, [vp]),
    "gs_status_string": (C.c_char_p, [i32]),
    "gs_abi_version": (u32, []),
    "gs_hip_versions": (None, [vp, vp, vp]),
    "gs_pod_size": (sz, [i32, i32]),
    "gs_pod_features": (i32, [i32, i32, vp]),
    "gs_feature_name": (C.c_char_p, [u32]),
    "gs_pack": (i32, [i32, i32, vp, sz, vp]),
    "gs_unpack_to_gaussian": (i32, [i32, i32, vp, sz, vp]),
    "gs_gaussian_transform_pod_new": (i32, [f32, i32, u8, u8, f32, vp]),
    "gs_gaussian_transform_pod_default": (None, [vp]),
    "gs_max_std_dev_encode": (i32, [f32, vp]),
    "gs_max_std_dev_decode": (f32, [u8]),
    "gs_model_transform_pod_new": (None, [vp, vp, vp, vp]),
    "gs_model_transform_pod_default": (None, [vp]),
    "gs_ply_property_name": (C.c_char_p, [u32]),
    "gs_gaussian_from_ply": (None, [vp, sz, vp]),
    "gs_gaussian_to_ply": (None, [vp, sz, vp]),
    "gs_expf": (f32, [f32]),
    "gs_ply_read": (i32, [vp, sz, vp, sz, vp, vp]),
    "gs_ply_write": (i32, [vp, sz, vp, sz, vp]),
    "gs_spz_options_default": (None, [vp]),
    "gs_spz_decode": (i32, [vp, sz, vp, vp, sz, vp]),
    "gs_spz_decode_decompressed": (i32, [vp, sz, vp, vp, sz, vp]),
    "gs_spz_encode": (i32, [vp, sz, vp, vp, sz, vp]),
    "gs_spz_encode_decompressed": (i32, [vp, sz, vp, vp, sz, vp]),
    "gs_spz_decompress": (i32, [vp, sz, vp, sz, vp]),
    "gs_spz_compress": (i32, [vp, sz, vp, sz, vp]),
    "gs_device_create": (i32, [i32, vp]),
    "gs_device_destroy": (None, [vp]),
    "gs_device_limits": (i32, [vp, vp]),
    "gs_device_synchronize": (i32, [vp]),
    "gs_device_fast_rank": (i32, [vp]),
    "gs_stream_create": (i32, [vp, vp]),
    "gs_stream_create_with_priority": (i32, [vp, i32, vp]),
    "gs_device_stream_priority_range": (i32, [vp, vp, vp]),
    "gs_stream_wrap": (i32, [vp, vp, vp]),
    "gs_stream_native": (vp, [vp]),
    "gs_stream_synchronize": (i32, [vp]),
    "gs_stream_destroy": (None, [vp]),
    "gs_buffer_create": (i32, [vp, sz, vp, vp]),
    "gs_buffer_from_raw": (i32, [vp, vp, sz, vp]),
    "gs_buffer_retain": (vp, [vp]),
    "gs_buffer_release": (None, [vp]),
    "gs_buffer_size": (sz, [vp]),
    "gs_buffer_device_ptr": (vp, [vp]),
    "gs_buffer_write": (i32, [vp, vp, sz, vp, sz]),
    "gs_buffer_download": (i32, [vp, vp, vp, sz]),
    "gs_gaussians_buffer_create": (i32, [vp, i32, i32, vp, sz, vp]),
    "gs_gaussians_buffer_create_from_gaussians": (i32, [vp, i32, i32, vp, sz, vp]),
    "gs_gaussians_buffer_create_from_ply": (i32, [vp, i32, i32, vp, sz, vp]),
    "gs_gaussians_buffer_update_range_ply": (i32, [vp, vp, sz, vp, sz]),
    "gs_pack_device_from_ply": (i32, [vp, vp, i32, i32, vp, sz, vp]),
    "gs_gaussians_buffer_create_from_spz": (i32, [vp, i32, i32, vp, sz, vp, vp]),
    "gs_gaussians_buffer_create_from_spz_decompressed": (i32, [vp, i32, i32, vp, sz, vp, vp]),
    "gs_gaussians_buffer_from_buffer": (i32, [vp, i32, i32, vp]),
    "gs_gaussians_buffer_destroy": (None, [vp]),
    "gs_gaussians_buffer_len": (sz, [vp]),
    "gs_gaussians_buffer_buffer": (vp, [vp]),
    "gs_gaussians_buffer_sh": (i32, [vp]),
    "gs_gaussians_buffer_cov3d": (i32, [vp]),
    "gs_gaussians_buffer_update": (i32, [vp, vp, vp, sz]),
    "gs_gaussians_buffer_update_range": (i32, [vp, vp, sz, vp, sz]),
    "gs_gaussians_buffer_update_gaussians": (i32, [vp, vp, vp, sz]),
    "gs_gaussians_buffer_update_range_gaussians": (i32, [vp, vp, sz, vp, sz]),
    "gs_gaussians_buffer_download": (i32, [vp, vp, vp, sz]),
    "gs_gaussians_buffer_download_gaussians": (i32, [vp, vp, vp, sz]),
    "gs_gaussians_buffer_mark_dirty": (None, [vp]),
    "gs_gaussians_buffer_set_spatial_order": (i32, [vp, i32]),
    "gs_gaussians_buffer_spatial_order": (i32, [vp]),
    "gs_gaussians_buffer_download_order": (i32, [vp, vp, vp, sz]),
    "gs_gaussian_transform_buffer_create": (i32, [vp, vp]),
    "gs_gaussian_transform_buffer_update": (i32, [vp, vp, vp]),
    "gs_gaussian_transform_buffer_from_buffer": (i32, [vp]),
    "gs_model_transform_buffer_create": (i32, [vp, vp]),
    "gs_model_transform_buffer_update": (i32, [vp, vp, vp]),
    "gs_model_transform_buffer_from_buffer": (i32, [vp]),
    "gs_bundle_create": (i32, [vp, vp, vp]),
    "gs_bundle_create_with_bind_groups": (i32, [vp, vp, vp, vp, u32, vp]),
    "gs_bundle_create_from_source": (i32, [vp, vp, vp]),
    "gs_bundle_attach_bind_groups": (i32, [vp, vp, vp, u32]),
    "gs_bundle_destroy": (None, [vp]),
    "gs_bundle_workgroup_size": (u32, [vp]),
    "gs_bundle_label": (C.c_char_p, [vp]),
    "gs_bundle_bind_group_layout_count": (u32, [vp]),
    "gs_bundle_bind_group_count": (u32, [vp]),
    "gs_bundle_set_bind_group": (i32, [vp, u32, vp, u32]),
    "gs_bundle_dispatch": (i32, [vp, vp, u32]),
    "gs_bundle_dispatch_with_bind_groups": (i32, [vp, vp, u32, vp, vp, u32]),
    "gs_bundle_last_workgroup_count": (u32, [vp]),
    "gs_camera_look_at": (None, [vp, vp, vp, f32, u32, u32, f32, f32, vp]),
    "gs_gaussians_read": (i32, [vp, sz, i32, vp, sz, vp]),
    "gs_gaussians_write": (i32, [vp, sz, i32, vp, sz, vp]),
    "gs_buffer_prepare_download": (i32, [vp, vp, vp]),
    "gs_download_ready": (i32, [vp]),
    "gs_download_map": (i32, [vp, vp, vp]),
    "gs_download_release": (None, [vp]),
    "gs_pack_device": (i32, [vp, vp, i32, i32, vp, sz, vp]),
    "gs_renderer_create": (i32, [vp, vp]),
    "gs_renderer_destroy": (None, [vp]),
    "gs_renderer_set_timing": (i32, [vp, i32]),
    "gs_renderer_reset_stats": (i32, [vp]),
    "gs_renderer_set_frame_flags_target": (i32, [vp, vp]),
    "gs_renderer_stats": (i32, [vp, vp]),
    "gs_renderer_wait_frame": (i32, [vp, vp]),
    "gs_renderer_sort_info": (i32, [vp, vp]),
    "gs_renderer_set_sort_mode": (i32, [vp, i32, i32]),
    "gs_renderer_set_tile_masks": (i32, [vp, i32]),
    "gs_renderer_set_rounds": (i32, [vp, i32, u32]),
    "gs_render_frame": (i32, [vp, vp, vp, vp, vp, vp, u32, u32, vp]),
    "gs_renderer_download_projected": (i32, [vp, vp, vp, sz]),
    "gs_renderer_download_sorted": (i32, [vp, vp, vp, u64, vp]),
    "gs_renderer_download_ranges": (i32, [vp, vp, sz]),
    "gs_sort_pairs_u64": (i32, [vp, vp, vp, vp, u64, u32]),
    "gs_exclusive_scan_u32": (i32, [vp, vp, vp, vp, u64, vp]),
}

_lib = None


def library_path():
    return _build.LIB_PATH


def load():
    """Load (building first if the sources are newer) libgs3d_hip.so.  Raises when it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    # GS3D_LIB=<path>: load ANOTHER build of libgs3d_hip.so (tools/build_rev.sh builds one from a git
    # revision) — same-box A/B runs of two versions of the kernels; never set in production
    override = os.environ.get("GS3D_LIB")
    if override:
        _hiprt.prepare()
        lib = C.CDLL(override)
        _hiprt.check()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is None:
                continue                 # an older revision may lack the newest entry points
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib
    if not os.path.exists(_build.LIB_PATH) or _build.needs_build():
        try:
            _build.build()
        except Exception as exc:  # no hipcc on this machine and no prebuilt library
            if not os.path.exists(_build.LIB_PATH):
                raise ImportError(
                    "libgs3d_hip.so is missing and could not be built (%s); this package has no "
                    "CPU fallback" % exc) from exc
    # one HIP runtime per process whatever the import order of torch and this package (_hiprt.py)
    _hiprt.prepare()
    lib = C.CDLL(_build.LIB_PATH)
    _hiprt.check()
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
