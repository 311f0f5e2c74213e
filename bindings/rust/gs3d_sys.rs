// gs3d_sys.rs — GENERATED from include/gs3d.h by tools/gen_rust_sys.py; do not edit.
// Raw FFI of libgs3d_hip.so for the reference crate.  Not compiled in the build image.
#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]
use std::os::raw::{c_char, c_void};

pub type gs_status = i32;

pub const GS_OK: gs_status = 0;
pub const GS_ERR_INVALID_ARGUMENT: gs_status = -1;
pub const GS_ERR_NO_DEVICE: gs_status = -2;
pub const GS_ERR_HIP: gs_status = -3;
pub const GS_ERR_OUT_OF_MEMORY: gs_status = -4;
pub const GS_ERR_COUNT_MISMATCH: gs_status = -10;
pub const GS_ERR_RANGE_COUNT_MISMATCH: gs_status = -11;
pub const GS_ERR_BUFFER_SIZE_NOT_MULTIPLE: gs_status = -12;
pub const GS_ERR_BUFFER_SIZE_MISMATCHED: gs_status = -13;
pub const GS_ERR_RESOURCE_COUNT_MISMATCH: gs_status = -14;
pub const GS_ERR_WORKGROUP_SIZE_EXCEEDS_LIMIT: gs_status = -15;
pub const GS_ERR_MISSING_BIND_GROUP_LAYOUT: gs_status = -16;
pub const GS_ERR_MISSING_RESOLVER: gs_status = -17;
pub const GS_ERR_MISSING_ENTRY_POINT: gs_status = -18;
pub const GS_ERR_MISSING_MAIN_SHADER: gs_status = -19;
pub const GS_ERR_KERNEL_COMPILE: gs_status = -20;
pub const GS_ERR_LOSSY_CONFIG: gs_status = -21;
pub const GS_ERR_DOWNLOAD: gs_status = -22;
pub const GS_ERR_PAIR_OVERFLOW: gs_status = -23;
pub const GS_ERR_PLY: gs_status = -24;
pub const GS_ERR_SPZ: gs_status = -25;
pub const GS_ERR_PAIR_CAPACITY: gs_status = -26;
pub const GS_ERR_RANK_ORDER: gs_status = -27;

// enum gs_sh_config (passed as u32)
pub const GS_SH_SINGLE: u32 = 0;
pub const GS_SH_HALF: u32 = 1;
pub const GS_SH_NORM8: u32 = 2;
pub const GS_SH_NONE: u32 = 3;

// enum gs_cov3d_config (passed as u32)
pub const GS_COV3D_ROT_SCALE: u32 = 0;
pub const GS_COV3D_SINGLE: u32 = 1;
pub const GS_COV3D_HALF: u32 = 2;

// enum gs_display_mode (passed as u32)
pub const GS_DISPLAY_SPLAT: u32 = 0;
pub const GS_DISPLAY_ELLIPSE: u32 = 1;
pub const GS_DISPLAY_POINT: u32 = 2;

// enum gs_gaussians_source (passed as u32)
pub const GS_SOURCE_INTERNAL: u32 = 0;
pub const GS_SOURCE_PLY: u32 = 1;
pub const GS_SOURCE_SPZ: u32 = 2;

// enum gs_kernel_id (passed as u32)
pub const GS_KERNEL_ARRAY_MAP_ADD: u32 = 0;
pub const GS_KERNEL_TEST_GAUSSIAN: u32 = 1;
pub const GS_KERNEL_TEST_GAUSSIAN_TRANSFORM: u32 = 2;
pub const GS_KERNEL_TEST_MODEL_TRANSFORM: u32 = 3;
pub const GS_KERNEL_UNPACK_SOA: u32 = 4;
pub const GS_KERNEL_COUNT_: u32 = 5;

#[repr(C)] pub struct gs_device { _private: [u8; 0] }
#[repr(C)] pub struct gs_stream { _private: [u8; 0] }
#[repr(C)] pub struct gs_buffer { _private: [u8; 0] }
#[repr(C)] pub struct gs_download { _private: [u8; 0] }
#[repr(C)] pub struct gs_gaussians_buffer { _private: [u8; 0] }
#[repr(C)] pub struct gs_bundle { _private: [u8; 0] }
#[repr(C)] pub struct gs_renderer { _private: [u8; 0] }

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_error_info {
    pub code: i32,
    pub a: u64,
    pub b: u64,
    pub c: u64,
    pub message: [c_char; 256],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_gaussian {
    pub rot: [f32; 4],
    pub pos: [f32; 3],
    pub color: [u8; 4],
    pub sh: [f32; 45],
    pub scale: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_gaussian_transform_pod {
    pub size: f32,
    pub flags: [u8; 4],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_model_transform_pod {
    pub pos: [f32; 3],
    pub _pad0: f32,
    pub rot: [f32; 4],
    pub scale: [f32; 3],
    pub _pad1: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_ply_gaussian_pod {
    pub pos: [f32; 3],
    pub normal: [f32; 3],
    pub color: [f32; 3],
    pub sh: [f32; 45],
    pub alpha: f32,
    pub scale: [f32; 3],
    pub rot: [f32; 4],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_spz_header {
    pub magic: u32,
    pub version: u32,
    pub num_points: u32,
    pub sh_degree: u8,
    pub fractional_bits: u8,
    pub flags: u8,
    pub reserved: u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_spz_options {
    pub version: u32,
    pub sh_degree: u8,
    pub fractional_bits: u8,
    pub antialiased: u8,
    pub _pad: u8,
    pub sh_quantize_bits: [u32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_limits {
    pub max_compute_workgroup_size_x: u32,
    pub max_compute_invocations_per_workgroup: u32,
    pub compute_units: u32,
    pub wavefront_size: u32,
    pub total_memory_bytes: u64,
    pub arch_name: [c_char; 64],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_bundle_desc {
    pub label: *const c_char,
    pub kernel: u32,
    pub sh: u32,
    pub cov: u32,
    pub bind_group_count: u32,
    pub bindings_per_group: *const u32,
    pub workgroup_size: u32,
    pub constant_names: *const *const c_char,
    pub constant_values: *const f64,
    pub constant_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_bundle_source_desc {
    pub label: *const c_char,
    pub source: *const c_char,
    pub entry_point: *const c_char,
    pub sh: u32,
    pub cov: u32,
    pub bind_group_count: u32,
    pub bindings_per_group: *const u32,
    pub workgroup_size: u32,
    pub constant_names: *const *const c_char,
    pub constant_values: *const f64,
    pub constant_count: u32,
    pub defines: *const *const c_char,
    pub define_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_camera {
    pub view: [f32; 16],
    pub pos: [f32; 3],
    pub fx: f32,
    pub fy: f32,
    pub cx: f32,
    pub cy: f32,
    pub near_plane: f32,
    pub far_plane: f32,
    pub width: u32,
    pub height: u32,
    pub background: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_projected {
    pub mx: f32,
    pub my: f32,
    pub ca: f32,
    pub cb: f32,
    pub cc: f32,
    pub opacity: f32,
    pub r: f32,
    pub g: f32,
    pub b: f32,
    pub depth: f32,
    pub tx0: u16,
    pub ty0: u16,
    pub tx1: u16,
    pub ty1: u16,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_frame_stats {
    pub gaussians: u64,
    pub visible: u64,
    pub pairs: u64,
    pub tiles_x: u32,
    pub tiles_y: u32,
    pub sort_passes: u32,
    pub timed_frames: u32,
    pub stage_ms: [f64; 12],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_frame_result {
    pub gaussians: u64,
    pub visible: u64,
    pub pairs: u64,
    pub pair_capacity: u64,
    pub flags: u32,
    pub launches: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gs_sort_info {
    pub depth_msd: u32,
    pub depth_bucket_max: u32,
    pub bucket_capacity: u32,
    pub tile_msd: u32,
    pub tile_bucket_max: u32,
    pub tile_masks: u32,
    pub rounds: u32,
    pub round1: u32,
    pub tiles_done: u32,
    pub partitioned: u32,
}

#[link(name = "gs3d_hip")]
extern "C" {
    pub fn gs_last_error(out: *mut gs_error_info);
    pub fn gs_status_string(s: gs_status) -> *const c_char;
    pub fn gs_abi_version() -> u32;
    pub fn gs_hip_versions(compiled: *mut i32, runtime: *mut i32, driver: *mut i32);
    pub fn gs_pod_size(sh: u32, cov: u32) -> usize;
    pub fn gs_pod_features(sh: u32, cov: u32, out: *mut u8) -> gs_status;
    pub fn gs_feature_name(index: u32) -> *const c_char;
    pub fn gs_pack(sh: u32, cov: u32, r#in: *const gs_gaussian, n: usize, out: *mut c_void) -> gs_status;
    pub fn gs_unpack_to_gaussian(sh: u32, cov: u32, pods: *const c_void, n: usize, out: *mut gs_gaussian) -> gs_status;
    pub fn gs_gaussian_transform_pod_new(size: f32, mode: u32, sh_deg: u8, no_sh0: u8, max_std_dev: f32, out: *mut gs_gaussian_transform_pod) -> gs_status;
    pub fn gs_gaussian_transform_pod_default(out: *mut gs_gaussian_transform_pod);
    pub fn gs_max_std_dev_encode(max_std_dev: f32, out: *mut u8) -> gs_status;
    pub fn gs_max_std_dev_decode(v: u8) -> f32;
    pub fn gs_model_transform_pod_new(pos: *const f32, rot_xyzw: *const f32, scale: *const f32, out: *mut gs_model_transform_pod);
    pub fn gs_model_transform_pod_default(out: *mut gs_model_transform_pod);
    pub fn gs_ply_property_name(index: u32) -> *const c_char;
    pub fn gs_gaussian_from_ply(r#in: *const gs_ply_gaussian_pod, n: usize, out: *mut gs_gaussian);
    pub fn gs_expf(x: f32) -> f32;
    pub fn gs_gaussian_to_ply(r#in: *const gs_gaussian, n: usize, out: *mut gs_ply_gaussian_pod);
    pub fn gs_ply_read(bytes: *const c_void, len: usize, out: *mut gs_ply_gaussian_pod, capacity: usize, count_out: *mut usize, is_inria_out: *mut i32) -> gs_status;
    pub fn gs_ply_write(pods: *const gs_ply_gaussian_pod, n: usize, out: *mut c_void, capacity: usize, bytes_out: *mut usize) -> gs_status;
    pub fn gs_spz_options_default(out: *mut gs_spz_options);
    pub fn gs_spz_decode(bytes: *const c_void, len: usize, header_out: *mut gs_spz_header, out: *mut gs_gaussian, capacity: usize, count_out: *mut usize) -> gs_status;
    pub fn gs_spz_decode_decompressed(bytes: *const c_void, len: usize, header_out: *mut gs_spz_header, out: *mut gs_gaussian, capacity: usize, count_out: *mut usize) -> gs_status;
    pub fn gs_spz_encode(r#in: *const gs_gaussian, n: usize, options: *const gs_spz_options, out: *mut c_void, capacity: usize, bytes_out: *mut usize) -> gs_status;
    pub fn gs_spz_encode_decompressed(r#in: *const gs_gaussian, n: usize, options: *const gs_spz_options, out: *mut c_void, capacity: usize, bytes_out: *mut usize) -> gs_status;
    pub fn gs_spz_decompress(bytes: *const c_void, len: usize, out: *mut c_void, capacity: usize, bytes_out: *mut usize) -> gs_status;
    pub fn gs_spz_compress(bytes: *const c_void, len: usize, out: *mut c_void, capacity: usize, bytes_out: *mut usize) -> gs_status;
    pub fn gs_gaussians_read(bytes: *const c_void, len: usize, source: gs_gaussians_source, out: *mut gs_gaussian, capacity: usize, count_out: *mut usize) -> gs_status;
    pub fn gs_gaussians_write(r#in: *const gs_gaussian, n: usize, source: gs_gaussians_source, out: *mut c_void, capacity: usize, bytes_out: *mut usize) -> gs_status;
    pub fn gs_device_create(hip_ordinal: i32, out: *mut *mut gs_device) -> gs_status;
    pub fn gs_device_destroy(dev: *mut gs_device);
    pub fn gs_device_limits(dev: *const gs_device, out: *mut gs_limits) -> gs_status;
    pub fn gs_device_synchronize(dev: *mut gs_device) -> gs_status;
    pub fn gs_device_fast_rank(dev: *const gs_device) -> i32;
    pub fn gs_stream_create(dev: *mut gs_device, out: *mut *mut gs_stream) -> gs_status;
    pub fn gs_stream_create_with_priority(dev: *mut gs_device, priority: i32, out: *mut *mut gs_stream) -> gs_status;
    pub fn gs_device_stream_priority_range(dev: *mut gs_device, least: *mut i32, greatest: *mut i32) -> gs_status;
    pub fn gs_stream_wrap(dev: *mut gs_device, hip_stream: *mut c_void, out: *mut *mut gs_stream) -> gs_status;
    pub fn gs_stream_native(s: *const gs_stream) -> *mut c_void;
    pub fn gs_stream_synchronize(s: *mut gs_stream) -> gs_status;
    pub fn gs_stream_destroy(s: *mut gs_stream);
    pub fn gs_buffer_create(dev: *mut gs_device, bytes: usize, init_or_null: *const c_void, out: *mut *mut gs_buffer) -> gs_status;
    pub fn gs_buffer_from_raw(dev: *mut gs_device, device_ptr: *mut c_void, bytes: usize, out: *mut *mut gs_buffer) -> gs_status;
    pub fn gs_buffer_retain(b: *mut gs_buffer) -> *mut gs_buffer;
    pub fn gs_buffer_release(b: *mut gs_buffer);
    pub fn gs_buffer_size(b: *const gs_buffer) -> usize;
    pub fn gs_buffer_device_ptr(b: *const gs_buffer) -> *mut c_void;
    pub fn gs_buffer_write(b: *mut gs_buffer, s: *mut gs_stream, offset: usize, src: *const c_void, bytes: usize) -> gs_status;
    pub fn gs_buffer_download(b: *mut gs_buffer, s: *mut gs_stream, dst: *mut c_void, bytes: usize) -> gs_status;
    pub fn gs_buffer_prepare_download(b: *mut gs_buffer, s: *mut gs_stream, out: *mut *mut gs_download) -> gs_status;
    pub fn gs_download_ready(d: *mut gs_download) -> i32;
    pub fn gs_download_map(d: *mut gs_download, data_out: *mut *const c_void, bytes_out: *mut usize) -> gs_status;
    pub fn gs_download_release(d: *mut gs_download);
    pub fn gs_gaussians_buffer_create(dev: *mut gs_device, sh: u32, cov: u32, pods_or_null: *const c_void, len: usize, out: *mut *mut gs_gaussians_buffer) -> gs_status;
    pub fn gs_gaussians_buffer_create_from_gaussians(dev: *mut gs_device, sh: u32, cov: u32, gaussians: *const gs_gaussian, len: usize, out: *mut *mut gs_gaussians_buffer) -> gs_status;
    pub fn gs_gaussians_buffer_create_from_ply(dev: *mut gs_device, sh: u32, cov: u32, ply: *const gs_ply_gaussian_pod, len: usize, out: *mut *mut gs_gaussians_buffer) -> gs_status;
    pub fn gs_gaussians_buffer_update_range_ply(g: *mut gs_gaussians_buffer, s: *mut gs_stream, start: usize, ply: *const gs_ply_gaussian_pod, count: usize) -> gs_status;
    pub fn gs_gaussians_buffer_create_from_spz(dev: *mut gs_device, sh: u32, cov: u32, bytes: *const c_void, len: usize, header_out: *mut gs_spz_header, out: *mut *mut gs_gaussians_buffer) -> gs_status;
    pub fn gs_gaussians_buffer_create_from_spz_decompressed(dev: *mut gs_device, sh: u32, cov: u32, bytes: *const c_void, len: usize, header_out: *mut gs_spz_header, out: *mut *mut gs_gaussians_buffer) -> gs_status;
    pub fn gs_pack_device_from_ply(dev: *mut gs_device, s: *mut gs_stream, sh: u32, cov: u32, ply_device: *const gs_ply_gaussian_pod, n: usize, pods_device: *mut c_void) -> gs_status;
    pub fn gs_pack_device(dev: *mut gs_device, s: *mut gs_stream, sh: u32, cov: u32, gaussians_device: *const gs_gaussian, n: usize, pods_device: *mut c_void) -> gs_status;
    pub fn gs_gaussians_buffer_from_buffer(buffer: *mut gs_buffer, sh: u32, cov: u32, out: *mut *mut gs_gaussians_buffer) -> gs_status;
    pub fn gs_gaussians_buffer_destroy(g: *mut gs_gaussians_buffer);
    pub fn gs_gaussians_buffer_len(g: *const gs_gaussians_buffer) -> usize;
    pub fn gs_gaussians_buffer_buffer(g: *const gs_gaussians_buffer) -> *mut gs_buffer;
    pub fn gs_gaussians_buffer_sh(g: *const gs_gaussians_buffer) -> u32;
    pub fn gs_gaussians_buffer_cov3d(g: *const gs_gaussians_buffer) -> u32;
    pub fn gs_gaussians_buffer_update(g: *mut gs_gaussians_buffer, s: *mut gs_stream, pods: *const c_void, count: usize) -> gs_status;
    pub fn gs_gaussians_buffer_update_range(g: *mut gs_gaussians_buffer, s: *mut gs_stream, start: usize, pods: *const c_void, count: usize) -> gs_status;
    pub fn gs_gaussians_buffer_update_gaussians(g: *mut gs_gaussians_buffer, s: *mut gs_stream, gaussians: *const gs_gaussian, count: usize) -> gs_status;
    pub fn gs_gaussians_buffer_update_range_gaussians(g: *mut gs_gaussians_buffer, s: *mut gs_stream, start: usize, gaussians: *const gs_gaussian, count: usize) -> gs_status;
    pub fn gs_gaussians_buffer_download(g: *mut gs_gaussians_buffer, s: *mut gs_stream, pods_out: *mut c_void, count: usize) -> gs_status;
    pub fn gs_gaussians_buffer_download_gaussians(g: *mut gs_gaussians_buffer, s: *mut gs_stream, out: *mut gs_gaussian, count: usize) -> gs_status;
    pub fn gs_gaussians_buffer_mark_dirty(g: *mut gs_gaussians_buffer);
    pub fn gs_gaussians_buffer_set_spatial_order(g: *mut gs_gaussians_buffer, enabled: i32) -> gs_status;
    pub fn gs_gaussians_buffer_spatial_order(g: *const gs_gaussians_buffer) -> i32;
    pub fn gs_gaussians_buffer_download_order(g: *mut gs_gaussians_buffer, s: *mut gs_stream, order_out: *mut u32, count: usize) -> gs_status;
    pub fn gs_gaussian_transform_buffer_create(dev: *mut gs_device, out: *mut *mut gs_buffer) -> gs_status;
    pub fn gs_gaussian_transform_buffer_update(b: *mut gs_buffer, s: *mut gs_stream, pod: *const gs_gaussian_transform_pod) -> gs_status;
    pub fn gs_gaussian_transform_buffer_from_buffer(b: *mut gs_buffer) -> gs_status;
    pub fn gs_model_transform_buffer_create(dev: *mut gs_device, out: *mut *mut gs_buffer) -> gs_status;
    pub fn gs_model_transform_buffer_update(b: *mut gs_buffer, s: *mut gs_stream, pod: *const gs_model_transform_pod) -> gs_status;
    pub fn gs_model_transform_buffer_from_buffer(b: *mut gs_buffer) -> gs_status;
    pub fn gs_bundle_create(dev: *mut gs_device, desc: *const gs_bundle_desc, out: *mut *mut gs_bundle) -> gs_status;
    pub fn gs_bundle_create_with_bind_groups(dev: *mut gs_device, desc: *const gs_bundle_desc, resources: *const *const *mut gs_buffer, resource_counts: *const u32, resource_group_count: u32, out: *mut *mut gs_bundle) -> gs_status;
    pub fn gs_bundle_create_from_source(dev: *mut gs_device, desc: *const gs_bundle_source_desc, out: *mut *mut gs_bundle) -> gs_status;
    pub fn gs_bundle_attach_bind_groups(b: *mut gs_bundle, resources: *const *const *mut gs_buffer, resource_counts: *const u32, resource_group_count: u32) -> gs_status;
    pub fn gs_bundle_destroy(b: *mut gs_bundle);
    pub fn gs_bundle_workgroup_size(b: *const gs_bundle) -> u32;
    pub fn gs_bundle_label(b: *const gs_bundle) -> *const c_char;
    pub fn gs_bundle_bind_group_layout_count(b: *const gs_bundle) -> u32;
    pub fn gs_bundle_bind_group_count(b: *const gs_bundle) -> u32;
    pub fn gs_bundle_set_bind_group(b: *mut gs_bundle, index: u32, buffers: *const *mut gs_buffer, count: u32) -> gs_status;
    pub fn gs_bundle_dispatch(b: *mut gs_bundle, s: *mut gs_stream, count: u32) -> gs_status;
    pub fn gs_bundle_dispatch_with_bind_groups(b: *mut gs_bundle, s: *mut gs_stream, count: u32, groups: *const *const *mut gs_buffer, group_counts: *const u32, group_count: u32) -> gs_status;
    pub fn gs_bundle_last_workgroup_count(b: *const gs_bundle) -> u32;
    pub fn gs_camera_look_at(eye: *const f32, target: *const f32, up: *const f32, vfov_radians: f32, width: u32, height: u32, near_plane: f32, far_plane: f32, out: *mut gs_camera);
    pub fn gs_renderer_create(dev: *mut gs_device, out: *mut *mut gs_renderer) -> gs_status;
    pub fn gs_renderer_destroy(r: *mut gs_renderer);
    pub fn gs_renderer_set_timing(r: *mut gs_renderer, enabled: i32) -> gs_status;
    pub fn gs_renderer_reset_stats(r: *mut gs_renderer) -> gs_status;
    pub fn gs_renderer_set_frame_flags_target(r: *mut gs_renderer, device_word: *mut u32) -> gs_status;
    pub fn gs_renderer_stats(r: *mut gs_renderer, out: *mut gs_frame_stats) -> gs_status;
    pub fn gs_render_frame(r: *mut gs_renderer, s: *mut gs_stream, gaussians: *mut gs_gaussians_buffer, gaussian_transform: *const gs_gaussian_transform_pod, model_transform: *const gs_model_transform_pod, camera: *const gs_camera, band_ty0: u32, band_ty1: u32, rgba_out_device: *mut f32) -> gs_status;
    pub fn gs_renderer_wait_frame(r: *mut gs_renderer, out: *mut gs_frame_result) -> gs_status;
    pub fn gs_renderer_sort_info(r: *mut gs_renderer, out: *mut gs_sort_info) -> gs_status;
    pub fn gs_renderer_set_sort_mode(r: *mut gs_renderer, depth_msd: i32, tile_msd: i32) -> gs_status;
    pub fn gs_renderer_set_tile_masks(r: *mut gs_renderer, mode: i32) -> gs_status;
    pub fn gs_renderer_set_rounds(r: *mut gs_renderer, mode: i32, first_round: u32) -> gs_status;
    pub fn gs_renderer_download_projected(r: *mut gs_renderer, proj_out: *mut gs_projected, tiles_touched_out: *mut u32, n: usize) -> gs_status;
    pub fn gs_renderer_download_sorted(r: *mut gs_renderer, keys_out: *mut u64, idx_out: *mut u32, capacity: u64, pairs_out: *mut u64) -> gs_status;
    pub fn gs_renderer_download_ranges(r: *mut gs_renderer, ranges_out: *mut u32, num_tiles: usize) -> gs_status;
    pub fn gs_sort_pairs_u64(dev: *mut gs_device, s: *mut gs_stream, keys: *mut u64, values: *mut u32, count: u64, end_bit: u32) -> gs_status;
    pub fn gs_exclusive_scan_u32(dev: *mut gs_device, s: *mut gs_stream, r#in: *const u32, out: *mut u32, count: u64, total_out: *mut u64) -> gs_status;
}
