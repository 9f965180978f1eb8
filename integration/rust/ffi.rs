// ffi.rs — mirrors include/rt_abi.h one to one (tests/test_rust_shim_cpu.py checks names and arities).
// Uncompiled: no Rust toolchain in the build image.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_ABI_VERSION: c_int = 2;
pub const RT_NOALLOC_MAX_PIXELS: u32 = 1 << 24;

pub const RT_OK: c_int = 0;
pub const RT_E_INVALID_ARG: c_int = -1;
pub const RT_E_HIP: c_int = -2;
pub const RT_E_OOM: c_int = -3;
pub const RT_E_UNSUPPORTED: c_int = -4;
pub const RT_E_INTERNAL: c_int = -5;

#[repr(C)] pub struct rt_ctx { _p: [u8; 0] }
#[repr(C)] pub struct rt_scene { _p: [u8; 0] }
#[repr(C)] pub struct rt_multi { _p: [u8; 0] }

// rt_prim_type
pub const RT_PRIM_SPHERE: u32 = 0;
pub const RT_PRIM_MOVING_SPHERE: u32 = 1;
pub const RT_PRIM_QUAD: u32 = 2;
pub const RT_PRIM_MEDIUM_SPHERE: u32 = 3;
pub const RT_PRIM_MEDIUM_BOX: u32 = 4;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_prim {
    pub type_: u32, pub material_id: u32, pub object_id: i32, pub reserved: u32,
    pub a: [f64; 3], pub b: [f64; 3], pub c: [f64; 3], pub d: [f64; 3],
    pub radius: f64, pub density: f64,
}

// rt_material_kind
pub const RT_MAT_LAMBERTIAN: u32 = 0;
pub const RT_MAT_METAL: u32 = 1;
pub const RT_MAT_DIELECTRIC: u32 = 2;
pub const RT_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const RT_MAT_ISOTROPIC: u32 = 4;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_material { pub kind: u32, pub tex: i32, pub albedo: [f64; 3], pub fuzz: f64, pub ior: f64 }

// rt_texture_kind
pub const RT_TEX_SOLID: u32 = 0;
pub const RT_TEX_CHECKER: u32 = 1;
pub const RT_TEX_IMAGE: u32 = 2;
pub const RT_TEX_NOISE: u32 = 3;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_texture {
    pub kind: u32, pub odd: i32, pub even: i32, pub width: u32, pub height: u32, pub reserved: u32,
    pub texel_offset: u64, pub rgb: [f64; 3], pub inv_scale: f64,
}

#[repr(C)]
pub struct rt_scene_desc {
    pub prims: *const rt_prim, pub n_prims: u64,
    pub materials: *const rt_material, pub n_materials: u64,
    pub textures: *const rt_texture, pub n_textures: u64,
    pub texels: *const u8, pub n_texel_bytes: u64,
}

pub const RT_SAMPLER_UNIFORM: u32 = 0;
pub const RT_SAMPLER_RANDOM: u32 = 1;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_camera {
    pub width: u32, pub height: u32, pub sampler_kind: u32, pub sampler_n: u32,
    pub max_depth: u32, pub background_mode: u32,
    pub center: [f64; 3], pub pixel00: [f64; 3], pub du: [f64; 3], pub dv: [f64; 3],
    pub defocus_u: [f64; 3], pub defocus_v: [f64; 3], pub defocus_angle: f64, pub background: [f64; 3],
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_ray { pub o: [f64; 3], pub d: [f64; 3], pub time: f64 }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_hit {
    pub object_id: i32, pub front_face: i32, pub t: f64, pub p: [f64; 3], pub n: [f64; 3], pub u: f64, pub v: f64,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_stats {
    pub paths: u64, pub rays: u64, pub passes: u32, pub bounces_run: u32,
    pub ms_total: f32, pub ms_generate: f32, pub ms_intersect: f32, pub ms_shade: f32, pub ms_accumulate: f32,
    pub intersect_launches: u64, pub probe_rays: u64, pub probe_node_visits: u64, pub probe_prim_tests: u64,
    pub dbg: [f32; 16],
    pub samples_nan: u64, pub samples_clamped: u64, pub dbg_counts: [u64; 12], pub lib_alloc_calls: u64,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_multi_stats {
    pub n_devices: u32, pub reserved: u32, pub paths: u64, pub rays: u64,
    pub ms_render: [f32; 16], pub ms_reduce: f32, pub ms_copy: f32,
}

extern "C" {
    pub fn rt_init(device_id: c_int, out: *mut *mut rt_ctx) -> c_int;
    pub fn rt_shutdown(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_last_error(ctx: *mut rt_ctx) -> *const c_char;
    pub fn rt_abi_version() -> c_int;
    pub fn rt_scene_upload(ctx: *mut rt_ctx, desc: *const rt_scene_desc, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_free(ctx: *mut rt_ctx, scene: *mut rt_scene) -> c_int;
    pub fn rt_intersect(ctx: *mut rt_ctx, scene: *const rt_scene, rays: *const rt_ray, n: u64, out_hits: *mut rt_hit) -> c_int;
    pub fn rt_render(ctx: *mut rt_ctx, scene: *const rt_scene, cam: *const rt_camera, seed: u64, sample_begin: u32,
                     sample_end: u32, sample_stride: u32, out_sum: *mut f32) -> c_int;
    pub fn rt_render_device(ctx: *mut rt_ctx, scene: *const rt_scene, cam: *const rt_camera, seed: u64, sample_begin: u32,
                            sample_end: u32, sample_stride: u32, d_out_sum: *mut f32, hip_stream: *mut c_void) -> c_int;
    pub fn rt_render_device_noalloc(ctx: *mut rt_ctx, scene: *const rt_scene, cam: *const rt_camera, seed: u64,
                                    sample_begin: u32, sample_end: u32, sample_stride: u32, d_out_sum: *mut f32,
                                    hip_stream: *mut c_void) -> c_int;
    pub fn rt_render_batch_device(ctx: *mut rt_ctx, scene: *const rt_scene, cam: *const rt_camera, n_frames: u32, seeds: *const u64,
                                  sample_begins: *const u32, sample_end: u32, sample_stride: u32, d_out_sums: *const *mut f32,
                                  hip_stream: *mut c_void) -> c_int;
    pub fn rt_resolve(ctx: *mut rt_ctx, sum: *const f32, width: u32, height: u32, spp: u32, gamma: f64,
                      out_rgb: *mut f64, out_rgb8: *mut u8) -> c_int;
    pub fn rt_resolve_device(ctx: *mut rt_ctx, d_sum: *const f32, width: u32, height: u32, spp: u32, gamma: f64,
                             d_out_rgb: *mut f64, d_out_rgb8: *mut u8, hip_stream: *mut c_void) -> c_int;
    pub fn rt_get_stats(ctx: *mut rt_ctx, out: *mut rt_stats) -> c_int;
    pub fn rt_set_option(ctx: *mut rt_ctx, key: *const c_char, value: i64) -> c_int;
    pub fn rt_philox4x32_10(ctr: *const u32, key: *const u32, out: *mut u32);
    pub fn rt_half_round_bits(x: f32, up: c_int) -> u32;
    pub fn rt_synchronize(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_multi_init(device_ids: *const c_int, n_devices: c_int, out: *mut *mut rt_multi) -> c_int;
    pub fn rt_multi_shutdown(m: *mut rt_multi) -> c_int;
    pub fn rt_multi_last_error(m: *mut rt_multi) -> *const c_char;
    pub fn rt_multi_device_count(m: *mut rt_multi) -> c_int;
    pub fn rt_multi_scene_upload(m: *mut rt_multi, desc: *const rt_scene_desc) -> c_int;
    pub fn rt_multi_render(m: *mut rt_multi, cam: *const rt_camera, seed: u64, sample_begin: u32, sample_end: u32,
                           sample_stride: u32, out_sum: *mut f32) -> c_int;
    pub fn rt_multi_render_device(m: *mut rt_multi, cam: *const rt_camera, seed: u64, sample_begin: u32, sample_end: u32,
                                  sample_stride: u32, d_out_sum_root: *mut f32) -> c_int;
    pub fn rt_multi_set_option(m: *mut rt_multi, key: *const c_char, value: i64) -> c_int;
    pub fn rt_multi_get_stats(m: *mut rt_multi, out: *mut rt_multi_stats) -> c_int;
}

/// Turns a negative rt_status into a panic carrying rt_last_error — the reference's own error style
/// (expect()/unwrap() at builder.rs:58-72, mod.rs:63,103,115).
pub unsafe fn check(rc: c_int, ctx: *mut rt_ctx) {
    if rc != RT_OK {
        let msg = std::ffi::CStr::from_ptr(rt_last_error(ctx)).to_string_lossy().into_owned();
        panic!("librt_mi355x: error {}: {}", rc, msg);
    }
}
pub unsafe fn check_multi(rc: c_int, m: *mut rt_multi) {
    if rc != RT_OK {
        let msg = std::ffi::CStr::from_ptr(rt_multi_last_error(m)).to_string_lossy().into_owned();
        panic!("librt_mi355x: error {}: {}", rc, msg);
    }
}
