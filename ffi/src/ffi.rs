//! `#[repr(C)]` mirrors of include/mirt.h and the functions of libmirt.so the host calls.
//! Not compiled in this repository's image (no Rust toolchain); layouts are those
//! `tests/test_abi.py` checks for the ctypes mirror.
#![allow(dead_code)]
use std::os::raw::{c_char, c_int};

pub const MIRT_OK: c_int = 0;

// enum values of include/mirt.h
pub const SHAPE_SPHERE: u32 = 0;
pub const SHAPE_XY_RECT: u32 = 1;
pub const SHAPE_XZ_RECT: u32 = 2;
pub const SHAPE_YZ_RECT: u32 = 3;
pub const SHAPE_CUBE: u32 = 4;
pub const SHAPE_MOVING_SPHERE: u32 = 5;
pub const MAT_LAMBERTIAN: u32 = 0;
pub const MAT_DIELECTRIC: u32 = 1;
pub const MAT_METAL: u32 = 2;
pub const MAT_DIFFUSE_LIGHT: u32 = 3;
pub const TEX_CONSTANT: u32 = 0;
pub const TEX_CHECKER: u32 = 1;
pub const TEX_NOISE: u32 = 2;
pub const TEX_IMAGE: u32 = 3;
pub const TRAVERSE_FAST: i32 = 1;
pub const SPLIT_TILE: i32 = 0;
pub const SPLIT_SAMPLE: i32 = 1;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MirtShape { pub kind: u32, pub material: u32, pub f: [f64; 10] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MirtMaterial { pub kind: u32, pub texture: u32, pub albedo: [f64; 3], pub param: f64 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MirtTexture { pub kind: u32, pub even: u32, pub odd: u32, pub perlin: u32, pub image: u32, pub _pad: u32,
                         pub color: [f64; 3], pub scale: f64 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct MirtPerlin { pub vecs: [[f64; 3]; 256], pub perm_x: [i32; 256], pub perm_y: [i32; 256], pub perm_z: [i32; 256] }
#[repr(C)] #[derive(Clone, Copy)]
pub struct MirtImage { pub width: u32, pub height: u32, pub rgb: *const u8 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MirtBvhMember { pub is_leaf: u32, pub left: u32, pub right: u32, pub shape: u32, pub bmin: [f64; 3], pub bmax: [f64; 3] }
#[repr(C)]
pub struct MirtSceneDesc {
    pub shapes: *const MirtShape, pub n_shapes: u32,
    pub materials: *const MirtMaterial, pub n_materials: u32,
    pub textures: *const MirtTexture, pub n_textures: u32,
    pub perlins: *const MirtPerlin, pub n_perlins: u32,
    pub images: *const MirtImage, pub n_images: u32,
    pub nodes: *const MirtBvhMember, pub n_nodes: u32,
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MirtCamera {
    pub origin: [f64; 3], pub lower_left_corner: [f64; 3], pub horizontal: [f64; 3], pub vertical: [f64; 3],
    pub u: [f64; 3], pub v: [f64; 3], pub w: [f64; 3], pub lens_radius: f64, pub time_start: f64, pub time_end: f64,
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MirtRenderOpts {
    pub struct_size: u32, pub device: i32, pub use_bvh: i32, pub traversal: i32,
    pub pixel_begin: u32, pub pixel_end: u32, pub sample_begin: u32, pub sample_end: u32,
    pub paths_per_pass: u32, pub profile: i32, pub tile_size: u32, pub tile_count: u32, pub tile_index: u32,
    pub handover_bounce: i32, pub handover_paths: u32,
}
impl MirtRenderOpts {
    pub fn new(use_bounding_volume: bool) -> Self {
        MirtRenderOpts { struct_size: std::mem::size_of::<MirtRenderOpts>() as u32, device: -1,
                         use_bvh: use_bounding_volume as i32, traversal: TRAVERSE_FAST, ..Default::default() }
    }
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MirtMultiOpts { pub struct_size: u32, pub split: i32, pub reduce: i32, pub tile_size: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MirtStats {
    pub samples: u64, pub segments: u64, pub passes: u64, pub launches_extend: u64, pub launches_shade: u64,
    pub ms_total: f64, pub ms_raygen: f64, pub ms_extend: f64, pub ms_bin: f64, pub ms_shade: f64, pub ms_accumulate: f64,
    pub ms_resolve: f64, pub ms_paths: f64, pub launches_paths: u64, pub segments_paths: u64, pub handover_bounce: u32,
    pub reserved: u32,
}
#[repr(C)] pub struct MirtScene { _private: [u8; 0] }

extern "C" {
    pub fn mirt_last_error() -> *const c_char;
    pub fn mirt_device_count() -> c_int;
    pub fn mirt_scene_create(desc: *const MirtSceneDesc, device: c_int, out: *mut *mut MirtScene) -> c_int;
    pub fn mirt_scene_destroy(scene: *mut MirtScene);
    pub fn mirt_build_id() -> *const c_char;
    pub fn mirt_multi_trim() -> c_int;
    pub fn mirt_render(scene: *const MirtScene, camera: *const MirtCamera, background: *const f64, width: u32, height: u32,
                       spp: u32, max_depth: u32, seed: u64, opts: *const MirtRenderOpts, out_rgb: *mut u8,
                       stats: *mut MirtStats) -> c_int;
    /// One frame over several devices from this process: `scenes[i]` = the same description created on device i.
    pub fn mirt_render_multi(scenes: *const *const MirtScene, n_scenes: u32, camera: *const MirtCamera, background: *const f64,
                             width: u32, height: u32, spp: u32, max_depth: u32, seed: u64, opts: *const MirtRenderOpts,
                             multi: *const MirtMultiOpts, out_rgb: *mut u8, stats: *mut MirtStats) -> c_int;
}

/// The reference panics on every failure (config.rs:211,216; bvh.rs:153,156,229): so does the binding.
pub fn check(rc: c_int) {
    if rc != MIRT_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(mirt_last_error()) }.to_string_lossy().into_owned();
        panic!("libmirt status {}: {}", rc, msg);
    }
}
