//! Raw bindings to `librt_hip.so`, the MI355X render core, mirroring `include/rt_hip.h` field for
//! field, plus a small safe wrapper (`GpuScene`) for the two calls the UI crate makes.
//!
//! UNCOMPILED (written without a Rust toolchain); `tests/test_rust_binding.py` checks this file
//! against the C header.  Keep one field or parameter per `pub name: type` / `name: type` item:
//! the checker reads them that way.
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const RT_ABI_VERSION: u32 = 1;

pub const RT_OK: c_int = 0;
pub const RT_E_INVALID: c_int = -1;
pub const RT_E_HIP: c_int = -2;
pub const RT_E_NO_DEVICE: c_int = -3;
pub const RT_E_DEPTH: c_int = -4;
pub const RT_E_IO: c_int = -5;
pub const RT_E_LIMIT: c_int = -6;
pub const RT_E_CANCELLED: c_int = -7;

pub const RT_MULTI_TILES: u32 = 0;
pub const RT_MULTI_SPP: u32 = 1;

pub const RT_HITABLE_SPHERE: u32 = 0;
pub const RT_HITABLE_PLANE: u32 = 1;
pub const RT_HITABLE_MESH: u32 = 2;

pub const RT_MAT_LAMBERTIAN: u32 = 0;
pub const RT_MAT_SPECULAR: u32 = 1;
pub const RT_MAT_DIELECTRIC: u32 = 2;

pub const RT_TEX_COLOR: u32 = 0;
pub const RT_TEX_CHECKER: u32 = 1;
pub const RT_TEX_UV_TEST: u32 = 2;
pub const RT_TEX_XYZ_TEST: u32 = 3;
pub const RT_TEX_COLOR_MAP: u32 = 4;
pub const RT_TEX_MESH: u32 = 5;
pub const RT_TEX_MESH_SET: u32 = 6;

pub const RT_LIGHT_POINT: u32 = 0;
pub const RT_LIGHT_LAMP: u32 = 1;
pub const RT_LIGHT_DIRECTIONAL: u32 = 2;

pub const RT_SKY_DAY: u32 = 0;
pub const RT_SKY_BLACK: u32 = 1;

pub const RT_DEVICE_HOST_ONLY: c_int = -1;

pub const RT_OPT_LENS_SAMPLER: u32 = 1;
pub const RT_OPT_FINE_SPLIT: u32 = 2;
pub const RT_OPT_ROUND_UNITS: u32 = 3;
pub const RT_OPT_LENS_COC_PX: u32 = 4;
pub const RT_OPT_CULL_EMPTY_TILES: u32 = 5;
pub const RT_OPT_DEBUG_STACK_ENTRIES: u32 = 100;
pub const RT_OPT_DEBUG_REBUILD_STACK_LIMIT: u32 = 101;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtV3 {
    pub x: f32,
    pub y: f32,
    pub z: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtV2 {
    pub x: f32,
    pub y: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtCamera {
    pub lower_left_corner: RtV3,
    pub horizontal: RtV3,
    pub vertical: RtV3,
    pub origin: RtV3,
    pub u: RtV3,
    pub v: RtV3,
    pub lens_radius: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtSphere {
    pub radius: f32,
    pub mat_id: u32,
    pub tex_id: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtPlane {
    pub normal: RtV3,
    pub u_basis: RtV3,
    pub v_basis: RtV3,
    pub radius: f32,
    pub has_radius: u32,
    pub mat_id: u32,
    pub tex_id: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtMeshTri {
    pub a: RtV3,
    pub b: RtV3,
    pub c: RtV3,
    pub a_uv: RtV2,
    pub b_uv: RtV2,
    pub c_uv: RtV2,
    pub tex_key: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct RtMesh {
    pub tris: *const RtMeshTri,
    pub tri_count: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtMeshObject {
    pub mesh_index: u32,
    pub mat_id: u32,
    pub tex_id: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtRotation {
    pub axis: RtV3,
    pub theta: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtEntity {
    pub hitable_kind: u32,
    pub hitable_index: u32,
    pub id: i32,
    pub translation_first: u32,
    pub translation_count: u32,
    pub rotation_first: u32,
    pub rotation_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtMaterial {
    pub kind: u32,
    pub reflectivity: f32,
    pub fuzz: f32,
    pub opacity: f32,
    pub ref_index: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtTexture {
    pub kind: u32,
    pub color: RtV3,
    pub color2: RtV3,
    pub size: f32,
    pub child0: i32,
    pub child1: i32,
    pub width: u32,
    pub height: u32,
    pub texel_first: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtLight {
    pub kind: u32,
    pub origin: RtV3,
    pub direction: RtV3,
    pub color: RtV3,
    pub intensity: f32,
    pub angle_deg: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct RtSceneDesc {
    pub abi_version: u32,
    pub camera: RtCamera,
    pub sky: u32,
    pub entities: *const RtEntity,
    pub entity_count: u64,
    pub translations: *const RtV3,
    pub translation_count: u64,
    pub rotations: *const RtRotation,
    pub rotation_count: u64,
    pub spheres: *const RtSphere,
    pub sphere_count: u64,
    pub planes: *const RtPlane,
    pub plane_count: u64,
    pub mesh_objects: *const RtMeshObject,
    pub mesh_object_count: u64,
    pub meshes: *const RtMesh,
    pub mesh_count: u64,
    pub materials: *const RtMaterial,
    pub material_count: u64,
    pub textures: *const RtTexture,
    pub texture_count: u64,
    pub tex_set_entries: *const i32,
    pub tex_set_entry_count: u64,
    pub texels: *const RtV3,
    pub texel_count: u64,
    pub lights: *const RtLight,
    pub light_count: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtSettings {
    pub width: u32,
    pub height: u32,
    pub max_reflections: u32,
    pub samples_per_pixel: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtRenderChunk {
    pub left: u32,
    pub top: u32,
    pub width: u32,
    pub height: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtRayCounters {
    pub primary: u64,
    pub secondary: u64,
    pub shadow: u64,
    pub samples: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtSceneInfo {
    pub entity_count: u64,
    pub mesh_count: u64,
    pub triangle_count: u64,
    pub instanced_triangle_count: u64,
    pub tlas_node_count: u64,
    pub blas_node_count: u64,
    pub tlas_depth: u32,
    pub blas_depth: u32,
    pub device_bytes: u64,
}

/// Opaque scene handle (`typedef struct RtScene RtScene`).
#[repr(C)]
pub struct RtScene {
    _private: [u8; 0],
}

extern "C" {
    pub fn rt_device_count() -> c_int;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_scene_create(desc: *const RtSceneDesc, device: c_int) -> *mut RtScene;
    pub fn rt_scene_destroy(scene: *mut RtScene);
    pub fn rt_scene_update_transforms(scene: *mut RtScene, translations: *const RtV3, translation_count: u64, rotations: *const RtRotation, rotation_count: u64, stream: *mut c_void) -> c_int;
    pub fn rt_scene_rebuild_tlas(scene: *mut RtScene, translations: *const RtV3, translation_count: u64, rotations: *const RtRotation, rotation_count: u64, stream: *mut c_void) -> c_int;
    pub fn rt_scene_rebuild_blas(scene: *mut RtScene, mesh_index: u32, tris: *const RtMeshTri, tri_count: u64, stream: *mut c_void) -> c_int;
    pub fn rt_scene_set_option(scene: *mut RtScene, option: u32, value: f64) -> c_int;
    pub fn rt_render_chunk(scene: *mut RtScene, settings: *const RtSettings, chunk: *const RtRenderChunk, seed: u64, rgb_out: *mut f32, rgba_out: *mut u8, counters: *mut RtRayCounters) -> c_int;
    pub fn rt_render_chunk_device(scene: *mut RtScene, settings: *const RtSettings, chunk: *const RtRenderChunk, seed: u64, rgb_dev: *mut c_void, rgba_dev: *mut c_void, pitch_pixels: u32, out_x: u32, out_y: u32, counters_dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rt_render_chunk_device_noalloc(scene: *mut RtScene, settings: *const RtSettings, chunk: *const RtRenderChunk, seed: u64, rgb_dev: *mut c_void, rgba_dev: *mut c_void, pitch_pixels: u32, out_x: u32, out_y: u32, counters_dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rt_noalloc_max_pixels() -> u64;
    pub fn rt_render_strips_device(scene: *mut RtScene, settings: *const RtSettings, strip_height: u32, strip_first: u32, strip_period: u32, seed: u64, rgb_dev: *mut c_void, rgba_dev: *mut c_void, counters_dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rt_strip_rows(height: u32, strip_height: u32, strip_first: u32, strip_period: u32) -> c_int;
    pub fn rt_render_tile_columns_device(scene: *mut RtScene, settings: *const RtSettings, col_first: u32, col_period: u32, seed: u64, rgb_dev: *mut c_void, rgba_dev: *mut c_void, counters_dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rt_tile_columns_width(width: u32, col_first: u32, col_period: u32) -> c_int;
    pub fn rt_accumulate_chunk_device(scene: *mut RtScene, settings: *const RtSettings, chunk: *const RtRenderChunk, seed: u64, sample_first: u32, sample_count: u32, accum_dev: *mut c_void, counters_dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rt_render_frame_multi(scenes: *const *mut RtScene, n_scenes: u32, settings: *const RtSettings, mode: u32, seed: u64, rgb_dev0: *mut c_void, rgba_dev0: *mut c_void, counters: *mut RtRayCounters) -> c_int;
    pub fn rt_render_frame_multi_host(scenes: *const *mut RtScene, n_scenes: u32, settings: *const RtSettings, mode: u32, seed: u64, rgb_out: *mut f32, rgba_out: *mut u8, counters: *mut RtRayCounters) -> c_int;
    pub fn rt_scene_cancel(scene: *mut RtScene, cancel: c_int) -> c_int;
    pub fn rt_scene_progress(scene: *const RtScene, blocks_done: *mut u64, blocks_total: *mut u64) -> c_int;
    pub fn rt_scene_culled_pixels(scene: *const RtScene, left_out: *mut u64, without_walk: *mut u64) -> c_int;
    pub fn rt_resolve_device(device: c_int, accum_dev: *const c_void, n_pixels: u64, samples_per_pixel: u32, rgb_dev: *mut c_void, rgba_dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rt_primary_hits(scene: *mut RtScene, settings: *const RtSettings, chunk: *const RtRenderChunk, t_out: *mut f32, normal_out: *mut f32, entity_out: *mut i32, tri_out: *mut i32) -> c_int;
    pub fn rt_create_render_chunks(width: u32, height: u32, count_x: u32, count_y: u32, out: *mut RtRenderChunk, capacity: u32) -> c_int;
    pub fn rt_v3_to_rgba(rgb: *const f32, n_pixels: u64, rgba_out: *mut u8);
    pub fn rt_sample_segment() -> u32;
    pub fn rt_scene_info(scene: *const RtScene, out: *mut RtSceneInfo) -> c_int;
    pub fn rt_scene_bvh_order(scene: *const RtScene, mesh_index: i64, out: *mut u32, capacity: u64) -> i64;
    pub fn rt_scene_profile_enable(scene: *mut RtScene, enable: c_int) -> c_int;
    pub fn rt_scene_profile_read(scene: *mut RtScene, kernel_ms_total: *mut f64, launches: *mut u64) -> c_int;
}

// ---------------------------------------------------------------------------------------------
// Safe wrapper: what `src/job_constructing.rs` and `src/render.rs` of the UI crate use.
// ---------------------------------------------------------------------------------------------

/// The message `rt_last_error()` holds for the calling thread.
pub fn last_error() -> String {
    unsafe {
        let p = rt_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}

/// An error of the render core: the `RT_E_*` code and its message.
#[derive(Debug, Clone)]
pub struct GpuError {
    pub code: c_int,
    pub message: String,
}

impl std::fmt::Display for GpuError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "rt_hip error {}: {}", self.code, self.message)
    }
}

impl std::error::Error for GpuError {}

fn check(code: c_int) -> Result<(), GpuError> {
    if code >= 0 { Ok(()) } else { Err(GpuError { code, message: last_error() }) }
}

/// A scene resident on one GPU: acceleration structures built and uploaded by `rt_scene_create`
/// (what `Scene::build_bvh` does on the CPU).  Immutable after creation and internally serialised,
/// so it is shared by the worker threads like the `Arc<(Scene, RenderSettings)>` it stands in for.
pub struct GpuScene {
    handle: *mut RtScene,
}

unsafe impl Send for GpuScene {}
unsafe impl Sync for GpuScene {}

impl GpuScene {
    /// `desc` and everything it points to only have to live for the duration of this call.
    pub fn create(desc: &RtSceneDesc, device: i32) -> Result<GpuScene, GpuError> {
        let handle = unsafe { rt_scene_create(desc as *const RtSceneDesc, device) };
        if handle.is_null() {
            return Err(GpuError { code: RT_E_INVALID, message: last_error() });
        }
        Ok(GpuScene { handle })
    }

    /// The pixel loop of `start_render_thread` for one chunk: RGBA8 bytes, row-major, `v3_to_rgba`
    /// already applied.  Blocks until the pixels are in `rgba`.
    pub fn render_chunk_rgba(&self, settings: &RtSettings, chunk: &RtRenderChunk, seed: u64, rgba: &mut [u8]) -> Result<RtRayCounters, GpuError> {
        let need = chunk.width as usize * chunk.height as usize * 4;
        if rgba.len() < need {
            return Err(GpuError { code: RT_E_INVALID, message: format!("rgba buffer holds {} bytes, chunk needs {}", rgba.len(), need) });
        }
        let mut counters = RtRayCounters::default();
        check(unsafe { rt_render_chunk(self.handle, settings, chunk, seed, std::ptr::null_mut(), rgba.as_mut_ptr(), &mut counters) })?;
        Ok(counters)
    }

    pub fn info(&self) -> Result<RtSceneInfo, GpuError> {
        let mut out = RtSceneInfo::default();
        check(unsafe { rt_scene_info(self.handle, &mut out) })?;
        Ok(out)
    }

    pub fn set_option(&self, option: u32, value: f64) -> Result<(), GpuError> {
        check(unsafe { rt_scene_set_option(self.handle, option, value) })
    }

    /// Animated scenes (`scene_interceptor_spin.rs:34-40` rebuilds everything per frame): new transforms for the same entities.
    /// `rebuild == false`: the TLAS is refitted on the device (hits bit-identical to a fresh scene); `rebuild == true`: a new
    /// TLAS is built on the device (for entities that travel; nearest hits are the reference's, the tree is not).
    /// `&mut self`: no render call may be in flight.
    pub fn move_entities(&mut self, translations: &[RtV3], rotations: &[RtRotation], rebuild: bool) -> Result<(), GpuError> {
        let (t, nt, r, nr) = (translations.as_ptr(), translations.len() as u64, rotations.as_ptr(), rotations.len() as u64);
        check(unsafe {
            if rebuild { rt_scene_rebuild_tlas(self.handle, t, nt, r, nr, std::ptr::null_mut()) }
            else { rt_scene_update_transforms(self.handle, t, nt, r, nr, std::ptr::null_mut()) }
        })
    }

    /// The cancellation token of `src/render.rs:131,149` reaching INTO a running call: every launch on the scene stops handing
    /// out work, the blocked `render_chunk_rgba` returns `RT_E_CANCELLED` within milliseconds.  `cancel(false)` re-arms the scene.
    pub fn cancel(&self, cancel: bool) -> Result<(), GpuError> {
        check(unsafe { rt_scene_cancel(self.handle, if cancel { 1 } else { 0 }) })
    }

    /// (blocks handed out so far - an estimate -, blocks of the call in flight): what the 800 ms snapshot of
    /// `src/render.rs:156-158` becomes when the pixel loop is one GPU call.
    pub fn progress(&self) -> Result<(u64, u64), GpuError> {
        let (mut done, mut total) = (0u64, 0u64);
        check(unsafe { rt_scene_progress(self.handle, &mut done, &mut total) })?;
        Ok((done, total))
    }

    /// What the empty-tile packet test (`RT_OPT_CULL_EMPTY_TILES`) did in the latest call: (pixels left out under a black sky,
    /// pixels rendered without a walk under a sky that depends on the ray).
    pub fn culled_pixels(&self) -> Result<(u64, u64), GpuError> {
        let (mut left_out, mut without_walk) = (0u64, 0u64);
        check(unsafe { rt_scene_culled_pixels(self.handle, &mut left_out, &mut without_walk) })?;
        Ok((left_out, without_walk))
    }

    /// A deforming mesh: new vertex positions for mesh `mesh_index` (same triangle count); its tree is rebuilt on the device and the
    /// TLAS above its instances refitted (`MeshBvhRoot::new`, mesh.rs:84-90, per frame on the CPU in the reference).
    pub fn rebuild_mesh(&mut self, mesh_index: u32, tris: &[RtMeshTri]) -> Result<(), GpuError> {
        check(unsafe { rt_scene_rebuild_blas(self.handle, mesh_index, tris.as_ptr(), tris.len() as u64, std::ptr::null_mut()) })
    }

    pub fn as_ptr(&self) -> *mut RtScene {
        self.handle
    }
}

/// One frame on several GPUs, composited in the first replica's device memory (`rt_render_frame_multi`): the worker-thread
/// pool of `src/render.rs:186-224` across devices.  `replicas[i]` is the same scene created on device i.
/// The same frame as RGBA8 bytes on the host (what `job_running.rs:37-41` blits into the egui texture).
pub fn render_frame_multi_rgba(replicas: &[&GpuScene], settings: &RtSettings, mode: u32, seed: u64, rgba: &mut [u8]) -> Result<RtRayCounters, GpuError> {
    let need = settings.width as usize * settings.height as usize * 4;
    if rgba.len() < need {
        return Err(GpuError { code: RT_E_INVALID, message: format!("rgba buffer holds {} bytes, the frame needs {}", rgba.len(), need) });
    }
    let handles: Vec<*mut RtScene> = replicas.iter().map(|s| s.as_ptr()).collect();
    let mut counters = RtRayCounters::default();
    check(unsafe { rt_render_frame_multi_host(handles.as_ptr(), handles.len() as u32, settings, mode, seed, std::ptr::null_mut(), rgba.as_mut_ptr(), &mut counters) })?;
    Ok(counters)
}

pub fn render_frame_multi(replicas: &[&GpuScene], settings: &RtSettings, mode: u32, seed: u64, rgb_dev0: *mut c_void, rgba_dev0: *mut c_void) -> Result<RtRayCounters, GpuError> {
    let handles: Vec<*mut RtScene> = replicas.iter().map(|s| s.as_ptr()).collect();
    let mut counters = RtRayCounters::default();
    check(unsafe { rt_render_frame_multi(handles.as_ptr(), handles.len() as u32, settings, mode, seed, rgb_dev0, rgba_dev0, &mut counters) })?;
    Ok(counters)
}

impl Drop for GpuScene {
    fn drop(&mut self) {
        unsafe { rt_scene_destroy(self.handle) }
    }
}
