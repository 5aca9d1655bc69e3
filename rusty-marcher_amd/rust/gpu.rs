//! engine/src/gpu.rs -- Rust side of the MI355X backend (drop-in for the Rayon patch
//! loop of renderer.rs:63-108).  UNVERIFIED: written against include/rusty_marcher_amd.h,
//! not compiled -- the build image has no rustc/cargo.
//!
//! Edition 2015 like the rest of the crate (bare `use geometry::..` paths).
//! Link with:  cargo:rustc-link-lib=dylib=rusty_marcher_amd  (build.rs)

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::ptr;

use framebuffer::FrameBuffer;
use geometry::Vec3f;
use shapes::Reflectance;

// ---- mirrors of the C structs (include/rusty_marcher_amd.h) -------------------------

#[repr(C)]
#[derive(Copy, Clone)]
pub struct RmVec3 {
    pub x: f64,
    pub y: f64,
    pub z: f64,
}

impl From<Vec3f> for RmVec3 {
    fn from(v: Vec3f) -> RmVec3 {
        RmVec3 { x: v.x, y: v.y, z: v.z }
    }
}

#[repr(C)]
#[derive(Copy, Clone)]
pub struct RmReflectance {
    pub diffusion: f64,
    pub diffuse_color: RmVec3,
    pub specular: f64,
    pub specular_exponent: f64,
    pub is_glass_like: i32,
    pub _pad: i32,
    pub reflection: f64,
    pub refractive_index: f64,
}

impl From<Reflectance> for RmReflectance {
    fn from(r: Reflectance) -> RmReflectance {
        RmReflectance {
            diffusion: r.diffusion,
            diffuse_color: r.diffuse_color.into(),
            specular: r.specular,
            specular_exponent: r.specular_exponent,
            is_glass_like: r.is_glass_like as i32,
            _pad: 0,
            reflection: r.reflection,
            refractive_index: r.refractive_index,
        }
    }
}

#[repr(C)]
pub struct RmSceneDesc {
    pub shapes: *const c_void,
    pub n_shapes: u32,
    pub spheres: *const c_void,
    pub n_spheres: u32,
    pub polygons: *const c_void,
    pub n_polygons: u32,
    pub polygon_vertices: *const RmVec3,
    pub n_polygon_vertices: u32,
    pub triangles: *const c_void,
    pub n_triangles: u32,
    pub lights: *const c_void,
    pub n_lights: u32,
    pub camera: RmVec3,
}

#[repr(C)]
pub struct RmParams {
    pub fov: f64,
    pub half_fov: f64,
    pub height: f64,
    pub width: f64,
    pub ratio: f64,
    pub frame_width: u32,
    pub frame_height: u32,
    pub max_depth: u32,
    pub patch_size: u32,
    pub background: RmVec3,
    pub patch_row_begin: u32,
    pub patch_row_end: u32,
    pub flags: u32,
    pub patch_row_stride: u32,
}

#[repr(C)]
#[derive(Default)]
pub struct RmTiming {
    pub kernel_ms: f64,
    pub d2h_ms: f64,
    pub total_ms: f64,
}

#[repr(C)]
#[derive(Default)]
pub struct RmFrameTimes {
    pub kernel_ms: f64,
    pub gather_ms: f64,
    pub total_ms: f64,
}

pub enum RmScene {}
pub enum RmCtx {}

// Every entry point of include/rusty_marcher_amd.h, in its order (tests/test_rust_binding.py
// checks names, arity and argument classes against the header).
#[link(name = "rusty_marcher_amd")]
#[allow(dead_code)]
extern "C" {
    fn rm_reflectance_default(out: *mut RmReflectance);
    fn rm_create_renderer(fov: f64, height: f64, width: f64, out: *mut RmParams);
    fn rm_scene_new(out: *mut *mut RmScene) -> c_int;
    fn rm_scene_free(scene: *mut RmScene);
    fn rm_scene_create_default(out: *mut *mut RmScene) -> c_int;
    fn rm_scene_add_sphere(scene: *mut RmScene, center: RmVec3, radius: f64, r: *const RmReflectance) -> c_int;
    fn rm_scene_add_polygon(scene: *mut RmScene, vertices: *const RmVec3, n_vertices: u32, r: *const RmReflectance) -> c_int;
    fn rm_scene_add_mesh(scene: *mut RmScene, tri_xyz: *const f64, n_triangles: u32, offset: RmVec3) -> c_int;
    fn rm_scene_add_light(scene: *mut RmScene, position: RmVec3, color: RmVec3, intensity: f64) -> c_int;
    fn rm_scene_offset_shape(scene: *mut RmScene, shape_index: u32, offset: RmVec3) -> c_int;
    fn rm_scene_set_camera(scene: *mut RmScene, camera: RmVec3) -> c_int;
    fn rm_scene_offset_camera(scene: *mut RmScene, offset: RmVec3) -> c_int;
    fn rm_scene_load_obj(scene: *mut RmScene, path: *const c_char, offset: RmVec3, n_models_out: *mut u32) -> c_int;
    fn rm_scene_open_obj(path: *const c_char, out: *mut *mut RmScene) -> c_int;
    fn rm_scene_get_desc(scene: *const RmScene, out: *mut RmSceneDesc) -> c_int;
    fn rm_format_status(buf: *mut c_char, buflen: usize, ms: u64, frame_width: u32, frame_height: u32) -> c_int;
    fn rm_init(device_ordinal: c_int, out: *mut *mut RmCtx) -> c_int;
    fn rm_destroy(ctx: *mut RmCtx);
    fn rm_last_error(ctx: *const RmCtx) -> *const c_char;
    fn rm_scene_upload(ctx: *mut RmCtx, desc: *const RmSceneDesc) -> c_int;
    fn rm_scene_uploads(ctx: *mut RmCtx, calls: *mut u64, copies: *mut u64) -> c_int;
    fn rm_camera_update(ctx: *mut RmCtx, camera: RmVec3) -> c_int;
    fn rm_render(ctx: *mut RmCtx, params: *const RmParams, host_rgb: *mut f64, timing: *mut RmTiming) -> c_int;
    fn rm_render_rows(ctx: *mut RmCtx, params: *const RmParams, rows: *const *mut f64, timing: *mut RmTiming) -> c_int;
    fn rm_render_display(ctx: *mut RmCtx, params: *const RmParams, host_rgb8: *mut u8, timing: *mut RmTiming) -> c_int;
    fn rm_fetch_rows(ctx: *mut RmCtx, rows: *const *mut f64, frame_width: u32, frame_height: u32, patch_row_begin: u32, patch_row_end: u32) -> c_int;
    fn rm_hostio_stats(ctx: *mut RmCtx, bytes_copied: *mut u64, patches: *mut u64, patches_sent: *mut u64, threads: *mut c_int) -> c_int;
    fn rm_render_device(ctx: *mut RmCtx, params: *const RmParams, device_rgb: *mut c_void, hip_stream: *mut c_void) -> c_int;
    fn rm_render_device_u8(ctx: *mut RmCtx, params: *const RmParams, device_rgb: *mut c_void, device_rgb8: *mut c_void, hip_stream: *mut c_void) -> c_int;
    fn rm_tile_stats(ctx: *mut RmCtx, hip_stream: *mut c_void, tiles: *mut u32, tiles_listed: *mut u32) -> c_int;
    fn rm_launch_stats(ctx: *mut RmCtx, workgroups: *mut u32, tail_patches: *mut u32) -> c_int;
    fn rm_device_framebuffer(ctx: *mut RmCtx, device_rgb: *mut *mut c_void, bytes: *mut usize) -> c_int;
    fn rm_postprocess(ctx: *mut RmCtx, device_rgb: *mut c_void, frame_width: u32, frame_height: u32, normalize: c_int, host_rgb8: *mut u8, max_out: *mut f64) -> c_int;
    // interactive loop / several GPUs (one process per GPU): host/rm_walk.cpp is the same
    // sequence in C++ -- rm_camera_update between frames, up to 4 frames in flight
    fn rm_buffer_alloc(ctx: *mut RmCtx, bytes: usize, device_ptr: *mut *mut c_void) -> c_int;
    fn rm_buffer_free(ctx: *mut RmCtx, device_ptr: *mut c_void);
    fn rm_buffer_read(ctx: *mut RmCtx, device_ptr: *const c_void, host_dst: *mut c_void, bytes: usize) -> c_int;
    fn rm_buffer_write(ctx: *mut RmCtx, device_ptr: *mut c_void, host_src: *const c_void, bytes: usize) -> c_int;
    fn rm_host_alloc(ctx: *mut RmCtx, bytes: usize, host_ptr: *mut *mut c_void) -> c_int;
    fn rm_host_free(ctx: *mut RmCtx, host_ptr: *mut c_void);
    fn rm_comm_unique_id(id_out: *mut c_void) -> c_int; // 128 bytes
    fn rm_comm_init(ctx: *mut RmCtx, id: *const c_void, rank: c_int, world: c_int) -> c_int;
    fn rm_comm_destroy(ctx: *mut RmCtx);
    fn rm_comm_exchange(ctx: *mut RmCtx, all_ranks: c_int) -> c_int;
    fn rm_exchange_layout(params: *const RmParams, world: c_int, rows_per_rank: *mut u32, chunk_bytes: *mut usize) -> c_int;
    fn rm_frame_submit(ctx: *mut RmCtx, params: *const RmParams, device_rgb: *mut c_void, device_gather8: *mut c_void, device_display8: *mut c_void, slot: u32) -> c_int;
    fn rm_frame_submit_to_host(ctx: *mut RmCtx, params: *const RmParams, device_rgb: *mut c_void, device_gather8: *mut c_void, device_display8: *mut c_void, host_display8: *mut c_void, slot: u32) -> c_int;
    fn rm_frame_wait(ctx: *mut RmCtx, slot: u32) -> c_int;
    fn rm_frame_wait_for(ctx: *mut RmCtx, slot: u32, timeout_ms: u32) -> c_int;
    fn rm_frame_submit_f64(ctx: *mut RmCtx, params: *const RmParams, device_gather64: *mut c_void, device_frame64: *mut c_void, slot: u32) -> c_int;
    fn rm_frame_timing_enable(ctx: *mut RmCtx, on: c_int) -> c_int;
    fn rm_frame_timing(ctx: *mut RmCtx, slot: u32, out: *mut RmFrameTimes) -> c_int;
    fn rm_comm_info(ctx: *mut RmCtx, rank: *mut c_int, world: *mut c_int, n_communicators: *mut c_int) -> c_int;
    fn rm_abi_version() -> u32;
    fn rm_build_info() -> *const c_char;
    fn rm_device_info(ctx: *mut RmCtx, name_buf: *mut c_char, buflen: usize, n_cus: *mut c_int, lds_bytes: *mut usize) -> c_int;
    fn rm_kernel_name(ctx: *mut RmCtx, params: *const RmParams, buf: *mut c_char, buflen: usize) -> c_int;
}

/// The reference's failure mode on this path is a panic.
fn check(status: c_int, ctx: *const RmCtx) {
    if status != 0 {
        let msg = unsafe { CStr::from_ptr(rm_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("rusty_marcher_amd: status {}: {}", status, msg);
    }
}

/// What `trait Shape` lacks for a GPU backend: a way to hand over the primitive's
/// parameters (sphere.rs:6-11 and polygon.rs:6-12 keep their fields private).
/// Each implementor adds itself to the flat scene; see INTEGRATION.md for the three
/// five-line impls.
pub struct SceneSink {
    scene: *mut RmScene,
}

impl SceneSink {
    pub fn sphere(&mut self, center: Vec3f, radius: f64, r: Reflectance) {
        let rr: RmReflectance = r.into();
        check(unsafe { rm_scene_add_sphere(self.scene, center.into(), radius, &rr) }, ptr::null());
    }
    pub fn polygon(&mut self, vertices: &[Vec3f], r: Reflectance) {
        let v: Vec<RmVec3> = vertices.iter().map(|p| (*p).into()).collect();
        let rr: RmReflectance = r.into();
        check(unsafe { rm_scene_add_polygon(self.scene, v.as_ptr(), v.len() as u32, &rr) }, ptr::null());
    }
    /// One `Obj`: triangles as 9 f64 each (already offset: pass Vec3f::zero()), or the
    /// un-offset vertices plus the offsets applied so far through `offset_last`.
    pub fn mesh(&mut self, tri_xyz: &[f64], offset: Vec3f) {
        check(
            unsafe { rm_scene_add_mesh(self.scene, tri_xyz.as_ptr(), (tri_xyz.len() / 9) as u32, offset.into()) },
            ptr::null(),
        );
    }
    pub fn offset_shape(&mut self, index: u32, off: Vec3f) {
        check(unsafe { rm_scene_offset_shape(self.scene, index, off.into()) }, ptr::null());
    }
}

/// One GPU.  Owned by `Renderer` (renderer.rs:17-23 gains a `gpu: RefCell<Gpu>` field) or by
/// `Win`.  The page-locked staging frame the copies go through belongs to the library's context
/// (include/rusty_marcher_amd.h, rm_render_rows): nothing to allocate, grow or free here.
pub struct Gpu {
    ctx: *mut RmCtx,
}

/// `FrameBuffer.buffer` is `Vec<Vec<Vec3f>>` (framebuffer.rs:6-10): one heap allocation per scan
/// line.  With `#[repr(C)]` on `Vec3f` (geometry.rs:4-8: three f64, x y z -- INTEGRATION.md
/// section 3d, the fourth one-line patch) a row is `width * 3` doubles and the library fills the
/// rows in place.
///
/// `buffer`, `width` and `height` are all `pub` (framebuffer.rs:6-10), so nothing but this check stands between a
/// FrameBuffer whose fields disagree and a write beyond a row's allocation: the library writes `width * 3` doubles
/// into each of `height` rows.  A panic here is the reference's own failure mode for such a frame (its scatter,
/// renderer.rs:92-108, indexes out of bounds).
fn row_pointers(frame: &mut FrameBuffer) -> Vec<*mut f64> {
    // the cast below reads a Vec3f as three consecutive f64: true only with `#[repr(C)]`
    const _VEC3F_IS_THREE_F64: [(); 24] = [(); ::std::mem::size_of::<Vec3f>()];
    assert!(frame.buffer.len() >= frame.height, "FrameBuffer: {} rows for a height of {}", frame.buffer.len(), frame.height);
    for (y, row) in frame.buffer.iter().enumerate() {
        assert!(row.len() == frame.width, "FrameBuffer: row {} holds {} pixels for a width of {}", y, row.len(), frame.width);
    }
    frame.buffer.iter_mut().map(|row| row.as_mut_ptr() as *mut f64).collect()
}

impl Gpu {
    pub fn new(device: i32) -> Gpu {
        let mut ctx: *mut RmCtx = ptr::null_mut();
        check(unsafe { rm_init(device, &mut ctx) }, ptr::null());
        Gpu { ctx }
    }

    /// flatten Scene -> rm_scene (shapes in list order: ties, shapes.rs:130) and hand it to the
    /// context.  render() gets the whole Scene every call (main.rs:331-333), so the flat copy is
    /// rebuilt every call (microseconds for the reference's scenes); rm_scene_upload compares
    /// it with the description the resident device image was built from and copies nothing when
    /// they are the same -- camera moves included, the camera is a kernel argument.
    fn upload(&mut self, scene: &::scene::Scene) {
        let mut raw: *mut RmScene = ptr::null_mut();
        check(unsafe { rm_scene_new(&mut raw) }, ptr::null());
        {
            let mut sink = SceneSink { scene: raw };
            for shape in &scene.shapes {
                shape.describe(&mut sink); // the added trait method
            }
        }
        for l in &scene.lights {
            // colours are already L-inf normalised (lights.rs:10-16); normalising again is a no-op
            check(unsafe { rm_scene_add_light(raw, l.position.into(), l.color.into(), l.intensity) }, ptr::null());
        }
        check(unsafe { rm_scene_set_camera(raw, scene.camera.into()) }, ptr::null());
        let mut desc: RmSceneDesc = unsafe { ::std::mem::zeroed() };
        check(unsafe { rm_scene_get_desc(raw, &mut desc) }, ptr::null());
        let st = unsafe { rm_scene_upload(self.ctx, &desc) };
        unsafe { rm_scene_free(raw) };
        check(st, self.ctx);
    }

    fn params(fov: f64, height: f64, width: f64, frame_width: usize, frame_height: usize) -> RmParams {
        if (frame_height % 32 != 0) || (frame_width % 32 != 0) {
            println!("Dimensions mismatch") // renderer.rs:49-51
        }
        let mut p: RmParams = unsafe { ::std::mem::zeroed() };
        unsafe { rm_create_renderer(fov, height, width, &mut p) };
        p.frame_width = frame_width as u32; // renderer.rs:53-54 read the FrameBuffer, not the Renderer
        p.frame_height = frame_height as u32;
        p.max_depth = 3; // renderer.rs:262
        p
    }

    fn status(now: ::std::time::Instant, frame_width: usize, frame_height: usize) -> String {
        // renderer.rs:111-125
        let ms = now.elapsed().as_secs() * 1_000 + u64::from(now.elapsed().subsec_nanos()) / 1_000_000;
        let mut buf = [0 as c_char; 256];
        unsafe { rm_format_status(buf.as_mut_ptr(), buf.len(), ms, frame_width as u32, frame_height as u32) };
        let message = unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
        println!("{}", message);
        message
    }

    /// Body of `Renderer::render` (renderer.rs:36-126) with the Rayon loop and the serial
    /// scatter replaced by one library call that fills `frame.buffer`'s rows in place.  The rows
    /// below the last whole patch row keep their previous contents (renderer.rs:53): the library
    /// never touches them.
    pub fn render(
        &mut self,
        fov: f64,
        height: f64,
        width: f64,
        frame: &mut FrameBuffer,
        scene: &::scene::Scene,
    ) -> String {
        let now = ::std::time::Instant::now();
        self.upload(scene);
        let p = Gpu::params(fov, height, width, frame.width, frame.height);
        let rows = row_pointers(frame);
        let mut timing = RmTiming::default();
        check(unsafe { rm_render_rows(self.ctx, &p, rows.as_ptr(), &mut timing) }, self.ctx);
        Gpu::status(now, frame.width, frame.height)
    }

    /// `render` with a device-resident FrameBuffer: the f64 frame stays on the GPU and only
    /// `fb.to_vec()` comes back -- what update_raytrace_image hands to the pixbuf
    /// (main.rs:337-346).  `display` is resized to width * height * 3; bytes of rows below the
    /// last whole patch row keep their contents.  `fetch` brings the f64 rows over when
    /// save_to_file wants them (main.rs:353-357).
    pub fn render_display(
        &mut self,
        fov: f64,
        height: f64,
        width: f64,
        frame_width: usize,
        frame_height: usize,
        scene: &::scene::Scene,
        display: &mut Vec<u8>,
    ) -> String {
        let now = ::std::time::Instant::now();
        self.upload(scene);
        let p = Gpu::params(fov, height, width, frame_width, frame_height);
        display.resize(frame_width * frame_height * 3, 0);
        let mut timing = RmTiming::default();
        check(unsafe { rm_render_display(self.ctx, &p, display.as_mut_ptr(), &mut timing) }, self.ctx);
        Gpu::status(now, frame_width, frame_height)
    }

    /// The f64 rows of the frame the last render left on the device, into `frame.buffer`.
    pub fn fetch(&mut self, frame: &mut FrameBuffer) {
        let rows = row_pointers(frame);
        // (the library refuses a FrameBuffer of another size than the frame it holds: a window resized since the render)
        check(unsafe { rm_fetch_rows(self.ctx, rows.as_ptr(), frame.width as u32, frame.height as u32, 0, 0) }, self.ctx);
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { rm_destroy(self.ctx) }
    }
}
