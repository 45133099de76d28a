//! src/render/hip.rs — binding of libptrace_hip.so (include/ptrace.h, ABI 5) for filippo-orru/path-tracer-rust.
//!
//! NOT COMPILED HERE: the build image has no Rust toolchain.  This is the file a maintainer adds as `mod hip;` in
//! `src/render/mod.rs` (a child module of `render`, so it may read the private fields of `Mesh` and
//! `StandaloneSphere`), plus the few lines in `render()` shown at the bottom.  Every struct mirrors a `typedef struct`
//! of include/ptrace.h field by field (sizes 36 / 36 / 72 / 56 / 56 bytes, checked from Python in tests/test_abi.py).
//!
//! Line references are to the reference's src/render/mod.rs.
use super::{ReflectType, RenderConfig, SceneData, SceneObject, SceneObjectData};
use glam::Vec3;
use std::ffi::{c_void, CStr};
use std::os::raw::c_char;
use std::sync::atomic::{AtomicBool, AtomicUsize, Ordering};
use std::sync::Mutex;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct PtCamera {
    pub position: [f32; 3],
    pub direction: [f32; 3],
    pub focal_length: f32,
    pub sensor_width: f32,
    pub aspect_ratio: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct PtTriangle {
    pub a: [f32; 3],
    pub b: [f32; 3],
    pub c: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct PtObject {
    pub kind: u32, // 0 = PT_SPHERE, 1 = PT_MESH
    pub position: [f32; 3],
    pub radius: f32,
    pub color: [f32; 3],
    pub emission: [f32; 3],
    pub reflect_type: u32, // enum order of ReflectType (:71-76)
    pub tri_offset: u32,
    pub tri_count: u32,
    pub bs_center: [f32; 3], // Mesh.bounding_sphere.position, object-local, as stored (:268 adds `position`)
    pub bs_radius: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct PtConfig {
    pub width: u32,
    pub height: u32,
    pub spp: u32,
    pub backend: u32, // 0 wavefront, 1 megakernel
    pub seed: u64,
    pub idx_begin: u32, // 0, 0 = whole frame
    pub idx_end: u32,
    pub rays_per_pass: u32, // 0 = library default
    pub flags: u32,
    pub chunk_pixels: u32,
    pub chunk_first: u32,
    pub chunk_step: u32,
    pub progress_ms: u32, // 0 = 500 ms, the cadence of the reference's progress thread (:965-982)
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct PtStats {
    pub ray_bounces: u64,
    pub samples: u64,
    pub intersect_rays: u64,
    pub intersect_launches: u32,
    pub passes: u32,
    pub ms_total: f64,
    pub ms_device: f64,
    pub ms_intersect: f64,
}

pub const PT_OK: i32 = 0;
pub const PT_CANCELLED: i32 = -4;

pub type PtProgressFn = extern "C" fn(user: *mut c_void, fraction: f32);

/// opaque `pt_ctx` of include/ptrace.h: one GPU, one stream, the device copies of one scene and the ray queues
#[repr(C)]
pub struct PtCtx {
    _private: [u8; 0],
}

#[link(name = "ptrace_hip")]
extern "C" {
    pub fn pt_render(
        cfg: *const PtConfig,
        cam: *const PtCamera,
        objs: *const PtObject,
        n_objs: u32,
        tris: *const PtTriangle,
        n_tris: u32,
        out_rgb: *mut f32,
        cancel: *const u8,
        cb: Option<PtProgressFn>,
        user: *mut c_void,
        stats: *mut PtStats,
    ) -> i32;
    // the resident form: scene tables and ray queues stay in HBM across frames, the frame is written to device memory,
    // and the progress callback may pull what has been accumulated so far (pt_ctx_snapshot)
    pub fn pt_ctx_create(device: i32, out: *mut *mut PtCtx) -> i32;
    pub fn pt_ctx_destroy(ctx: *mut PtCtx);
    pub fn pt_ctx_set_scene(
        ctx: *mut PtCtx,
        cam: *const PtCamera,
        objs: *const PtObject,
        n_objs: u32,
        tris: *const PtTriangle,
        n_tris: u32,
    ) -> i32;
    pub fn pt_ctx_render(
        ctx: *mut PtCtx,
        cfg: *const PtConfig,
        d_out_rgb: *mut c_void,
        hip_stream: *mut c_void,
        cancel: *const u8,
        cb: Option<PtProgressFn>,
        user: *mut c_void,
        stats: *mut PtStats,
    ) -> i32;
    pub fn pt_ctx_snapshot(ctx: *mut PtCtx, d_out_rgb: *mut c_void, spp_done: *mut u32) -> i32;
    pub fn pt_device_malloc(device: i32, bytes: usize, out: *mut *mut c_void) -> i32;
    pub fn pt_device_free(device: i32, p: *mut c_void) -> i32;
    pub fn pt_device_download(device: i32, dst_host: *mut c_void, src_device: *const c_void, bytes: usize) -> i32;
    pub fn pt_last_error() -> *const c_char;
    pub fn pt_device_count() -> i32;
    pub fn pt_image_hash(rgb: *const f32, n_floats: usize) -> u64;
}

fn v3(v: Vec3) -> [f32; 3] {
    [v.x, v.y, v.z]
}

/// SceneData -> the flat arrays of the C ABI.  Nothing is recomputed: every value is copied as the reference holds it
/// (`direction` un-normalised as stored, `bounding_sphere` as deserialised or as Mesh::new made it, :450-499), and the
/// triangles of all meshes are laid end to end in object order (object-local coordinates; the library adds `position`
/// exactly as Triangle::transformed does, :546-552).
pub fn flatten(scene: &SceneData) -> (PtCamera, Vec<PtObject>, Vec<PtTriangle>) {
    let cam = PtCamera {
        position: v3(scene.camera.position),
        direction: v3(scene.camera.direction()),
        focal_length: scene.camera.focal_length,
        sensor_width: scene.camera.sensor_width,
        aspect_ratio: scene.camera.aspect_ratio,
    };
    let mut objs: Vec<PtObject> = Vec::with_capacity(scene.objects.len());
    let mut tris: Vec<PtTriangle> = Vec::new();
    for o in scene.objects.iter() {
        let o: &SceneObjectData = o;
        let mut p = PtObject {
            position: v3(o.position),
            color: v3(o.material.color),
            emission: v3(o.material.emmission), // sic
            reflect_type: match o.material.reflect_type {
                ReflectType::Diffuse => 0,
                ReflectType::Specular => 1,
                ReflectType::Refract => 2,
            },
            ..Default::default()
        };
        match &o.type_ {
            SceneObject::Sphere { radius } => {
                p.kind = 0;
                p.radius = *radius;
            }
            SceneObject::Mesh { mesh, file: _ } => {
                p.kind = 1;
                p.tri_offset = tris.len() as u32;
                p.tri_count = mesh.triangles.len() as u32;
                p.bs_center = v3(mesh.bounding_sphere.position);
                p.bs_radius = mesh.bounding_sphere.radius;
                tris.extend(mesh.triangles.iter().map(|t| PtTriangle {
                    a: v3(t.a),
                    b: v3(t.b),
                    c: v3(t.c),
                }));
            }
        }
        objs.push(p);
    }
    (cam, objs, tris)
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(pt_last_error()) }.to_string_lossy().into_owned()
}

/// what the progress callback needs: the counter and the pixel buffer the reference's 500 ms thread reads (:960-976)
struct Preview<'a> {
    ctx: *mut PtCtx,
    d_snap: *mut c_void,
    pixels: &'a Mutex<Vec<Vec3>>,
    processed_pixel_count: &'a AtomicUsize,
    grid_size: usize,
}

/// Called by the library between passes, at most every 500 ms (pt_config.progress_ms = 0: the reference's RenderUpdate
/// cadence, :965-982), on the thread that called pt_ctx_render.  Drives the counter the progress thread reads (:966-968) and
/// puts the picture accumulated so far into `pixels`, so that the RenderUpdate that thread sends next (:969-972) carries a
/// partial image: pt_ctx_snapshot resolves the accumulators into device memory (every pixel over the samples it has so far
/// - the reference's partial image is a random subset of finished pixels, this one is the whole frame at partial spp),
/// one download, one copy under the mutex (held for a memcpy, as render_pixel_to_vec holds it for one store, :1013-1014).
extern "C" fn on_progress(user: *mut c_void, fraction: f32) {
    let p = unsafe { &*(user as *const Preview) };
    p.processed_pixel_count
        .store((fraction * p.grid_size as f32) as usize, Ordering::Relaxed);
    if fraction >= 1.0 {
        return; // the finished frame is copied by render_pixels_hip itself
    }
    let mut spp_done: u32 = 0;
    if unsafe { pt_ctx_snapshot(p.ctx, p.d_snap, &mut spp_done) } != PT_OK {
        return; // nothing accumulated yet
    }
    let mut local = vec![Vec3::default(); p.grid_size];
    let bytes = p.grid_size * 3 * std::mem::size_of::<f32>();
    if unsafe { pt_device_download(0, local.as_mut_ptr() as *mut c_void, p.d_snap, bytes) } == PT_OK {
        p.pixels.lock().unwrap().copy_from_slice(&local);
    }
}

/// Replaces the parallel section :1017-1024: fills `pixels` (index (H-1-y)*W+x, :805-806; glam::Vec3 is
/// #[repr(C)] 3 x f32, so the Vec's memory IS the out_rgb layout) and keeps it filled with the partial picture while the
/// frame renders.  `cancel` is the flag render() already owns (`stop_render`, :943): one byte, read by the library between
/// passes (a few milliseconds apart; the reference polls it every 100 ms, :947-958).
pub fn render_pixels_hip(
    cfg: &RenderConfig,
    pixels: &Mutex<Vec<Vec3>>,
    cancel: &AtomicBool,
    processed_pixel_count: &AtomicUsize,
    seed: u64,
) -> Result<PtStats, String> {
    let (cam, objs, tris) = flatten(&cfg.scene);
    let grid_size = cfg.resolution.width * cfg.resolution.height;
    let bytes = grid_size * 3 * std::mem::size_of::<f32>();
    let c = PtConfig {
        width: cfg.resolution.width as u32,
        height: cfg.resolution.height as u32,
        spp: cfg.samples_per_pixel as u32,
        seed,
        ..Default::default() // progress_ms = 0: a callback at most every 500 ms
    };
    let mut ctx: *mut PtCtx = std::ptr::null_mut();
    let (mut d_out, mut d_snap): (*mut c_void, *mut c_void) = (std::ptr::null_mut(), std::ptr::null_mut());
    let mut st = PtStats::default();
    let rc = unsafe {
        let mut rc = pt_ctx_create(0, &mut ctx);
        if rc == PT_OK {
            rc = pt_ctx_set_scene(ctx, &cam, objs.as_ptr(), objs.len() as u32, tris.as_ptr(), tris.len() as u32);
        }
        if rc == PT_OK {
            rc = pt_device_malloc(0, bytes, &mut d_out);
        }
        if rc == PT_OK {
            rc = pt_device_malloc(0, bytes, &mut d_snap);
        }
        if rc == PT_OK {
            let preview = Preview {
                ctx,
                d_snap,
                pixels,
                processed_pixel_count,
                grid_size,
            };
            rc = pt_ctx_render(
                ctx,
                &c,
                d_out,
                std::ptr::null_mut(),
                cancel.as_ptr() as *const u8,
                Some(on_progress),
                &preview as *const Preview as *mut c_void,
                &mut st,
            );
        }
        if rc == PT_OK || rc == PT_CANCELLED {
            // the finished frame - or, cancelled, every pixel over the samples that were accumulated (the reference's
            // cancelled image holds its finished pixels and black elsewhere, :1003-1016)
            let mut local = vec![Vec3::default(); grid_size];
            let rc2 = pt_device_download(0, local.as_mut_ptr() as *mut c_void, d_out, bytes);
            if rc2 == PT_OK {
                pixels.lock().unwrap().copy_from_slice(&local);
            } else {
                rc = rc2;
            }
        }
        rc
    };
    let msg = if rc == PT_OK || rc == PT_CANCELLED { String::new() } else { last_error() };
    unsafe {
        pt_device_free(0, d_snap);
        pt_device_free(0, d_out);
        pt_ctx_destroy(ctx);
    }
    match rc {
        PT_OK | PT_CANCELLED => Ok(st),
        _ => Err(msg),
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The edit in render() (:1001-1024).  `pixels` is the Arc<Mutex<Vec<Vec3>>> render() already owns (:938) and the 500 ms
// thread clones under its lock (:969-972); render_pixels_hip only takes the lock for a memcpy.
//
//     let render_pixel_to_vec = ...;                       // unchanged (CPU path)
//     if hip::pt_device_count() > 0 && !MOCK_RANDOM {
//         let seed = rand::random::<u64>();                // the reference is OS-seeded too (:53)
//         match hip::render_pixels_hip(&render_config, &pixels, &stop_render, &processed_pixel_count, seed) {
//             Ok(stats) => println!("GPU: {} ray bounces in {:.1} ms", stats.ray_bounces, stats.ms_total),
//             Err(msg) => panic!("libptrace_hip: {msg}"),  // the reference unwraps its own errors (:1032,1042)
//         }
//     } else if MOCK_RANDOM { ... } else { ... rayon ... } // unchanged
//
// RenderUpdate { progress, image } (:969-972) then carries the growing picture every 500 ms exactly as with the CPU path:
// the progress thread is untouched, it finds `pixels` refreshed by on_progress.  A host that renders many frames of one
// scene (the GUI re-renders on every camera move) keeps the PtCtx and the two device buffers instead of creating them per
// frame: pt_ctx_set_scene only when the scene changed.
