//! src/render/hip.rs — binding of libptrace_hip.so (include/ptrace.h, ABI 3) for filippo-orru/path-tracer-rust.
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

struct Progress<'a> {
    processed_pixel_count: &'a AtomicUsize,
    grid_size: usize,
}

extern "C" fn on_progress(user: *mut c_void, fraction: f32) {
    // drives the counter the 500 ms thread reads (:960-975); no image copy here - see the note at the bottom
    let p = unsafe { &*(user as *const Progress) };
    p.processed_pixel_count
        .store((fraction * p.grid_size as f32) as usize, Ordering::Relaxed);
}

/// Replaces the parallel section :1017-1024: fills `pixels` (index (H-1-y)*W+x, :805-806; glam::Vec3 is
/// #[repr(C)] 3 x f32, so the Vec's memory IS the out_rgb layout).  `cancel` is the flag render() already owns
/// (`stop_render`, :943): one byte, read by the library between passes.
pub fn render_pixels_hip(
    cfg: &RenderConfig,
    pixels: &mut [Vec3],
    cancel: &AtomicBool,
    processed_pixel_count: &AtomicUsize,
    seed: u64,
) -> Result<PtStats, String> {
    let (cam, objs, tris) = flatten(&cfg.scene);
    let grid_size = cfg.resolution.width * cfg.resolution.height;
    assert_eq!(pixels.len(), grid_size);
    let c = PtConfig {
        width: cfg.resolution.width as u32,
        height: cfg.resolution.height as u32,
        spp: cfg.samples_per_pixel as u32,
        seed,
        ..Default::default()
    };
    let progress = Progress {
        processed_pixel_count,
        grid_size,
    };
    let mut st = PtStats::default();
    let rc = unsafe {
        pt_render(
            &c,
            &cam,
            objs.as_ptr(),
            objs.len() as u32,
            tris.as_ptr(),
            tris.len() as u32,
            pixels.as_mut_ptr() as *mut f32,
            cancel.as_ptr() as *const u8,
            Some(on_progress),
            &progress as *const Progress as *mut c_void,
            &mut st,
        )
    };
    match rc {
        PT_OK | PT_CANCELLED => Ok(st), // cancelled: the partial image is in `pixels`, as with the reference (:1003-1016)
        _ => Err(unsafe { CStr::from_ptr(pt_last_error()) }
            .to_string_lossy()
            .into_owned()),
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The edit in render() (:1001-1024).  The mutex around `pixels` is held by the 500 ms thread while it clones the
// buffer (:972-976), so the GPU call must not hold it for the whole frame: render into a local buffer and copy.
//
//     let render_pixel_to_vec = ...;                       // unchanged (CPU path)
//     if hip::pt_device_count() > 0 && !MOCK_RANDOM {
//         let mut local = vec![Vec3::default(); grid_size];
//         let seed = rand::random::<u64>();                // the reference is OS-seeded too (:53)
//         match hip::render_pixels_hip(&render_config, &mut local, &stop_render, &processed_pixel_count, seed) {
//             Ok(stats) => println!("GPU: {} ray bounces in {:.1} ms", stats.ray_bounces, stats.ms_total),
//             Err(msg) => panic!("libptrace_hip: {msg}"),  // the reference unwraps its own errors (:1032,1042)
//         }
//         pixels.lock().unwrap().copy_from_slice(&local);
//     } else if MOCK_RANDOM { ... } else { ... rayon ... } // unchanged
//
// Progressive images in RenderUpdate (:972-976) stay black until the copy above; a host that wants the partial frame
// every 500 ms uses the resident form instead (pt_ctx_create / pt_ctx_set_scene / pt_ctx_render with a callback that
// calls pt_ctx_snapshot and copies the snapshot into `pixels`): include/ptrace.h, "Progressive preview".
