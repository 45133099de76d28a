"""ctypes views of the C ABI (include/ptrace.h) and of the oracle (oracle/pt_oracle.h) for the tests.

The scene loader in this file is an independent Python reading of the reference's JSON/OFF formats
(src/render/mod.rs:85-90,236-241,297-324; src/render/load_off.rs:8-85) used to cross-check the
product's C++ loader; it is test code only.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "path-tracer-rust_amd")
ORACLE_SO = os.path.join(ROOT, "oracle", "libpt_oracle.so")
PRODUCT_SO = os.path.join(PKG, "libptrace_hip.so")

f3 = C.c_float * 3


class PtCamera(C.Structure):
    _fields_ = [("position", f3), ("direction", f3), ("focal_length", C.c_float),
                ("sensor_width", C.c_float), ("aspect_ratio", C.c_float)]


class PtTriangle(C.Structure):
    _fields_ = [("a", f3), ("b", f3), ("c", f3)]


class PtObject(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("position", f3), ("radius", C.c_float), ("color", f3),
                ("emission", f3), ("reflect_type", C.c_uint32), ("tri_offset", C.c_uint32),
                ("tri_count", C.c_uint32), ("bs_center", f3), ("bs_radius", C.c_float)]


class PtConfig(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("backend", C.c_uint32),
                ("seed", C.c_uint64), ("idx_begin", C.c_uint32), ("idx_end", C.c_uint32),
                ("rays_per_pass", C.c_uint32), ("flags", C.c_uint32), ("chunk_pixels", C.c_uint32),
                ("chunk_first", C.c_uint32), ("chunk_step", C.c_uint32), ("progress_ms", C.c_uint32)]


class PtStats(C.Structure):
    _fields_ = [("ray_bounces", C.c_uint64), ("samples", C.c_uint64), ("intersect_rays", C.c_uint64),
                ("intersect_launches", C.c_uint32), ("passes", C.c_uint32), ("ms_total", C.c_double),
                ("ms_device", C.c_double), ("ms_intersect", C.c_double)]


class PtoScene(C.Structure):
    _fields_ = [("camera", PtCamera), ("objs", C.POINTER(PtObject)), ("n_objs", C.c_uint32),
                ("tris", C.POINTER(PtTriangle)), ("n_tris", C.c_uint32)]


class PtoConfig(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("_pad", C.c_uint32),
                ("seed", C.c_uint64)]


class PtoCounters(C.Structure):
    _fields_ = [("ray_bounces", C.c_uint64), ("misses", C.c_uint64), ("splits", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("triangle_tests", C.c_uint64)]


PT_SPHERE, PT_MESH = 0, 1
REFLECT = {"Diffuse": 0, "Specular": 1, "Refract": 2}
BACKEND_WAVEFRONT, BACKEND_MEGAKERNEL = 0, 1
FLAG_NO_BVH = 1
PROGRESS_EVERY_PASS = 0xffffffff
PT_CANCELLED = -4

fp = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def _np_f(a):
    return a.ctypes.data_as(fp)


_oracle = None


def oracle():
    """Load (building if needed) the CPU restatement."""
    global _oracle
    if _oracle is not None:
        return _oracle
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ROOT, "oracle", "pt_oracle.c")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    L = C.CDLL(ORACLE_SO)
    L.pto_vec_ops.argtypes = [fp, fp, C.c_float, fp]
    L.pto_sinf.argtypes = [C.c_float]
    L.pto_sinf.restype = C.c_float
    L.pto_cosf.argtypes = [C.c_float]
    L.pto_cosf.restype = C.c_float
    L.pto_sincos_vs_libm.argtypes = [C.c_uint32, C.c_uint32, u64p, u64p]
    L.pto_philox4x32_10.argtypes = [u32p, u32p, u32p]
    L.pto_philox4x32_7.argtypes = [u32p, u32p, u32p]
    L.pto_u32_to_unit.argtypes = [C.c_uint32]
    L.pto_u32_to_unit.restype = C.c_float
    L.pto_draw4.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, fp]
    L.pto_gamma_correction.argtypes = [C.c_float]
    L.pto_gamma_correction.restype = C.c_float
    L.pto_to_int_with_gamma_correction.argtypes = [C.c_float]
    L.pto_to_int_with_gamma_correction.restype = C.c_uint32
    L.pto_camera_basis.argtypes = [C.POINTER(PtCamera), fp, fp, fp]
    L.pto_mesh_bounding_sphere.argtypes = [C.POINTER(PtTriangle), C.c_uint32, fp, fp]
    L.pto_intersect_sphere.argtypes = [fp, C.c_float, fp, fp, fp, fp, fp]
    L.pto_intersect_sphere.restype = C.c_int
    L.pto_intersect_batch.argtypes = [C.POINTER(PtoScene), fp, fp, C.c_uint32, fp, i32p, i32p, fp, fp]
    L.pto_radiance_mean.argtypes = [C.POINTER(PtoScene), fp, fp, C.c_uint64, C.c_uint32, C.c_uint32, fp,
                                    C.POINTER(PtoCounters)]
    L.pto_radiance_mean_at.argtypes = [C.POINTER(PtoScene), fp, fp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, fp,
                                       C.POINTER(PtoCounters)]
    L.pto_primary_ray.argtypes = [C.POINTER(PtCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_uint64, fp, fp]
    L.pto_render_pixel.argtypes = [C.POINTER(PtoScene), C.POINTER(PtoConfig), C.c_uint32, fp,
                                   C.POINTER(PtoCounters)]
    L.pto_dump_rays.argtypes = [C.POINTER(PtoScene), C.POINTER(PtoConfig), C.c_uint32, C.c_uint32, fp,
                                C.c_uint64]
    L.pto_dump_rays.restype = C.c_uint64
    L.pto_render.argtypes = [C.POINTER(PtoScene), C.POINTER(PtoConfig), C.c_uint32, C.c_uint32, fp, C.c_int,
                             C.POINTER(PtoCounters), C.POINTER(C.c_double)]
    L.pto_render.restype = C.c_int
    L.pto_max_threads.restype = C.c_int
    L.pto_render_mock.argtypes = [C.POINTER(PtoScene), C.POINTER(PtoConfig), fp, C.POINTER(PtoCounters), u64p]
    L.pto_render_mock.restype = C.c_int
    L.pto_mesh_bounding_box.argtypes = [C.POINTER(PtTriangle), C.c_uint32, C.POINTER(PtTriangle)]
    L.pto_mesh_bounding_box.restype = None
    L.pto_intersect_bounds_batch.argtypes = [C.POINTER(PtoScene), C.POINTER(PtTriangle), C.c_uint32, fp, fp, C.c_uint32,
                                             i32p, fp, fp, fp]
    L.pto_intersect_bounds_batch.restype = None
    L.pto_orbit_point_batch.argtypes = [C.POINTER(PtoScene), C.POINTER(PtTriangle), fp, fp, C.c_uint32, i32p, fp, i32p, fp]
    L.pto_orbit_point_batch.restype = None
    L.pto_format_ppm.argtypes = [fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64, C.c_char_p,
                                 C.c_size_t]
    L.pto_format_ppm.restype = C.c_size_t
    L.pto_siphash13.argtypes = [C.c_char_p, C.c_size_t]
    L.pto_siphash13.restype = C.c_uint64
    L.pto_image_hash.argtypes = [fp, C.c_size_t]
    L.pto_image_hash.restype = C.c_uint64
    _oracle = L
    return L


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_float)

_product = None


def product():
    """Load libptrace_hip.so (the product).  Never falls back to anything."""
    global _product
    if _product is not None:
        return _product
    if not os.path.exists(PRODUCT_SO):
        raise RuntimeError("libptrace_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(PRODUCT_SO)
    L.pt_version.restype = C.c_char_p
    L.pt_last_error.restype = C.c_char_p
    L.pt_abi_version.restype = C.c_int
    L.pt_device_count.restype = C.c_int
    L.pt_camera_basis.argtypes = [C.POINTER(PtCamera), fp, fp, fp]
    L.pt_mesh_bounding_sphere.argtypes = [C.POINTER(PtTriangle), C.c_uint32, fp, fp]
    L.pt_config_pixels.argtypes = [C.POINTER(PtConfig)]
    L.pt_config_pixels.restype = C.c_uint32
    L.pt_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.pt_ctx_destroy.argtypes = [C.c_void_p]
    L.pt_ctx_destroy.restype = None
    L.pt_ctx_set_scene.argtypes = [C.c_void_p, C.POINTER(PtCamera), C.POINTER(PtObject), C.c_uint32,
                                   C.POINTER(PtTriangle), C.c_uint32]
    L.pt_ctx_render.argtypes = [C.c_void_p, C.POINTER(PtConfig), C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.POINTER(PtStats)]
    L.pt_ctx_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.pt_ctx_set_memory_budget.argtypes = [C.c_void_p, C.c_size_t]
    L.pt_ctx_pass_kernel.argtypes = [C.c_void_p, C.c_uint32]
    L.pt_ctx_pass_kernel.restype = C.c_char_p
    L.pt_render_multi.argtypes = [C.POINTER(PtConfig), C.c_uint32, C.POINTER(PtCamera), C.POINTER(PtObject), C.c_uint32,
                                  C.POINTER(PtTriangle), C.c_uint32, fp, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(PtStats)]
    L.pt_ctx_snapshot.argtypes = [C.c_void_p, C.c_void_p, u32p]
    L.pt_device_malloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    L.pt_device_free.argtypes = [C.c_int, C.c_void_p]
    L.pt_device_download.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.pt_image_hash.argtypes = [fp, C.c_size_t]
    L.pt_image_hash.restype = C.c_uint64
    L.pt_siphash.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_char_p, C.c_size_t]
    L.pt_siphash.restype = C.c_uint64
    L.pt_host_sincos.argtypes = [C.c_float, fp, fp]
    L.pt_host_sincos.restype = None
    L.pt_ctx_numerics_probe.argtypes = [C.c_void_p, fp, C.c_uint32, fp, fp, fp, fp, u32p]
    L.pt_ctx_radiance.argtypes = [C.c_void_p, fp, fp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                  fp, C.POINTER(PtStats)]
    L.pt_ctx_numerics_sweep.argtypes = [C.c_void_p, u64p]
    L.pt_ctx_sincos_sweep.argtypes = [C.c_void_p, u64p]
    L.pt_ctx_primary_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, u32p, u32p, C.c_uint32, C.c_uint32, fp, fp]
    L.pt_ctx_intersect_streams.argtypes = [C.c_void_p, fp, fp, C.c_uint32, C.c_uint32, fp, i32p]
    L.pt_ctx_intersect.argtypes = [C.c_void_p, fp, fp, C.c_uint32, fp, i32p, i32p, fp, fp]
    L.pt_ctx_intersect_bounds.argtypes = [C.c_void_p, C.c_uint32, fp, fp, C.c_uint32, i32p, fp, fp, fp]
    L.pt_ctx_orbit_point.argtypes = [C.c_void_p, fp, fp, C.c_uint32, i32p, fp, i32p, fp]
    L.pt_ctx_set_mesh_bounds.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(PtTriangle)]
    L.pt_mesh_bounding_box.argtypes = [C.POINTER(PtTriangle), C.c_uint32, C.POINTER(PtTriangle)]
    L.pt_scene_bounding_box.argtypes = [C.c_void_p, C.c_uint32]
    L.pt_scene_bounding_box.restype = C.POINTER(PtTriangle)
    L.pt_comm_unique_id.argtypes = [C.c_char_p]
    L.pt_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
    L.pt_comm_destroy.argtypes = [C.c_void_p]
    L.pt_comm_destroy.restype = None
    L.pt_comm_gather_frame.argtypes = [C.c_void_p, C.POINTER(PtConfig), C.c_void_p, C.c_void_p, C.c_void_p]
    L.pt_render.argtypes = [C.POINTER(PtConfig), C.POINTER(PtCamera), C.POINTER(PtObject), C.c_uint32,
                            C.POINTER(PtTriangle), C.c_uint32, fp, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.POINTER(PtStats)]
    L.pt_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    L.pt_scene_load_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.pt_load_off_ex.argtypes = [C.c_char_p, C.c_float, C.c_uint32, C.POINTER(C.POINTER(PtTriangle)), u32p]
    L.pt_scene_free.argtypes = [C.c_void_p]
    L.pt_scene_free.restype = None
    L.pt_scene_save.argtypes = [C.c_void_p, C.c_char_p]
    L.pt_scene_set_camera.argtypes = [C.c_void_p, C.POINTER(PtCamera)]
    L.pt_scene_id.argtypes = [C.c_void_p]
    L.pt_scene_id.restype = C.c_char_p
    L.pt_scene_camera.argtypes = [C.c_void_p]
    L.pt_scene_camera.restype = C.POINTER(PtCamera)
    L.pt_scene_objects.argtypes = [C.c_void_p, u32p]
    L.pt_scene_objects.restype = C.POINTER(PtObject)
    L.pt_scene_triangles.argtypes = [C.c_void_p, u32p]
    L.pt_scene_triangles.restype = C.POINTER(PtTriangle)
    L.pt_load_off.argtypes = [C.c_char_p, C.c_float, C.POINTER(C.POINTER(PtTriangle)), u32p]
    L.pt_free.argtypes = [C.c_void_p]
    L.pt_free.restype = None
    L.pt_gamma_correction.argtypes = [C.c_float]
    L.pt_gamma_correction.restype = C.c_float
    L.pt_to_int_with_gamma_correction.argtypes = [C.c_float]
    L.pt_to_int_with_gamma_correction.restype = C.c_uint32
    L.pt_write_ppm.argtypes = [C.c_char_p, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64]
    _product = L
    return L


# ---------------------------------------------------------------------------- python scene reader
def _v3(v):
    return f3(*[float(np.float32(x)) for x in v])


def load_off_py(path, scale, triangulate=False):
    """load_off.rs:8-85 in Python (f32 arithmetic through numpy); triangulate=True: fan over polygon faces."""
    lines = []
    for l in open(path):
        l = l.strip()
        if l and not l.startswith("#"):
            lines.append(l)
    assert lines[0] == "OFF"
    nv, nf, _ = [int(t) for t in lines[1].split()]
    sc = np.float32(scale)
    verts = [np.array([np.float32(t) for t in l.split()], dtype=np.float32) * sc for l in lines[2:2 + nv]]
    tris = []
    for l in lines[2 + nv:2 + nv + nf]:
        t = l.split()
        k = int(t[0])
        assert k == 3 or (triangulate and k > 3)
        for j in range(2, k):
            tris.append((verts[int(t[1])], verts[int(t[j])], verts[int(t[j + 1])]))
    return tris


class Scene:
    """Flattened scene: pt_camera + pt_object[] + pt_triangle[] (kept alive by this object)."""

    def __init__(self, sid, cam, objs, tris):
        self.id = sid
        self.cam = cam
        self.n_objs = len(objs)
        self.n_tris = len(tris)
        self.objs = (PtObject * max(1, len(objs)))(*objs)
        self.tris = (PtTriangle * max(1, len(tris)))(*tris)

    def pto(self):
        s = PtoScene()
        s.camera = self.cam
        s.objs = C.cast(self.objs, C.POINTER(PtObject))
        s.n_objs = self.n_objs
        s.tris = C.cast(self.tris, C.POINTER(PtTriangle))
        s.n_tris = self.n_tris
        return s


def make_camera(position, direction, focal_length=0.035, sensor_width=0.036, aspect_ratio=1.5):
    cam = PtCamera()
    cam.position = _v3(position)
    cam.direction = _v3(direction)
    cam.focal_length = float(np.float32(focal_length))
    cam.sensor_width = float(np.float32(sensor_width))
    cam.aspect_ratio = float(np.float32(aspect_ratio))
    return cam


def make_sphere(position, radius, color, emission, reflect):
    o = PtObject()
    o.kind = PT_SPHERE
    o.position = _v3(position)
    o.radius = float(np.float32(radius))
    o.color = _v3(color)
    o.emission = _v3(emission)
    o.reflect_type = REFLECT[reflect] if isinstance(reflect, str) else reflect
    return o


def make_mesh(position, color, emission, reflect, tri_offset, tri_count, bs_center, bs_radius):
    o = PtObject()
    o.kind = PT_MESH
    o.position = _v3(position)
    o.color = _v3(color)
    o.emission = _v3(emission)
    o.reflect_type = REFLECT[reflect] if isinstance(reflect, str) else reflect
    o.tri_offset = tri_offset
    o.tri_count = tri_count
    o.bs_center = _v3(bs_center)
    o.bs_radius = float(np.float32(bs_radius))
    return o


def make_tri(a, b, c):
    t = PtTriangle()
    t.a = _v3(a)
    t.b = _v3(b)
    t.c = _v3(c)
    return t


def load_scene_py(path, base_dir=None, triangulate=False):
    """SceneDescriptor::load + to_data (mod.rs:92-110,304-318) in Python."""
    base_dir = base_dir or os.path.dirname(os.path.dirname(os.path.abspath(path)))
    d = json.load(open(path))
    c = d["camera"]
    cam = make_camera(c["position"], c["direction"], c["focal_length"], c["sensor_width"], c["aspect_ratio"])
    objs, tris = [], []
    for od in d["objects"]:
        m = od["material"]
        ty = od["type_"]
        (kind, val), = ty.items()
        if kind == "Sphere":
            objs.append(make_sphere(od["position"], val["radius"], m["color"], m["emmission"], m["reflect_type"]))
            continue
        if kind == "MeshFile":
            tl = load_off_py(os.path.join(base_dir, val["path"]), val["scale"], triangulate)
            tlist = [make_tri(a, b, c_) for a, b, c_ in tl]
            arr = (PtTriangle * len(tlist))(*tlist)
            ctr = (C.c_float * 3)()
            rad = C.c_float()
            oracle().pto_mesh_bounding_sphere(arr, len(tlist), ctr, C.byref(rad))
            bs_c, bs_r = list(ctr), rad.value
        else:
            tlist = [make_tri(t["a"], t["b"], t["c"]) for t in val["triangles"]]
            bs_c, bs_r = val["bounding_sphere"]["position"], val["bounding_sphere"]["radius"]
        objs.append(make_mesh(od["position"], m["color"], m["emmission"], m["reflect_type"], len(tris),
                              len(tlist), bs_c, bs_r))
        tris.extend(tlist)
    return Scene(d["id"], cam, objs, tris)


def oracle_boxes(scene):
    """Mesh.bounding_box of every object as Mesh::new computes it (12 object-local triangles per object; zeros for
    spheres): the `boxes` argument of pto_intersect_bounds_batch / pto_orbit_point_batch."""
    L = oracle()
    boxes = (PtTriangle * (12 * max(1, scene.n_objs)))()
    for i in range(scene.n_objs):
        o = scene.objs[i]
        if o.kind != PT_MESH:
            continue
        arr = (PtTriangle * o.tri_count)(*[scene.tris[o.tri_offset + k] for k in range(o.tri_count)])
        out = (PtTriangle * 12)()
        L.pto_mesh_bounding_box(arr, o.tri_count, out)
        for k in range(12):
            boxes[12 * i + k] = out[k]
    return boxes


def oracle_render_mock(scene, width, height, spp):
    """The reference's frame with MOCK_RANDOM = true (mod.rs:31-51): image, counters, number of rand01() calls."""
    L = oracle()
    cfg = PtoConfig(width, height, spp, 0, 0)
    out = np.zeros((width * height, 3), dtype=np.float32)
    cnt = PtoCounters()
    draws = C.c_uint64()
    ps = scene.pto()
    assert L.pto_render_mock(C.byref(ps), C.byref(cfg), _np_f(out), C.byref(cnt), C.byref(draws)) == 0
    return out, cnt, draws.value


def oracle_intersect(scene, o, d):
    """intersect_scene ray by ray: (t, object_id, tri_id, x, n) arrays."""
    L = oracle()
    o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
    m = len(o)
    t, oid, tid = np.zeros(m, np.float32), np.zeros(m, np.int32), np.zeros(m, np.int32)
    x, nr = np.zeros((m, 3), np.float32), np.zeros((m, 3), np.float32)
    ps = scene.pto()
    L.pto_intersect_batch(C.byref(ps), _np_f(o), _np_f(d), m, _np_f(t), oid.ctypes.data_as(i32p),
                          tid.ctypes.data_as(i32p), _np_f(x), _np_f(nr))
    return t, oid, tid, x, nr


def oracle_radiance(scene, o, d, depth, n, seed, pixel):
    """radiance(&ray, depth, scene) averaged over n samples keyed (seed; pixel, i): (mean rgb, counters)."""
    L = oracle()
    o = np.ascontiguousarray(o, dtype=np.float32)
    d = np.ascontiguousarray(d, dtype=np.float32)
    out = np.zeros(3, np.float32)
    cnt = PtoCounters()
    ps = scene.pto()
    L.pto_radiance_mean_at(C.byref(ps), _np_f(o), _np_f(d), depth, seed, pixel, n, _np_f(out), C.byref(cnt))
    return out, cnt


def scene_path(sid):
    return os.path.join(ROOT, "scenes", sid + ".json")


def oracle_render(scene, width, height, spp, seed, idx_begin=0, idx_end=None, threads=0):
    L = oracle()
    cfg = PtoConfig(width, height, spp, 0, seed)
    idx_end = width * height if idx_end is None else idx_end
    out = np.zeros((width * height, 3), dtype=np.float32)
    cnt = PtoCounters()
    secs = C.c_double()
    ps = scene.pto()
    rc = L.pto_render(C.byref(ps), C.byref(cfg), idx_begin, idx_end, _np_f(out), threads, C.byref(cnt),
                      C.byref(secs))
    assert rc == 0
    return out, cnt, secs.value
