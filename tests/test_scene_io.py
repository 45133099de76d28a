"""The formats either side of the path: the product's C++ JSON/OFF loader against an independent Python
reading of the same files, the committed scene fixtures against the reference's own data files (when the
reference tree is mounted), and the loader's error behaviour where the reference panics."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import ptlib

L = ptlib.product()
SCENES = ["single-sphere", "two-spheres", "three-spheres", "cartesian", "cornell", "mesh"]
REF = "/root/reference"


def load_c(path, base=None):
    h = C.c_void_p()
    rc = L.pt_scene_load(path.encode(), (base or ptlib.ROOT).encode(), C.byref(h))
    return rc, h


def as_bytes(arr, n):
    return bytes(C.cast(arr, C.POINTER(C.c_char * (C.sizeof(arr._type_) * n))).contents) if n else b""


@pytest.mark.parametrize("sid", SCENES)
def test_cpp_loader_equals_python_loader(sid):
    rc, h = load_c(ptlib.scene_path(sid))
    assert rc == 0, L.pt_last_error()
    py = ptlib.load_scene_py(ptlib.scene_path(sid))
    n, m = C.c_uint32(), C.c_uint32()
    objs = L.pt_scene_objects(h, C.byref(n))
    tris = L.pt_scene_triangles(h, C.byref(m))
    assert L.pt_scene_id(h).decode() == py.id == sid
    assert (n.value, m.value) == (py.n_objs, py.n_tris)
    assert bytes(L.pt_scene_camera(h).contents) == bytes(py.cam)
    assert as_bytes(objs, n.value) == bytes(py.objs)[: C.sizeof(ptlib.PtObject) * n.value]
    assert as_bytes(tris, m.value) == bytes(py.tris)[: C.sizeof(ptlib.PtTriangle) * m.value]
    L.pt_scene_free(h)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
@pytest.mark.parametrize("sid", SCENES)
def test_committed_fixtures_equal_reference_data(sid):
    """scenes/*.json in this repo are compact re-emissions of the reference's data files: every number must
    parse to the same f32 (via f64, as serde_json does) and the structure must be identical."""
    def f32ify(o):
        if isinstance(o, float):
            return float(np.float32(o))
        if isinstance(o, list):
            return [f32ify(x) for x in o]
        if isinstance(o, dict):
            return {k: f32ify(v) for k, v in o.items()}
        return o
    ours = json.load(open(ptlib.scene_path(sid)))
    theirs = json.load(open(os.path.join(REF, "scenes", sid + ".json")))
    assert f32ify(ours) == f32ify(theirs)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_committed_mesh_equals_reference_mesh():
    a = ptlib.load_off_py(os.path.join(ptlib.ROOT, "meshes", "mctri.off"), 0.16)
    b = ptlib.load_off_py(os.path.join(REF, "meshes", "mctri.off"), 0.16)
    assert len(a) == len(b) == 810
    for ta, tb in zip(a, b):
        for va, vb in zip(ta, tb):
            assert va.tobytes() == vb.tobytes()


def test_mesh_scene_facts():
    """mctri.off at scale 0.16 -> bounding sphere c=(0.24,0.64,-0.16) r=1.60997; 824 triangles."""
    rc, h = load_c(ptlib.scene_path("mesh"))
    assert rc == 0
    n, m = C.c_uint32(), C.c_uint32()
    objs = L.pt_scene_objects(h, C.byref(n))
    L.pt_scene_triangles(h, C.byref(m))
    assert (n.value, m.value) == (8, 824)
    assert objs[0].tri_count == 810 and objs[0].kind == ptlib.PT_MESH
    assert np.allclose(list(objs[0].bs_center), [0.24, 0.64, -0.16], atol=1e-6)
    assert abs(objs[0].bs_radius - 1.60997) < 1e-5
    L.pt_scene_free(h)


def write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


GOOD_CAM = '"camera":{"position":[0,0,1],"direction":[0,0,-1],"focal_length":0.035,"sensor_width":0.036,"aspect_ratio":1.5}'
GOOD_MAT = '"material":{"color":[1,1,1],"emmission":[0,0,0],"reflect_type":"Diffuse"}'


def test_loader_accepts_pretty_json_unknown_keys_and_escapes(tmp_path):
    text = '{\n "id" : "a\\u0041\\n", "extra": {"x":[1,2,{"y":null}]},\n "objects": [ {"type_": {"Sphere": {"radius": 1e0}}, ' \
           '"position": [1.5, -2, 3E-1], ' + GOOD_MAT + ', "unknown": true} ],\n ' + GOOD_CAM[:-1] + ',"updating_direction":null}\n}\n'
    rc, h = load_c(write(tmp_path, "s.json", text))
    assert rc == 0, L.pt_last_error()
    assert L.pt_scene_id(h) == b"aA\n"
    n = C.c_uint32()
    o = L.pt_scene_objects(h, C.byref(n))
    assert n.value == 1 and o[0].radius == 1.0 and list(o[0].position) == [1.5, -2.0, float(np.float32(0.3))]
    L.pt_scene_free(h)


@pytest.mark.parametrize("text", [
    "",                                                     # empty file
    "{",                                                    # truncated
    '{"id":"x","objects":[],' + GOOD_CAM + '} trailing',    # trailing characters
    '{"id":"x","objects":[]}',                              # missing camera (serde: missing field)
    '{"id":"x","objects":[{"type_":{"Cube":{}},"position":[0,0,0],' + GOOD_MAT + '}],' + GOOD_CAM + '}',
    '{"id":"x","objects":[{"type_":{"Sphere":{}},"position":[0,0,0],' + GOOD_MAT + '}],' + GOOD_CAM + '}',
    '{"id":"x","objects":[{"type_":{"Sphere":{"radius":1}},"position":[0,0],' + GOOD_MAT + '}],' + GOOD_CAM + '}',
    '{"id":"x","objects":[{"type_":{"Sphere":{"radius":1}},"position":[0,0,0],"material":{"color":[1,1,1],'
    '"emmission":[0,0,0],"reflect_type":"Glossy"}}],' + GOOD_CAM + '}',
    '{"id":"x","objects":[{"type_":{"Mesh":{"triangles":[]}},"position":[0,0,0],' + GOOD_MAT + '}],' + GOOD_CAM + '}',
])
def test_loader_rejects_what_serde_rejects(tmp_path, text):
    rc, h = load_c(write(tmp_path, "bad.json", text))
    assert rc == -7, (rc, L.pt_last_error())
    assert not h.value


def test_off_loader_semantics(tmp_path):
    """load_off.rs: blank and '#' lines skipped anywhere, `OFF` header, vertex*scale, colour tokens after the
    4th face token ignored, non-triangles rejected."""
    good = "# c\nOFF\n\n# c2\n3 1 0\n0 0 0\n 1.0e+00 0 0 \n0 2 0\n# mid\n3 0 1 2 0.5 0.5 0.5 1.0\n"
    tris = C.POINTER(ptlib.PtTriangle)()
    n = C.c_uint32()
    assert L.pt_load_off(write(tmp_path, "g.off", good).encode(), 0.5, C.byref(tris), C.byref(n)) == 0
    assert n.value == 1 and list(tris[0].b) == [0.5, 0.0, 0.0] and list(tris[0].c) == [0.0, 1.0, 0.0]
    L.pt_free(tris)
    bad = {
        "header": "COFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n",
        "counts": "OFF\n3 1\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n",
        "vertex": "OFF\n3 1 0\n0 0\n1 0 0\n0 1 0\n3 0 1 2\n",
        "vertex_tok": "OFF\n3 1 0\n0 0 x\n1 0 0\n0 1 0\n3 0 1 2\n",
        "quad": "OFF\n4 1 0\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n",
        "short_face": "OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n3 0 1\n",
        "index": "OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n3 0 1 7\n",
        "eof": "OFF\n3 2 0\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n",
    }
    for name, text in bad.items():
        rc = L.pt_load_off(write(tmp_path, name + ".off", text).encode(), 1.0, C.byref(tris), C.byref(n))
        assert rc == -7, (name, rc)
    assert L.pt_load_off(b"/nonexistent.off", 1.0, C.byref(tris), C.byref(n)) == -6


def test_hdodec_fan_triangulation_extension():
    """meshes/hdodec.off (12 pentagons) is rejected as the reference rejects it, and loads as 36 fan triangles
    with PT_LOAD_TRIANGULATE; the flattened scene equals the Python reading of the same rule."""
    path = ptlib.scene_path("mesh-hdodec")
    h = C.c_void_p()
    assert L.pt_scene_load(path.encode(), ptlib.ROOT.encode(), C.byref(h)) == -7
    assert L.pt_scene_load_ex(path.encode(), ptlib.ROOT.encode(), 1, C.byref(h)) == 0, L.pt_last_error()
    py = ptlib.load_scene_py(path, triangulate=True)
    n, m = C.c_uint32(), C.c_uint32()
    objs = L.pt_scene_objects(h, C.byref(n))
    tris = L.pt_scene_triangles(h, C.byref(m))
    assert objs[0].tri_count == 36 and m.value == 36 + 14
    assert as_bytes(objs, n.value) == bytes(py.objs)[: C.sizeof(ptlib.PtObject) * n.value]
    assert as_bytes(tris, m.value) == bytes(py.tris)[: C.sizeof(ptlib.PtTriangle) * m.value]
    L.pt_scene_free(h)


def test_hdodec_style_pentagons_are_rejected(tmp_path):
    """The reference cannot load meshes/hdodec.off (pentagon faces, load_off.rs:73-76); neither can we."""
    text = "OFF\n5 1 0\n0 0 0\n1 0 0\n1 1 0\n0.5 1.5 0\n0 1 0\n5 0 1 2 3 4\n"
    tris = C.POINTER(ptlib.PtTriangle)()
    n = C.c_uint32()
    assert L.pt_load_off(write(tmp_path, "p.off", text).encode(), 1.0, C.byref(tris), C.byref(n)) == -7
    assert b"Invalid face" in L.pt_last_error()


@pytest.mark.parametrize("sid", SCENES)
def test_save_roundtrip(sid, tmp_path):
    """pt_scene_save then pt_scene_load gives back identical flattened scenes (all six shipped scenes)."""
    rc, h = load_c(ptlib.scene_path(sid))
    assert rc == 0
    out = tmp_path / "scenes"
    out.mkdir()
    path = str(out / (sid + ".json"))
    assert L.pt_scene_save(h, path.encode()) == 0, L.pt_last_error()
    rc2, h2 = load_c(path)  # MeshFile paths resolve against the repo root
    assert rc2 == 0, L.pt_last_error()
    for hh in (h, h2):
        pass
    n1, n2, m1, m2 = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    o1, o2 = L.pt_scene_objects(h, C.byref(n1)), L.pt_scene_objects(h2, C.byref(n2))
    t1, t2 = L.pt_scene_triangles(h, C.byref(m1)), L.pt_scene_triangles(h2, C.byref(m2))
    assert (n1.value, m1.value) == (n2.value, m2.value)
    assert as_bytes(o1, n1.value) == as_bytes(o2, n2.value) and as_bytes(t1, m1.value) == as_bytes(t2, m2.value)
    assert bytes(L.pt_scene_camera(h).contents) == bytes(L.pt_scene_camera(h2).contents)
    json.load(open(path))  # and it is valid JSON
    L.pt_scene_free(h)
    L.pt_scene_free(h2)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
@pytest.mark.parametrize("sid", SCENES)
def test_saved_bytes_equal_the_files_the_reference_wrote(sid, tmp_path):
    """The reference's scenes/*.json ARE outputs of SceneDescriptor::save (serde_json::to_string_pretty): loading
    one and saving it again must reproduce the file byte for byte (float formatting = ryu f32, layout = 2-space
    pretty printer).  Five of the six files predate the removal of the camera's `updating_direction` field."""
    src = os.path.join(REF, "scenes", sid + ".json")
    rc, h = load_c(src, base=REF)
    assert rc == 0, L.pt_last_error()
    path = str(tmp_path / "o.json")
    assert L.pt_scene_save(h, path.encode()) == 0
    ours = open(path).read()
    theirs = open(src).read().replace('    "updating_direction": null,\n', "")
    assert ours == theirs
    L.pt_scene_free(h)


# ---------------------------------------------------------------------------------------------------------------
# built-in scenes (setup_scenes, scenes.rs:43-318)
L.pt_scene_builtin.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
L.pt_builtin_scene_id.restype = C.c_char_p
L.pt_builtin_scene_id.argtypes = [C.c_uint32]
BUILTIN_ORDER = ["single-sphere", "cartesian", "two-spheres", "three-spheres", "cornell", "mesh"]


def builtin(sid, base=None):
    h = C.c_void_p()
    rc = L.pt_scene_builtin(sid.encode(), (base or ptlib.ROOT).encode(), C.byref(h))
    return rc, h


def test_builtin_listing_is_the_reference_order():
    assert [L.pt_builtin_scene_id(i).decode() for i in range(L.pt_builtin_scene_count())] == BUILTIN_ORDER
    assert L.pt_builtin_scene_id(L.pt_builtin_scene_count()) is None


@pytest.mark.parametrize("sid", BUILTIN_ORDER)
def test_builtin_scene_equals_the_shipped_file(sid, tmp_path):
    """The shipped scenes/*.json were written by SceneDescriptor::save from setup_scenes' values, so generating
    a scene in code and saving it must give the bytes that loading the shipped file and saving it gives - objects,
    quads, bounding spheres (Mesh::new's min + max*0.5 centre), the 12 bounding-box triangles and the camera.
    Only mesh.json's camera was moved in the GUI before it was saved: there the built-in camera is checked against
    CameraData::new((0.9, -0.2, 7.8), normalize(-0.09, -0.06, -1)) and the rest against the file."""
    rc, h = builtin(sid)
    assert rc == 0, L.pt_last_error()
    rc, g = load_c(ptlib.scene_path(sid))
    assert rc == 0, L.pt_last_error()
    if sid == "mesh":
        cam = L.pt_scene_camera(h).contents
        f = np.float32
        d = np.array([-0.09, -0.06, -1.0], f)
        inv = f(1.0) / np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], dtype=f)
        assert list(cam.direction) == [float(x) for x in d * inv]
        assert list(cam.position) == [float(f(0.9)), float(f(-2.0) + f(1.8)), float(f(8.8) - f(1.0))]
        assert (cam.focal_length, cam.sensor_width, cam.aspect_ratio) == (float(f(0.035)), float(f(0.036)), 1.5)
        assert L.pt_scene_set_camera(h, L.pt_scene_camera(g)) == 0
    a, b = str(tmp_path / "builtin.json"), str(tmp_path / "loaded.json")
    assert L.pt_scene_save(h, a.encode()) == 0 and L.pt_scene_save(g, b.encode()) == 0
    assert open(a, "rb").read() == open(b, "rb").read()
    n, m = C.c_uint32(), C.c_uint32()
    n2, m2 = C.c_uint32(), C.c_uint32()
    assert as_bytes(L.pt_scene_objects(h, C.byref(n)), n.value) == as_bytes(L.pt_scene_objects(g, C.byref(n2)), n2.value)
    assert as_bytes(L.pt_scene_triangles(h, C.byref(m)), m.value) == as_bytes(L.pt_scene_triangles(g, C.byref(m2)), m2.value)
    L.pt_scene_free(h)
    L.pt_scene_free(g)


def test_builtin_errors(tmp_path):
    h = C.c_void_p()
    assert L.pt_scene_builtin(b"no-such-scene", b".", C.byref(h)) == -1 and not h  # PT_ERR_INVALID
    assert L.pt_scene_builtin(None, b".", C.byref(h)) == -1
    # "mesh" needs meshes/mctri.off under base_dir (the reference panics in load_off, mod.rs:309)
    assert L.pt_scene_builtin(b"mesh", str(tmp_path).encode(), C.byref(h)) < 0 and not h
    assert b"mctri.off" in L.pt_last_error()
