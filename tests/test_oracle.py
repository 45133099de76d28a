"""Pin the CPU restatement (oracle/) against every known-answer test the reference holds for the path:
the 7 tests of src/render/test.rs, restated with the same inputs and the same exact expectations."""
import ctypes as C
import os
import math

import numpy as np
import pytest

import ptlib
from ptlib import PtoCounters, _np_f, f3, make_camera, make_sphere, Scene

L = ptlib.oracle()


def vec_ops(a, b, s):
    out = np.zeros(23, dtype=np.float32)
    L.pto_vec_ops(_np_f(np.array(a, dtype=np.float32)), _np_f(np.array(b, dtype=np.float32)), s, _np_f(out))
    return dict(add=out[0:3], sub=out[3:6], mul=out[6:9], scale=out[9:12], div=out[12:15], dot=out[15],
                cross=out[16:19], normalize=out[19:22], length=out[22])


def test_vector_operations():
    """src/render/test.rs:3-27 (glam Vec3 semantics)."""
    v1, v2, v3 = (1.0, 2.0, 3.0), (2.0, 3.0, 4.0), (3.0, 4.0, 5.0)
    r = vec_ops(v1, v2, 2.0)
    assert list(r["add"]) == [3.0, 5.0, 7.0]
    assert list(vec_ops(v3, v2, 1.0)["sub"]) == [1.0, 1.0, 1.0]
    assert list(r["mul"]) == [2.0, 6.0, 12.0]
    assert list(r["scale"]) == [2.0, 4.0, 6.0]
    assert list(vec_ops(v2, v1, 2.0)["div"]) == [1.0, 1.5, 2.0]
    assert r["dot"] == 20.0
    assert list(r["cross"]) == [-1.0, 2.0, -1.0]
    assert list(vec_ops((1.0, 0.0, 0.0), v1, 1.0)["normalize"]) == [1.0, 0.0, 0.0]
    n = vec_ops((1.0, 1.0, 0.0), v1, 1.0)["normalize"]
    assert list(n) == [np.float32(0.7071067811865475), np.float32(0.7071067811865475), 0.0]
    assert r["length"] == np.float32(3.7416573867739413)


def test_helpers():
    """src/render/test.rs:29-35."""
    for x, want in [(0.0, 0), (0.5, 186), (0.75, 224), (1.0, 255)]:
        assert L.pto_to_int_with_gamma_correction(x) == want
    # outside the reference's test: clamp and NaN behaviour of `as usize`
    assert L.pto_to_int_with_gamma_correction(-3.0) == 0
    assert L.pto_to_int_with_gamma_correction(7.0) == 255
    assert L.pto_to_int_with_gamma_correction(float("nan")) == 0


TEST_MAT = dict(color=(1.0, 0.0, 0.0), emission=(0.0, 0.0, 0.0), reflect="Diffuse")


def one_sphere_scene(pos):
    cam = make_camera((0, 0, 0), (0, 0, -1))
    return Scene("t", cam, [make_sphere(pos, 1.0, **TEST_MAT)], [])


def isect(scene, o, d):
    o = np.array(o, dtype=np.float32)
    d = np.array(d, dtype=np.float32)
    t = np.zeros(1, dtype=np.float32)
    oid = np.zeros(1, dtype=np.int32)
    tid = np.zeros(1, dtype=np.int32)
    x = np.zeros(3, dtype=np.float32)
    n = np.zeros(3, dtype=np.float32)
    ps = scene.pto()
    L.pto_intersect_batch(C.byref(ps), _np_f(o), _np_f(d), 1, _np_f(t), oid.ctypes.data_as(ptlib.i32p),
                          tid.ctypes.data_as(ptlib.i32p), _np_f(x), _np_f(n))
    return int(oid[0]), float(t[0]), list(x), list(n)


SPHERE_KATS = [
    # (sphere position, ray origin, ray direction, expected (object_id, distance, x, n) or None)
    ((0, 0, -3), (0, 0, 0), (0, 0, -1), (0, 2.0, [0, 0, -2], [0, 0, 1])),        # test.rs:43-69
    ((0, 0, -3), (2, 0, 0), "norm(1,0,-1)", None),                                # test.rs:72-87
    ((0, 0, 0), (0, 0, 0), (0, 0, -1), (0, 1.0, [0, 0, -1], [0, 0, -1])),         # test.rs:90-116
    ((0, 0, -3), (0, 1, 0), (0, 0, -1), (0, 3.0, [0, 1, -3], [0, 1, 0])),         # test.rs:119-144
]


@pytest.mark.parametrize("pos,o,d,want", SPHERE_KATS)
def test_intersect_scene_kats(pos, o, d, want):
    if isinstance(d, str):
        d = list(vec_ops((1.0, 0.0, -1.0), (0, 0, 0), 1.0)["normalize"])
    oid, t, x, n = isect(one_sphere_scene(pos), o, d)
    if want is None:
        assert oid == -1
    else:
        assert (oid, t, x, n) == (want[0], want[1], [float(v) for v in want[2]], [float(v) for v in want[3]])


def test_radiance():
    """src/render/test.rs:146-183: mean of 10 000 samples of one ray, .x > 0.3 (analytic 50/144 = 0.34722),
    y and z exactly 0 (red diffuse sphere)."""
    cam = make_camera((0, 0, 0), (0, 0, -1))
    sc = Scene("t", cam, [make_sphere((0, 0, -3), 1.0, (1, 0, 0), (0, 0, 0), "Diffuse"),
                          make_sphere((0, 0, 10), 1.0, (0, 0, 0), (50, 50, 50), "Diffuse")], [])
    ps = sc.pto()
    o = np.array([0, 0, 0], dtype=np.float32)
    d = np.array([0, 0, -1], dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    cnt = PtoCounters()
    L.pto_radiance_mean(C.byref(ps), _np_f(o), _np_f(d), 1, 0, 10000, _np_f(out), C.byref(cnt))
    assert out[0] > 0.3, out
    assert out[1] == 0.0 and out[2] == 0.0
    # tighter, with 400k samples: analytic expectation 50 * (1/12)^2
    L.pto_radiance_mean(C.byref(ps), _np_f(o), _np_f(d), 7, 3, 400000, _np_f(out), C.byref(cnt))
    assert abs(out[0] - 50.0 / 144.0) < 0.02, out


# ------------------------------------------------------------------ third-party arithmetic
PHILOX_KATS = [  # Random123 kat_vectors, philox4x32 10 rounds: counter, key, expected
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


PHILOX7_KATS = [  # Random123 kat_vectors, philox4x32 7 rounds (the rounds the RNG contract draws with)
    ([0, 0, 0, 0], [0, 0], [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]),
]


@pytest.mark.parametrize("ctr,key,want", PHILOX_KATS)
def test_philox_kat(ctr, key, want):
    out = (C.c_uint32 * 4)()
    L.pto_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    assert list(out) == want


@pytest.mark.parametrize("ctr,key,want", PHILOX7_KATS)
def test_philox7_kat_and_the_draws_use_it(ctr, key, want):
    out = (C.c_uint32 * 4)()
    L.pto_philox4x32_7((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    assert list(out) == want
    # pto_draw4 (what radiance() and render_pixel draw from): the same words through rand 0.8.5's u32 -> f32 map
    u = (C.c_float * 4)()
    L.pto_draw4(key[0] | (key[1] << 32), ctr[0], ctr[1], ctr[2], u)
    if ctr[3] == 0:
        assert list(u) == [L.pto_u32_to_unit(w) for w in want]


def test_u32_to_unit():
    """rand 0.8.5 Standard<f32>: 24 high bits, [0,1)."""
    assert L.pto_u32_to_unit(0) == 0.0
    assert L.pto_u32_to_unit(0xffffffff) == float(np.float32(1.0) - np.float32(2.0 ** -24))
    assert L.pto_u32_to_unit(0x80000000) == 0.5
    assert L.pto_u32_to_unit(0x000000ff) == 0.0


def test_sincos_equals_platform_libm_on_whole_domain():
    """Rust f32::sin/cos -> libm sinf/cosf.  The path only evaluates them at r1 = 2*PI*(k*2^-24)
    (mod.rs:691,703): check all 2^24 arguments bit for bit against this machine's libm."""
    ms, mc = C.c_uint64(), C.c_uint64()
    L.pto_sincos_vs_libm(0, 1 << 24, C.byref(ms), C.byref(mc))
    # glibc >= 2.28 gives 0; allow a handful for libms whose contraction of the double polynomial differs
    assert ms.value <= 4 and mc.value <= 4, (ms.value, mc.value)


def test_sincos_accuracy():
    xs = np.linspace(0, 2 * math.pi, 20001).astype(np.float32)
    for x in xs[::37]:
        assert abs(L.pto_sinf(float(x)) - math.sin(float(x))) < 1.2e-7
        assert abs(L.pto_cosf(float(x)) - math.cos(float(x))) < 1.2e-7


def test_camera_basis_cornell():
    """SURVEY Appendix A numbers (numpy f32 probe) for the shared camera."""
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    lens, su, sv = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    L.pto_camera_basis(C.byref(sc.cam), lens, su, sv)
    assert np.allclose(list(lens), [0, -0.20209628, 7.765063], atol=1e-6)
    assert np.allclose(list(su), [0.036, 0, 0], atol=1e-7)
    assert np.allclose(list(sv), [0, 0.02395691, -0.00143741], atol=1e-7)


def test_mesh_bounding_sphere_matches_stored():
    """Mesh::new (mod.rs:450-499) recomputed on the inline quads reproduces the bounding spheres the
    reference serialised into cornell.json (incl. the min+max*0.5 centre)."""
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    for i in range(sc.n_objs):
        o = sc.objs[i]
        if o.kind != ptlib.PT_MESH:
            continue
        arr = (ptlib.PtTriangle * o.tri_count)(*[sc.tris[o.tri_offset + k] for k in range(o.tri_count)])
        ctr, rad = (C.c_float * 3)(), C.c_float()
        L.pto_mesh_bounding_sphere(arr, o.tri_count, ctr, C.byref(rad))
        assert list(ctr) == list(o.bs_center), i
        assert rad.value == o.bs_radius, i


def _siphash_py(c_rounds, d_rounds, k0, k1, data):
    """SipHash-c-d (Aumasson & Bernstein), written independently of oracle/ and of the product, for the KATs below."""
    M = (1 << 64) - 1
    rotl = lambda x, b: ((x << b) | (x >> (64 - b))) & M
    v = [k0 ^ 0x736f6d6570736575, k1 ^ 0x646f72616e646f6d, k0 ^ 0x6c7967656e657261, k1 ^ 0x7465646279746573]

    def rnd():
        v[0] = (v[0] + v[1]) & M; v[1] = rotl(v[1], 13); v[1] ^= v[0]; v[0] = rotl(v[0], 32)
        v[2] = (v[2] + v[3]) & M; v[3] = rotl(v[3], 16); v[3] ^= v[2]
        v[0] = (v[0] + v[3]) & M; v[3] = rotl(v[3], 21); v[3] ^= v[0]
        v[2] = (v[2] + v[1]) & M; v[1] = rotl(v[1], 17); v[1] ^= v[2]; v[2] = rotl(v[2], 32)

    n = len(data)
    for i in range(0, n - n % 8, 8):
        m = int.from_bytes(data[i:i + 8], "little")
        v[3] ^= m
        for _ in range(c_rounds):
            rnd()
        v[0] ^= m
    b = ((n & 0xff) << 56) | int.from_bytes(data[n - n % 8:], "little")
    v[3] ^= b
    for _ in range(c_rounds):
        rnd()
    v[0] ^= b
    v[2] ^= 0xff
    for _ in range(d_rounds):
        rnd()
    return v[0] ^ v[1] ^ v[2] ^ v[3]


def test_siphash13_kat():
    """Image.hash is Rust's DefaultHasher = SipHash-1-3 with a zero key (mod.rs:916-926).  The Python restatement
    above is pinned on the published SipHash-2-4 vector (reference paper, key 00..0f, message 00..0e) and on the first
    SipHash-1-3 vector of Rust's own library tests (library/core/tests/hash/sip.rs, key 00..0f, empty message:
    dc c4 0f 05 58 01 ac ab); the oracle then has to agree with it, zero key, on fixed and random messages."""
    k0, k1 = 0x0706050403020100, 0x0f0e0d0c0b0a0908
    assert _siphash_py(2, 4, k0, k1, bytes(range(15))) == 0xa129ca6149be45e5
    assert _siphash_py(1, 3, k0, k1, b"") == 0xabac0158050fc4dc
    assert L.pto_siphash13(b"", 0) == 0xd1fba762150c532c == _siphash_py(1, 3, 0, 0, b"")
    rng = np.random.default_rng(5)
    for n in (1, 7, 8, 9, 12, 16, 31, 4096):
        msg = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert L.pto_siphash13(msg, n) == _siphash_py(1, 3, 0, 0, msg), n
    img = rng.random(33, dtype=np.float32)
    assert L.pto_image_hash(_np_f(img), img.size) == _siphash_py(1, 3, 0, 0, img.tobytes())


# ---------------------------------------------------------------------------------------------------------------
# hand-derived KATs for what the reference's tests do not reach (tests/kats.py): triangles, the gate, the tie rules
import kats


@pytest.mark.parametrize("name,build,o,d,want", kats.CASES, ids=[c[0] for c in kats.CASES])
def test_hand_derived_kats_on_the_oracle(name, build, o, d, want):
    sc = build()
    t, oid, tid, x, n = ptlib.oracle_intersect(sc, o, d)
    if want is None:
        assert oid[0] == -1, (name, oid[0], t[0])
        return
    assert (oid[0], tid[0]) == (want["object_id"], want["tri_id"]), name
    assert t[0] == np.float32(want["t"]), (name, t[0])
    assert list(x[0]) == [np.float32(v) for v in want["x"]], (name, x[0])
    assert list(n[0]) == [np.float32(v) for v in want["n"]], (name, n[0])


# ---------------------------------------------------------------------------------------------------------------
# MOCK_RANDOM (mod.rs:31-51): the reference's only deterministic mode, restated so that a cargo holder can diff a frame
def test_mock_random_mode():
    """rand01() with MOCK_RANDOM = true walks a 9-entry table through one global counter, pixels in index order on one
    thread (mod.rs:1017-1018).  Checked here: determinism; the draw count lies between the bounds the path's rand01()
    calls allow (2 per sample, 2 per surviving diffuse bounce, 1 per roulette); the PPM header of the reference's
    writer; and that this stream is not the Philox stream of the parity contract."""
    sc = ptlib.load_scene_py(ptlib.scene_path("two-spheres"))  # diffuse spheres only: every bounce draws r1, r2
    w, h, spp = 24, 16, 4
    img, cnt, draws = ptlib.oracle_render_mock(sc, w, h, spp)
    img2, cnt2, draws2 = ptlib.oracle_render_mock(sc, w, h, spp)
    assert np.array_equal(img, img2) and draws == draws2 and cnt.ray_bounces == cnt2.ray_bounces
    samples = w * h * spp
    hits = cnt.ray_bounces - cnt.misses
    assert hits > 0 and img.max() > 0.0
    # a hit at new_depth <= 5 draws r1, r2; at new_depth > 5 the roulette number first and r1, r2 only if it survives
    assert 2 * samples < draws <= 2 * samples + 3 * hits
    # the counter is global: the frame's first row rendered as a frame of its own geometry would restart at entry 0;
    # here the same frame at twice the samples must not simply repeat the 4-spp stream (9 does not divide the draws)
    img8, _, draws8 = ptlib.oracle_render_mock(sc, w, h, 2 * spp)
    assert draws8 > draws and not np.array_equal(img8, img)
    ppm_len = L.pto_format_ppm(_np_f(img), w, h, spp, b"two-spheres", 0, None, 0)
    buf = C.create_string_buffer(ppm_len)
    assert L.pto_format_ppm(_np_f(img), w, h, spp, b"two-spheres", 0, buf, ppm_len) == ppm_len
    assert buf.raw.startswith(b"P3\n# samplesPerPixel: 4, resolution_y: 16, scene_id: two-spheres\n# rendering time: 0 s\n24 16\n255\n")
    # Philox frames (the parity contract of the GPU path) are a different stream: not the same picture
    ph, _, _ = ptlib.oracle_render(sc, w, h, spp, 1)
    assert not np.array_equal(ph, img)


def test_mock_random_first_sample_by_hand():
    """First sample of pixel 0 of single-sphere.json at 4x4, by hand from mod.rs:805-843 with the first two table
    entries (0.75902418, 0.023879213): r1 = 1.5180483 >= 1 -> xfilter = 1 - sqrt(2 - r1); r2 = 0.047758426 < 1 ->
    yfilter = sqrt(r2) - 1.  Pixel 0 is x = 0, y = H-1 = 3 (mod.rs:805-806).  The ray is compared with the oracle's
    through its own primary-ray arithmetic evaluated in numpy f32."""
    sc = ptlib.load_scene_py(ptlib.scene_path("single-sphere"))
    f = np.float32
    r1, r2 = f(2.0) * f(0.75902418061906407), f(2.0) * f(0.023879213030728041)
    xf = f(1.0) - np.sqrt(f(2.0) - r1)
    yf = np.sqrt(r2) - f(1.0)
    sx = (f(0) + f(0.5) * (f(0.5) + f(0) + xf)) / f(4) - f(0.5)
    sy = (f(3) + f(0.5) * (f(0.5) + f(0) + yf)) / f(4) - f(0.5)
    lens, su, sv = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    L.pto_camera_basis(C.byref(sc.cam), lens, su, sv)
    pos = np.array(list(sc.cam.position), np.float32)
    sensor = (pos + np.array(list(su), np.float32) * sx) + np.array(list(sv), np.float32) * sy
    dvec = np.array(list(lens), np.float32) - sensor
    dlen = np.sqrt((dvec[0] * dvec[0] + dvec[1] * dvec[1]) + dvec[2] * dvec[2])
    dvec = dvec * (f(1.0) / dlen)
    # with 1 spp the whole 4x4 frame makes 16 primary rays; single-sphere has one emissive diffuse sphere: a ray that
    # hits it returns emission + color * radiance(next), so pixel values depend on the draw order as well
    img, cnt, draws = ptlib.oracle_render_mock(sc, 4, 4, 1)
    t, oid, _, _, _ = ptlib.oracle_intersect(sc, list(lens), dvec)
    # pixel 0 is a corner of the frame: its ray misses the unit sphere at the origin -> exactly 2 draws, black
    assert oid[0] == -1 and list(img[0]) == [0.0, 0.0, 0.0]
    assert draws >= 32


def test_intersect_bounds_and_orbit_point():
    """SceneObjectData::intersect_bounds (mod.rs:282-290) and get_orbit_point (viewport_tab.rs:401-431) by hand on the
    unit triangle: its Mesh::new box is the degenerate slab [0,1]x[0,1]x{0}; a ray through (0.75, 0.75) misses the
    triangle (u + v = 1.5) but hits box triangle (0,1,2) = (0,0,0),(1,0,0),(1,1,0) at t = 1 - the orbit point falls
    back to the bounds hit; through (0.25, 0.25) the real hit is used."""
    sc = kats.mesh_scene([kats.UNIT])
    boxes = ptlib.oracle_boxes(sc)
    assert [list(boxes[0].a), list(boxes[0].b), list(boxes[0].c)] == [[0, 0, 0], [1, 0, 0], [1, 1, 0]]
    assert [list(boxes[1].a), list(boxes[1].b), list(boxes[1].c)] == [[0, 0, 0], [1, 1, 0], [0, 1, 0]]
    o = np.array([[0.75, 0.75, 1.0], [0.25, 0.25, 1.0], [2.0, 2.0, 1.0]], np.float32)
    d = np.array([[0, 0, -1.0]] * 3, np.float32)
    hit = np.zeros(3, np.int32)
    t, x, n = np.zeros(3, np.float32), np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32)
    ps = sc.pto()
    L.pto_intersect_bounds_batch(C.byref(ps), boxes, 0, _np_f(o), _np_f(d), 3, hit.ctypes.data_as(ptlib.i32p), _np_f(t),
                                 _np_f(x), _np_f(n))
    assert list(hit) == [1, 1, 0] and list(t[:2]) == [1.0, 1.0]
    assert list(x[0]) == [0.75, 0.75, 0.0]
    found, obj = np.zeros(3, np.int32), np.zeros(3, np.int32)
    pt, tt = np.zeros((3, 3), np.float32), np.zeros(3, np.float32)
    L.pto_orbit_point_batch(C.byref(ps), boxes, _np_f(o), _np_f(d), 3, found.ctypes.data_as(ptlib.i32p), _np_f(pt),
                            obj.ctypes.data_as(ptlib.i32p), _np_f(tt))
    assert list(found) == [1, 1, 0] and list(obj) == [0, 0, -1]
    assert list(pt[0]) == [0.75, 0.75, 0.0] and list(pt[1]) == [0.25, 0.25, 0.0]
    # shipped scene: the boxes Mesh::new computes are the ones the reference serialised into cornell.json
    import json
    cj = json.load(open(ptlib.scene_path("cornell")))
    sc2 = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    b2 = ptlib.oracle_boxes(sc2)
    for i, od in enumerate(cj["objects"]):
        if "Mesh" not in od["type_"]:
            continue
        stored = od["type_"]["Mesh"]["bounding_box"]
        for k in range(12):
            for key in "abc":
                assert [np.float32(v) for v in stored[k][key]] == list(getattr(b2[12 * i + k], key)), (i, k, key)


# ---------------------------------------------------------------------------------------------------------------
# committed golden vectors (tests/golden/*.npz, written by tools/make_golden.py)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_SCENES = ["single-sphere", "two-spheres", "three-spheres", "cartesian", "cornell", "mesh"]


@pytest.mark.parametrize("sid", GOLDEN_SCENES)
def test_oracle_reproduces_golden_scene(sid):
    """The oracle of today gives the committed frame, counters and per-ray intersections bit for bit (any thread
    count): the fixtures pin the restatement against drift."""
    g = np.load(os.path.join(GOLDEN, "scene_%s.npz" % sid))
    O = ptlib.oracle()
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    w, h, spp, seed = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["seed"])
    img, cnt, _ = ptlib.oracle_render(sc, w, h, spp, seed)
    assert cnt.ray_bounces == int(g["ray_bounces"])
    assert np.array_equal(img.view(np.uint32), g["image"].view(np.uint32))
    o, d = np.ascontiguousarray(g["ray_o"]), np.ascontiguousarray(g["ray_d"])
    m = len(o)
    t, oid, tid = np.zeros(m, np.float32), np.zeros(m, np.int32), np.zeros(m, np.int32)
    x, nr = np.zeros((m, 3), np.float32), np.zeros((m, 3), np.float32)
    ps = sc.pto()
    O.pto_intersect_batch(C.byref(ps), ptlib._np_f(o), ptlib._np_f(d), m, ptlib._np_f(t), oid.ctypes.data_as(ptlib.i32p),
                          tid.ctypes.data_as(ptlib.i32p), ptlib._np_f(x), ptlib._np_f(nr))
    assert np.array_equal(oid, g["hit_object"]) and np.array_equal(tid, g["hit_triangle"])
    for got, key in ((t, "hit_t"), (x, "hit_x"), (nr, "hit_n")):
        assert np.array_equal(got.view(np.uint32), g[key].view(np.uint32)), key


def test_oracle_reproduces_golden_numerics():
    g = np.load(os.path.join(GOLDEN, "numerics.npz"))
    O = ptlib.oracle()
    s = np.array([O.pto_sinf(float(v)) for v in g["sincos_x"]], np.float32)
    c = np.array([O.pto_cosf(float(v)) for v in g["sincos_x"]], np.float32)
    assert np.array_equal(s.view(np.uint32), g["sin"].view(np.uint32))
    assert np.array_equal(c.view(np.uint32), g["cos"].view(np.uint32))
    out = np.zeros(4, np.uint32)
    for ctr, key, want in zip(g["philox_ctr"], g["philox_key"], g["philox_out"]):
        ctr, key = np.ascontiguousarray(ctr), np.ascontiguousarray(key)
        O.pto_philox4x32_7(ctr.ctypes.data_as(ptlib.u32p), key.ctypes.data_as(ptlib.u32p), out.ctypes.data_as(ptlib.u32p))
        assert np.array_equal(out, want)
    gi = np.array([O.pto_to_int_with_gamma_correction(float(v)) for v in g["gamma_x"]], np.uint32)
    assert np.array_equal(gi, g["gamma_int"])


# ---------------------------------------------------------------------------------------------------------------
# round 3: hand-derived KATs for the shading half of radiance() (tests/kats_shading.py), on the oracle
import kats_shading


@pytest.mark.parametrize("name,build,o,d,depth,want", kats_shading.CASES, ids=[c[0] for c in kats_shading.CASES])
def test_hand_derived_shading_kats(name, build, o, d, depth, want):
    """mod.rs:661-792 worked out by hand on exact geometry: the oracle's radiance() must return the closed expression."""
    sc = build()
    if want[0] == "exact":
        for pixel in range(16):
            got, cnt = ptlib.oracle_radiance(sc, o, d, depth, 1, 11, pixel)
            assert np.array_equal(got, want[1]), (name, pixel, got, want[1])
        got, _ = ptlib.oracle_radiance(sc, o, d, depth, 64, 11, 3)
        assert np.allclose(got, want[1], rtol=1e-6, atol=0), (name, got)
        return
    _, a, b, p_a = want
    seen_a = seen_b = 0
    for pixel in range(64):  # one sample each: one of the two outcomes, exactly
        got, _ = ptlib.oracle_radiance(sc, o, d, depth, 1, 11, pixel)
        is_a, is_b = np.array_equal(got, a), np.array_equal(got, b)
        assert is_a or is_b, (name, pixel, got, a, b)
        seen_a += is_a
        seen_b += is_b
    assert seen_a > 0 and seen_b > 0, (name, seen_a, seen_b)
    n = 20000
    got, _ = ptlib.oracle_radiance(sc, o, d, depth, n, 11, 5)
    mean = p_a * a.astype(np.float64) + (1.0 - p_a) * b.astype(np.float64)
    sd = np.abs(a.astype(np.float64) - b.astype(np.float64)) * np.sqrt(p_a * (1.0 - p_a) / n)
    assert np.all(np.abs(got - mean) <= 5.0 * sd + 1e-6), (name, got, mean, sd)


# ------------------------------------------------------------------ render_pixel's sensor mapping (mod.rs:805-843)
def _cam_of(d):
    return ptlib.make_camera(d["position"], d["direction"], d["focal_length"], d["sensor_width"], d["aspect_ratio"])


def test_kats_camera_philox_restatement_against_random123():
    import kats_camera as K

    for ctr, key, want in K.PHILOX7_KATS:
        assert K.philox4x32(ctr, key) == want
    # the numpy restatement's draws are the oracle's (pto_draw4: words through rand 0.8.5's map), tag 0
    u = (C.c_float * 4)()
    for _, _, _, pix, smp, seed in K.CASES:
        L.pto_draw4(seed, pix, smp, 0, u)
        u1, u2 = K.camera_draws(seed, pix, smp)
        assert (u[0], u[1]) == (float(u1), float(u2))
    # the cases reach both branches of the tent filter with both draws, and the branch point itself (r = 1.0 exactly)
    draws = [K.camera_draws(seed, pix, smp) for _, _, _, pix, smp, seed in K.CASES]
    for i in (0, 1):
        assert any(d[i] < 0.5 for d in draws) and any(d[i] > 0.5 for d in draws) and any(d[i] == 0.5 for d in draws)
    assert K.camera_draws(K.SEED_R1_IS_ONE, 0, 0)[0] == 0.5 and K.camera_draws(K.SEED_R2_IS_ONE, 0, 0)[1] == 0.5
    assert K.tent(np.float32(1.0)) == 0.0 and K.tent(np.float32(2.0) * np.float32(0.5 - 2.0 ** -24)) < 0.0


def test_primary_ray_against_the_independent_restatement():
    """pto_primary_ray (what the oracle's render_pixel casts) == tests/kats_camera.py's numpy-f32 reading of
    mod.rs:805-843 + mod.rs:211-232, bit for bit, on every case: corners, centre, the four sub-pixels, r on both sides of
    1.0 and exactly 1.0, non-square pixels, both `up` vectors of orthogonals()."""
    import kats_camera as K

    o, d = (C.c_float * 3)(), (C.c_float * 3)()
    for case in K.CASES:
        cam, w, h, pix, smp, seed = case
        want_o, want_d = K.expected(case)
        pc = _cam_of(cam)
        L.pto_primary_ray(C.byref(pc), w, h, pix, smp, seed, o, d)
        got_o, got_d = np.array(list(o), np.float32), np.array(list(d), np.float32)
        assert np.array_equal(got_o.view(np.uint32), want_o.view(np.uint32)), (case, got_o, want_o)
        assert np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32)), (case, got_d, want_d)
    # by hand: the centre of the branch point.  r1 = r2 = 1.0 -> xfilter = yfilter = 0; pixel 0 of a 1x1 frame: x = y = 0,
    # sample 3 -> xsub = ysub = 1: sx = sy = (0 + 0.5 * 1.5) / 1 - 0.5 = 0.25 exactly
    cam = dict(position=(0.0, 0.0, 0.0), direction=(0.0, 0.0, -1.0), focal_length=0.5, sensor_width=2.0, aspect_ratio=2.0)
    ro, rd = K.primary_ray(cam, 1, 1, 0, 3, np.float32(0.5), np.float32(0.5))
    # su = normalize((0,0,-1) x (0,1,0)) * 2 = (1,0,0) * 2, sv = (su x dir) * 1 = (0,1,0); sensor = (0.5, 0.25, 0); lens = (0,0,-0.5)
    assert list(ro) == [0.0, 0.0, -0.5]
    want = np.array([-0.5, -0.25, -0.5], np.float32)
    want = want * (np.float32(1.0) / np.sqrt(np.float32(0.5625)))
    assert np.array_equal(rd, want) and list(rd) == [np.float32(-0.5) * np.float32(1.0 / 0.75), np.float32(-0.25) * np.float32(1.0 / 0.75),
                                                      np.float32(-0.5) * np.float32(1.0 / 0.75)]


# ------------------------------------------------------------------ whole frames in the reference's MOCK_RANDOM mode
MOCK_FRAMES = [("cornell", 32, 4), ("mesh", 16, 2), ("three-spheres", 16, 4)]  # (scene, res_y, spp); width = res_y * 3 / 2


@pytest.mark.parametrize("sid,res_y,spp", MOCK_FRAMES)
def test_mock_random_frames_against_committed_ppm(sid, res_y, spp):
    """tests/golden/mock_<scene>_<res_y>_<spp>.ppm = the file the reference writes for this frame with MOCK_RANDOM = true
    (mod.rs:31-51, 1017-1018, 1031-1076) as the ORACLE renders it (tools/make_golden.py) - the fixed target of the cargo
    diff INTEGRATION.md describes, and a drift guard on pto_render_mock / pto_format_ppm until someone can run it."""
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    width = res_y * 3 // 2
    img, cnt, draws = ptlib.oracle_render_mock(sc, width, res_y, spp)
    n = L.pto_format_ppm(_np_f(img), width, res_y, spp, sid.encode(), 0, None, 0)
    buf = C.create_string_buffer(n)
    L.pto_format_ppm(_np_f(img), width, res_y, spp, sid.encode(), 0, buf, n)
    want = open(os.path.join(ptlib.ROOT, "tests", "golden", "mock_%s_%d_%d.ppm" % (sid, res_y, spp)), "rb").read()
    assert buf.raw[:n] == want
