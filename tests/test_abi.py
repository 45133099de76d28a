"""No-GPU checks of the boundary: the C-ABI library loads, exports every symbol include/ptrace.h declares,
and its host-only entry points (camera basis, Mesh::new bounds, gamma, PPM bytes) agree with the oracle.
No compute entry point is called with real work here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ptlib
from ptlib import _np_f

L = ptlib.product()
O = ptlib.oracle()
HEADER = os.path.join(ptlib.ROOT, "include", "ptrace.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if n not in ("pt_progress_fn",)))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 45
    for must in ("pt_ctx_intersect_bounds", "pt_ctx_orbit_point", "pt_comm_gather_frame", "pt_scene_bounding_box"):
        assert must in names
    for n in names:
        assert hasattr(L, n), "libptrace_hip.so does not export " + n


def test_version_and_struct_sizes():
    assert L.pt_abi_version() == 5
    assert b"gfx950" in L.pt_version()
    # POD layout the Rust/cgo/ctypes side must match (include/ptrace.h)
    assert C.sizeof(ptlib.PtCamera) == 36
    assert C.sizeof(ptlib.PtTriangle) == 36
    assert C.sizeof(ptlib.PtObject) == 72
    assert C.sizeof(ptlib.PtConfig) == 56
    assert C.sizeof(ptlib.PtStats) == 56


def test_no_device_means_error_not_fallback():
    """Without a GPU the compute entry points refuse to run (there is no CPU path in the product)."""
    if L.pt_device_count() > 0:
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    assert L.pt_ctx_create(0, C.byref(ctx)) == -2  # PT_ERR_NO_DEVICE
    assert b"no HIP device" in L.pt_last_error()
    sc = ptlib.load_scene_py(ptlib.scene_path("single-sphere"))
    cfg = ptlib.PtConfig(8, 8, 1, 0, 1, 0, 0, 0, 0)
    out = np.full((64, 3), 7.0, np.float32)
    st = ptlib.PtStats()
    rc = L.pt_render(C.byref(cfg), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None, None,
                     None, C.byref(st))
    assert rc == -2 and (out == 7.0).all()


@pytest.mark.parametrize("sid", ["cornell", "mesh"])
def test_camera_basis_matches_oracle(sid):
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    a = [(C.c_float * 3)() for _ in range(3)]
    b = [(C.c_float * 3)() for _ in range(3)]
    assert L.pt_camera_basis(C.byref(sc.cam), *a) == 0
    O.pto_camera_basis(C.byref(sc.cam), *b)
    for x, y in zip(a, b):
        assert bytes(x) == bytes(y)


def test_mesh_bounding_sphere_matches_oracle_and_reference_data():
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    for i in range(sc.n_objs):
        o = sc.objs[i]
        if o.kind != ptlib.PT_MESH:
            continue
        arr = (ptlib.PtTriangle * o.tri_count)(*[sc.tris[o.tri_offset + k] for k in range(o.tri_count)])
        ctr, rad = (C.c_float * 3)(), C.c_float()
        assert L.pt_mesh_bounding_sphere(arr, o.tri_count, ctr, C.byref(rad)) == 0
        assert list(ctr) == list(o.bs_center) and rad.value == o.bs_radius  # values serialised by the reference


def test_gamma_kats_and_oracle_agreement():
    for x, want in [(0.0, 0), (0.5, 186), (0.75, 224), (1.0, 255)]:  # src/render/test.rs:29-35
        assert L.pt_to_int_with_gamma_correction(x) == want
    xs = np.concatenate([np.linspace(-0.5, 1.5, 4001), [np.nan, np.inf, -np.inf]]).astype(np.float32)
    for x in xs:
        assert L.pt_to_int_with_gamma_correction(float(x)) == O.pto_to_int_with_gamma_correction(float(x))


def test_ppm_bytes_match_oracle(tmp_path):
    """P3 writer (mod.rs:1043-1076): header lines, reversed pixel order, trailing space, no final newline."""
    rng = np.random.default_rng(1)
    w, h = 7, 5
    img = rng.uniform(-0.2, 1.3, size=(w * h, 3)).astype(np.float32)
    img[3] = [np.nan, 0.0, 1.0]
    path = str(tmp_path / "x.ppm")
    assert L.pt_write_ppm(path.encode(), _np_f(img), w, h, 12, b"cornell", 3) == 0
    got = open(path, "rb").read()
    n = O.pto_format_ppm(_np_f(img), w, h, 12, b"cornell", 3, None, 0)
    buf = C.create_string_buffer(n)
    O.pto_format_ppm(_np_f(img), w, h, 12, b"cornell", 3, buf, n)
    assert got == buf.raw
    assert got.startswith(b"P3\n# samplesPerPixel: 12, resolution_y: 5, scene_id: cornell\n# rendering time: 3 s\n7 5\n255\n")
    assert got.endswith(b" ") and got.count(b"\n") == 5
    # first pixel written is the LAST framebuffer entry
    body = got.split(b"255\n", 1)[1].split()
    assert [int(v) for v in body[:3]] == [O.pto_to_int_with_gamma_correction(float(v)) for v in img[-1]]


def test_invalid_arguments_return_codes():
    assert L.pt_camera_basis(None, None, None, None) == -1
    assert L.pt_mesh_bounding_sphere(None, 0, None, None) == -1
    h = C.c_void_p()
    assert L.pt_scene_load(b"/nonexistent/x.json", b".", C.byref(h)) == -6
    assert L.pt_write_ppm(b"/nonexistent_dir/x.ppm", _np_f(np.zeros(3, np.float32)), 1, 1, 1, b"s", 0) == -6


def test_product_sincos_equals_oracle_on_whole_domain():
    """The kernels' branch-free sincos (csrc/pt_math.h, host instantiation) against the oracle's restatement of
    glibc's sinf/cosf on every reachable argument 2*PI*(k*2^-24): bit-identical (the oracle itself equals the
    platform libm there, tests/test_oracle.py)."""
    import subprocess, tempfile, textwrap
    src = textwrap.dedent("""
        #include <stdio.h>
        #include <string.h>
        #include <stdint.h>
        float pto_sinf(float); float pto_cosf(float); void pt_host_sincos(float, float*, float*);
        int main(void) { const float two_pi = 2.0f * 3.141592653589793f; long bad = 0;
          for (uint32_t k = 0; k < (1u << 24); k++) { float a = two_pi * ((float)k * (1.0f / 16777216.0f)), s, c;
            pt_host_sincos(a, &s, &c); float s0 = pto_sinf(a), c0 = pto_cosf(a);
            bad += memcmp(&s, &s0, 4) != 0; bad += memcmp(&c, &c0, 4) != 0; }
          printf("%ld ", bad); return bad != 0; }
    """)
    with tempfile.TemporaryDirectory() as td:
        cfile = os.path.join(td, "chk.c")
        open(cfile, "w").write(src)
        exe = os.path.join(td, "chk")
        subprocess.check_call(["gcc", "-O2", cfile, "-o", exe, ptlib.PRODUCT_SO, ptlib.ORACLE_SO,
                               "-Wl,-rpath," + os.path.dirname(ptlib.PRODUCT_SO),
                               "-Wl,-rpath," + os.path.dirname(ptlib.ORACLE_SO), "-lm"])
        out = subprocess.check_output([exe]).decode().strip()
    assert out == "0"


def test_siphash_pinned_and_image_hash():
    """SipHash-2-4 official test vector (key 00..0f, message 00..0e -> a129ca6149be45e5) pins the implementation;
    Image.hash is the same code with 1 and 3 rounds and a zero key (Rust DefaultHasher).  The oracle's separate
    SipHash-1-3 agrees."""
    key = bytes(range(16))
    k0, k1 = int.from_bytes(key[:8], "little"), int.from_bytes(key[8:], "little")
    assert L.pt_siphash(2, 4, k0, k1, bytes(range(15)), 15) == 0xa129ca6149be45e5
    assert L.pt_siphash(2, 4, k0, k1, b"", 0) == 0x726fdb47dd0e0e31
    rng = np.random.default_rng(2)
    for n in (0, 1, 2, 3, 7, 24, 1000):
        img = rng.uniform(0, 1, size=n).astype(np.float32)
        assert L.pt_image_hash(_np_f(img), n) == O.pto_image_hash(_np_f(img), n)
        assert L.pt_image_hash(_np_f(img), n) == L.pt_siphash(1, 3, 0, 0, img.tobytes(), 4 * n)


def test_mesh_bounding_box_matches_oracle_and_stored_boxes():
    """pt_mesh_bounding_box == the oracle's Mesh::new box (mod.rs:452-476, 501-536); pt_scene_bounding_box returns the
    boxes the reference serialised into cornell.json (inline meshes) and Mesh::new's for a MeshFile object."""
    for sid in ("cornell", "mesh"):
        sc = ptlib.load_scene_py(ptlib.scene_path(sid))
        want = ptlib.oracle_boxes(sc)
        h = C.c_void_p()
        assert L.pt_scene_load(ptlib.scene_path(sid).encode(), ptlib.ROOT.encode(), C.byref(h)) == 0, L.pt_last_error()
        for i in range(sc.n_objs):
            o = sc.objs[i]
            stored = L.pt_scene_bounding_box(h, i)
            if o.kind != ptlib.PT_MESH:
                assert not stored
                continue
            arr = (ptlib.PtTriangle * o.tri_count)(*[sc.tris[o.tri_offset + k] for k in range(o.tri_count)])
            got = (ptlib.PtTriangle * 12)()
            assert L.pt_mesh_bounding_box(arr, o.tri_count, got) == 0
            for k in range(12):
                for key in "abc":
                    assert list(getattr(got[k], key)) == list(getattr(want[12 * i + k], key)), (sid, i, k, key)
                    assert list(getattr(stored[k], key)) == list(getattr(want[12 * i + k], key)), (sid, i, k, key)
        L.pt_scene_free(h)


def test_comm_entry_points_fail_cleanly_without_a_gpu():
    """pt_comm_* (RCCL behind the C ABI): argument checks and the no-device answer; no collective is attempted here."""
    assert L.pt_comm_create(0, 0, 1, None, None) == -1
    if L.pt_device_count() > 0:
        pytest.skip("a GPU is present")
    comm = C.c_void_p()
    ident = C.create_string_buffer(128)
    assert L.pt_comm_create(0, 0, 1, ident, C.byref(comm)) == -2
    assert not comm.value
