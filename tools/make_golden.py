#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the oracle (the CPU restatement of the reference; the Rust reference cannot be
built in this image, so these vectors pin the restatement and everything checked against it, not the Rust binary).

    python tools/make_golden.py          # rewrites tests/golden/

Per scene: a small frame (48x32 @ 8 spp, seed 7) with its ray-bounce / sample counters, and the first rays the path
tracer casts for it with the intersection result of each (distance, object, triangle, hit point, normal).
Plus the RNG contract: Philox4x32-7 words, sin/cos on arguments of the 2*pi*k/2^24 lattice, gamma integers.
tests/test_oracle.py checks that the oracle still reproduces these bits; tests/test_gpu_parity.py checks the HIP
path against them (decisions bit-exact, images within 1e-4)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptlib  # noqa: E402
from ptlib import PtoConfig, _np_f  # noqa: E402

SCENES = ["single-sphere", "two-spheres", "three-spheres", "cartesian", "cornell", "mesh"]
W, H, SPP, SEED, N_RAYS = 48, 32, 8, 7, 4096


def scene_vectors(sid):
    O = ptlib.oracle()
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    spp = 2 if sid == "mesh" else SPP
    img, cnt, _ = ptlib.oracle_render(sc, W, H, spp, SEED, threads=1)
    ps = sc.pto()
    cfg = PtoConfig(W, H, spp, 0, SEED)
    cap = W * H * spp * 16
    rays = np.zeros((cap, 6), dtype=np.float32)
    n = int(O.pto_dump_rays(C.byref(ps), C.byref(cfg), 0, W * H, _np_f(rays), cap))
    assert 0 < n < cap
    step = max(1, n // N_RAYS)
    pick = rays[:n:step][:N_RAYS].copy()
    o, d = np.ascontiguousarray(pick[:, :3]), np.ascontiguousarray(pick[:, 3:])
    m = len(o)
    t, oid, tid = np.zeros(m, np.float32), np.zeros(m, np.int32), np.zeros(m, np.int32)
    x, nr = np.zeros((m, 3), np.float32), np.zeros((m, 3), np.float32)
    O.pto_intersect_batch(C.byref(ps), _np_f(o), _np_f(d), m, _np_f(t), oid.ctypes.data_as(ptlib.i32p),
                          tid.ctypes.data_as(ptlib.i32p), _np_f(x), _np_f(nr))
    return dict(width=W, height=H, spp=spp, seed=SEED, image=img, ray_bounces=np.uint64(cnt.ray_bounces),
                rays_total=np.uint64(n), ray_o=o, ray_d=d, hit_t=t, hit_object=oid, hit_triangle=tid, hit_x=x, hit_n=nr)


def numerics_vectors():
    O = ptlib.oracle()
    rng = np.random.default_rng(11)
    k = rng.integers(0, 1 << 24, size=4096, dtype=np.uint32)
    two_pi = np.float32(2.0) * np.float32(3.141592653589793)
    x = (two_pi * (k.astype(np.float32) * np.float32(1.0 / 16777216.0))).astype(np.float32)
    s = np.array([O.pto_sinf(float(v)) for v in x], np.float32)
    c = np.array([O.pto_cosf(float(v)) for v in x], np.float32)
    ctr = rng.integers(0, 1 << 32, size=(256, 4), dtype=np.uint64).astype(np.uint32)
    key = rng.integers(0, 1 << 32, size=(256, 2), dtype=np.uint64).astype(np.uint32)
    out = np.zeros((256, 4), np.uint32)
    for i in range(256):
        O.pto_philox4x32_7(ctr[i].ctypes.data_as(ptlib.u32p), key[i].ctypes.data_as(ptlib.u32p),
                            out[i].ctypes.data_as(ptlib.u32p))
    g = np.linspace(0.0, 1.0, 513, dtype=np.float32)
    gi = np.array([O.pto_to_int_with_gamma_correction(float(v)) for v in g], np.uint32)
    return dict(sincos_x=x, sin=s, cos=c, philox_ctr=ctr, philox_key=key, philox_out=out, gamma_x=g, gamma_int=gi)


# (scene, res_y, spp): whole frames as the reference writes them with MOCK_RANDOM = true (mod.rs:31-51), width = res_y*3/2
# as its GUI sets it (mod.rs:872-879) - rendered by the ORACLE (pto_render_mock + pto_format_ppm); the target a cargo holder
# diffs against and a drift guard until then (tests/test_oracle.py holds the list too)
MOCK_FRAMES = [("cornell", 32, 4), ("mesh", 16, 2), ("three-spheres", 16, 4)]


def mock_ppm(sid, res_y, spp):
    O = ptlib.oracle()
    width = res_y * 3 // 2
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    img, _, _ = ptlib.oracle_render_mock(sc, width, res_y, spp)
    n = O.pto_format_ppm(_np_f(img), width, res_y, spp, sid.encode(), 0, None, 0)
    buf = C.create_string_buffer(n)
    O.pto_format_ppm(_np_f(img), width, res_y, spp, sid.encode(), 0, buf, n)
    return buf.raw[:n]


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for sid, res_y, spp in MOCK_FRAMES:
        with open(os.path.join(out_dir, "mock_%s_%d_%d.ppm" % (sid, res_y, spp)), "wb") as f:
            f.write(mock_ppm(sid, res_y, spp))
    if len(sys.argv) > 1 and sys.argv[1] == "mock":  # only the MOCK_RANDOM frames (the .npz vectors stay as they are)
        return
    for sid in SCENES:
        np.savez_compressed(os.path.join(out_dir, "scene_%s.npz" % sid), **scene_vectors(sid))
    np.savez_compressed(os.path.join(out_dir, "numerics.npz"), **numerics_vectors())
    print("wrote", sorted(os.listdir(out_dir)))


if __name__ == "__main__":
    main()
