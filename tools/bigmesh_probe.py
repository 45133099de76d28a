"""A scene with one large mesh (a bumpy sphere of 2 x n x 2n triangles inside an emissive sphere, next to a glass ball):
frame time through the walk queue (k_pass_cand_bvh) and through k_pass_bvh (PT_CAND_BVH=0), images compared.
python tools/bigmesh_probe.py [n=256] [spp=64] [width=1024] [height=768] [rays_per_pass=0: the library's default]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import ptlib
from ptlib import PtConfig, PtStats

if os.environ.get("PT_LIB"):
    ptlib.PRODUCT_SO = os.environ["PT_LIB"]
L = ptlib.product()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
H = int(sys.argv[4]) if len(sys.argv) > 4 else 768
RPP = int(sys.argv[5]) if len(sys.argv) > 5 else 0
t = np.linspace(0, np.pi, steps + 1)[:, None]
p = np.linspace(0, 2 * np.pi, 2 * steps + 1)[None, :]
r = 1.0 + 0.04 * np.sin(9 * t) * np.cos(7 * p)
P = np.stack([r * np.sin(t) * np.cos(p), r * np.cos(t) + 0 * p, r * np.sin(t) * np.sin(p)], -1).astype(np.float32)
a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
tri = np.concatenate([np.stack([a, b, d], -2).reshape(-1, 3, 3), np.stack([b, c, d], -2).reshape(-1, 3, 3)])
tris = (ptlib.PtTriangle * len(tri))()
C.memmove(tris, np.ascontiguousarray(tri).ctypes.data, tri.nbytes)
cam = ptlib.make_camera((0, 0.3, 4.5), (0, -0.05, -1))
objs = [ptlib.make_sphere((0, 0, 0), 12.0, (0.7, 0.7, 0.7), (0.5, 0.5, 0.5), "Diffuse"),
        ptlib.make_sphere((1.6, -0.4, 0.8), 0.5, (0.9, 0.9, 0.9), (0, 0, 0), "Refract"),
        ptlib.make_mesh((-0.3, 0, 0), (0.8, 0.5, 0.3), (0, 0, 0), "Diffuse", 0, len(tri), (0, 0, 0), 1.05)]
sc = ptlib.Scene("big", cam, objs, [])
sc.tris, sc.n_tris = tris, len(tri)
res = []
for env in ({}, {"PT_CAND_BVH": "0"}):
    os.environ.pop("PT_CAND_BVH", None)
    os.environ.update(env)
    ctx = C.c_void_p()
    assert L.pt_ctx_create(0, C.byref(ctx)) == 0
    t0 = time.perf_counter()
    assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0, L.pt_last_error()
    t_scene = time.perf_counter() - t0
    cfg = PtConfig(W, H, spp, 0, 1, 0, 0, RPP, 0, 0, 0, 0, 0)
    dev = C.c_void_p()
    assert L.pt_device_malloc(0, W * H * 12, C.byref(dev)) == 0
    st = PtStats()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        assert L.pt_ctx_render(ctx, C.byref(cfg), dev, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
        best = min(best, time.perf_counter() - t0)
    img = np.empty(W * H * 3, np.float32)
    assert L.pt_device_download(0, img.ctypes.data_as(C.c_void_p), dev, img.nbytes) == 0
    print("%d triangles, %-16s scene set up in %.2f s; %.1f ms per frame, %d bounces, %.2f G bounces/s" % (
        len(tri), L.pt_ctx_pass_kernel(ctx, 0).decode(), t_scene, best * 1e3, st.ray_bounces, st.ray_bounces / best / 1e9), flush=True)
    res.append((img, st.ray_bounces))
    L.pt_device_free(0, dev)
    L.pt_ctx_destroy(ctx)
print("same frame, bit for bit:", bool(np.array_equal(res[0][0], res[1][0])) and res[0][1] == res[1][1])
