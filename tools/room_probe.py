"""What the BVH mesh of a scene costs: the frame with and without its first object (mesh.json: the 810-triangle mesh),
time per ray bounce of each.  python tools/room_probe.py [scene] [spp]   (PT_LIB to test another build)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ptlib
from ptlib import PtConfig, PtStats

if os.environ.get("PT_LIB"):
    ptlib.PRODUCT_SO = os.environ["PT_LIB"]
L = ptlib.product()
scene = sys.argv[1] if len(sys.argv) > 1 else "mesh"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W, H = 1024, 768
full = ptlib.load_scene_py(ptlib.scene_path(scene))
room = ptlib.Scene("room", full.cam, list(full.objs)[1:full.n_objs], list(full.tris)[:full.n_tris])


def run(sc):
    ctx = C.c_void_p()
    assert L.pt_ctx_create(0, C.byref(ctx)) == 0
    assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0, L.pt_last_error()
    cfg = PtConfig(W, H, spp, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0)
    d = C.c_void_p()
    assert L.pt_device_malloc(0, W * H * 12, C.byref(d)) == 0
    st = PtStats()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        assert L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
        best = min(best, time.perf_counter() - t0)
    L.pt_device_free(0, d)
    L.pt_ctx_destroy(ctx)
    return best, st.ray_bounces


for name, sc in (("full", full), ("without object 0", room)):
    t, n = run(sc)
    print("%-18s %.1f ms  %d bounces  %.2f G bounces/s  %.1f ps per bounce" % (name, t * 1e3, n, n / t / 1e9, t / n * 1e12), flush=True)
