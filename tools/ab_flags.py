"""A/B through the C ABI on one GPU: the accelerated scan (quad candidates, BVH) against PT_FLAG_NO_BVH (reference
order over every triangle) - identical image bits and bounce counts, and the time of each.
usage: python tools/ab_flags.py [scene ...]   (PT_LIB=/path/to/other/libptrace_hip.so to test another build)"""
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
W, H = 1024, 768


def render(scene, spp, flags, backend=0, reps=1, seed=1):
    sc = ptlib.load_scene_py(ptlib.scene_path(scene))
    ctx = C.c_void_p()
    assert L.pt_ctx_create(0, C.byref(ctx)) == 0
    assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0
    cfg = PtConfig(W, H, spp, backend, seed, 0, 0, 0, flags, 0, 0, 0, 0)
    d = C.c_void_p()
    assert L.pt_device_malloc(0, W * H * 12, C.byref(d)) == 0
    st = PtStats()
    best = 1e9
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        rc = L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st))
        assert rc == 0, L.pt_last_error()
        best = min(best, time.perf_counter() - t0) if rep or not reps else best
    img = np.empty(W * H * 3, np.float32)
    assert L.pt_device_download(0, img.ctypes.data_as(C.c_void_p), d, img.nbytes) == 0
    L.pt_device_free(0, d)
    L.pt_ctx_destroy(ctx)
    return img, st.ray_bounces, best


if __name__ == "__main__":
    for scene in sys.argv[1:] or ["cornell", "mesh"]:
        for backend in (0, 1):
            a, na, ta = render(scene, 64, 0, backend)
            b, nb, tb = render(scene, 64, 1, backend)
            print("%-8s backend %d @64spp: identical bits %s, bounces %d vs %d (%s)  %.1f ms vs %.1f ms" % (
                scene, backend, bool((a.view(np.uint32) == b.view(np.uint32)).all()), na, nb,
                "equal" if na == nb else "DIFFERENT", ta * 1e3, tb * 1e3), flush=True)
        _, n, t = render(scene, 1024, 0, 0)
        print("%-8s wavefront @1024spp accelerated: %.1f ms, %.2f G bounces/s" % (scene, t * 1e3, n / t / 1e9), flush=True)
        _, n, t = render(scene, 256, 1, 0)
        print("%-8s wavefront @256spp  NO_BVH:      %.1f ms, %.2f G bounces/s" % (scene, t * 1e3, n / t / 1e9), flush=True)
