"""Probe: do two resident contexts on ONE GPU, each rendering every second image row on its own stream, finish
a frame faster than one context rendering the whole frame?  (k_intersect is VALU-bound, k_shade HBM-bound.)"""
import os
if os.environ.get("PROBE_TORCH"):
    import torch  # noqa: F401  (loads torch's bundled HIP runtime first)
import ctypes as C
import sys
import threading
import time

sys.path.insert(0, "tests")
import ptlib
from ptlib import PtConfig, PtStats

L = ptlib.product()
sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
w, h, spp = 1024, 768, 512


def make_ctx():
    ctx = C.c_void_p()
    assert L.pt_ctx_create(0, C.byref(ctx)) == 0
    assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0
    return ctx


def run(n, reps=3):
    ctxs = [make_ctx() for _ in range(n)]
    cfgs = [PtConfig(w, h, spp, 0, 1, 0, 0, 0, 0, w if n > 1 else 0, r, n if n > 1 else 0, 0) for r in range(n)]
    outs = []
    for r in range(n):
        d = C.c_void_p()
        assert L.pt_device_malloc(0, L.pt_config_pixels(C.byref(cfgs[r])) * 12, C.byref(d)) == 0
        outs.append(d)
    stats = [PtStats() for _ in range(n)]

    def work(r):
        rc = L.pt_ctx_render(ctxs[r], C.byref(cfgs[r]), outs[r], None, None, None, None, C.byref(stats[r]))
        assert rc == 0, L.pt_last_error()

    best = 1e9
    for rep in range(reps + 1):
        th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        if rep > 0:
            best = min(best, dt)
    total = sum(s.ray_bounces for s in stats)
    print("%d context(s): best wall %.1f ms, %.2f G bounces/s" % (n, best * 1e3, total / best / 1e9))
    for r in range(n):
        L.pt_device_free(0, outs[r])
        L.pt_ctx_destroy(ctxs[r])


for n in (1, 2, 1, 2, 3, 4):
    run(n)


def run_flag(n, reps=3):
    ctx = make_ctx()
    cfg = PtConfig(w, h, spp, 0, 1, 0, 0, 0, (n << 8) if n > 1 else 0)
    d = C.c_void_p()
    assert L.pt_device_malloc(0, w * h * 12, C.byref(d)) == 0
    st = PtStats()
    best = 1e9
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        rc = L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st))
        assert rc == 0, L.pt_last_error()
        dt = time.perf_counter() - t0
        if rep > 0:
            best = min(best, dt)
    print("PT_FLAG_PIPELINES(%d): best wall %.1f ms, %.2f G bounces/s" % (n, best * 1e3, st.ray_bounces / best / 1e9))
    L.pt_device_free(0, d)
    L.pt_ctx_destroy(ctx)


for n in (1, 2, 3):
    run_flag(n)
