"""The compute side of strong scaling on ONE GPU, and the fixed cost of the gather (no multi-GPU node exists for this build):
  python tools/share_probe.py [spp]
1. rank 0's share of the 1024x768 frame (rows 0, N, 2N, ...) for N = 1, 2, 4, 8: wall time per frame and HIP-event time from
   the first to the last kernel; 2. the framebuffer gather through the C ABI's communicator at world size 1
   (pt_comm_gather_frame: copy into the staging buffer, ncclAllGather, un-permute kernel, synchronisation) for a frame of a
   rank's share at N = 8 (1024x96) and for the whole frame (what every rank receives): what a frame pays on top of its
   kernels, whatever the links then add for 8.3 MB."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ptlib
from ptlib import PtConfig, PtStats

L = ptlib.product()
sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
ctx = C.c_void_p()
assert L.pt_ctx_create(0, C.byref(ctx)) == 0
assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0
W, H = 1024, 768
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d, f = C.c_void_p(), C.c_void_p()
assert L.pt_device_malloc(0, W * H * 12, C.byref(d)) == 0 and L.pt_device_malloc(0, W * H * 12, C.byref(f)) == 0
whole = None
for step in (1, 2, 4, 8):
    cfg = PtConfig(W, H, spp, 0, 1, 0, 0, 0, 0, W if step > 1 else 0, 0, step if step > 1 else 0, 0)
    st = PtStats()
    best, best_dev = 1e9, 0.0
    for r in range(3):
        t0 = time.perf_counter()
        assert L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
        dt = time.perf_counter() - t0
        if dt < best:
            best, best_dev = dt, st.ms_device
    whole = whole or best
    print("share 1/%d: %.1f ms per frame (kernels %.1f ms, %d passes), %.2f G bounces/s, %.2fx the whole frame's rate per GPU-second" %
          (step, best * 1e3, best_dev, st.passes, st.ray_bounces / best / 1e9, whole / (best * step)), flush=True)
ident = C.create_string_buffer(128)
assert L.pt_comm_unique_id(ident) == 0, L.pt_last_error()
comm = C.c_void_p()
assert L.pt_comm_create(0, 0, 1, ident, C.byref(comm)) == 0, L.pt_last_error()
for h in (H // 8, H):
    cfg = PtConfig(W, h, 16, 0, 1, 0, 0, 0, 0)
    st = PtStats()
    assert L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st)) == 0
    cfg.chunk_pixels = W
    L.pt_comm_gather_frame(comm, C.byref(cfg), d, f, None)
    ts = []
    for r in range(20):
        t0 = time.perf_counter()
        assert L.pt_comm_gather_frame(comm, C.byref(cfg), d, f, None) == 0, L.pt_last_error()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("gather at world size 1, %dx%d (%.1f MB): median %.3f ms, best %.3f ms per frame" %
          (W, h, W * h * 12 / 1e6, ts[len(ts) // 2] * 1e3, ts[0] * 1e3), flush=True)
L.pt_comm_destroy(comm)
