"""pt_comm_* (RCCL behind the C ABI) at world size 1 on GPU 0, in a process of its own: no torch here - a process that
holds torch's bundled HIP/HSA runtime next to /opt/rocm's gives RCCL an HSA instance that was never initialised.
Renders a small frame, gathers it through ncclAllGather + the un-permute kernel, compares.  Prints "gather ok"."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ptlib
from ptlib import PtConfig, PtStats

L = ptlib.product()
assert L.pt_device_count() >= 1, "needs a GPU"
sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
ctx = C.c_void_p()
assert L.pt_ctx_create(0, C.byref(ctx)) == 0, L.pt_last_error()
assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0, L.pt_last_error()
w, h, spp = 96, 50, 4
npix = w * h
ident = C.create_string_buffer(128)
assert L.pt_comm_unique_id(ident) == 0, L.pt_last_error()
comm = C.c_void_p()
assert L.pt_comm_create(0, 0, 1, ident, C.byref(comm)) == 0, L.pt_last_error()
d_local, d_frame = C.c_void_p(), C.c_void_p()
assert L.pt_device_malloc(0, npix * 12, C.byref(d_local)) == 0 and L.pt_device_malloc(0, npix * 12, C.byref(d_frame)) == 0
cfg = PtConfig(w, h, spp, 0, 9, 0, 0, 0, 0)
st = PtStats()
assert L.pt_ctx_render(ctx, C.byref(cfg), d_local, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
cfg.chunk_pixels = w
for _ in range(3):  # the staging buffer is reused across frames
    assert L.pt_comm_gather_frame(comm, C.byref(cfg), d_local, d_frame, None) == 0, L.pt_last_error()
a, b = np.zeros((npix, 3), np.float32), np.zeros((npix, 3), np.float32)
assert L.pt_device_download(0, a.ctypes.data_as(C.c_void_p), d_local, npix * 12) == 0
assert L.pt_device_download(0, b.ctypes.data_as(C.c_void_p), d_frame, npix * 12) == 0
assert a.any() and np.array_equal(a, b), "gathered frame differs from the rank's own"
# a band of the frame: only the band's pixels are gathered
cfg2 = PtConfig(w, h, spp, 0, 9, 7 * w, 31 * w, 0, 0)
assert L.pt_ctx_render(ctx, C.byref(cfg2), d_local, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
cfg2.chunk_pixels = w
assert L.pt_comm_gather_frame(comm, C.byref(cfg2), d_local, d_frame, None) == 0, L.pt_last_error()
band = np.zeros((24 * w, 3), np.float32)
assert L.pt_device_download(0, band.ctypes.data_as(C.c_void_p), d_frame, band.nbytes) == 0
assert np.array_equal(band, a[7 * w:31 * w])
L.pt_comm_destroy(comm)
L.pt_device_free(0, d_local)
L.pt_device_free(0, d_frame)
L.pt_ctx_destroy(ctx)
print("gather ok: %d pixels through ncclAllGather at world size 1" % npix)
