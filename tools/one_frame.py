"""One or more frames through the C ABI without torch (for profilers): python tools/one_frame.py [scene] [spp] [frames] [backend]
PT_LIB selects another build of the library."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ptlib
from ptlib import PtConfig, PtStats

if os.environ.get("PT_LIB"):
    ptlib.PRODUCT_SO = os.environ["PT_LIB"]
L = ptlib.product()
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1
backend = int(sys.argv[4]) if len(sys.argv) > 4 else 0
W, H = 1024, 768
sc = ptlib.load_scene_py(ptlib.scene_path(scene))
ctx = C.c_void_p()
assert L.pt_ctx_create(0, C.byref(ctx)) == 0
assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0
rpp = int(os.environ.get("PT_ONE_FRAME_RPP", "0"))  # explicit pt_config.rays_per_pass (0: the library's own, timed passes)
cfg = PtConfig(W, H, spp, backend, 1, 0, 0, rpp, 0, 0, 0, 0, 0)
d = C.c_void_p()
assert L.pt_device_malloc(0, W * H * 12, C.byref(d)) == 0
st = PtStats()
for _ in range(frames):
    assert L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
    print("%s @%d spp: %d bounces, %.1f ms, %.2f G bounces/s" % (scene, spp, st.ray_bounces, st.ms_device,
                                                                st.ray_bounces / st.ms_device / 1e6), flush=True)
