"""Walk counters of a -DPT_WALK_STATS build of the library (never the shipped one) on one GPU:
  hipcc <Makefile FLAGS> -DPT_WALK_STATS -x hip -shared -o scratch/libptrace_stats.so <SRC> -ldl
  PT_LIB=scratch/libptrace_stats.so python tools/walk_stats.py [scene] [spp]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ptlib
from ptlib import PtConfig, PtStats

ptlib.PRODUCT_SO = os.environ["PT_LIB"]
L = ptlib.product()
scene = sys.argv[1] if len(sys.argv) > 1 else "mesh"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H = 1024, 768
sc = ptlib.load_scene_py(ptlib.scene_path(scene))
ctx = C.c_void_p()
assert L.pt_ctx_create(0, C.byref(ctx)) == 0
assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0
cfg = PtConfig(W, H, spp, 0, 1, 0, 0, 512 << 20, 0, 0, 0, 0, 0)  # (the default pass size, given explicitly: no short timed passes, whose sessions are nearly empty)
d = C.c_void_p()
assert L.pt_device_malloc(0, W * H * 12, C.byref(d)) == 0
st = PtStats()
assert L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
out = (C.c_ulonglong * 16)()
L.pt_debug_walk_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert L.pt_debug_walk_stats(ctx, out) == 0
v = list(out)
names = ["walks", "box_batches", "items", "node_items", "popped_live", "leaf_batches", "leaf_tests", "rays", "parked",
         "scan_trips", "wave_walks", "overflows", "leaves_live", "walks_won", "refills", "popped"]
for n, x in zip(names, v):
    print("%-16s %d" % (n, x))
rays = float(v[7])
print("bounces %d (stats rays %d)" % (st.ray_bounces, v[7]))
print("parked / ray                %.3f" % (v[8] / rays))
print("walks / ray                 %.3f  (rays entering a mesh walk)" % (v[0] / rays))
print("node items / walk           %.2f" % (v[3] / max(v[0], 1)))
print("box batches / wave-walk     %.2f" % (v[1] / max(v[10], 1)))
print("items / box batch           %.2f of 64   (nodes %.2f)" % (v[2] / max(v[1], 1), v[3] / max(v[1], 1)))
print("leaves / walk               %.2f   per leaf batch %.1f" % (v[6] / max(v[0], 1), v[6] / max(v[5], 1)))
print("leaf batches / wave-walk    %.2f" % (v[5] / max(v[10], 1)))
print("leaves still worth testing    %.3f of those popped" % (v[12] / max(v[6], 1)))
print("walks that found the hit     %.3f of the walks" % (v[13] / max(v[0], 1)))
if v[14]:
    print("pool refills / wave-walk    %.2f   items popped per refill %.1f, still worth testing %.3f" % (v[14] / max(v[10], 1), v[15] / v[14], v[4] / max(v[15], 1)))
print("batches with dropped pushes %d (%.4f per wave-walk)" % (v[11], v[11] / max(v[10], 1)))
