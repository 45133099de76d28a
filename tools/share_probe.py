import ctypes as C, sys, time, os
sys.path.insert(0,'tests'); import ptlib
from ptlib import PtConfig, PtStats
L=ptlib.product()
sc=ptlib.load_scene_py(ptlib.scene_path("cornell"))
ctx=C.c_void_p(); assert L.pt_ctx_create(0,C.byref(ctx))==0
assert L.pt_ctx_set_scene(ctx,C.byref(sc.cam),sc.objs,sc.n_objs,sc.tris,sc.n_tris)==0
W,H,spp=1024,768,4096
d=C.c_void_p(); assert L.pt_device_malloc(0,W*H*12,C.byref(d))==0
for step in (1,2,4,8):
    cfg=PtConfig(W,H,spp,0,1,0,0,0,0,W if step>1 else 0,0,step if step>1 else 0,0)
    st=PtStats()
    best=1e9
    for r in range(3):
        t0=time.perf_counter(); rc=L.pt_ctx_render(ctx,C.byref(cfg),d,None,None,None,None,C.byref(st)); assert rc==0
        best=min(best,time.perf_counter()-t0)
    print("share 1/%d: %.1f ms  (ideal %.1f)  bounces %d  %.2f G/s passes %d" % (step,best*1e3, 0, st.ray_bounces, st.ray_bounces/best/1e9, st.passes),flush=True)
