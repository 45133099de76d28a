"""Parity of the HIP path (through the C ABI) with the oracle on the same seeded inputs.  -m gpu.

Bars: geometry and control flow are bit-exact (hit distance/ids/points/normals, RNG words, bounce counts);
radiance is summed top-down in 32.32 fixed point on the GPU and bottom-up in f32 by the reference, so the
image is compared with the tolerance the north star states: 1e-4 per channel (observed ~1e-6)."""
import ctypes as C
import os

import numpy as np
import pytest

import ptlib
from ptlib import PtConfig, PtStats, PtoConfig, _np_f

pytestmark = pytest.mark.gpu

TOL = 1e-4
SCENES = ["single-sphere", "two-spheres", "three-spheres", "cartesian", "cornell", "mesh"]


@pytest.fixture(scope="module")
def gpu():
    L = ptlib.product()
    assert L.pt_device_count() >= 1, "no HIP device visible: the product has no CPU fallback"
    ctx = C.c_void_p()
    rc = L.pt_ctx_create(0, C.byref(ctx))
    assert rc == 0, L.pt_last_error()
    yield L, ctx
    L.pt_ctx_destroy(ctx)


def set_scene(gpu, sc):
    L, ctx = gpu
    rc = L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris)
    assert rc == 0, L.pt_last_error()


def gpu_render(gpu, sc, w, h, spp, seed, backend=0, band=None, rays_per_pass=0):
    """pt_render: host buffers in and out, the drop-in entry point."""
    L, _ = gpu
    cfg = PtConfig(w, h, spp, backend, seed, 0, 0, rays_per_pass, 0)
    if band:
        cfg.idx_begin, cfg.idx_end = band
    out = np.zeros((w * h, 3), dtype=np.float32)
    st = PtStats()
    rc = L.pt_render(C.byref(cfg), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None, None,
                     None, C.byref(st))
    assert rc == 0, L.pt_last_error()
    return out, st


def test_device_numerics_contract(gpu):
    """sin/cos/sqrt/reciprocal and Philox on the device are bit-identical to the oracle's host evaluation."""
    L, ctx = gpu
    O = ptlib.oracle()
    rng = np.random.default_rng(0)
    k = rng.integers(0, 1 << 24, size=200000, dtype=np.uint32)
    two_pi = np.float32(2.0) * np.float32(3.141592653589793)
    x = (two_pi * (k.astype(np.float32) * np.float32(1.0 / 16777216.0))).astype(np.float32)
    x[:8] = [0.0, 1e-5, 0.25, 0.7853981, 0.7853982, 3.1415927, 6.2831845, 1.5707964]
    n = len(x)
    s, c, q, r = (np.zeros(n, np.float32) for _ in range(4))
    ph = np.zeros(4 * n, np.uint32)
    rc = L.pt_ctx_numerics_probe(ctx, _np_f(x), n, _np_f(s), _np_f(c), _np_f(q), _np_f(r),
                                 ph.ctypes.data_as(ptlib.u32p))
    assert rc == 0, L.pt_last_error()
    want_s = np.array([O.pto_sinf(float(v)) for v in x[:20000]], np.float32)
    want_c = np.array([O.pto_cosf(float(v)) for v in x[:20000]], np.float32)
    assert np.array_equal(s[:20000].view(np.uint32), want_s.view(np.uint32))
    assert np.array_equal(c[:20000].view(np.uint32), want_c.view(np.uint32))
    # IEEE sqrt and division are correctly rounded on both sides: compare against numpy f32
    assert np.array_equal(q.view(np.uint32), np.sqrt(x).view(np.uint32))
    ok = x >= np.float32(2.0 ** -126)  # f_rcp's contract: normal, in-range divisors
    assert np.array_equal(r[ok].view(np.uint32), (np.float32(1.0) / x[ok]).view(np.uint32))
    # the device's short correctly-rounded sqrt / reciprocal sequences over the whole exponent range, including
    # the inputs that force the scaled path (0 < x < 2^-96), denormals, zeros, infinities, negatives and NaN
    m = 400000
    bits = rng.integers(0, 1 << 32, size=m, dtype=np.uint64).astype(np.uint32)
    y = bits.view(np.float32).copy()
    y[:12] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-30, 1.4e-45, 1e-38, 2.0 ** -96, np.float32(2.0 ** -96) * np.float32(0.9999999),
              1.0, 3.4e38]
    y[12:20000] = np.abs(y[12:20000])  # plenty of positive values (sqrt domain)
    s2, c2, q2, r2 = (np.zeros(m, np.float32) for _ in range(4))
    ph2 = np.zeros(4 * m, np.uint32)
    rc = L.pt_ctx_numerics_probe(ctx, _np_f(y), m, _np_f(s2), _np_f(c2), _np_f(q2), _np_f(r2),
                                 ph2.ctypes.data_as(ptlib.u32p))
    assert rc == 0, L.pt_last_error()
    with np.errstate(all="ignore"):
        want_q = np.sqrt(y)
        want_r = np.float32(1.0) / y
    nan_q = np.isnan(want_q)
    assert np.array_equal(np.isnan(q2), nan_q)
    assert np.array_equal(q2[~nan_q].view(np.uint32), want_q[~nan_q].view(np.uint32))
    ay = np.abs(y)
    dom = np.isfinite(y) & (ay >= np.float32(2.0 ** -126)) & (ay <= np.float32(2.0 ** 126))  # f_rcp's contract
    assert dom.sum() > 0.9 * m * 0.98
    assert np.array_equal(r2[dom].view(np.uint32), want_r[dom].view(np.uint32))
    out = (C.c_uint32 * 4)()
    for i in list(range(64)) + [n - 1]:
        ctr = (C.c_uint32 * 4)(i, int(x[i:i + 1].view(np.uint32)[0]), ((i << 8) | (i & 15)) & 0xffffffff, 0)
        O.pto_philox4x32_7(ctr, (C.c_uint32 * 2)(0x89abcdef, 0x01234567), out)
        assert list(out) == list(ph[4 * i:4 * i + 4])


def test_device_sincos_on_every_reachable_argument(gpu):
    """The diffuse bounce's cos / sin (mod.rs:703) on ALL 2^24 arguments r1 = 2 pi k 2^-24 it can see, on the device,
    against the host instantiation of the same source - which is held to the oracle's restatement of glibc's algorithm here
    (a sample) and, in tests/test_abi.py / test_oracle.py, to the platform libm on the same 2^24 arguments."""
    L, ctx = gpu
    out = (C.c_uint64 * 2)()
    rc = L.pt_ctx_sincos_sweep(ctx, out)
    assert rc == 0, L.pt_last_error()
    assert out[1] == 1 << 24 and out[0] == 0, list(out)
    O = ptlib.oracle()
    rng = np.random.default_rng(5)
    two_pi = np.float32(2.0) * np.float32(3.141592653589793)
    s, c = C.c_float(), C.c_float()
    for k in [0, 1, (1 << 24) - 1, 1 << 23, 1 << 22, 3 << 22] + [int(v) for v in rng.integers(0, 1 << 24, size=2000)]:
        x = float(two_pi * (np.float32(k) * np.float32(2.0 ** -24)))
        L.pt_host_sincos(x, C.byref(s), C.byref(c))
        assert (s.value, c.value) == (O.pto_sinf(x), O.pto_cosf(x)), k


def test_primary_rays_against_the_independent_restatement(gpu):
    """render_pixel's sensor mapping on the DEVICE (pt_ctx_primary_rays: the functions the frame kernels call, both forms)
    == tests/kats_camera.py's numpy-f32 reading of mod.rs:805-843 == the oracle's pto_primary_ray, bit for bit: corners,
    centre, the four sub-pixels, r on both sides of 1.0 and exactly 1.0, non-square pixels, both `up` vectors."""
    import kats_camera as K

    L, ctx = gpu
    O = ptlib.oracle()
    groups = {}
    for case in K.CASES:
        cam, w, h, pix, smp, seed = case
        groups.setdefault((id(cam), w, h, seed), []).append(case)
    n_checked = 0
    for cases in groups.values():
        cam, w, h, _, _, seed = cases[0]
        pc = ptlib.make_camera(cam["position"], cam["direction"], cam["focal_length"], cam["sensor_width"], cam["aspect_ratio"])
        sc = ptlib.Scene("cam", pc, [ptlib.make_sphere((0, 0, -3), 1.0, (1, 1, 1), (0, 0, 0), "Diffuse")], [])
        set_scene(gpu, sc)
        pix = np.array([c[3] for c in cases], np.uint32)
        smp = np.array([c[4] for c in cases], np.uint32)
        n = len(cases)
        for form in (0, 1):
            o, d = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
            rc = L.pt_ctx_primary_rays(ctx, w, h, seed, pix.ctypes.data_as(ptlib.u32p), smp.ctypes.data_as(ptlib.u32p), n, form,
                                       _np_f(o), _np_f(d))
            assert rc == 0, L.pt_last_error()
            for i, case in enumerate(cases):
                want_o, want_d = K.expected(case)
                assert np.array_equal(o[i].view(np.uint32), want_o.view(np.uint32)), (form, case, o[i], want_o)
                assert np.array_equal(d[i].view(np.uint32), want_d.view(np.uint32)), (form, case, d[i], want_d)
                n_checked += 1
    assert n_checked == 2 * len(K.CASES)
    # and in bulk against the oracle: every sample of a block of pixels of the bench frame, both forms
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    set_scene(gpu, sc)
    w, h, seed = 1024, 768, 1
    rng = np.random.default_rng(3)
    pix = rng.integers(0, w * h, size=4096, dtype=np.uint64).astype(np.uint32)
    smp = rng.integers(0, 4096, size=4096, dtype=np.uint64).astype(np.uint32)
    want_o, want_d = np.zeros((4096, 3), np.float32), np.zeros((4096, 3), np.float32)
    oo, dd = (C.c_float * 3)(), (C.c_float * 3)()
    for i in range(4096):
        O.pto_primary_ray(C.byref(sc.cam), w, h, int(pix[i]), int(smp[i]), seed, oo, dd)
        want_o[i], want_d[i] = list(oo), list(dd)
    for form in (0, 1):
        o, d = np.zeros((4096, 3), np.float32), np.zeros((4096, 3), np.float32)
        rc = L.pt_ctx_primary_rays(ctx, w, h, seed, pix.ctypes.data_as(ptlib.u32p), smp.ctypes.data_as(ptlib.u32p), 4096, form,
                                   _np_f(o), _np_f(d))
        assert rc == 0, L.pt_last_error()
        assert np.array_equal(o.view(np.uint32), want_o.view(np.uint32)) and np.array_equal(d.view(np.uint32), want_d.view(np.uint32))
    # bad arguments are errors, not faults
    assert L.pt_ctx_primary_rays(ctx, w, h, seed, pix.ctypes.data_as(ptlib.u32p), smp.ctypes.data_as(ptlib.u32p), 4096, 2, _np_f(o), _np_f(d)) == -1
    bad = np.array([w * h], np.uint32)
    assert L.pt_ctx_primary_rays(ctx, w, h, seed, bad.ctypes.data_as(ptlib.u32p), smp.ctypes.data_as(ptlib.u32p), 1, 0, _np_f(o), _np_f(d)) == -1


@pytest.mark.parametrize("sid", ["cornell", "mesh", "three-spheres"])
def test_intersect_ray_by_ray(gpu, sid):
    """Every ray the path tracer casts for a block of pixels: hit distance, object, triangle, hit point and
    normal from the device intersection code are bit-identical to intersect_scene of the oracle."""
    L, ctx = gpu
    O = ptlib.oracle()
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    set_scene(gpu, sc)
    w, h, spp = 64, 48, 2 if sid == "mesh" else 8
    cfg = PtoConfig(w, h, spp, 0, 11)
    cap = w * h * spp * 16
    rays = np.zeros((cap, 6), dtype=np.float32)
    ps = sc.pto()
    n = O.pto_dump_rays(C.byref(ps), C.byref(cfg), 0, w * h, _np_f(rays), cap)
    assert 0 < n < cap
    o = np.ascontiguousarray(rays[:n, :3])
    d = np.ascontiguousarray(rays[:n, 3:])

    def run(fn, handle):
        t = np.zeros(n, np.float32)
        oid = np.zeros(n, np.int32)
        tid = np.zeros(n, np.int32)
        x = np.zeros((n, 3), np.float32)
        nr = np.zeros((n, 3), np.float32)
        rc = fn(handle, _np_f(o), _np_f(d), n, _np_f(t), oid.ctypes.data_as(ptlib.i32p),
                tid.ctypes.data_as(ptlib.i32p), _np_f(x), _np_f(nr))
        return rc, t, oid, tid, x, nr

    rc, t, oid, tid, x, nr = run(L.pt_ctx_intersect, ctx)
    assert rc == 0, L.pt_last_error()
    _, t0, oid0, tid0, x0, nr0 = run(O.pto_intersect_batch, C.byref(ps))
    assert np.array_equal(oid, oid0)
    assert np.array_equal(tid, tid0)
    assert np.array_equal(t.view(np.uint32), t0.view(np.uint32))
    assert np.array_equal(x.view(np.uint32), x0.view(np.uint32))
    assert np.array_equal(nr.view(np.uint32), nr0.view(np.uint32))
    assert (oid0 < 0).sum() > 0 or sid != "cornell"  # the closed box still leaks rays


@pytest.mark.parametrize("backend", [ptlib.BACKEND_WAVEFRONT, ptlib.BACKEND_MEGAKERNEL])
@pytest.mark.parametrize("sid", SCENES)
def test_frame_parity(gpu, sid, backend):
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    w, h, spp = (48, 32, 4) if sid == "mesh" else (96, 64, 16)
    want, cnt, _ = ptlib.oracle_render(sc, w, h, spp, 3)
    got, st = gpu_render(gpu, sc, w, h, spp, 3, backend)
    assert st.ray_bounces == cnt.ray_bounces  # same number of intersect_scene calls: same paths
    assert st.samples == w * h * spp
    assert float(np.abs(got - want).max()) <= TOL


def test_many_passes_and_odd_sizes(gpu):
    """Passes of 1 spp, a width that is not a multiple of the wave, spp that is not a multiple of the pass."""
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    w, h, spp = 67, 41, 7
    want, cnt, _ = ptlib.oracle_render(sc, w, h, spp, 9)
    got, st = gpu_render(gpu, sc, w, h, spp, 9, 0, rays_per_pass=2 * w * h)
    assert st.passes == 4 and st.ray_bounces == cnt.ray_bounces
    assert float(np.abs(got - want).max()) <= TOL
    got1, st1 = gpu_render(gpu, sc, w, h, spp, 9, 0, rays_per_pass=1)
    assert st1.passes == 7
    assert np.array_equal(got1, got)  # the image does not depend on the pass size (integer accumulation)


@pytest.mark.gpu
@pytest.mark.parametrize("sid", ["cornell", "mesh"])
def test_primary_ray_order_of_the_pass_kernel(gpu, sid):
    """k_pass_cand starts its primary rays SAMPLE-MAJOR - a trip's 64 primaries are consecutive samples of one pixel - in chunks of 64
    that are dealt to the workgroup's four waves in turn (csrc/pt_kernels.hip).  Which lane traces which (pixel, sample) must not
    show anywhere: the frame is the megakernel's (its own, unrelated order) bit for bit and the bounce count the same, for passes
    whose samples per pixel sit on every edge of that arithmetic - 1, just under / at / just over a chunk of 64, just under / at /
    just over the four waves' 256, a prime - for frames of 1, 3, 22 and 23 pixels per stream, and for a stream count that leaves the
    last streams a pixel short."""
    L, ctx = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    set_scene(gpu, sc)
    cases = [(16, 8, 1, 0), (16, 8, 63, 0), (16, 8, 64, 0), (16, 8, 65, 0), (9, 7, 255, 0), (9, 7, 256, 0), (9, 7, 257, 0),
             (9, 7, 683, 0), (31, 3, 130, 0), (64, 33, 67, 0), (40, 30, 97, 1), (40, 30, 300, 5)]
    for (w, h, spp, per_pass) in cases:
        npix = w * h
        d_out = C.c_void_p()
        assert L.pt_device_malloc(0, npix * 12, C.byref(d_out)) == 0
        frames, bounces = [], []
        for backend in (0, 1):
            rpp = npix * per_pass if (per_pass and backend == 0) else 0  # (several passes: the last one shorter)
            cfg = PtConfig(w, h, spp, backend, 77, 0, 0, rpp, 0)
            st = PtStats()
            assert L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
            host = np.zeros((npix, 3), dtype=np.float32)
            assert L.pt_device_download(0, host.ctypes.data_as(C.c_void_p), d_out, npix * 12) == 0
            frames.append(host)
            bounces.append(st.ray_bounces)
            assert st.samples == npix * spp
        L.pt_device_free(0, d_out)
        assert bounces[0] == bounces[1], (w, h, spp, bounces)
        assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)), (w, h, spp)


def test_bands_equal_whole_frame(gpu):
    """Rendering the frame as N bands (what N ranks do) gives bit-identical pixels to one call."""
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    w, h, spp = 80, 60, 8
    whole, st = gpu_render(gpu, sc, w, h, spp, 4)
    for backend in (0, 1):
        for n in (2, 3, 8):
            img = np.zeros_like(whole)
            total = 0
            for r in range(n):
                b, e = (w * h * r) // n, (w * h * (r + 1)) // n
                part, s = gpu_render(gpu, sc, w, h, spp, 4, backend, band=(b, e))
                img[b:e] = part[b:e]
                assert not part[:b].any() and not part[e:].any()
                total += s.ray_bounces
            assert np.array_equal(img, whole)
            assert total == st.ray_bounces


def test_single_sphere_analytic(gpu):
    """single-sphere.json: pixels inside the silhouette are exactly 1.0 (emission clamps), background 0.0."""
    sc = ptlib.load_scene_py(ptlib.scene_path("single-sphere"))
    got, st = gpu_render(gpu, sc, 256, 256, 64, 1)
    vals = np.unique(got)
    assert vals.min() == 0.0 and vals.max() == 1.0
    frac = ((got > 0) & (got < 1)).any(axis=1).mean()
    assert frac < 0.05  # only silhouette pixels are fractional
    assert (got == 1.0).all(axis=1).mean() > 0.05


def test_errors_and_cancel(gpu):
    L, ctx = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    out = np.zeros((16 * 16, 3), np.float32)
    st = PtStats()
    bad = PtConfig(0, 16, 4, 0, 1, 0, 0, 0, 0)
    assert L.pt_render(C.byref(bad), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None, None,
                       None, C.byref(st)) == -1
    bad = PtConfig(16, 16, 4, 7, 1, 0, 0, 0, 0)
    assert L.pt_render(C.byref(bad), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None, None,
                       None, C.byref(st)) == -1
    bad = PtConfig(16, 16, 4, 0, 1, 10, 300, 0, 0)
    assert L.pt_render(C.byref(bad), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None, None,
                       None, C.byref(st)) == -1
    # cancel flag already set: PT_CANCELLED, framebuffer all zero (the reference leaves unrendered pixels 0)
    flag = (C.c_uint8 * 1)(1)
    cfg = PtConfig(16, 16, 4, 0, 1, 0, 0, 0, 0)
    rc = L.pt_render(C.byref(cfg), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out),
                     C.cast(flag, C.c_void_p), None, None, C.byref(st))
    assert rc == -4 and not out.any()
    # progress callback reaches 1.0
    seen = []
    cb = ptlib.PROGRESS_FN(lambda user, frac: seen.append(frac))
    rc = L.pt_render(C.byref(cfg), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None,
                     C.cast(cb, C.c_void_p), None, C.byref(st))
    assert rc == 0 and seen and seen[-1] == 1.0


def test_full_size_properties(gpu):
    """BASELINE size (cornell 1024x768), reduced spp: determinism, backend agreement, bounce statistics and
    frame mean against the survey's strict-f32 probe (sanity, MC noise ~0.002)."""
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    w, h, spp = 1024, 768, 32
    a, sa = gpu_render(gpu, sc, w, h, spp, 1, 0)
    b, sb = gpu_render(gpu, sc, w, h, spp, 1, 0)
    m, sm = gpu_render(gpu, sc, w, h, spp, 1, 1)
    assert np.array_equal(a, b) and sa.ray_bounces == sb.ray_bounces
    assert np.array_equal(a, m) and sa.ray_bounces == sm.ray_bounces
    per_sample = sa.ray_bounces / (w * h * spp)
    assert 8.5 < per_sample < 8.85, per_sample  # 8.68 in the survey probe
    # a sample of pixels against the oracle at full resolution
    O = ptlib.oracle()
    ps = sc.pto()
    cfg = PtoConfig(w, h, spp, 0, 1)
    rng = np.random.default_rng(5)
    px = np.zeros(3, np.float32)
    worst = 0.0
    for idx in rng.integers(0, w * h, size=300):
        O.pto_render_pixel(C.byref(ps), C.byref(cfg), int(idx), _np_f(px), None)
        worst = max(worst, float(np.abs(px - a[idx]).max()))
    assert worst <= TOL, worst


def test_bvh_equals_reference_scan(gpu):
    """mesh.json (810-triangle mctri.off): the LDS-staged BVH returns exactly what the reference's linear scan
    returns.  (a) frame with BVH == frame with PT_FLAG_NO_BVH, same bounce count; (b) 400k random rays with
    origins all over the scene volume, incl. rays aimed at mesh vertices/edges and grazing rays, against the oracle."""
    L, ctx = gpu
    O = ptlib.oracle()
    sc = ptlib.load_scene_py(ptlib.scene_path("mesh"))
    w, h, spp = 160, 120, 8

    def render(flags, backend):
        cfg = PtConfig(w, h, spp, backend, 2, 0, 0, 0, flags)
        out = np.zeros((w * h, 3), dtype=np.float32)
        st = PtStats()
        rc = L.pt_render(C.byref(cfg), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None,
                         None, None, C.byref(st))
        assert rc == 0, L.pt_last_error()
        return out, st.ray_bounces

    a, na = render(0, 0)
    b, nb = render(ptlib.FLAG_NO_BVH, 0)
    m, nm = render(0, 1)
    assert na == nb == nm and np.array_equal(a, b) and np.array_equal(a, m)

    set_scene(gpu, sc)
    rng = np.random.default_rng(3)
    n = 400000
    o = rng.uniform([-2.5, -1.9, -8.0], [2.5, 1.9, 8.5], size=(n, 3)).astype(np.float32)
    # targets: mesh vertices (world space) jittered by 0 .. 1e-3, so many rays hit edges/vertices or graze
    ob = sc.objs[0]
    verts = np.array([[list(sc.tris[k].a), list(sc.tris[k].b), list(sc.tris[k].c)] for k in range(ob.tri_count)],
                     dtype=np.float32).reshape(-1, 3) + np.array(list(ob.position), dtype=np.float32)
    tgt = verts[rng.integers(0, len(verts), size=n)] + (rng.normal(size=(n, 3)) *
                                                         rng.choice([0.0, 1e-6, 1e-4, 1e-2, 0.3], size=(n, 1))).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    # origins sitting on the mesh surface itself (self-hit cases, t == 0 rejects)
    o[: n // 8] = tgt[: n // 8]
    d[: n // 8] = rng.normal(size=(n // 8, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    # axis-parallel and plane-parallel rays: direction components that are exactly +-0 (the path tracer makes
    # them: a diffuse draw r2 == 0 off an axis-aligned wall, mirror bounces)
    k = n // 8
    for j, zero_axes in enumerate([(0,), (1,), (2,), (0, 1), (0, 2), (1, 2)]):
        sl = slice(k + j * (k // 6), k + (j + 1) * (k // 6))
        dd = d[sl].copy()
        for ax in zero_axes:
            dd[:, ax] = np.where(rng.random(len(dd)) < 0.5, 0.0, -0.0)
        dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        d[sl] = dd
        o[sl] = (verts[rng.integers(0, len(verts), size=len(dd))] - dd * rng.uniform(0.5, 3.0, size=(len(dd), 1))
                 + rng.normal(size=(len(dd), 3)) * 0.05).astype(np.float32)
    d = np.ascontiguousarray(d.astype(np.float32))
    o = np.ascontiguousarray(o)

    def run(fn, handle):
        t = np.zeros(n, np.float32)
        oid = np.zeros(n, np.int32)
        tid = np.zeros(n, np.int32)
        rc = fn(handle, _np_f(o), _np_f(d), n, _np_f(t), oid.ctypes.data_as(ptlib.i32p),
                tid.ctypes.data_as(ptlib.i32p), None, None)
        return rc, t, oid, tid

    rc, t, oid, tid = run(L.pt_ctx_intersect, ctx)
    assert rc == 0, L.pt_last_error()
    ps = sc.pto()
    _, t0, oid0, tid0 = run(O.pto_intersect_batch, C.byref(ps))
    assert (oid0 == 0).mean() > 0.3  # the mesh is really being hit
    assert np.array_equal(oid, oid0) and np.array_equal(tid, tid0)
    assert np.array_equal(t.view(np.uint32), t0.view(np.uint32))


def test_full_size_mesh_bvh_vs_scan(gpu):
    """BASELINE config 4 size (mesh.json 1024x768), reduced spp: the BVH frame equals the reference-style
    full triangle scan bit for bit, with the same number of intersect_scene evaluations (~200 M rays)."""
    L, _ = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("mesh"))
    w, h, spp = 1024, 768, 32
    res = []
    for flags in (0, ptlib.FLAG_NO_BVH):
        cfg = PtConfig(w, h, spp, 0, 1, 0, 0, 0, flags)
        out = np.zeros((w * h, 3), dtype=np.float32)
        st = PtStats()
        rc = L.pt_render(C.byref(cfg), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None,
                         None, None, C.byref(st))
        assert rc == 0, L.pt_last_error()
        res.append((out, st.ray_bounces))
    assert res[0][1] == res[1][1]
    assert np.array_equal(res[0][0], res[1][0])
    assert 7.0 < res[0][1] / (w * h * spp) < 8.2  # 7.4-7.7 bounces per sample (survey probe / oracle)


def test_mesh_walk_forms_agree(gpu):
    """mesh.json through the three forms of the BVH walk - k_pass_cand_bvh (candidate scan, walks as a per-wave queue of
    box tests), the same with a queue so small that most rays take the depth-first second walk, and k_pass_bvh (scan +
    depth-first parked walks) - and through the reference-style full scan: the same frame, bit for bit, and the same
    number of intersect_scene evaluations."""
    L, _ = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("mesh"))
    w, h, spp = 320, 240, 16
    res = {}
    for name, env, flags in (("queue", {}, 0), ("tiny queue", {"PT_WALK_QUEUE_CAP": "128"}, 0),
                             ("k_pass_bvh", {"PT_CAND_BVH": "0"}, 0), ("full scan", {}, ptlib.FLAG_NO_BVH)):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = C.c_void_p()
            assert L.pt_ctx_create(0, C.byref(ctx)) == 0  # the tuning variables are read here
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0, L.pt_last_error()
        kernel = L.pt_ctx_pass_kernel(ctx, flags).decode()
        cfg = PtConfig(w, h, spp, 0, 3, 0, 0, 0, flags)
        d = C.c_void_p()
        assert L.pt_device_malloc(0, w * h * 12, C.byref(d)) == 0
        st = PtStats()
        assert L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
        img = np.empty((w * h, 3), np.float32)
        assert L.pt_device_download(0, _np_f(img), d, img.nbytes) == 0
        L.pt_device_free(0, d)
        L.pt_ctx_destroy(ctx)
        res[name] = (kernel, img, st.ray_bounces)
    assert res["queue"][0] == "k_pass_cand_bvh" and res["tiny queue"][0] == "k_pass_cand_bvh"
    assert res["k_pass_bvh"][0] == "k_pass_bvh" and res["full scan"][0] == "k_pass"
    for name in ("tiny queue", "k_pass_bvh", "full scan"):
        assert res[name][2] == res["queue"][2], name
        assert np.array_equal(res[name][1], res["queue"][1]), name


def test_wide_bvh_through_the_walk_queue(gpu):
    """A mesh of 36 480 triangles: its BVH has more leaves than 16-bit child references can number, so the depth-first
    walkers keep u32 stacks.  The walk queue (k_pass_cand_bvh), the same with a queue so small that the depth-first
    second walk (u32 stacks in the queue's LDS) does most of the work, and k_pass_bvh give the same frame, bit for bit;
    single rays agree with the oracle's linear scan."""
    L, _ = gpu
    O = ptlib.oracle()
    steps = 96
    tris = []
    for i in range(steps):
        t1, t2 = np.pi * i / steps, np.pi * (i + 1) / steps
        for j in range(2 * steps):
            p1, p2 = np.pi * j / steps, np.pi * (j + 1) / steps
            r = lambda t, p: 1.0 + 0.04 * np.sin(9 * t) * np.cos(7 * p)
            P = lambda t, p: (r(t, p) * np.sin(t) * np.cos(p), r(t, p) * np.cos(t), r(t, p) * np.sin(t) * np.sin(p))
            a, b, c, d = P(t1, p1), P(t2, p1), P(t2, p2), P(t1, p2)
            if i == 0:
                tris.append((a, c, d))
            elif i + 1 == steps:
                tris.append((a, b, c))
            else:
                tris.append((a, b, d))
                tris.append((b, c, d))
    tris = [ptlib.make_tri(*t) for t in tris]
    assert len(tris) == 36480
    cam = ptlib.make_camera((0, 0.3, 4.5), (0, -0.05, -1))
    objs = [ptlib.make_sphere((0, 0, 0), 12.0, (0.7, 0.7, 0.7), (0.5, 0.5, 0.5), "Diffuse"),
            ptlib.make_sphere((1.6, -0.4, 0.8), 0.5, (0.9, 0.9, 0.9), (0, 0, 0), "Refract"),
            ptlib.make_mesh((-0.3, 0, 0), (0.8, 0.5, 0.3), (0, 0, 0), "Diffuse", 0, len(tris), (0, 0, 0), 1.05)]
    sc = ptlib.Scene("wide", cam, objs, tris)
    w, h, spp = 160, 120, 8
    res = {}
    for name, env in (("queue", {}), ("tiny queue", {"PT_WALK_QUEUE_CAP": "128"}), ("k_pass_bvh", {"PT_CAND_BVH": "0"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = C.c_void_p()
            assert L.pt_ctx_create(0, C.byref(ctx)) == 0
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0, L.pt_last_error()
        kernel = L.pt_ctx_pass_kernel(ctx, 0).decode()
        cfg = PtConfig(w, h, spp, 0, 5, 0, 0, 0, 0)
        d = C.c_void_p()
        assert L.pt_device_malloc(0, w * h * 12, C.byref(d)) == 0
        st = PtStats()
        assert L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
        img = np.empty((w * h, 3), np.float32)
        assert L.pt_device_download(0, _np_f(img), d, img.nbytes) == 0
        L.pt_device_free(0, d)
        if name == "queue":  # single rays against the oracle's scan (k_query: depth-first, u32 stacks)
            rng = np.random.default_rng(11)
            n = 4000
            o = rng.uniform(-2.5, 2.5, size=(n, 3)).astype(np.float32)
            dd = rng.normal(size=(n, 3)).astype(np.float32) * 0.7 - o
            dd /= np.linalg.norm(dd, axis=1, keepdims=True)
            o, dd = np.ascontiguousarray(o), np.ascontiguousarray(dd.astype(np.float32))
            out = []
            ps = sc.pto()
            for fn, handle in ((L.pt_ctx_intersect, ctx), (O.pto_intersect_batch, C.byref(ps))):
                t, oid, tid = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.int32)
                assert fn(handle, _np_f(o), _np_f(dd), n, _np_f(t), oid.ctypes.data_as(ptlib.i32p),
                          tid.ctypes.data_as(ptlib.i32p), None, None) == 0
                out.append((t, oid, tid))
            assert (out[1][1] == 2).mean() > 0.2
            assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
            assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
        L.pt_ctx_destroy(ctx)
        res[name] = (kernel, img, st.ray_bounces)
    assert res["queue"][0] == "k_pass_cand_bvh" and res["k_pass_bvh"][0] == "k_pass_bvh"
    assert res["queue"][1].max() > 0.0
    for name in ("tiny queue", "k_pass_bvh"):
        assert res[name][2] == res["queue"][2], name
        assert np.array_equal(res[name][1], res["queue"][1]), name


def test_render_multi_and_snapshot(gpu):
    """pt_render_multi with 1, 3 and 5 bands (all on this box's one GPU) == pt_render, bit for bit; the progress
    callback can pull partial frames with pt_ctx_snapshot and the last one (all spp) equals the final image."""
    L, ctx = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    w, h, spp = 64, 40, 12
    whole, st = gpu_render(gpu, sc, w, h, spp, 8)
    for n in (1, 3, 5):
        cfg = PtConfig(w, h, spp, 0, 8, 0, 0, 0, 0)
        out = np.zeros((w * h, 3), dtype=np.float32)
        s2 = PtStats()
        rc = L.pt_render_multi(C.byref(cfg), n, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out),
                               None, None, None, C.byref(s2))
        assert rc == 0, L.pt_last_error()
        assert np.array_equal(out, whole) and s2.ray_bounces == st.ray_bounces
    # snapshots: 12 passes of 1 spp; the callback fires between passes
    set_scene(gpu, sc)
    nbytes = w * h * 3 * 4
    d_out, d_snap = C.c_void_p(), C.c_void_p()
    assert L.pt_device_malloc(0, nbytes, C.byref(d_out)) == 0 and L.pt_device_malloc(0, nbytes, C.byref(d_snap)) == 0

    def download(ptr):
        host = np.zeros((w * h, 3), dtype=np.float32)
        assert L.pt_device_download(0, host.ctypes.data_as(C.c_void_p), ptr, nbytes) == 0
        return host

    seen = []

    def on_progress(user, frac):
        n = C.c_uint32()
        if L.pt_ctx_snapshot(ctx, d_snap, C.byref(n)) == 0:
            seen.append((frac, n.value, download(d_snap)))

    cb = ptlib.PROGRESS_FN(on_progress)
    cfg = PtConfig(w, h, spp, 0, 8, 0, 0, 1, 0)
    cfg.progress_ms = ptlib.PROGRESS_EVERY_PASS  # the default cadence is the reference's 500 ms: this frame takes a few ms
    s3 = PtStats()
    rc = L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, None, C.cast(cb, C.c_void_p), None, C.byref(s3))
    assert rc == 0, L.pt_last_error()
    assert np.array_equal(download(d_out), whole)
    assert len(seen) >= 3 and all(1 <= n <= spp for _, n, _ in seen)
    # a snapshot over k samples is the k-spp frame of the same seed (sample indices 0..k-1)
    frac, k, img = seen[1]
    ref, _ = gpu_render(gpu, sc, w, h, k, 8)
    assert np.array_equal(img, ref)
    # ... and, against the oracle: every snapshot is the oracle's frame over that many samples per pixel
    for _, k2, img2 in (seen[0], seen[1], seen[-1]):
        want, _, _ = ptlib.oracle_render(sc, w, h, k2, 8)
        assert np.abs(img2 - want).max() <= TOL, k2
    assert L.pt_device_free(0, d_out) == 0 and L.pt_device_free(0, d_snap) == 0


def test_edge_scenes(gpu):
    """Edge cases the domain has: an empty scene (every ray misses), a 1x1 frame, a scene made only of meshes,
    degenerate (zero-area) and tiny triangles (|det| < 1e-4 rejects them, mod.rs:571), a black object
    (max_reflection = 0: roulette always stops, 1/0 = inf never used), a sphere the camera sits inside."""
    cam = ptlib.make_camera((0, 0, 5), (0, 0, -1))
    cases = {
        "empty": ptlib.Scene("e", cam, [], []),
        "only_meshes": ptlib.Scene("m", cam, [
            ptlib.make_mesh((0, 0, 0), (0.8, 0.8, 0.8), (0.5, 0.5, 0.5), "Diffuse", 0, 3, (0, 0, 0), 10.0)],
            [ptlib.make_tri((-1, -1, 0), (1, -1, 0), (0, 1, 0)),
             ptlib.make_tri((0, 0, 1), (0, 0, 1), (0, 0, 1)),                 # zero area
             ptlib.make_tri((0, 0, 2), (1e-3, 0, 2), (0, 1e-3, 2))]),         # tiny: det below 1e-4
        "black_and_inside": ptlib.Scene("b", cam, [
            ptlib.make_sphere((0, 0, 5), 30.0, (0.0, 0.0, 0.0), (0.3, 0.2, 0.1), "Diffuse"),   # camera inside, black
            ptlib.make_sphere((0, 0, 0), 1.0, (0.9, 0.9, 0.9), (0, 0, 0), "Refract"),
            ptlib.make_sphere((2, 0, 0), 1.0, (0.9, 0.9, 0.9), (0, 0, 0), "Specular")], []),
    }
    # an inline mesh whose stored bounding sphere does not enclose it (the reference uses the stored sphere
    # verbatim, mod.rs:315): the quad at z = 0 is only "seen" by rays that also hit the small sphere placed (a) in
    # front of it, (b) behind the camera - there the line hits the sphere but intersect_sphere returns None, which
    # is exactly the case the speculative gate of the device code has to catch and redo
    quad = [ptlib.make_tri((-2, -2, 0), (2, -2, 0), (2, 2, 0)), ptlib.make_tri((-2, -2, 0), (2, 2, 0), (-2, 2, 0))]
    cases["lying_sphere_front"] = ptlib.Scene("l1", cam, [
        ptlib.make_sphere((0, 0, 0), 30.0, (0.5, 0.5, 0.5), (0.2, 0.2, 0.2), "Diffuse"),
        ptlib.make_mesh((0, 0, 0), (0.9, 0.2, 0.2), (0.1, 0, 0), "Diffuse", 0, 2, (0.3, 0.2, 1.0), 0.7)], quad)
    cases["lying_sphere_behind"] = ptlib.Scene("l2", cam, [
        ptlib.make_sphere((0, 0, 0), 30.0, (0.5, 0.5, 0.5), (0.2, 0.2, 0.2), "Diffuse"),
        ptlib.make_mesh((0, 0, 0), (0.9, 0.2, 0.2), (0.1, 0, 0), "Diffuse", 0, 2, (0.0, 0.0, 9.0), 3.0)], quad)
    for name, sc in cases.items():
        for (w, h, spp) in ((1, 1, 5), (33, 17, 6)):
            want, cnt, _ = ptlib.oracle_render(sc, w, h, spp, 21)
            for backend in (0, 1):
                got, st = gpu_render(gpu, sc, w, h, spp, 21, backend)
                assert st.ray_bounces == cnt.ray_bounces, (name, backend)
                assert float(np.abs(got - want).max()) <= TOL, (name, backend)
    # malformed scenes are rejected, not rendered
    L, ctx = gpu
    bad = ptlib.Scene("x", cam, [ptlib.make_mesh((0, 0, 0), (1, 1, 1), (0, 0, 0), "Diffuse", 0, 5, (0, 0, 0), 1.0)],
                      [ptlib.make_tri((0, 0, 0), (1, 0, 0), (0, 1, 0))])
    assert L.pt_ctx_set_scene(ctx, C.byref(bad.cam), bad.objs, bad.n_objs, bad.tris, bad.n_tris) == -1
    o = ptlib.make_sphere((0, 0, 0), 1.0, (1, 1, 1), (0, 0, 0), "Diffuse")
    o.reflect_type = 9
    bad = ptlib.Scene("y", cam, [o], [])
    assert L.pt_ctx_set_scene(ctx, C.byref(bad.cam), bad.objs, bad.n_objs, bad.tris, bad.n_tris) == -1


def test_largest_frame_geometry(gpu):
    """BASELINE config 5 geometry (cornell 4096x4096) at 2 spp: 16.7 M pixels, 16 384 streams of 1 024 pixels,
    33 M primary rays in one pass.  Wavefront == megakernel bit for bit, bounce counts equal, and a sample of
    pixels agrees with the oracle."""
    O = ptlib.oracle()
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    w = h = 4096
    spp = 2
    a, sa = gpu_render(gpu, sc, w, h, spp, 1, 0)
    b, sb = gpu_render(gpu, sc, w, h, spp, 1, 1)
    assert sa.ray_bounces == sb.ray_bounces and sa.samples == w * h * spp
    assert np.array_equal(a, b)
    ps = sc.pto()
    cfg = PtoConfig(w, h, spp, 0, 1)
    rng = np.random.default_rng(9)
    px = np.zeros(3, np.float32)
    idxs = list(rng.integers(0, w * h, size=400)) + [0, w * h - 1, w * h // 2, 1024 * 1024 - 1, 1024 * 1024]
    for idx in idxs:
        O.pto_render_pixel(C.byref(ps), C.byref(cfg), int(idx), _np_f(px), None)
        assert float(np.abs(px - a[idx]).max()) <= TOL, idx


def test_hdodec_extension_parity(gpu):
    """Fan-triangulated hdodec.off (36 triangles -> BVH path) inside the cornell walls: GPU == oracle on the same
    triangles (the triangulation itself has no reference behaviour to match)."""
    sc = ptlib.load_scene_py(ptlib.scene_path("mesh-hdodec"), triangulate=True)
    w, h, spp = 96, 64, 8
    want, cnt, _ = ptlib.oracle_render(sc, w, h, spp, 4)
    for backend in (0, 1):
        got, st = gpu_render(gpu, sc, w, h, spp, 4, backend)
        assert st.ray_bounces == cnt.ray_bounces
        assert float(np.abs(got - want).max()) <= TOL
    assert (want[:, 0] != want[:, 2]).any()


def test_interleaved_chunks_equal_whole_frame(gpu):
    """The multi-GPU partition of bench.py: rank r renders chunks r, r+N, ... of the frame into a dense buffer.
    Reassembled by chunk index the N buffers are the one-call frame, bit for bit (incl. a partial last chunk, more
    ranks than chunks, and chunking inside a band)."""
    L, ctx = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    set_scene(gpu, sc)
    w, h, spp = 50, 33, 6
    npix = w * h
    whole, st = gpu_render(gpu, sc, w, h, spp, 12)

    def chunked(backend, C_, n, band=(0, 0)):
        b0, e0 = band if band != (0, 0) else (0, npix)
        img = np.zeros_like(whole)
        total = 0
        for r in range(n):
            cfg = PtConfig(w, h, spp, backend, 12, band[0], band[1], 0, 0, C_, r, n, 0)
            own = L.pt_config_pixels(C.byref(cfg))
            span = e0 - b0
            idx = np.arange(span)
            mine = idx[(idx // C_) % n == r] + b0
            assert own == len(mine)
            if own == 0:
                continue
            d_out = C.c_void_p()
            assert L.pt_device_malloc(0, own * 12, C.byref(d_out)) == 0
            s2 = PtStats()
            rc = L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, None, None, None, C.byref(s2))
            assert rc == 0, L.pt_last_error()
            part = np.zeros((own, 3), dtype=np.float32)
            assert L.pt_device_download(0, part.ctypes.data_as(C.c_void_p), d_out, own * 12) == 0
            L.pt_device_free(0, d_out)
            img[mine] = part
            total += s2.ray_bounces
        return img, total

    for backend in (0, 1):
        for C_, n in ((w, 2), (w, 8), (64, 3), (7, 5), (npix, 4), (1000, 40)):
            img, total = chunked(backend, C_, n)
            assert np.array_equal(img, whole), (backend, C_, n)
            assert total == st.ray_bounces
    img, _ = chunked(0, 32, 3, band=(100, 1300))
    assert np.array_equal(img[100:1300], whole[100:1300]) and not img[:100].any() and not img[1300:].any()
    bad = PtConfig(w, h, spp, 0, 12, 0, 0, 0, 0, 0, 1, 2, 0)   # chunk_pixels == 0
    assert L.pt_config_pixels(C.byref(bad)) == 0
    bad = PtConfig(w, h, spp, 0, 12, 0, 0, 0, 0, 8, 2, 2, 0)   # chunk_first >= chunk_step
    assert L.pt_config_pixels(C.byref(bad)) == 0


def test_large_mesh_bvh_in_global_memory(gpu):
    """A 3 968-triangle tessellated sphere (2 000 BVH nodes: more than the LDS budget, so the nodes stay in global
    memory and the traversal stack is the u32 form) next to a 200-triangle one (nodes in LDS), glass and diffuse,
    inside an emissive enclosure: ray-by-ray hits and a small frame against the oracle's linear scan."""
    L, ctx = gpu
    O = ptlib.oracle()

    def uv_sphere(radius, steps):
        tris = []
        for i in range(steps):
            t1, t2 = np.pi * i / steps, np.pi * (i + 1) / steps
            for j in range(2 * steps):
                p1, p2 = 2 * np.pi * j / (2 * steps), 2 * np.pi * (j + 1) / (2 * steps)
                P = lambda t, p: (radius * np.sin(t) * np.cos(p), radius * np.cos(t), radius * np.sin(t) * np.sin(p))
                a, b, c, d = P(t1, p1), P(t2, p1), P(t2, p2), P(t1, p2)
                if i == 0:
                    tris.append((a, c, d))
                elif i + 1 == steps:
                    tris.append((a, b, c))
                else:
                    tris.append((a, b, d))
                    tris.append((b, c, d))
        return [ptlib.make_tri(*t) for t in tris]

    big, small = uv_sphere(1.0, 32), uv_sphere(0.6, 8)
    assert len(big) == 3968 and len(small) == 224
    cam = ptlib.make_camera((0, 0, 6), (0, 0, -1))
    objs = [ptlib.make_sphere((0, 0, 0), 20.0, (0.6, 0.6, 0.6), (0.4, 0.4, 0.4), "Diffuse"),
            ptlib.make_mesh((-1.1, 0, 0), (0.9, 0.9, 0.9), (0, 0, 0), "Refract", 0, len(big), (0, 0, 0), 1.01),
            ptlib.make_mesh((1.0, 0.2, 0.5), (0.9, 0.3, 0.3), (0, 0, 0), "Diffuse", len(big), len(small), (0, 0, 0), 0.61)]
    sc = ptlib.Scene("tess", cam, objs, big + small)
    set_scene(gpu, sc)
    rng = np.random.default_rng(4)
    n = 60000
    o = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    tgt = rng.normal(size=(n, 3)).astype(np.float32) * 0.8 + np.array([[-1.1, 0, 0]], np.float32) * (rng.random((n, 1)) < 0.6)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.ascontiguousarray(d.astype(np.float32))
    o = np.ascontiguousarray(o)

    def run(fn, handle):
        t = np.zeros(n, np.float32)
        oid = np.zeros(n, np.int32)
        tid = np.zeros(n, np.int32)
        rc = fn(handle, _np_f(o), _np_f(d), n, _np_f(t), oid.ctypes.data_as(ptlib.i32p),
                tid.ctypes.data_as(ptlib.i32p), None, None)
        return rc, t, oid, tid

    rc, t, oid, tid = run(L.pt_ctx_intersect, ctx)
    assert rc == 0, L.pt_last_error()
    ps = sc.pto()
    _, t0, oid0, tid0 = run(O.pto_intersect_batch, C.byref(ps))
    assert (oid0 == 1).mean() > 0.2 and (oid0 == 2).mean() > 0.01
    assert np.array_equal(oid, oid0) and np.array_equal(tid, tid0)
    assert np.array_equal(t.view(np.uint32), t0.view(np.uint32))
    w, h, spp = 40, 30, 2
    want, cnt, _ = ptlib.oracle_render(sc, w, h, spp, 2)
    for backend in (0, 1):
        got, st = gpu_render(gpu, sc, w, h, spp, 2, backend)
        assert st.ray_bounces == cnt.ray_bounces
        assert float(np.abs(got - want).max()) <= TOL


def test_concurrent_pipelines_same_image(gpu):
    """PT_FLAG_PIPELINES(n): n wavefront pipelines on n streams share the call's pixels chunk by chunk.  Same image
    and bounce count as the single pipeline, also on top of a rank's own chunking and with a partial last chunk."""
    L, ctx = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    set_scene(gpu, sc)
    w, h, spp = 50, 33, 6
    npix = w * h
    whole, st = gpu_render(gpu, sc, w, h, spp, 12)

    def render(cfg):
        own = L.pt_config_pixels(C.byref(cfg))
        d_out = C.c_void_p()
        assert L.pt_device_malloc(0, own * 12, C.byref(d_out)) == 0
        s2 = PtStats()
        rc = L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, None, None, None, C.byref(s2))
        assert rc == 0, L.pt_last_error()
        part = np.zeros((own, 3), dtype=np.float32)
        assert L.pt_device_download(0, part.ctypes.data_as(C.c_void_p), d_out, own * 12) == 0
        L.pt_device_free(0, d_out)
        return part, s2.ray_bounces

    for n in (2, 3, 5, 8):  # rays_per_pass kept small: every pipeline allocates its own ray streams
        img, nb = render(PtConfig(w, h, spp, 0, 12, 0, 0, 4000, n << 8))
        assert np.array_equal(img, whole) and nb == st.ray_bounces, n
    # inside a rank's partition: rank 1 of 3 with 64-pixel chunks, 2 pipelines
    cfg = PtConfig(w, h, spp, 0, 12, 0, 0, 4000, 2 << 8, 64, 1, 3, 0)
    img, _ = render(cfg)
    idx = np.arange(npix)
    mine = idx[(idx // 64) % 3 == 1]
    assert np.array_equal(img, whole[mine])
    # a band, 3 pipelines
    img, _ = render(PtConfig(w, h, spp, 0, 12, 130, 1500, 4000, 3 << 8))
    assert np.array_equal(img, whole[130:1500])


PT_FLAG_SEPARATE_KERNELS = 2


def _render_flags(gpu, sc, w, h, spp, seed, flags, rays_per_pass=0, chunks=None):
    L, _ = gpu
    cfg = PtConfig(w, h, spp, ptlib.BACKEND_WAVEFRONT, seed, 0, 0, rays_per_pass, flags)
    if chunks:
        cfg.chunk_pixels, cfg.chunk_first, cfg.chunk_step = chunks
    out = np.zeros((w * h, 3), dtype=np.float32)  # pt_render puts the chunks of a share at their places in the frame
    st = PtStats()
    rc = L.pt_render(C.byref(cfg), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None, None,
                     None, C.byref(st))
    assert rc == 0, L.pt_last_error()
    return out, st


@pytest.mark.parametrize("sid", ["cornell", "three-spheres", "cartesian", "single-sphere", "mesh"])
def test_pass_kernel_equals_separate_kernels(gpu, sid):
    """k_pass / k_pass_bvh (a whole pass per launch, the default) and the generate / intersect / shade kernels give the
    same image bits and the same bounce count: one pass, many ragged passes, an interleaved share of the frame."""
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    for (w, h, spp, rpp, chunks) in [(96, 64, 24, 0, None), (97, 53, 50, 97 * 53 * 7, None), (96, 64, 9, 0, (96, 1, 3)),
                                     (33, 17, 3, 33 * 17, None)]:
        a, sa = _render_flags(gpu, sc, w, h, spp, 11, 0, rpp, chunks)
        b, sb = _render_flags(gpu, sc, w, h, spp, 11, PT_FLAG_SEPARATE_KERNELS, rpp, chunks)
        assert sa.ray_bounces == sb.ray_bounces and sa.samples == sb.samples and sa.passes == sb.passes
        assert sa.intersect_launches == sa.passes and sb.intersect_launches == 12 * sb.passes  # really two paths
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (sid, w, h, spp)
        # ... and the stand-alone intersect step in both its forms: candidate scan (k_intersect_cand, above) and every
        # triangle per ray (k_intersect<false>: PT_FLAG_NO_BVH)
        c, sc3 = _render_flags(gpu, sc, w, h, spp, 11, PT_FLAG_SEPARATE_KERNELS | ptlib.FLAG_NO_BVH, rpp, chunks)
        assert sc3.ray_bounces == sa.ray_bounces and np.array_equal(a.view(np.uint32), c.view(np.uint32)), (sid, w, h, spp)


def test_pass_kernel_full_size(gpu):
    """Bench geometry (1024x768, 16 K streams): k_pass against the separate kernels and against the megakernel."""
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    a, sa = _render_flags(gpu, sc, 1024, 768, 48, 5, 0, rays_per_pass=32 << 20)
    b, sb = _render_flags(gpu, sc, 1024, 768, 48, 5, PT_FLAG_SEPARATE_KERNELS, rays_per_pass=32 << 20)
    assert sa.ray_bounces == sb.ray_bounces and sa.passes == 2
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    m, sm = gpu_render(gpu, sc, 1024, 768, 48, 5, backend=ptlib.BACKEND_MEGAKERNEL)
    assert sm.ray_bounces == sa.ray_bounces and np.array_equal(a.view(np.uint32), m.view(np.uint32))


def test_gate_shortcut_at_the_rim(gpu):
    """The winner's bounding-sphere gate is skipped when the hit point lies well inside the sphere (rr_in,
    intersect_scene_dev).  Meshes whose stored sphere does not enclose them - small, large, off-centre - with rays
    aimed at the rim of the sphere on the mesh, from inside and outside the sphere, towards and away from it: hit
    or miss, distance, triangle, point and normal must be the oracle's, bit for bit."""
    L, ctx = gpu
    O = ptlib.oracle()
    cam = ptlib.make_camera((0, 0, 3), (0, 0, -1))
    rng = np.random.default_rng(123)
    quad = [ptlib.make_tri((-2, -2, 0), (2, -2, 0), (2, 2, 0)), ptlib.make_tri((-2, -2, 0), (2, 2, 0), (-2, 2, 0))]
    for (centre, radius) in (((0.0, 0.0, 0.0), 1.5), ((0.4, -0.3, 0.2), 1.1), ((0.0, 0.0, 0.0), 3.5),
                             ((1.0, 1.0, -0.5), 2.0)):
        sc = ptlib.Scene("rim", cam, [ptlib.make_mesh((0, 0, 0), (0.9, 0.9, 0.9), (0, 0, 0), "Diffuse", 0, 2, centre,
                                                      radius)], quad)
        set_scene(gpu, sc)
        n = 200000
        c = np.array(centre, np.float32)
        # targets on the quad plane at distances around the circle where the sphere cuts the plane
        rho = np.sqrt(max(radius * radius - float(c[2]) ** 2, 1e-3))
        ang = rng.uniform(0, 2 * np.pi, n)
        rr = rho * (1.0 + rng.choice([0.0, 1e-7, -1e-7, 1e-4, -1e-4, 1e-3, -1e-3, 2e-3, -2e-3, 0.05, -0.5], n)
                    + rng.normal(0, 1e-6, n))
        tgt = np.stack([c[0] + rr * np.cos(ang), c[1] + rr * np.sin(ang), np.zeros(n)], 1)
        org = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.choice([3.0, 0.5, 1e-3, -1.0, 2.9], n)], 1)
        d = tgt - org
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[::7] *= -1.0  # some rays point away from the plane
        o = np.ascontiguousarray(org.astype(np.float32))
        d = np.ascontiguousarray(d.astype(np.float32))
        ps = sc.pto()

        def run(fn, handle):
            t = np.zeros(n, np.float32)
            oid = np.zeros(n, np.int32)
            tid = np.zeros(n, np.int32)
            x = np.zeros((n, 3), np.float32)
            nr = np.zeros((n, 3), np.float32)
            rc = fn(handle, _np_f(o), _np_f(d), n, _np_f(t), oid.ctypes.data_as(ptlib.i32p),
                    tid.ctypes.data_as(ptlib.i32p), _np_f(x), _np_f(nr))
            return rc, t, oid, tid, x

        rc, t, oid, tid, x = run(L.pt_ctx_intersect, ctx)
        assert rc == 0, L.pt_last_error()
        _, t0, oid0, tid0, x0 = run(O.pto_intersect_batch, C.byref(ps))
        assert 0.05 < (oid0 >= 0).mean() < 0.95  # both outcomes are well represented
        assert np.array_equal(oid, oid0) and np.array_equal(tid, tid0)
        assert np.array_equal(t.view(np.uint32), t0.view(np.uint32))
        assert np.array_equal(x.view(np.uint32), x0.view(np.uint32))


def _random_scene(rng, k):
    """A random room: spheres of all three materials (some emissive, some huge, some touching), meshes of 1-40
    triangles (axis-aligned quads, slivers, soups) with Mesh::new's bounding sphere, a lying one, or a loose one."""
    cam = ptlib.make_camera((float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), float(rng.uniform(3, 7))),
                            (float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2)), -1.0))
    mats = ["Diffuse", "Specular", "Refract"]
    objs, tris = [], []
    for _ in range(int(rng.integers(0, 6))):
        r = float(rng.choice([0.3, 0.8, 1.5, 25.0]))
        pos = tuple(float(v) for v in rng.uniform(-3, 3, 3))
        em = tuple(float(v) for v in (rng.uniform(0, 6, 3) if rng.random() < 0.4 else np.zeros(3)))
        col = tuple(float(v) for v in rng.uniform(0.05, 0.999, 3))
        objs.append(ptlib.make_sphere(pos, r, col, em, mats[int(rng.integers(0, 3))]))
    for _ in range(int(rng.integers(1, 7))):
        n = int(rng.choice([1, 2, 2, 2, 5, 17, 40]))
        base = len(tris)
        if n == 2:  # a wall
            ax = int(rng.integers(0, 3))
            sx, sy = float(rng.uniform(1, 6)), float(rng.uniform(1, 6))
            v = np.zeros((4, 3), np.float32)
            for i in range(2):
                for j in range(2):
                    v[2 * i + j][(ax + 1) % 3] = -sx if i == 0 else sx
                    v[2 * i + j][(ax + 2) % 3] = -sy if j == 0 else sy
            local = [(v[0], v[1], v[2]), (v[2], v[1], v[3])]
        else:
            c0 = rng.uniform(-1, 1, 3)
            local = [tuple(c0 + rng.normal(0, rng.choice([0.02, 0.5, 1.5]), 3) for _ in range(3)) for _ in range(n)]
        for (a, b, c) in local:
            tris.append(ptlib.make_tri(tuple(map(float, a)), tuple(map(float, b)), tuple(map(float, c))))
        pts = np.array([p for t in local for p in t], np.float32)
        lo, hi = pts.min(0), pts.max(0)
        centre = lo + hi * np.float32(0.5)  # Mesh::new's centre (mod.rs:463)
        radius = float(max(np.linalg.norm(lo - centre), np.linalg.norm(hi - centre)))
        mode = rng.random()
        if mode < 0.25:  # a sphere that does not enclose the mesh (stored spheres are used verbatim)
            centre, radius = centre + rng.normal(0, 0.5, 3).astype(np.float32), radius * float(rng.uniform(0.2, 0.7))
        elif mode < 0.4:
            radius *= 3.0
        pos = tuple(float(v) for v in rng.uniform(-3, 3, 3))
        em = tuple(float(v) for v in (rng.uniform(0, 3, 3) if rng.random() < 0.3 else np.zeros(3)))
        col = tuple(float(v) for v in rng.uniform(0.05, 0.999, 3))
        objs.append(ptlib.make_mesh(pos, col, em, mats[int(rng.integers(0, 3))], base, n,
                                    tuple(float(v) for v in centre), max(radius, 1e-3)))
    order = rng.permutation(len(objs))
    return ptlib.Scene("fuzz%d" % k, cam, [objs[i] for i in order], tris)


def test_random_scenes_against_oracle(gpu):
    """40 random scenes through every device path (k_pass, separate kernels, megakernel, with and without the
    acceleration structures): bounce counts equal the oracle's exactly, images within the tolerance, and the device
    paths agree with each other bit for bit."""
    rng = np.random.default_rng(int(os.environ.get("PT_FUZZ_SEED", "2026")))
    total = 0
    for k in range(int(os.environ.get("PT_FUZZ_SCENES", "40"))):  # one-off long runs: PT_FUZZ_SCENES=500 PT_FUZZ_SEED=...
        sc = _random_scene(rng, k)
        w, h, spp = 40, 28, int(os.environ.get("PT_FUZZ_SPP", "6"))  # (PT_FUZZ_SPP=80: passes whose chunks of 64 primaries are one pixel's samples)
        want, cnt, _ = ptlib.oracle_render(sc, w, h, spp, 100 + k)
        ref_img = None
        for (backend, flags) in ((0, 0), (0, PT_FLAG_SEPARATE_KERNELS), (0, 1), (1, 0)):
            L, _ = gpu
            cfg = PtConfig(w, h, spp, backend, 100 + k, 0, 0, 0, flags)
            out = np.zeros((w * h, 3), dtype=np.float32)
            st = PtStats()
            rc = L.pt_render(C.byref(cfg), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None,
                             None, None, C.byref(st))
            assert rc == 0, L.pt_last_error()
            assert st.ray_bounces == cnt.ray_bounces, (k, backend, flags)
            assert float(np.abs(out - want).max()) <= TOL, (k, backend, flags)
            if ref_img is None:
                ref_img = out
            else:
                assert np.array_equal(out.view(np.uint32), ref_img.view(np.uint32)), (k, backend, flags)
        total += cnt.ray_bounces
    assert total > 100000


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("sid", ["single-sphere", "two-spheres", "three-spheres", "cartesian", "cornell", "mesh"])
def test_gpu_against_golden_scene(gpu, sid):
    """The committed golden vectors (tests/golden, tools/make_golden.py) without the oracle in the loop: the frame of
    every device path within the tolerance and with exactly the recorded number of ray bounces; every recorded ray's
    intersection - distance, object, triangle, hit point, normal - bit for bit."""
    L, ctx = gpu
    g = np.load(os.path.join(GOLDEN, "scene_%s.npz" % sid))
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    w, h, spp, seed = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["seed"])
    for backend, flags in ((0, 0), (0, PT_FLAG_SEPARATE_KERNELS), (1, 0)):
        cfg = PtConfig(w, h, spp, backend, seed, 0, 0, 0, flags)
        out = np.zeros((w * h, 3), dtype=np.float32)
        st = PtStats()
        rc = L.pt_render(C.byref(cfg), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None, None,
                         None, C.byref(st))
        assert rc == 0, L.pt_last_error()
        assert st.ray_bounces == int(g["ray_bounces"]), (backend, flags)
        assert float(np.abs(out - g["image"]).max()) <= TOL, (backend, flags)
    set_scene(gpu, sc)
    o, d = np.ascontiguousarray(g["ray_o"]), np.ascontiguousarray(g["ray_d"])
    m = len(o)
    t, oid, tid = np.zeros(m, np.float32), np.zeros(m, np.int32), np.zeros(m, np.int32)
    x, nr = np.zeros((m, 3), np.float32), np.zeros((m, 3), np.float32)
    rc = L.pt_ctx_intersect(ctx, _np_f(o), _np_f(d), m, _np_f(t), oid.ctypes.data_as(ptlib.i32p),
                            tid.ctypes.data_as(ptlib.i32p), _np_f(x), _np_f(nr))
    assert rc == 0, L.pt_last_error()
    assert np.array_equal(oid, g["hit_object"]) and np.array_equal(tid, g["hit_triangle"])
    for got, key in ((t, "hit_t"), (x, "hit_x"), (nr, "hit_n")):
        assert np.array_equal(got.view(np.uint32), g[key].view(np.uint32)), key


def test_gpu_against_golden_numerics(gpu):
    L, ctx = gpu
    g = np.load(os.path.join(GOLDEN, "numerics.npz"))
    x = np.ascontiguousarray(g["sincos_x"])
    n = len(x)
    s, c, q, r = (np.zeros(n, np.float32) for _ in range(4))
    ph = np.zeros(4 * n, np.uint32)
    rc = L.pt_ctx_numerics_probe(ctx, _np_f(x), n, _np_f(s), _np_f(c), _np_f(q), _np_f(r), ph.ctypes.data_as(ptlib.u32p))
    assert rc == 0, L.pt_last_error()
    assert np.array_equal(s.view(np.uint32), g["sin"].view(np.uint32))
    assert np.array_equal(c.view(np.uint32), g["cos"].view(np.uint32))


REF_SPHERE_KATS = [  # src/render/test.rs:43-144, the reference's own intersect_scene fixtures
    ((0, 0, -3), (0, 0, 0), (0, 0, -1), (0, 2.0, [0, 0, -2], [0, 0, 1])),
    ((0, 0, -3), (2, 0, 0), "norm(1,0,-1)", None),
    ((0, 0, 0), (0, 0, 0), (0, 0, -1), (0, 1.0, [0, 0, -1], [0, 0, -1])),
    ((0, 0, -3), (0, 1, 0), (0, 0, -1), (0, 3.0, [0, 1, -3], [0, 1, 0])),
]


@pytest.mark.parametrize("pos,o,d,want", REF_SPHERE_KATS)
def test_reference_sphere_kats_through_the_c_abi(gpu, pos, o, d, want):
    """The reference's own intersect_scene fixtures (src/render/test.rs:43-144) against the HIP path, exact values."""
    L, ctx = gpu
    if isinstance(d, str):
        v = np.array([1.0, 0.0, -1.0], np.float32)
        inv = np.float32(1.0) / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2], dtype=np.float32)
        d = [float(c) for c in v * inv]  # glam normalize: v * (1 / length)
    sc = ptlib.Scene("kat", ptlib.make_camera((0, 0, 5), (0, 0, -1)),
                     [ptlib.make_sphere(pos, 1.0, (1, 1, 1), (0, 0, 0), "Diffuse")], [])
    set_scene(gpu, sc)
    oa = np.array([o], np.float32)
    da = np.array([d], np.float32)
    t, oid, tid = np.zeros(1, np.float32), np.zeros(1, np.int32), np.zeros(1, np.int32)
    x, nr = np.zeros((1, 3), np.float32), np.zeros((1, 3), np.float32)
    rc = L.pt_ctx_intersect(ctx, _np_f(oa), _np_f(da), 1, _np_f(t), oid.ctypes.data_as(ptlib.i32p),
                            tid.ctypes.data_as(ptlib.i32p), _np_f(x), _np_f(nr))
    assert rc == 0, L.pt_last_error()
    if want is None:
        assert int(oid[0]) == -1
    else:
        assert (int(oid[0]), float(t[0]), list(x[0]), list(nr[0])) == \
               (want[0], want[1], [float(v) for v in want[2]], [float(v) for v in want[3]])


def test_reference_radiance_test_through_the_c_abi(gpu):
    """src/render/test.rs:146-183 (`test_radiance`): a red diffuse unit sphere in front, an emissive one behind the
    camera; the reference asserts mean radiance .x > 0.3 over 10 000 samples of the central ray (analytic 50/144).
    Here: the central pixel of a 1001x667 frame (its rays leave the lens centre within 0.03 degrees of -z), 65 536
    samples, on every device path."""
    cam = ptlib.make_camera((0, 0, 0.035), (0, 0, -1))  # lens centre at the origin, looking down -z
    sc = ptlib.Scene("t", cam, [ptlib.make_sphere((0, 0, -3), 1.0, (1, 0, 0), (0, 0, 0), "Diffuse"),
                                ptlib.make_sphere((0, 0, 10), 1.0, (0, 0, 0), (50, 50, 50), "Diffuse")], [])
    w, h = 1001, 667
    idx = (h - 1 - h // 2) * w + w // 2
    for backend in (0, 1):
        got, st = gpu_render(gpu, sc, w, h, 65536, 3, backend, band=(idx, idx + 1))
        px = got[idx]
        assert px[0] > 0.3 and abs(px[0] - 50.0 / 144.0) < 0.02, px
        assert px[1] == 0.0 and px[2] == 0.0


# ---------------------------------------------------------------------------------------------------------------
# round 2: hand-derived KATs, bounds queries, cancel / progress semantics, parity at the spp BASELINE.json names
import kats


@pytest.mark.parametrize("name,build,o,d,want", kats.CASES, ids=[c[0] for c in kats.CASES])
def test_hand_derived_kats_through_the_c_abi(gpu, name, build, o, d, want):
    """tests/kats.py (triangles, the gate, the tie rules - worked out by hand from mod.rs:554-659) on the HIP path."""
    L, ctx = gpu
    sc = build()
    set_scene(gpu, sc)
    ro, rd = np.array([o], np.float32), np.array([d], np.float32)
    t, oid, tid = np.zeros(1, np.float32), np.zeros(1, np.int32), np.zeros(1, np.int32)
    x, n = np.zeros((1, 3), np.float32), np.zeros((1, 3), np.float32)
    rc = L.pt_ctx_intersect(ctx, _np_f(ro), _np_f(rd), 1, _np_f(t), oid.ctypes.data_as(ptlib.i32p),
                            tid.ctypes.data_as(ptlib.i32p), _np_f(x), _np_f(n))
    assert rc == 0, L.pt_last_error()
    if want is None:
        assert oid[0] == -1, (name, oid[0], t[0])
        return
    assert (oid[0], tid[0]) == (want["object_id"], want["tri_id"]), name
    assert t[0] == np.float32(want["t"]), (name, t[0])
    assert list(x[0]) == [np.float32(v) for v in want["x"]], (name, x[0])
    assert list(n[0]) == [np.float32(v) for v in want["n"]], (name, n[0])


@pytest.mark.parametrize("sid", ["cornell", "mesh", "three-spheres"])
def test_intersect_bounds_and_orbit_point_ray_by_ray(gpu, sid):
    """SceneObjectData::intersect_bounds (mod.rs:282-290) for every object and get_orbit_point
    (viewport_tab.rs:401-431) against the oracle, bit for bit, on camera rays and on rays from inside the scene."""
    L, ctx = gpu
    O = ptlib.oracle()
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    set_scene(gpu, sc)
    boxes = ptlib.oracle_boxes(sc)
    rng = np.random.default_rng(11)
    m = 6000
    cfg = PtoConfig(96, 64, 1, 0, 3)
    o = np.zeros((m, 3), np.float32)
    d = np.zeros((m, 3), np.float32)
    for i in range(m // 2):  # picking rays: primary rays of random pixels
        oo, dd = (C.c_float * 3)(), (C.c_float * 3)()
        O.pto_primary_ray(C.byref(sc.cam), 96, 64, int(rng.integers(0, 96 * 64)), i, 3, oo, dd)
        o[i], d[i] = list(oo), list(dd)
    o[m // 2:] = rng.uniform(-2.0, 2.0, size=(m - m // 2, 3)).astype(np.float32)
    v = rng.normal(size=(m - m // 2, 3))
    d[m // 2:] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    ps = sc.pto()
    for k in range(sc.n_objs):
        hit_w, hit_g = np.zeros(m, np.int32), np.zeros(m, np.int32)
        tw, tg = np.zeros(m, np.float32), np.zeros(m, np.float32)
        xw, xg = np.zeros((m, 3), np.float32), np.zeros((m, 3), np.float32)
        nw, ng = np.zeros((m, 3), np.float32), np.zeros((m, 3), np.float32)
        O.pto_intersect_bounds_batch(C.byref(ps), boxes, k, _np_f(o), _np_f(d), m, hit_w.ctypes.data_as(ptlib.i32p),
                                     _np_f(tw), _np_f(xw), _np_f(nw))
        rc = L.pt_ctx_intersect_bounds(ctx, k, _np_f(o), _np_f(d), m, hit_g.ctypes.data_as(ptlib.i32p), _np_f(tg),
                                       _np_f(xg), _np_f(ng))
        assert rc == 0, L.pt_last_error()
        assert np.array_equal(hit_w, hit_g), (sid, k)
        assert hit_w.any() or sc.objs[k].kind == ptlib.PT_SPHERE
        for a, b, what in ((tw, tg, "t"), (xw, xg, "x"), (nw, ng, "n")):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (sid, k, what)
    fw, fg = np.zeros(m, np.int32), np.zeros(m, np.int32)
    ow, og = np.zeros(m, np.int32), np.zeros(m, np.int32)
    pw, pg = np.zeros((m, 3), np.float32), np.zeros((m, 3), np.float32)
    tw, tg = np.zeros(m, np.float32), np.zeros(m, np.float32)
    O.pto_orbit_point_batch(C.byref(ps), boxes, _np_f(o), _np_f(d), m, fw.ctypes.data_as(ptlib.i32p), _np_f(pw),
                            ow.ctypes.data_as(ptlib.i32p), _np_f(tw))
    rc = L.pt_ctx_orbit_point(ctx, _np_f(o), _np_f(d), m, fg.ctypes.data_as(ptlib.i32p), _np_f(pg),
                              og.ctypes.data_as(ptlib.i32p), _np_f(tg))
    assert rc == 0, L.pt_last_error()
    assert np.array_equal(fw, fg) and np.array_equal(ow, og) and fw.sum() > 100
    assert np.array_equal(pw.view(np.uint32), pg.view(np.uint32)) and np.array_equal(tw.view(np.uint32), tg.view(np.uint32))
    # an explicit box (what an inline Mesh of a scene file carries) replaces the computed one
    if sid == "cornell":
        k = 4
        box = (ptlib.PtTriangle * 12)(*[boxes[12 * k + j] for j in range(12)])
        for j in range(12):
            box[j].a[0] += 0.5  # move the box: hits must move with it
            box[j].b[0] += 0.5
            box[j].c[0] += 0.5
        assert L.pt_ctx_set_mesh_bounds(ctx, k, box) == 0
        moved = (ptlib.PtTriangle * (12 * sc.n_objs))(*[boxes[j] for j in range(12 * sc.n_objs)])
        for j in range(12):
            moved[12 * k + j] = box[j]
        O.pto_intersect_bounds_batch(C.byref(ps), moved, k, _np_f(o), _np_f(d), m, hit_w.ctypes.data_as(ptlib.i32p),
                                     _np_f(tw), _np_f(xw), _np_f(nw))
        assert L.pt_ctx_intersect_bounds(ctx, k, _np_f(o), _np_f(d), m, hit_g.ctypes.data_as(ptlib.i32p), _np_f(tg),
                                         _np_f(xg), _np_f(ng)) == 0
        assert np.array_equal(hit_w, hit_g) and np.array_equal(tw.view(np.uint32), tg.view(np.uint32))
    assert L.pt_ctx_intersect_bounds(ctx, sc.n_objs, _np_f(o), _np_f(d), 1, None, None, None, None) == -1


@pytest.mark.parametrize("backend", [ptlib.BACKEND_WAVEFRONT, ptlib.BACKEND_MEGAKERNEL])
def test_cancel_mid_frame_and_progress_cadence(gpu, backend):
    """Cancel raised from the progress callback in the middle of a frame (mod.rs:943-958): PT_CANCELLED, and the
    framebuffer is the frame over the samples that were accumulated - the spp_done-sample frame of the same seed, bit for
    bit - with stats.samples saying how many.  Progress cadence (mod.rs:965-982): by default at most every 500 ms, so
    a frame of a few milliseconds only reports its completion; PT_PROGRESS_EVERY_PASS reports every pass."""
    L, ctx = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    set_scene(gpu, sc)
    w, h, spp = 64, 48, 24
    npix = w * h
    nbytes = npix * 12
    d_out = C.c_void_p()
    assert L.pt_device_malloc(0, nbytes, C.byref(d_out)) == 0

    def download():
        host = np.zeros((npix, 3), dtype=np.float32)
        assert L.pt_device_download(0, host.ctypes.data_as(C.c_void_p), d_out, nbytes) == 0
        return host

    flag = (C.c_uint8 * 1)(0)
    calls = []
    armed = [True]

    def on_progress(user, frac):
        calls.append(frac)
        if armed[0] and len(calls) == 4:
            flag[0] = 1

    cb = ptlib.PROGRESS_FN(on_progress)
    # one sample per pixel and pass (wavefront) / round (megakernel): 24 boundaries
    cfg = PtConfig(w, h, spp, backend, 21, 0, 0, npix, 0)
    cfg.progress_ms = ptlib.PROGRESS_EVERY_PASS
    st = PtStats()
    rc = L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, C.cast(flag, C.c_void_p), C.cast(cb, C.c_void_p), None, C.byref(st))
    assert rc == ptlib.PT_CANCELLED, (rc, L.pt_last_error())
    assert len(calls) == 4 and all(0.0 < f < 1.0 for f in calls)
    assert st.samples % npix == 0
    done = st.samples // npix
    assert 4 <= done < spp, done
    part = download()
    assert part.any()
    cfg2 = PtConfig(w, h, done, backend, 21, 0, 0, npix, 0)
    st2 = PtStats()
    assert L.pt_ctx_render(ctx, C.byref(cfg2), d_out, None, None, None, None, C.byref(st2)) == 0, L.pt_last_error()
    assert np.array_equal(part, download()) and st2.ray_bounces == st.ray_bounces
    # ... and against the oracle: the cancelled frame is the reference's picture over `done` samples per pixel, with
    # exactly the intersect_scene evaluations of those samples
    want, cnt, _ = ptlib.oracle_render(sc, w, h, done, 21)
    assert st.ray_bounces == cnt.ray_bounces
    assert np.abs(part - want).max() <= TOL
    # default cadence: this frame takes milliseconds, far less than 500 ms -> only the completion is reported
    calls.clear()
    flag[0] = 0
    armed[0] = False
    cfg3 = PtConfig(w, h, spp, backend, 21, 0, 0, npix, 0)
    assert L.pt_ctx_render(ctx, C.byref(cfg3), d_out, None, C.cast(flag, C.c_void_p), C.cast(cb, C.c_void_p), None, C.byref(st)) == 0
    assert calls == [1.0]
    # an explicit interval of 1 ms: some reports, fewer than one per pass is allowed, the last one is the completion
    calls.clear()
    cfg3.progress_ms = 1
    cfg3.spp = 4 * spp
    assert L.pt_ctx_render(ctx, C.byref(cfg3), d_out, None, C.cast(flag, C.c_void_p), C.cast(cb, C.c_void_p), None, C.byref(st)) == 0
    assert calls[-1] == 1.0 and calls == sorted(calls)
    assert L.pt_device_free(0, d_out) == 0


def _dear_scene():
    """A room whose rays cost several times cornell's: 56 small tilted meshes of 14 triangles each (no BVH below 16; tilted, so
    no conservative filter applies) = 392 pair records that are candidates of EVERY ray, under an emissive sphere."""
    rng = np.random.default_rng(17)
    objs, tris = [], []
    for i in range(56):
        c = rng.uniform(-2.0, 2.0, 3).astype(np.float32)
        tl = []
        for _ in range(14):
            a = rng.uniform(-0.3, 0.3, 3)
            tl.append(ptlib.make_tri(a, a + rng.uniform(-0.25, 0.25, 3), a + rng.uniform(-0.25, 0.25, 3)))
        objs.append(ptlib.make_mesh(c, rng.uniform(0.3, 0.9, 3), (0, 0, 0), "Diffuse", len(tris), len(tl), (0, 0, 0), 1.0))
        tris.extend(tl)
    objs.append(ptlib.make_sphere((0, 0, 0), 9.0, (0.8, 0.8, 0.8), (0.7, 0.7, 0.7), "Diffuse"))
    cam = ptlib.make_camera((0.0, 0.0, 7.0), (0.0, 0.0, -1.0))
    return ptlib.Scene("dear", cam, objs, tris)


@pytest.mark.parametrize("backend", [0, 1])
def test_cancel_latency_follows_the_scene(gpu, backend):
    """The reference looks at its stop flag every 100 ms (mod.rs:947-958); here it is read between passes / rounds, whose
    length is measured - a short timed first pass, then as many samples as fit 100-120 ms - instead of a constant tuned on the
    bench scene.  On a scene several times dearer per ray than cornell.json, at the library's own pass size, a flag raised
    from another thread in the middle of a frame of many seconds brings the call back within 250 ms, with the samples that
    were accumulated: the frame of `done` samples per pixel, bit for bit."""
    import threading
    import time

    L, ctx = gpu
    sc = _dear_scene()
    set_scene(gpu, sc)
    w, h, spp = 768, 512, 8192  # 3.2 G primary samples: many seconds on this scene
    npix = w * h
    d_out = C.c_void_p()
    assert L.pt_device_malloc(0, npix * 12, C.byref(d_out)) == 0

    def download():
        host = np.zeros((npix, 3), dtype=np.float32)
        assert L.pt_device_download(0, host.ctypes.data_as(C.c_void_p), d_out, npix * 12) == 0
        return host

    flag = (C.c_uint8 * 1)(0)
    t_raised = [0.0]

    def raise_later():
        time.sleep(0.6)
        t_raised[0] = time.perf_counter()
        flag[0] = 1

    cfg = PtConfig(w, h, spp, backend, 9, 0, 0, 0, 0)  # rays_per_pass = 0: the library's own pass size
    st = PtStats()
    th = threading.Thread(target=raise_later)
    th.start()
    rc = L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, C.cast(flag, C.c_void_p), None, None, C.byref(st))
    t_back = time.perf_counter()
    th.join()
    assert rc == ptlib.PT_CANCELLED, (rc, L.pt_last_error())
    latency = t_back - t_raised[0]
    assert 0.0 <= latency < 0.25, latency
    assert st.samples % npix == 0
    done = st.samples // npix
    assert 0 < done < spp and st.passes >= 3  # a short timed pass, then passes of about a tenth of a second
    part = download()
    # the same samples in ONE pass (an explicit pass size): same bits, same bounce count - passes only batch the samples
    cfg2 = PtConfig(w, h, done, backend, 9, 0, 0, min(npix * done, 0xffffffff), 0)
    st2 = PtStats()
    assert L.pt_ctx_render(ctx, C.byref(cfg2), d_out, None, None, None, None, C.byref(st2)) == 0, L.pt_last_error()
    assert st2.ray_bounces == st.ray_bounces and np.array_equal(part, download())
    # the passes' length was learnt: the same frame again starts at full length (fewer passes for the same samples)
    cfg3 = PtConfig(w, h, done, backend, 9, 0, 0, 0, 0)
    st3 = PtStats()
    assert L.pt_ctx_render(ctx, C.byref(cfg3), d_out, None, None, None, None, C.byref(st3)) == 0, L.pt_last_error()
    assert st3.passes <= st.passes and st3.ray_bounces == st.ray_bounces and np.array_equal(part, download())
    per_pass_ms = st3.ms_device / st3.passes
    assert per_pass_ms < 160.0, per_pass_ms
    assert L.pt_device_free(0, d_out) == 0


def _pixels_against_oracle(gpu, sc, w, h, spp, seed, pixels, backend=0):
    """Render the image rows that hold `pixels` on the GPU (bands: the RNG is keyed on the global pixel index, so a band
    is the frame's own pixels) and compare those pixels with the oracle's render_pixel (mod.rs:794-857)."""
    L, ctx = gpu
    O = ptlib.oracle()
    ps = sc.pto()
    ocfg = PtoConfig(w, h, spp, 0, seed)
    worst = 0.0
    rows = sorted(set(int(p) // w for p in pixels))
    for r in rows:
        cfg = PtConfig(w, h, spp, backend, seed, r * w, (r + 1) * w, 0, 0)
        band = np.zeros((w * h, 3), dtype=np.float32)
        st = PtStats()
        rc = L.pt_render(C.byref(cfg), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(band), None, None,
                         None, C.byref(st))
        assert rc == 0, L.pt_last_error()
        for p in [int(q) for q in pixels if int(q) // w == r]:
            want = (C.c_float * 3)()
            O.pto_render_pixel(C.byref(ps), C.byref(ocfg), p, want, None)
            worst = max(worst, float(np.abs(band[p] - np.array(list(want), np.float32)).max()))
    return worst


def test_parity_at_baseline_spp(gpu):
    """The 1e-4 bar at the sample counts BASELINE.json names.  The GPU sums radiance exactly (32.32 fixed point, top
    down), the reference sequentially in f32 (mod.rs:846): the gap grows with spp, so it is measured where it is
    largest: 300 pixels (12 image rows) of cornell 1024x768 at 1024 and at 4096 spp (configs 2, 3), 300 pixels of mesh.json
    at 1024 spp and 60 at 4096 spp (config 4) and 50 pixels of cornell 4096x4096 at 16384 spp (config 5, both backends), HIP
    against the oracle's render_pixel.  The measured maxima are written to gpurun_out/ when that directory exists."""
    measured = {}
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    rng = np.random.default_rng(2026)
    w, h = 1024, 768
    rows = rng.choice(h, size=12, replace=False)
    pix = np.concatenate([r * w + rng.choice(w, size=25, replace=False) for r in rows])
    for spp in (1024, 4096):
        err = _pixels_against_oracle(gpu, sc, w, h, spp, 1, pix)
        measured["cornell_%dx%d_%dspp_%dpx" % (w, h, spp, len(pix))] = err
        assert err <= TOL, (spp, err)
    mesh = ptlib.load_scene_py(ptlib.scene_path("mesh"))
    rows = [150, 300, 384, 450, 520, 600]  # (the mesh covers the middle of the frame)
    pix = np.concatenate([r * w + rng.choice(w, size=50, replace=False) for r in rows])
    err = _pixels_against_oracle(gpu, mesh, w, h, 1024, 1, pix)
    measured["mesh_%dx%d_1024spp_%dpx" % (w, h, len(pix))] = err
    assert err <= TOL, err
    pix = np.concatenate([r * w + rng.choice(w, size=10, replace=False) for r in rows])
    err = _pixels_against_oracle(gpu, mesh, w, h, 4096, 1, pix)
    measured["mesh_%dx%d_4096spp_%dpx" % (w, h, len(pix))] = err
    assert err <= TOL, err
    w = h = 4096
    rows = rng.choice(h, size=5, replace=False)
    pix = np.concatenate([r * w + rng.choice(w, size=10, replace=False) for r in rows])
    for backend in (ptlib.BACKEND_WAVEFRONT, ptlib.BACKEND_MEGAKERNEL):
        err = _pixels_against_oracle(gpu, sc, w, h, 16384, 1, pix, backend)
        measured["cornell_%dx%d_16384spp_%dpx_backend%d" % (w, h, len(pix), backend)] = err
        assert err <= TOL, (backend, err)
    out_dir = os.path.join(ptlib.ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        import json
        with open(os.path.join(out_dir, "parity_at_baseline_spp.json"), "w") as f:
            json.dump({"max_abs_diff_vs_oracle": measured, "tolerance": TOL}, f, indent=1)


def _comm_gather_world_1(L, ctx):
    """pt_comm_* at world size 1 on GPU 0: a communicator of one rank, pt_comm_gather_frame (in-place ncclAllGather + the
    un-permute kernel) must return the rank's own frame, whole and as a band."""
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
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
    cfg2 = PtConfig(w, h, spp, 0, 9, 7 * w, 31 * w, 0, 0)  # a band: only the band's pixels are gathered
    assert L.pt_ctx_render(ctx, C.byref(cfg2), d_local, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
    cfg2.chunk_pixels = w
    assert L.pt_comm_gather_frame(comm, C.byref(cfg2), d_local, d_frame, None) == 0, L.pt_last_error()
    band = np.zeros((24 * w, 3), np.float32)
    assert L.pt_device_download(0, band.ctypes.data_as(C.c_void_p), d_frame, band.nbytes) == 0
    assert np.array_equal(band, a[7 * w:31 * w])
    L.pt_comm_destroy(comm)
    L.pt_device_free(0, d_local)
    L.pt_device_free(0, d_frame)


def test_comm_gather_world_1_in_this_process_with_torch(gpu):
    """pt_comm_* on hardware, IN the pytest process and with torch imported into it (round 2 could only run this in a clean
    subprocess: `ncclCommInitRank: unhandled cuda error`).  This process loaded libptrace_hip.so - and with it /opt/rocm's
    HIP runtime - before torch brings its bundled libamdhip64 / librccl (same sonames): pt_comm must pick the RCCL that
    sits next to the HIP runtime this library is bound to, not whichever copy answers to the soname (csrc/pt_comm.hip).
    No PT_RCCL_LIB.  More ranks need more GPUs than this box has; the partition arithmetic of the un-permute is what
    test_interleaved_chunks_equal_whole_frame covers."""
    assert "PT_RCCL_LIB" not in os.environ
    import torch  # noqa: F401  (maps torch/lib/libamdhip64.so, libhsa-runtime64.so, librccl.so into this process)
    assert torch.cuda.device_count() >= 1
    L, ctx = gpu
    _comm_gather_world_1(L, ctx)


@pytest.mark.parametrize("order", ["none", "first", "after"])
def test_comm_gather_world_1_load_orders(order):
    """The same in processes of their own, with torch imported before the library (bench.py's order: the library then binds
    to torch's HIP runtime and must use torch's RCCL), after it, or not at all (tools/comm_probe.py)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "PT_RCCL_LIB"}
    r = subprocess.run([sys.executable, os.path.join(ptlib.ROOT, "tools", "comm_probe.py"), order], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0 and "gather ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_large_call_in_parts_cancel_and_snapshot(gpu):
    """A call of more than 1.5 M pixels is rendered as parts of 2^20 call-local pixels (pt_ctx_render).  Same bits as the
    megakernel (one part); a cancel raised while the second part is in progress leaves part one final, part two
    averaged over its accumulated samples and nothing else; a snapshot taken at that moment shows exactly that."""
    L, ctx = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    set_scene(gpu, sc)
    w, h, spp = 2048, 1040, 3  # 2 129 920 pixels: parts of 1 048 576 + 1 048 576 + 32 768
    npix, part = w * h, 1 << 20
    nbytes = npix * 12
    d_out, d_snap = C.c_void_p(), C.c_void_p()
    assert L.pt_device_malloc(0, nbytes, C.byref(d_out)) == 0 and L.pt_device_malloc(0, nbytes, C.byref(d_snap)) == 0

    def download(ptr):
        host = np.zeros((npix, 3), dtype=np.float32)
        assert L.pt_device_download(0, host.ctypes.data_as(C.c_void_p), ptr, nbytes) == 0
        return host

    st = PtStats()
    cfg = PtConfig(w, h, spp, ptlib.BACKEND_MEGAKERNEL, 4, 0, 0, 0, 0)
    assert L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
    whole_mega, bounces = download(d_out), st.ray_bounces
    cfg = PtConfig(w, h, spp, ptlib.BACKEND_WAVEFRONT, 4, 0, 0, 0, 0)
    assert L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
    assert np.array_equal(download(d_out), whole_mega) and st.ray_bounces == bounces
    # one sample per pass; cancel from the callback once the second part is under way
    flag = (C.c_uint8 * 1)(0)
    snaps = []

    def on_progress(user, frac):
        if frac > 0.5 and not flag[0]:
            n = C.c_uint32()
            assert L.pt_ctx_snapshot(ctx, d_snap, C.byref(n)) == 0, L.pt_last_error()
            snaps.append((n.value, download(d_snap)))
            flag[0] = 1

    cb = ptlib.PROGRESS_FN(on_progress)
    spp2 = 8
    cfg = PtConfig(w, h, spp2, ptlib.BACKEND_WAVEFRONT, 4, 0, 0, part, 0)
    cfg.progress_ms = ptlib.PROGRESS_EVERY_PASS
    rc = L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, C.cast(flag, C.c_void_p), C.cast(cb, C.c_void_p), None, C.byref(st))
    assert rc == ptlib.PT_CANCELLED and len(snaps) == 1
    got = download(d_out)
    done2 = (st.samples - part * spp2) // part  # samples per pixel the second part got
    assert st.samples == part * spp2 + part * done2 and 1 <= done2 < spp2
    full = PtConfig(w, h, spp2, ptlib.BACKEND_WAVEFRONT, 4, 0, part, 0, 0)  # part one as a band of its own
    band = np.zeros((npix, 3), np.float32)
    assert L.pt_render(C.byref(full), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(band), None, None, None,
                       C.byref(st)) == 0
    assert np.array_equal(got[:part], band[:part])
    half = PtConfig(w, h, done2, ptlib.BACKEND_WAVEFRONT, 4, part, 2 * part, 0, 0)
    assert L.pt_render(C.byref(half), C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(band), None, None, None,
                       C.byref(st)) == 0
    assert np.array_equal(got[part:2 * part], band[part:2 * part])
    assert not got[2 * part:].any()
    # the snapshot: part one final, part two over the samples it had then, the rest black
    n_snap, snap = snaps[0]
    assert np.array_equal(snap[:part], got[:part]) and not snap[2 * part:].any() and 1 <= n_snap <= done2
    if n_snap == done2:
        assert np.array_equal(snap[part:2 * part], got[part:2 * part])
    assert L.pt_device_free(0, d_out) == 0 and L.pt_device_free(0, d_snap) == 0


# ---------------------------------------------------------------------------------------------------------------
# round 3: pt_ctx_radiance (radiance(&ray, depth, scene) of one ray on the device), the hand-derived shading KATs through
# it, the exhaustive sqrt / reciprocal sweep, memory-aware pass sizing
import kats_shading

FLAG_SEPARATE_KERNELS = 2
# every device path: (backend, flags, the kernel it must report for scenes without BVH meshes)
DEVICE_PATHS = [(0, 0, b"k_pass_cand"), (0, ptlib.FLAG_NO_BVH, b"k_pass"), (0, FLAG_SEPARATE_KERNELS, b"k_intersect_cand"),
                (0, FLAG_SEPARATE_KERNELS | ptlib.FLAG_NO_BVH, b"k_intersect"),
                (1, 0, None)]


def gpu_radiance(gpu, o, d, depth, n, seed, pixel, backend=0, flags=0):
    L, ctx = gpu
    o = np.ascontiguousarray(o, dtype=np.float32)
    d = np.ascontiguousarray(d, dtype=np.float32)
    out = np.zeros(3, np.float32)
    st = PtStats()
    rc = L.pt_ctx_radiance(ctx, _np_f(o), _np_f(d), depth, n, seed, pixel, backend, flags, _np_f(out), C.byref(st))
    assert rc == 0, L.pt_last_error()
    return out, st


def test_exhaustive_sqrt_and_reciprocal_sweep(gpu):
    """The device's short f_sqrt / f_rcp sequences (csrc/pt_math.h) against the compiler's IEEE expansions: f_sqrt on
    all 2^32 binary32 bit patterns, f_rcp on every normal divisor with 2^-126 <= |d| <= 2^126.  No input may differ."""
    L, ctx = gpu
    out = (C.c_uint64 * 4)()
    assert L.pt_ctx_numerics_sweep(ctx, out) == 0, L.pt_last_error()
    assert out[2] == 1 << 32 and out[3] == 2 * (252 * (1 << 23) + 1), list(out)
    assert out[0] == 0 and out[1] == 0, "f_sqrt mismatches %d, f_rcp mismatches %d" % (out[0], out[1])


@pytest.mark.parametrize("name,build,o,d,depth,want", kats_shading.CASES, ids=[c[0] for c in kats_shading.CASES])
def test_hand_derived_shading_kats_through_the_c_abi(gpu, name, build, o, d, depth, want):
    """tests/kats_shading.py (specular / refract / Fresnel / roulette / MAX_DEPTH worked out by hand from mod.rs:661-792)
    through pt_ctx_radiance on every device path: against the hand value and against the oracle's radiance() on the same
    RNG streams, with equal intersect_scene counts."""
    L, ctx = gpu
    sc = build()
    set_scene(gpu, sc)
    for backend, flags, kernel in DEVICE_PATHS:
        if kernel is not None:
            assert L.pt_ctx_pass_kernel(ctx, flags) == kernel
        tag = (name, backend, flags)
        if want[0] == "exact":
            for n, pixel in ((1, 0), (1, 7), (257, 3)):
                got, st = gpu_radiance(gpu, o, d, depth, n, 11, pixel, backend, flags)
                ref, cnt = ptlib.oracle_radiance(sc, o, d, depth, n, 11, pixel)
                assert st.ray_bounces == cnt.ray_bounces, tag
                assert np.allclose(got, want[1], rtol=2e-6, atol=0), (tag, got, want[1])
                # (the oracle sums its n samples one after the other in f32 as the reference's test does: 257 equal values
                # drift by a few 1e-7 each; the device's fixed-point sum does not)
                assert np.allclose(got, ref, rtol=2e-6 if n == 1 else 2e-5, atol=0), (tag, got, ref)
            continue
        _, a, b, p_a = want
        seen_a = seen_b = 0
        for pixel in range(24):  # one sample each: the oracle says which of the two outcomes
            got, st = gpu_radiance(gpu, o, d, depth, 1, 11, pixel, backend, flags)
            ref, cnt = ptlib.oracle_radiance(sc, o, d, depth, 1, 11, pixel)
            assert st.ray_bounces == cnt.ray_bounces, tag
            is_a = np.array_equal(ref, a)
            assert is_a or np.array_equal(ref, b), tag
            assert np.allclose(got, ref, rtol=2e-6, atol=0), (tag, pixel, got, ref)
            seen_a += is_a
            seen_b += not is_a
        assert seen_a > 0 and seen_b > 0, tag
        n = 20000
        got, st = gpu_radiance(gpu, o, d, depth, n, 11, 5, backend, flags)
        ref, cnt = ptlib.oracle_radiance(sc, o, d, depth, n, 11, 5)
        assert st.ray_bounces == cnt.ray_bounces, tag
        # (20 000 values summed one after the other in f32 by the oracle, as the reference's test does: 1e-5 of drift; the
        # bar is the north star's 1e-4)
        assert np.allclose(got, ref, rtol=1e-4, atol=1e-7), (tag, got, ref)
        mean = p_a * a.astype(np.float64) + (1.0 - p_a) * b.astype(np.float64)
        sd = np.abs(a.astype(np.float64) - b.astype(np.float64)) * np.sqrt(p_a * (1.0 - p_a) / n)
        assert np.all(np.abs(got - mean) <= 5.0 * sd + 1e-6), (tag, got, mean)


def test_reference_radiance_test_run_literally(gpu):
    """src/render/test.rs:146-183 (`test_radiance`) as the reference runs it: radiance(&ray, 0, &scene) of the ray
    (0,0,0) -> (0,0,-1), 10 000 times, the sum divided by their number; it asserts .x > 0.3 (analytic 50/144).  Through
    pt_ctx_radiance on every device path, and against the oracle's loop over the same RNG streams."""
    cam = ptlib.make_camera((0, 0, 0.035), (0, 0, -1))
    sc = ptlib.Scene("t", cam, [ptlib.make_sphere((0, 0, -3), 1.0, (1, 0, 0), (0, 0, 0), "Diffuse"),
                                ptlib.make_sphere((0, 0, 10), 1.0, (0, 0, 0), (50, 50, 50), "Diffuse")], [])
    set_scene(gpu, sc)
    o, d = (0, 0, 0), (0, 0, -1)
    ref, cnt = ptlib.oracle_radiance(sc, o, d, 0, 10000, 1, 0)
    assert ref[0] > 0.3
    for backend, flags, _ in DEVICE_PATHS:
        got, st = gpu_radiance(gpu, o, d, 0, 10000, 1, 0, backend, flags)
        # (a sample is 50 with probability 1/144, else 0: the mean of 10 000 has a standard deviation of 0.042 - the
        # reference's `> 0.3` holds for this seed; three standard deviations around the analytic value hold for any)
        assert got[0] > 0.3 and abs(got[0] - 50.0 / 144.0) < 0.125, got
        assert got[1] == 0.0 and got[2] == 0.0
        assert st.ray_bounces == cnt.ray_bounces
        assert abs(got[0] - ref[0]) <= 1e-4, (got, ref)


def test_radiance_probe_on_the_shipped_scenes(gpu):
    """pt_ctx_radiance against the oracle on cornell.json and mesh.json (diffuse sampling, glass, BVH walks): camera rays
    of a few pixels, 2 000 samples each, both backends."""
    for sid in ("cornell", "mesh"):
        sc = ptlib.load_scene_py(ptlib.scene_path(sid))
        set_scene(gpu, sc)
        O = ptlib.oracle()
        for pixel in (100 * 64 + 17, 300 * 64 + 40):
            ro, rd = np.zeros(3, np.float32), np.zeros(3, np.float32)
            O.pto_primary_ray(C.byref(sc.cam), 64, 48, pixel % (64 * 48), 0, 5, _np_f(ro), _np_f(rd))
            ref, cnt = ptlib.oracle_radiance(sc, ro, rd, 0, 2000, 9, pixel)
            for backend in (0, 1):
                got, st = gpu_radiance(gpu, ro, rd, 0, 2000, 9, pixel, backend, 0)
                assert st.ray_bounces == cnt.ray_bounces, (sid, pixel, backend)
                assert np.allclose(got, ref, rtol=2e-5, atol=1e-6), (sid, pixel, backend, got, ref)


def _render_dev(L, ctx, sc, w, h, spp, seed, flags=0, rays_per_pass=0):
    nbytes = w * h * 12
    d_out = C.c_void_p()
    assert L.pt_device_malloc(0, nbytes, C.byref(d_out)) == 0, L.pt_last_error()
    cfg = PtConfig(w, h, spp, 0, seed, 0, 0, rays_per_pass, flags)
    st = PtStats()
    rc = L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, None, None, None, C.byref(st))
    assert rc == 0, L.pt_last_error()
    host = np.zeros((w * h, 3), dtype=np.float32)
    assert L.pt_device_download(0, host.ctypes.data_as(C.c_void_p), d_out, nbytes) == 0
    assert L.pt_device_free(0, d_out) == 0
    return host, st


def test_small_wave_stacks_hold_primaries_back(gpu):
    """k_pass_cand's waves keep their waiting rays on stacks of 1024 slots, where the bound on what may ever wait (`phi`,
    DESIGN section 3) never holds a wave back.  With stacks of 512 slots (PT_WAVE_STACK) it does all the time - primaries start
    only when nothing waits - and a wave pops fewer than 64 rays whenever its primaries are held back: the other half of the
    loop's decisions.  Same frame bit for bit, same bounce count, no overflow - on cornell.json (glass deferral in the bound)
    and on mesh.json (parked rays in the bound), with streams long enough that their whole share does not fit the stack.
    Two more switches of the same kernel, same bits: glass hits collected per wave and shaded 64 at a time (PT_GLASS_DEFER=1, the
    default until the levels went) and the BVH nodes read through L2 instead of staged in LDS (PT_NODES_LDS=0)."""
    L, _ = gpu
    for sid, (w, h, spp) in (("cornell", (256, 192, 384)), ("mesh", (256, 192, 256))):
        sc = ptlib.load_scene_py(ptlib.scene_path(sid))
        res = {}
        for name, env in (("1024", {}), ("512", {"PT_WAVE_STACK": "512"}), ("glass deferred", {"PT_GLASS_DEFER": "1"}),
                          ("nodes from L2", {"PT_NODES_LDS": "0"})):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                ctx = C.c_void_p()
                assert L.pt_ctx_create(0, C.byref(ctx)) == 0  # the tuning variables are read here
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0, L.pt_last_error()
            assert L.pt_ctx_pass_kernel(ctx, 0).decode().startswith("k_pass_cand")
            img, st = _render_dev(L, ctx, sc, w, h, spp, 9)
            L.pt_ctx_destroy(ctx)
            res[name] = (img, st.ray_bounces)
        for name in res:
            assert res[name][1] == res["1024"][1], (sid, name)
            assert np.array_equal(res[name][0], res["1024"][0]), (sid, name)


def test_megakernel_items_and_rounds_do_not_change_the_image(gpu):
    """k_mega_cand (round 4): two paths per lane one trip apart, primary rays made in dense rounds of the whole wave, items -
    (pixel, part of a round's samples) - handed out from a counter 32 at a time, split children on a stack per lane in
    global memory.  None of that may show: the frame is the wavefront backend's, bit for bit, with the same number of
    intersect_scene evaluations - for the default item size, for items of a handful of samples (PT_MEGA_ITEMS=256: many
    hand-outs, every lane starved of spare rays all the time) and for one item per lane (PT_MEGA_ITEMS=1), with the rounds
    the library chooses and with explicit rounds of a few samples, on cornell.json (glass: splits on the lanes' stacks) and on
    mesh.json (walks inside the trip)."""
    L, _ = gpu
    for sid, (w, h, spp) in (("cornell", (320, 240, 96)), ("mesh", (256, 192, 48))):
        sc = ptlib.load_scene_py(ptlib.scene_path(sid))
        res = {}
        for name, env, backend, rpp in (("wavefront", {}, 0, 0), ("mega", {}, 1, 0), ("tiny items", {"PT_MEGA_ITEMS": "256"}, 1, 0),
                                        ("one item", {"PT_MEGA_ITEMS": "1"}, 1, 0), ("short rounds", {}, 1, w * h * 5)):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                ctx = C.c_void_p()
                assert L.pt_ctx_create(0, C.byref(ctx)) == 0  # the tuning variables are read here
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0, L.pt_last_error()
            cfg = PtConfig(w, h, spp, backend, 13, 0, 0, rpp, 0)
            d = C.c_void_p()
            assert L.pt_device_malloc(0, w * h * 12, C.byref(d)) == 0
            st = PtStats()
            assert L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
            img = np.empty((w * h, 3), np.float32)
            assert L.pt_device_download(0, _np_f(img), d, img.nbytes) == 0
            L.pt_device_free(0, d)
            L.pt_ctx_destroy(ctx)
            res[name] = (img, st.ray_bounces, st.passes)
        assert res["short rounds"][2] >= spp // 5
        for name in ("mega", "tiny items", "one item", "short rounds"):
            assert res[name][1] == res["wavefront"][1], (sid, name)
            assert np.array_equal(res[name][0], res["wavefront"][0]), (sid, name)


def test_memory_budget_changes_the_passes_not_the_image(gpu):
    """pt_ctx_set_memory_budget: 8 MiB is less than the wave stacks of the 64-sample pass of a 128x96 frame take (2048 streams x
    4 waves x 512 slots x 40 B = 168 MB), so the pass is halved down to one sample per pixel and pass (the smallest stacks, 128
    slots per wave) instead of all 64 in one pass.  Same bits, same bounce count."""
    L, ctx = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    set_scene(gpu, sc)
    w, h, spp = 128, 96, 64
    whole, st0 = _render_dev(L, ctx, sc, w, h, spp, 4)
    assert st0.passes == 1
    assert L.pt_ctx_set_memory_budget(ctx, 8 << 20) == 0
    try:
        small, st1 = _render_dev(L, ctx, sc, w, h, spp, 4)
    finally:
        assert L.pt_ctx_set_memory_budget(ctx, 0) == 0
    assert st1.passes == spp and st1.ray_bounces == st0.ray_bounces
    assert np.array_equal(small, whole)
    # the budget also bounds an EXPLICIT rays_per_pass (round 3 applied it to the default size only): asked for the whole
    # frame in one pass, got the passes the budget allows; without a budget the explicit size is taken as given
    assert L.pt_ctx_set_memory_budget(ctx, 8 << 20) == 0
    try:
        small2, st2 = _render_dev(L, ctx, sc, w, h, spp, 4, rays_per_pass=w * h * spp)
        sep, st3 = _render_dev(L, ctx, sc, w, h, spp, 4, flags=2, rays_per_pass=w * h * spp)  # level-by-level form: 352 B per primary
    finally:
        assert L.pt_ctx_set_memory_budget(ctx, 0) == 0
    assert st2.passes == spp and st2.ray_bounces == st0.ray_bounces and np.array_equal(small2, whole)
    assert st3.passes == -(-spp // ((8 << 20) // 352 // (w * h))) and np.array_equal(sep, whole)
    free, st4 = _render_dev(L, ctx, sc, w, h, spp, 4, rays_per_pass=w * h * spp)
    assert st4.passes == 1 and np.array_equal(free, whole)


def test_default_pass_size_is_shared_by_pipelines_and_ranks(gpu):
    """Eight pipelines (PT_FLAG_PIPELINES) and eight ranks of pt_render_multi on this box's ONE GPU with the default
    rays_per_pass on the baseline frame size: each context sizes its passes to an eighth of the default (round 2: 8 x 36 GB
    of queues, PT_ERR_HIP on a 288 GB device).  Same bits as the single pipeline."""
    L, ctx = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    set_scene(gpu, sc)
    w, h, spp = 1024, 768, 256
    one, st = _render_dev(L, ctx, sc, w, h, spp, 2)
    eight, st8 = _render_dev(L, ctx, sc, w, h, spp, 2, flags=8 << 8)
    assert st8.ray_bounces == st.ray_bounces and np.array_equal(eight, one)
    cfg = PtConfig(w, h, spp, 0, 2, 0, 0, 0, 0)
    out = np.zeros((w * h, 3), dtype=np.float32)
    s2 = PtStats()
    seen = []
    cb = ptlib.PROGRESS_FN(lambda user, frac: seen.append(frac))
    rc = L.pt_render_multi(C.byref(cfg), 8, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris, _np_f(out), None,
                           C.cast(cb, C.c_void_p), None, C.byref(s2))
    assert rc == 0, L.pt_last_error()
    assert s2.ray_bounces == st.ray_bounces and np.array_equal(out, one)
    assert seen and seen[-1] == 1.0 and seen.count(1.0) == 1  # the completion is reported once, by the caller's thread, at the end


def test_out_of_device_memory_halves_the_pass(gpu):
    """A pass whose ray queues do not fit what is left of the device is halved until it does (DevBuf::ensure reports the
    failed allocation, render_wavefront retries): nearly all of the HBM is taken by other allocations first (the refused one
    among them must not come back as the frame's error), then a frame is asked for in ONE pass of 2048 samples - 64 Ki streams
    with 10.7 GB of wave stacks where 2 to 4 GiB are free.  Same bits as the frame rendered at leisure."""
    L, _ = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    ctx = C.c_void_p()
    assert L.pt_ctx_create(0, C.byref(ctx)) == 0, L.pt_last_error()
    assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0
    w, h, spp = 1024, 768, 2048
    want, st0 = _render_dev(L, ctx, sc, w, h, spp, 6, rays_per_pass=8 << 20)
    hogs = []
    try:
        while len(hogs) < 160:  # 2 GiB at a time until the device refuses
            p = C.c_void_p()
            if L.pt_device_malloc(0, 2 << 30, C.byref(p)) != 0:
                break
            hogs.append(p)
        assert len(hogs) >= 32, "could not even take 64 GiB"
        L.pt_device_free(0, hogs.pop())  # leave 2 GiB (+ the remainder, less than 2 GiB) free
        got, st1 = _render_dev(L, ctx, sc, w, h, spp, 6, rays_per_pass=1 << 31)
    finally:
        for p in hogs:
            L.pt_device_free(0, p)
        L.pt_ctx_destroy(ctx)
    assert st1.passes > 1, st1.passes  # 2 Gi rays would have been one pass
    assert st1.ray_bounces == st0.ray_bounces and np.array_equal(got, want)


def _oracle_ids(sc, oid, tid):
    """(object, triangle-in-object) of the oracle -> the hit id of the stream kernels: object index for a sphere, n_objs +
    flattened triangle index for a triangle, -1 for a miss."""
    off = np.array([sc.objs[i].tri_offset for i in range(sc.n_objs)], np.int64)
    ids = np.where(oid < 0, -1, np.where(tid < 0, oid, sc.n_objs + off[np.maximum(oid, 0)] + tid))
    return ids.astype(np.int32)


@pytest.mark.parametrize("sid", ["cornell", "mesh", "cartesian"])
def test_stream_intersect_kernels_ray_by_ray(gpu, sid):
    """The intersect step of the wavefront pipeline ITSELF, ray by ray against intersect_scene of the oracle (bit-exact
    distance, same primitive): the candidate scan (k_intersect_cand: filters, ring, dense exact batches - the scan k_pass_cand
    runs), the every-triangle scan (PT_FLAG_NO_BVH) and, for mesh.json, scan + parked walks.  Rays: every ray the path
    tracer casts for a block of pixels - four fifths of them start ON a triangle, where the reference's epsilon-free
    Triangle::intersect decides a self-hit by rounding and the flat filter's exact sign rule (filter_flat) must agree - and
    the same rays with their origin moved by -2, -1, +1, +2 ulp along each axis: origins on either side of their wall."""
    L, ctx = gpu
    O = ptlib.oracle()
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    set_scene(gpu, sc)
    w, h, spp = 48, 32, 2 if sid == "mesh" else 6
    cfg = PtoConfig(w, h, spp, 0, 23)
    cap = w * h * spp * 16
    rays = np.zeros((cap, 6), dtype=np.float32)
    ps = sc.pto()
    n = O.pto_dump_rays(C.byref(ps), C.byref(cfg), 0, w * h, _np_f(rays), cap)
    assert 0 < n < cap
    o0 = np.ascontiguousarray(rays[:n, :3])
    d0 = np.ascontiguousarray(rays[:n, 3:])
    os_, ds_ = [o0], [d0]
    sub = slice(0, min(n, 6000))
    for axis in range(3):
        for k in (-2, -1, 1, 2):
            o = o0[sub].copy()
            col = o[:, axis]
            for _ in range(abs(k)):
                col = np.nextafter(col, np.float32(np.inf if k > 0 else -np.inf)).astype(np.float32)
            o[:, axis] = col
            os_.append(o)
            ds_.append(d0[sub])
    o = np.ascontiguousarray(np.concatenate(os_))
    d = np.ascontiguousarray(np.concatenate(ds_))
    m = len(o)
    t0, oid0, tid0, _, _ = ptlib.oracle_intersect(sc, o, d)
    want_id = _oracle_ids(sc, oid0, tid0)
    hit = want_id >= 0
    assert hit.sum() > (0.8 * m if sid != "cartesian" else 0)  # (cartesian.json: a few spheres in empty space)
    for flags in (0, ptlib.FLAG_NO_BVH):
        t = np.zeros(m, np.float32)
        ids = np.zeros(m, np.int32)
        rc = L.pt_ctx_intersect_streams(ctx, _np_f(o), _np_f(d), m, flags, _np_f(t), ids.ctypes.data_as(ptlib.i32p))
        assert rc == 0, L.pt_last_error()
        assert np.array_equal(ids, want_id), (sid, flags, int((ids != want_id).sum()))
        assert np.array_equal(t[hit].view(np.uint32), t0[hit].view(np.uint32)), (sid, flags)
        assert np.all(np.isinf(t[~hit]))
    # the self-hits the reference's arithmetic really produces are in the sample: rays that hit the triangle they start on
    if sid == "cornell":
        assert (t0[hit] < 1e-5).sum() > 0


def test_progress_at_part_boundaries(gpu):
    """A call rendered in parts (more than 1.5 M pixels) whose parts take one pass each makes no callback from inside a
    part (those start at the third pass); the part boundaries are progress points of their own: 1/3, 2/3, then 1.0."""
    L, ctx = gpu
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    set_scene(gpu, sc)
    w, h = 2048, 1040  # parts of 1 048 576 + 1 048 576 + 32 768 pixels
    d_out = C.c_void_p()
    assert L.pt_device_malloc(0, w * h * 12, C.byref(d_out)) == 0
    seen = []
    cb = ptlib.PROGRESS_FN(lambda user, frac: seen.append(frac))
    cfg = PtConfig(w, h, 1, 0, 3, 0, 0, 0, 0)
    cfg.progress_ms = ptlib.PROGRESS_EVERY_PASS
    st = PtStats()
    rc = L.pt_ctx_render(ctx, C.byref(cfg), d_out, None, None, C.cast(cb, C.c_void_p), None, C.byref(st))
    assert rc == 0, L.pt_last_error()
    assert st.passes == 3
    assert len(seen) == 3 and abs(seen[0] - 1 / 3) < 1e-6 and abs(seen[1] - 2 / 3) < 1e-6 and seen[2] == 1.0, seen
    assert L.pt_device_free(0, d_out) == 0
