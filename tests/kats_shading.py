"""Hand-derived known-answer tests for the SHADING half of radiance() (src/render/mod.rs:661-792) - the part the
reference's own tests reach only through one statistical bound (test.rs:146-183): the specular direction (mod.rs:722-723),
`into` / `nnt` / `cos2t` / `tdir`, Re = R0 + (1-R0) c^5, P, RP, TP (mod.rs:736-758), both subtrees at new_depth <= 2
(mod.rs:775-786), the choice at new_depth > 2 (mod.rs:760-774), total internal reflection (mod.rs:743-744), the
roulette with its rescale (mod.rs:676-683) and MAX_DEPTH (mod.rs:661,678).

Every scene is built so that hit points, normals and directions are EXACT in binary32 (axis-aligned rays or rays with
dyadic components, triangles with power-of-two edges, unit spheres hit along an axis); emitters have colour 0, so
whatever a path does after it reached one is multiplied by zero (`emission + color * radiance(..)`, mod.rs:685-686) and
the value of radiance() is a closed expression of the materials.  The cases run on the oracle (tests/test_oracle.py: the
oracle's f32 result must EQUAL the expression) and through pt_ctx_radiance on every device path (tests/test_gpu_parity.py:
the device sums throughput x emission top-down in fixed point, so it agrees to f32 round-off, 1e-6).

Worked example ("glass_normal_incidence_split"): triangle A=(-4,-4,0) B=(4,-4,0) C=(-4,4,0), ray o=(-2,-2,3) d=(0,0,-1).
  Triangle::intersect: va_vb=(8,0,0), va_vc=(0,8,0); pvec = d x va_vc = (8,0,0); det = 64; tvec = (2,2,3); u = 16/64 = 0.25;
  qvec = tvec x va_vb = (0,24,-16); v = 16/64 = 0.25; distance = 192/64 = 3; x = (-2,-2,0); normal = (0,0,64)/64 = (0,0,1).
  radiance, Refract arm: normal.d = -1 < 0 -> normal_towards_ray = normal; refl = d - normal*2*(normal.d) = (0,0,1);
  into = true, nnt = 1/1.5, ddn = -1, cos2t = 1 - nnt^2 * (1 - 1) = 1;
  tdir = normalize(d*nnt - normal*(ddn*nnt + 1)) = normalize((0,0,-nnt) - (0,0,1 - nnt)) = (0,0,-1)   [1 - nnt and the sum are exact]
  r0 = 0.5*0.5 / (2.5*2.5) = fl(0.04); c = 1 - (-ddn) = 0 -> re = r0 + (1-r0)*0 = r0; tr = 1 - r0.
  new_depth = 1 <= 2: color * (radiance(refl)*re + radiance(trans)*tr).  refl from (-2,-2,0) along +z meets emitter A
  (sphere (-2,-2,10) r=2: op=(0,0,10), b=10, det=4, t=8), trans along -z meets emitter B; the glass triangle itself gives
  distance = +-0 from x, which `distance <= 0.0` rejects (mod.rs:592).  Emitters have colour 0: radiance = emission.
  => radiance = color * (eA*re + eB*tr), evaluated in f32 in that order.
"""
import numpy as np

import ptlib
from ptlib import make_camera, make_mesh, make_sphere, make_tri, Scene

f32 = np.float32
CAM = dict(position=(0, 0, 50), direction=(0, 0, -1))
# Bounding spheres are given explicitly and generously: Mesh::new's own (`min + max*0.5` centre, mod.rs:488) need not
# contain its mesh (tests/kats.py), and the gate is not what these cases are about.
BOUNDS = ((0.0, 0.0, 0.0), 1000.0)
BLACK = (0.0, 0.0, 0.0)
GLASS_TRI = ((-4, -4, 0), (4, -4, 0), (-4, 4, 0))  # va_vb = (8,0,0), va_vc = (0,8,0): det and 1/det are powers of two


def scene(meshes=(), spheres=()):
    """meshes: (triangle, color, emission, reflect); spheres: (position, radius, color, emission, reflect)."""
    objs, tris = [], []
    for tri, color, emission, reflect in meshes:
        objs.append(make_mesh((0, 0, 0), color, emission, reflect, len(tris), 1, BOUNDS[0], BOUNDS[1]))
        tris.append(make_tri(*tri))
    for pos, rad, color, emission, reflect in spheres:
        objs.append(make_sphere(pos, rad, color, emission, reflect))
    return Scene("kat-shading", make_camera(**CAM), objs, tris)


def v3(*a):
    return np.array(a, dtype=f32)


# ---- Fresnel terms at normal incidence, in the reference's f32 operation order (mod.rs:750-758) ---------------------------
def fresnel_normal():
    nc, nt = f32(1.0), f32(1.5)
    a, b = nt - nc, nt + nc
    r0 = (a * a) / (b * b)
    c = f32(1.0) - f32(1.0)  # 1 - (-ddn), ddn = -1
    re = r0 + (f32(1.0) - r0) * (c * ((c * c) * (c * c)))
    tr = f32(1.0) - re
    p = f32(0.25) + f32(0.5) * re
    return re, tr, p, re / p, tr / (f32(1.0) - p)


RE, TR, P, RP, TP = fresnel_normal()
assert RE == f32(0.04) and P == f32(0.27)

GLASS_COLOR = v3(1.0, 0.5, 0.25)
EA, EB = v3(8.0, 4.0, 2.0), v3(1.0, 2.0, 4.0)


def glass_scene():
    return scene(meshes=[(GLASS_TRI, tuple(GLASS_COLOR), BLACK, "Refract")],
                 spheres=[((-2, -2, 10), 2.0, BLACK, tuple(EA), "Diffuse"), ((-2, -2, -10), 2.0, BLACK, tuple(EB), "Diffuse")])


# ---- the zigzag between two parallel mirrors ------------------------------------------------------------------------------
# ray o=(-3,-14,1), d=(0.75,0,-0.5): z=0 at t=2 (x=-1.5), then every 4 units of t the other plane, x advancing by 3:
# (-1.5, 0) (1.5, 2) (4.5, 0) (7.5, 2) (10.5, 0) (13.5, 2) (16.5, 0) ...  With va_vb=(32,0,0), va_vc=(0,32,0) (or 64) every
# determinant is a power of two, so distances 2 and 4 and the hit points are exact; reflecting off a plane of normal (0,0,1)
# flips the sign of d.z exactly (d - n*2*(n.d) = d - (0,0,2)*(-+0.5)).  The sphere an emitter needs along such a ray, given
# that intersect_sphere (mod.rs:412-427) assumes a unit direction and these have |d|^2 = 0.8125: centre 8 d ahead, r = 4 ->
# b = 6.5, det = 42.25 - 52 + 16 = 6.25, t = 6.5 - 2.5 = 4.
def tri_at(ax, ay, z, edge):
    return ((ax, ay, z), (ax + edge, ay, z), (ax, ay + edge, z))


def zigzag_scene(mirror_color, n_mirror_hits, emission):
    bottom = tri_at(-16, -16, 0, 32)  # holds x = -1.5, 4.5, 10.5 at y = -14 (u + v = 0.52, 0.70, 0.89), not 16.5 (1.08)
    if n_mirror_hits == 5:
        top = tri_at(-20, -16, 2, 32)  # holds x = 1.5, 7.5 (0.73, 0.92), not 13.5 (1.11): the sixth ray leaves upwards
        last, d_last = (10.5, -14.0, 0.0), (0.75, 0.0, 0.5)
    else:
        assert n_mirror_hits == 6
        top = tri_at(-20, -16, 2, 64)  # holds x = 13.5 as well (0.55); the seventh ray leaves downwards past x = 16.5
        last, d_last = (13.5, -14.0, 2.0), (0.75, 0.0, -0.5)
    centre = tuple(float(last[k] + 8.0 * d_last[k]) for k in range(3))
    return scene(meshes=[(bottom, mirror_color, BLACK, "Specular"), (top, mirror_color, BLACK, "Specular")],
                 spheres=[(centre, 4.0, BLACK, emission, "Diffuse")])


ZIG_O, ZIG_D = (-3.0, -14.0, 1.0), (0.75, 0.0, -0.5)

# name, scene builder, o, d, depth, expectation
#   ("exact", rgb)                     every sample returns rgb: the mean over any n is rgb
#   ("choice", rgb_a, rgb_b, p_a)      a sample returns rgb_a with probability p_a, else rgb_b (one random choice per sample)
CASES = [
    # mod.rs:716-728.  Unit mirror sphere at the origin hit along -z at (0,0,1) (b = 5, det = 1, t = 4; normal (0,0,1)):
    # d - n*2*(n.d) = (0,0,1); from (0,0,1) the mirror itself gives t0 = -2, t1 = 0 (both < 1e-4), the emitter (0,0,10) r=2
    # gives b = 9, det = 4, t = 7.  radiance = 0 + color * emission.
    ("mirror_facing_emitter",
     lambda: scene(spheres=[((0, 0, 0), 1.0, (0.5, 0.25, 1.0), BLACK, "Specular"), ((0, 0, 10), 2.0, BLACK, (12.0, 6.0, 3.0), "Diffuse")]),
     (0, 0, 5), (0, 0, -1), 0, ("exact", v3(6.0, 1.5, 3.0))),
    # mod.rs:729-786, new_depth = 1: both subtrees (module docstring)
    ("glass_normal_incidence_split", glass_scene, (-2, -2, 3), (0, 0, -1), 0,
     ("exact", GLASS_COLOR * (EA * RE + EB * TR))),
    # mod.rs:760-774, new_depth = 3: `rand01() < p` picks the reflected ray with weight rp, else the transmitted one with tp
    ("glass_normal_incidence_choice", glass_scene, (-2, -2, 3), (0, 0, -1), 2,
     ("choice", GLASS_COLOR * EA * RP, GLASS_COLOR * EB * TP, float(P))),
    # mod.rs:741-744.  The same triangle met from below by d = (0.75,0,0.5) (o = (-5,-2,-2): pvec = (-4,0,6), det = -32, u = v = 0.25,
    # distance = 4, x = (-2,-2,0)): normal.d = 0.5 > 0 -> normal_towards_ray = -normal, into = false, nnt = 1.5, ddn = -0.5,
    # cos2t = 1 - 2.25*(1 - 0.25) = -0.6875 < 0: `color * radiance(refl_ray)`, refl = d - (0,0,2)*0.5 = (0.75,0,-0.5), which meets
    # the emitter 8 d ahead (see above) at t = 4.  (The incoming ray misses that sphere: b = 5.75, det = -35.9.)
    ("glass_total_internal_reflection",
     lambda: scene(meshes=[(GLASS_TRI, tuple(GLASS_COLOR), BLACK, "Refract")],
                   spheres=[((4, -2, -4), 4.0, BLACK, (3.0, 5.0, 7.0), "Diffuse")]),
     (-5, -2, -2), (0.75, 0, 0.5), 0, ("exact", GLASS_COLOR * v3(3.0, 5.0, 7.0))),
    # mod.rs:676-683 with max_reflection = 0: `rand01() < 0.0` is false, the emission is returned
    ("emitter_behind_the_roulette",
     lambda: scene(spheres=[((0, 0, -5), 1.0, BLACK, (7.0, 11.0, 13.0), "Diffuse")]),
     (0, 0, 0), (0, 0, -1), 5, ("exact", v3(7.0, 11.0, 13.0))),
    # mod.rs:678: new_depth = 12 is not < MAX_DEPTH, whatever the draw: the emission is returned although the colour is 0.9
    ("max_depth_returns_emission",
     lambda: scene(spheres=[((0, 0, -5), 1.0, (0.9, 0.9, 0.9), (7.0, 11.0, 13.0), "Diffuse")]),
     (0, 0, 0), (0, 0, -1), 11, ("exact", v3(7.0, 11.0, 13.0))),
    # mod.rs:678-679: new_depth = 6, max_reflection = 0.5: with probability 0.5 the path goes on with color * (1/0.5) =
    # (1, 0.5, 0.5), else the mirror's emission (0) is returned.  Beyond: mirror_facing_emitter's geometry.
    ("roulette_rescale",
     lambda: scene(spheres=[((0, 0, 0), 1.0, (0.5, 0.25, 0.25), BLACK, "Specular"), ((0, 0, 10), 2.0, BLACK, (8.0, 16.0, 32.0), "Diffuse")]),
     (0, 0, 5), (0, 0, -1), 5, ("choice", v3(8.0, 8.0, 16.0), v3(0.0, 0.0, 0.0), 0.5)),
    # five specular bounces (new_depth 1..5: no roulette), then the emitter at new_depth = 6 with max_reflection = 0:
    # 0.5^5 * emission
    ("mirror_zigzag_then_emitter", lambda: zigzag_scene((0.5, 0.5, 0.5), 5, (32.0, 64.0, 96.0)), ZIG_O, ZIG_D, 0,
     ("exact", v3(1.0, 2.0, 3.0))),
    # six specular bounces: the sixth at new_depth = 6 survives with probability max_reflection = 0.5 and weight 1/0.5:
    # throughput (0.5, 0.25, 0.25)^5 * (1, 0.5, 0.5) = (1/32, 1/2048, 1/2048)
    ("mirror_zigzag_roulette", lambda: zigzag_scene((0.5, 0.25, 0.25), 6, (64.0, 4096.0, 8192.0)), ZIG_O, ZIG_D, 0,
     ("choice", v3(2.0, 2.0, 4.0), v3(0.0, 0.0, 0.0), 0.5)),
]
