"""Hand-derived known-answer tests for the parts of the path the reference's own tests do not reach:
Triangle::intersect (src/render/mod.rs:554-615), the bounding-sphere gate (mod.rs:261-280) and the tie rules of
intersect_scene (mod.rs:631-659).  Every expectation below is worked out by hand from the reference's lines on
numbers that are exact in binary32 (so the expected t / x / n are exact, not approximate).  The same cases run on the
oracle (tests/test_oracle.py, CPU) and through pt_ctx_intersect on the HIP path (tests/test_gpu_parity.py, GPU).

Worked example (case "centre"): triangle A=(0,0,0) B=(1,0,0) C=(0,1,0), ray o=(0.25,0.25,1) d=(0,0,-1).
  va_vb = (1,0,0), va_vc = (0,1,0)                                            mod.rs:560-561
  pvec = d x va_vc = (0*0 - 1*(-1), (-1)*0 - 0*0, 0*1 - 0*0) = (1, 0, 0)       mod.rs:563 (glam cross order)
  determinant = va_vb . pvec = 1        -> |1| >= 1e-4, kept                   mod.rs:564,571
  tvec = o - A = (0.25, 0.25, 1);  u = tvec . pvec * 1 = 0.25                  mod.rs:577-578
  qvec = tvec x va_vb = (0.25*0 - 0*1, 1*1 - 0*0.25, 0.25*0 - 1*0.25) = (0, 1, -0.25)   mod.rs:583
  v = d . qvec = 0.25;  u + v = 0.5 <= 1                                        mod.rs:584-585
  distance = va_vc . qvec = 1 > 0                                               mod.rs:589-592
  intersection = o + d*1 = (0.25, 0.25, 0);  normal = normalize((1,0,0) x (0,1,0)) = (0, 0, 1)   mod.rs:604-605
"""
import ptlib
from ptlib import make_camera, make_mesh, make_sphere, make_tri, Scene

MAT = dict(color=(1.0, 0.0, 0.0), emission=(0.0, 0.0, 0.0), reflect="Diffuse")
CAM = dict(position=(0, 0, 5), direction=(0, 0, -1))
UNIT = [((0, 0, 0), (1, 0, 0), (0, 1, 0))]
DOWN = (0.0, 0.0, -1.0)
MISS = None


def mesh_scene(tri_lists, positions=None, spheres=(), bounds=None):
    """One mesh object per triangle list (bounding sphere as Mesh::new computes it unless `bounds[i]` gives
    (centre, radius)), then the given spheres (position, radius)."""
    objs, tris = [], []
    for i, tl in enumerate(tri_lists):
        tlist = [make_tri(*t) for t in tl]
        if bounds and bounds[i] is not None:
            ctr, rad = bounds[i]
        else:
            import ctypes as C
            arr = (ptlib.PtTriangle * len(tlist))(*tlist)
            c, r = (C.c_float * 3)(), C.c_float()
            ptlib.oracle().pto_mesh_bounding_sphere(arr, len(tlist), c, C.byref(r))
            ctr, rad = list(c), r.value
        pos = positions[i] if positions else (0, 0, 0)
        objs.append(make_mesh(pos, MAT["color"], MAT["emission"], MAT["reflect"], len(tris), len(tlist), ctr, rad))
        tris.extend(tlist)
    for pos, rad in spheres:
        objs.append(make_sphere(pos, rad, **MAT))
    return Scene("kat", make_camera(**CAM), objs, tris)


def hit(obj, tri, t, x, n):
    return dict(object_id=obj, tri_id=tri, t=t, x=x, n=n)


# name, scene builder, ray origin, ray direction, expected (None = intersect_scene returns None)
CASES = [
    # -- Triangle::intersect on the unit triangle ------------------------------------------------------------------
    ("centre", lambda: mesh_scene([UNIT]), (0.25, 0.25, 1.0), DOWN, hit(0, 0, 1.0, (0.25, 0.25, 0.0), (0, 0, 1))),
    # u + v == 1 exactly: `(u + v) > 1.0` is false, the hypotenuse belongs to the triangle (mod.rs:585)
    ("edge_u_plus_v_is_1", lambda: mesh_scene([UNIT]), (0.5, 0.5, 1.0), DOWN, hit(0, 0, 1.0, (0.5, 0.5, 0.0), (0, 0, 1))),
    # u == 0, v == 0: `u < 0.0` false (mod.rs:579); u == 1: `u > 1.0` false.  (Bounding sphere given explicitly, see next.)
    ("vertex_a", lambda: mesh_scene([UNIT], bounds=[((0.5, 0.5, 0.0), 2.0)]), (0.0, 0.0, 1.0), DOWN,
     hit(0, 0, 1.0, (0.0, 0.0, 0.0), (0, 0, 1))),
    ("vertex_b", lambda: mesh_scene([UNIT], bounds=[((0.5, 0.5, 0.0), 2.0)]), (1.0, 0.0, 1.0), DOWN,
     hit(0, 0, 1.0, (1.0, 0.0, 0.0), (0, 0, 1))),
    # The same two rays against the sphere Mesh::new computes (mod.rs:478-492): centre (0.5,0.5,0), radius
    # sqrt(0.5) -> 0.70710677 in f32, whose square 0.49999997 is BELOW 0.5.  The vertices lie exactly on that sphere:
    # det = b*b - op.op + r*r = (1 - 1.5) + 0.49999997 = -2.98e-8 < 0 (mod.rs:416-417) - the gate fails by rounding and
    # the reference reports no hit although Moller-Trumbore accepts the ray ("a ray grazing it can fail
    # the gate").  Kept as is.
    ("vertex_a_fails_mesh_new_gate_by_rounding", lambda: mesh_scene([UNIT]), (0.0, 0.0, 1.0), DOWN, MISS),
    ("vertex_b_fails_mesh_new_gate_by_rounding", lambda: mesh_scene([UNIT]), (1.0, 0.0, 1.0), DOWN, MISS),
    # just outside the hypotenuse: u + v = 0.5 + 0.625 > 1
    ("outside", lambda: mesh_scene([UNIT]), (0.5, 0.625, 1.0), DOWN, MISS),
    # u < 0
    ("outside_u_negative", lambda: mesh_scene([UNIT]), (-0.125, 0.25, 1.0), DOWN, MISS),
    # origin on the triangle's plane: distance = va_vc . qvec = -0.0, `distance <= 0.0` rejects (mod.rs:592), no epsilon
    ("self_hit_t_is_0", lambda: mesh_scene([UNIT]), (0.25, 0.25, 0.0), DOWN, MISS),
    # triangle behind the origin: distance = -1
    ("behind", lambda: mesh_scene([UNIT]), (0.25, 0.25, -1.0), DOWN, MISS),
    # from below: USE_CULLING = false (mod.rs:27,566), determinant = -1, the geometric normal is NOT flipped (mod.rs:605)
    ("back_face", lambda: mesh_scene([UNIT]), (0.25, 0.25, -1.0), (0.0, 0.0, 1.0), hit(0, 0, 1.0, (0.25, 0.25, 0.0), (0, 0, 1))),
    # object position is added to every vertex first (Triangle::transformed, mod.rs:546-552): exact with these numbers
    ("translated", lambda: mesh_scene([UNIT], positions=[(2.0, -4.0, -8.0)]), (2.25, -3.75, 0.0), DOWN,
     hit(0, 0, 8.0, (2.25, -3.75, -8.0), (0, 0, 1))),
    # -- |determinant| < 1e-4 is an absolute threshold on an un-normalised determinant (mod.rs:571) -------------------
    # legs 2^-7: determinant = 2^-14 = 6.1e-5 < 1e-4 -> skipped although the ray passes through the triangle
    ("tiny_triangle_vanishes", lambda: mesh_scene([[((0, 0, 0), (0.0078125, 0, 0), (0, 0.0078125, 0))]]),
     (0.001953125, 0.001953125, 1.0), DOWN, MISS),
    # legs 2^-6: determinant = 2^-12 = 2.4e-4 >= 1e-4 -> hit; u = v = 2^-9 / 2^-6 = 0.125
    ("small_triangle_stays", lambda: mesh_scene([[((0, 0, 0), (0.015625, 0, 0), (0, 0.015625, 0))]]),
     (0.001953125, 0.001953125, 1.0), DOWN, hit(0, 0, 1.0, (0.001953125, 0.001953125, 0.0), (0, 0, 1))),
    # -- ties -----------------------------------------------------------------------------------------------------
    # the same triangle twice in one list: `distance < closest` is strict, the first in the list wins (mod.rs:598)
    ("equal_distance_first_triangle_wins", lambda: mesh_scene([UNIT + UNIT]), (0.25, 0.25, 1.0), DOWN,
     hit(0, 0, 1.0, (0.25, 0.25, 0.0), (0, 0, 1))),
    # a nearer triangle later in the list still wins
    ("nearer_later_triangle_wins", lambda: mesh_scene([UNIT + [((0, 0, 0.5), (1, 0, 0.5), (0, 1, 0.5))]]),
     (0.25, 0.25, 1.0), DOWN, hit(0, 1, 0.5, (0.25, 0.25, 0.5), (0, 0, 1))),
    # the same mesh as two objects: objects are visited from the last to the first with strict `<`, the HIGHER index
    # keeps an equal distance (mod.rs:637,649)
    ("equal_distance_higher_object_wins", lambda: mesh_scene([UNIT, UNIT]), (0.25, 0.25, 1.0), DOWN,
     hit(1, 0, 1.0, (0.25, 0.25, 0.0), (0, 0, 1))),
    # two identical spheres: same rule.  intersect_sphere (mod.rs:412-438): op = (0,0,-3), b = 3, det = 9 - 9 + 1 = 1,
    # t = b - 1 = 2, x = (0,0,1), n = (0,0,1)
    ("equal_distance_higher_sphere_wins", lambda: mesh_scene([], spheres=[((0, 0, 0), 1.0), ((0, 0, 0), 1.0)]),
     (0.0, 0.0, 3.0), DOWN, hit(1, -1, 2.0, (0.0, 0.0, 1.0), (0, 0, 1))),
    # a sphere (object 1) touching distance 1 exactly like the triangle (object 0): higher index wins -> the sphere.
    # sphere centre (0.25,0.25,-1) r 1: op = (0,0,-2), b = 2, det = 4 - 4 + 1, t = 1, x = (0.25,0.25,0), n = (0,0,1)
    ("triangle_and_sphere_at_equal_distance", lambda: mesh_scene([UNIT], spheres=[((0.25, 0.25, -1.0), 1.0)]),
     (0.25, 0.25, 1.0), DOWN, hit(1, -1, 1.0, (0.25, 0.25, 0.0), (0, 0, 1))),
    # -- the bounding-sphere gate (mod.rs:267-273) ------------------------------------------------------------------
    # a stored bounding sphere that does not contain the triangle (inline Mesh objects keep theirs verbatim, mod.rs:315):
    # the gate fails and the triangle list is never looked at, although Moller-Trumbore would accept the ray
    ("gate_rejects_what_moller_trumbore_accepts", lambda: mesh_scene([UNIT], bounds=[((10.0, 10.0, 10.0), 1.0)]),
     (0.25, 0.25, 1.0), DOWN, MISS),
    # origin inside the bounding sphere: the near root is negative, the far root b + sqrt(det) >= 1e-4 passes the gate
    ("gate_passes_through_far_root", lambda: mesh_scene([UNIT], bounds=[((0.25, 0.25, 1.0), 4.0)]),
     (0.25, 0.25, 1.0), DOWN, hit(0, 0, 1.0, (0.25, 0.25, 0.0), (0, 0, 1))),
    # sphere entirely behind the origin (both roots < 1e-4): gate fails even though the triangle is ahead
    ("gate_behind_origin", lambda: mesh_scene([UNIT], bounds=[((0.25, 0.25, 3.0), 1.0)]), (0.25, 0.25, 1.0), DOWN, MISS),
    # Mesh::new's centre is min + max*0.5 (mod.rs:478-482): for this triangle (2,2,0)-(3,2,0)-(2,3,0) the sphere is
    # centred at (3.5,3.5,0) with radius |min - c| = 1.5*sqrt(2); the ray below still passes it and hits: u = v = 0.25
    ("mesh_new_sphere_centre_is_min_plus_half_max", lambda: mesh_scene([[((2, 2, 0), (3, 2, 0), (2, 3, 0))]]),
     (2.25, 2.25, 1.0), DOWN, hit(0, 0, 1.0, (2.25, 2.25, 0.0), (0, 0, 1))),
]
