// pt_host.cpp — once-per-frame host arithmetic feeding the kernels.  Built with -ffp-contract=off: the
// values produced here must be the f32 values the reference computes (per frame for the camera, per ray
// for the triangle edges — see pt_device.h).
#include "pt_host.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace pt {
namespace host {

static vec3 ld(const float *p) { return mk(p[0], p[1], p[2]); }
static void st(float *p, vec3 v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}

void camera_basis(const pt_camera &cam, float lens_center[3], float su_out[3], float sv_out[3]) {
    const vec3 position = ld(cam.position), direction = ld(cam.direction);
    const float sensor_height = cam.sensor_width / cam.aspect_ratio;  // mod.rs:211-213
    const vec3 lens = position + direction * cam.focal_length;        // mod.rs:216-218
    const vec3 helper = f_abs(direction.y) < 0.9f ? mk(0.0f, 1.0f, 0.0f) : mk(0.0f, 0.0f, 1.0f);
    const vec3 su = normalize(cross(direction, helper));  // mod.rs:223-229
    const vec3 sv = cross(su, direction);                 // mod.rs:230
    st(lens_center, lens);
    st(su_out, su * cam.sensor_width);  // mod.rs:231
    st(sv_out, sv * sensor_height);
}

void mesh_bounding_sphere(const pt_triangle *tris, uint32_t n, float center[3], float *radius) {
    const float inf = std::numeric_limits<float>::infinity();
    vec3 lo = mk(inf, inf, inf), hi = mk(-inf, -inf, -inf);
    for (uint32_t i = 0; i < n; ++i) {
        const float *corners[3] = {tris[i].a, tris[i].b, tris[i].c};
        for (const float *v : corners) {
            lo.x = v[0] < lo.x ? v[0] : lo.x;
            lo.y = v[1] < lo.y ? v[1] : lo.y;
            lo.z = v[2] < lo.z ? v[2] : lo.z;
            hi.x = v[0] > hi.x ? v[0] : hi.x;
            hi.y = v[1] > hi.y ? v[1] : hi.y;
            hi.z = v[2] > hi.z ? v[2] : hi.z;
        }
    }
    // the reference's centre is min + max*0.5 (mod.rs:478-482), kept as is
    const vec3 c = mk(lo.x + hi.x * 0.5f, lo.y + hi.y * 0.5f, lo.z + hi.z * 0.5f);
    const float to_lo = length(lo - c), to_hi = length(hi - c);
    st(center, c);
    *radius = to_lo > to_hi ? to_lo : to_hi;  // max_by keeps the last of equal maxima
}

void mesh_bounding_box(const pt_triangle *tris, uint32_t n, pt_triangle out[12]) {
    const float inf = std::numeric_limits<float>::infinity();
    vec3 lo = mk(inf, inf, inf), hi = mk(-inf, -inf, -inf);
    for (uint32_t i = 0; i < n; ++i) {
        const float *corners[3] = {tris[i].a, tris[i].b, tris[i].c};
        for (const float *v : corners) {
            lo.x = v[0] < lo.x ? v[0] : lo.x;
            lo.y = v[1] < lo.y ? v[1] : lo.y;
            lo.z = v[2] < lo.z ? v[2] : lo.z;
            hi.x = v[0] > hi.x ? v[0] : hi.x;
            hi.y = v[1] > hi.y ? v[1] : hi.y;
            hi.z = v[2] > hi.z ? v[2] : hi.z;
        }
    }
    // bounding_box_to_triangles, mod.rs:501-536: vertex and index tables as written there
    const vec3 vtx[8] = {mk(lo.x, lo.y, lo.z), mk(hi.x, lo.y, lo.z), mk(hi.x, hi.y, lo.z), mk(lo.x, hi.y, lo.z),
                         mk(lo.x, lo.y, hi.z), mk(hi.x, lo.y, hi.z), mk(hi.x, hi.y, hi.z), mk(lo.x, hi.y, hi.z)};
    static const int idx[12][3] = {{0, 1, 2}, {0, 2, 3}, {4, 6, 5}, {4, 7, 6}, {0, 4, 5}, {0, 5, 1},
                                   {3, 2, 6}, {3, 6, 7}, {1, 5, 6}, {1, 6, 2}, {0, 3, 7}, {0, 7, 4}};
    for (int k = 0; k < 12; ++k) {
        st(out[k].a, vtx[idx[k][0]]);
        st(out[k].b, vtx[idx[k][1]]);
        st(out[k].c, vtx[idx[k][2]]);
    }
}

void box_pair_records(const pt_triangle box[12], const float position[3], TriPairRec out[6]) {
    const vec3 pos = ld(position);
    for (int k = 0; k < 12; ++k) {
        const vec3 a = ld(box[k].a) + pos, b = ld(box[k].b) + pos, c = ld(box[k].c) + pos;
        const vec3 e1 = b - a, e2 = c - a;
        TriPairRec &r = out[k / 2];
        const int hf = k & 1;
        r.ax[hf] = a.x, r.ay[hf] = a.y, r.az[hf] = a.z;
        r.e1x[hf] = e1.x, r.e1y[hf] = e1.y, r.e1z[hf] = e1.z;
        r.e2x[hf] = e2.x, r.e2y[hf] = e2.y, r.e2z[hf] = e2.z;
        r.id[hf] = (uint32_t)k;
    }
}

namespace {

struct BuildTri {
    vec3 lo, hi, mid;
    vec3 a, e1, e2;
    uint32_t id;  // flattened triangle index
};

struct BvhBuilder {
    FlatScene &out;
    std::vector<BuildTri> &t;  // lo/hi already padded per triangle
    bool use_sah = true;       // false: median split (balanced: depth <= log2(n) + 1)
    uint32_t depth_max = 0;

    static void grow(vec3 &lo, vec3 &hi, const vec3 &l, const vec3 &h) {
        lo = mk(std::fmin(lo.x, l.x), std::fmin(lo.y, l.y), std::fmin(lo.z, l.z));
        hi = mk(std::fmax(hi.x, h.x), std::fmax(hi.y, h.y), std::fmax(hi.z, h.z));
    }

    // returns the child reference of the subtree over t[b,e) and its padded box
    int32_t build(size_t b, size_t e, vec3 &lo, vec3 &hi, uint32_t depth) {
        const float inf = std::numeric_limits<float>::infinity();
        lo = mk(inf, inf, inf);
        hi = mk(-inf, -inf, -inf);
        depth_max = depth > depth_max ? depth : depth_max;
        // (splitting such a run further where the surface-area heuristic would - a node step priced at 1, 2 or 4 triangle
        // tests - loses on mesh.json: 15.8, 16.8, 17.5 against 17.8 G bounces/s; it is the node steps that cost)
        if (e - b <= 2u * kBvhLeafPairs) {  // leaf = up to kBvhLeafPairs consecutive TriPairRecs
            const size_t first = out.tri_pairs.size();
            for (size_t k0 = b; k0 < e; k0 += 2) {
                TriPairRec rec{};
                rec.id[0] = rec.id[1] = kNoTri;  // a filler half is all zeros: determinant 0, rejected (mod.rs:571)
                for (size_t k = k0; k < e && k < k0 + 2; ++k) {
                    const uint32_t hf = (uint32_t)(k - k0);
                    const BuildTri &q = t[k];
                    rec.ax[hf] = q.a.x, rec.ay[hf] = q.a.y, rec.az[hf] = q.a.z;
                    rec.e1x[hf] = q.e1.x, rec.e1y[hf] = q.e1.y, rec.e1z[hf] = q.e1.z;
                    rec.e2x[hf] = q.e2.x, rec.e2y[hf] = q.e2.y, rec.e2z[hf] = q.e2.z;
                    rec.id[hf] = q.id;
                    grow(lo, hi, q.lo, q.hi);
                }
                out.tri_pairs.push_back(rec);
            }
            const size_t count = out.tri_pairs.size() - first;
            return ~(int32_t)((first << kBvhLeafBits) | (count - 1));
        }
        // surface-area-heuristic split: for each axis sort by centroid and sweep; only even left counts are
        // considered so that leaves are full pairs wherever possible.  Ties are broken by triangle id: the tree
        // (and with it the traversal order) is a pure function of the scene.
        const size_t cnt = e - b;
        size_t half = 0;
        int best_axis = -1;
        float best_cost = inf;
        std::vector<float> right_area(cnt + 1);
        auto area = [](const vec3 &l, const vec3 &h) {
            const vec3 d = h - l;
            return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
        };
        for (int axis = 0; use_sah && axis < 3; ++axis) {
            auto key = [axis](const BuildTri &q) { return axis == 0 ? q.mid.x : (axis == 1 ? q.mid.y : q.mid.z); };
            std::sort(t.begin() + (long)b, t.begin() + (long)e, [&](const BuildTri &x, const BuildTri &y) {
                const float kx = key(x), ky = key(y);
                return kx < ky || (kx == ky && x.id < y.id);
            });
            vec3 l = mk(inf, inf, inf), h = mk(-inf, -inf, -inf);
            for (size_t k = cnt; k-- > 0;) {
                grow(l, h, t[b + k].lo, t[b + k].hi);
                right_area[k] = area(l, h);
            }
            l = mk(inf, inf, inf), h = mk(-inf, -inf, -inf);
            for (size_t k = 1; k < cnt; ++k) {
                grow(l, h, t[b + k - 1].lo, t[b + k - 1].hi);
                if (k & 1) continue;
                const float cost = area(l, h) * (float)k + right_area[k] * (float)(cnt - k);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    half = k;
                }
            }
        }
        if (best_axis < 0 && use_sah) {  // cnt == 3: one full pair + one half-filled leaf
            best_axis = 2;
            half = 2;
        }
        if (!use_sah) {  // median of the widest centroid axis, left count even
            vec3 clo = mk(inf, inf, inf), chi = mk(-inf, -inf, -inf);
            for (size_t k = b; k < e; ++k) grow(clo, chi, t[k].mid, t[k].mid);
            const vec3 ext = chi - clo;
            best_axis = (ext.x >= ext.y && ext.x >= ext.z) ? 0 : (ext.y >= ext.z ? 1 : 2);
            half = ((cnt / 2) + 1) & ~(size_t)1;
            if (half >= cnt) half = 2;
            const int axis = best_axis;
            auto key = [axis](const BuildTri &q) { return axis == 0 ? q.mid.x : (axis == 1 ? q.mid.y : q.mid.z); };
            std::sort(t.begin() + (long)b, t.begin() + (long)e, [&](const BuildTri &x, const BuildTri &y) {
                const float kx = key(x), ky = key(y);
                return kx < ky || (kx == ky && x.id < y.id);
            });
        } else if (best_axis != 2) {  // the array is currently sorted along z: restore the winning order
            const int axis = best_axis;
            auto key = [axis](const BuildTri &q) { return axis == 0 ? q.mid.x : q.mid.y; };
            std::sort(t.begin() + (long)b, t.begin() + (long)e, [&](const BuildTri &x, const BuildTri &y) {
                const float kx = key(x), ky = key(y);
                return kx < ky || (kx == ky && x.id < y.id);
            });
        }
        const size_t node_at = out.bvh_nodes.size();
        out.bvh_nodes.push_back(BvhNode{});
        vec3 l0, h0, l1, h1;
        const int32_t c0 = build(b, b + half, l0, h0, depth + 1);
        const int32_t c1 = build(b + half, e, l1, h1, depth + 1);
        BvhNode &n = out.bvh_nodes[node_at];
        n.lox[0] = l0.x, n.loy[0] = l0.y, n.loz[0] = l0.z, n.hix[0] = h0.x, n.hiy[0] = h0.y, n.hiz[0] = h0.z;
        n.lox[1] = l1.x, n.loy[1] = l1.y, n.loz[1] = l1.z, n.hix[1] = h1.x, n.hiy[1] = h1.y, n.hiz[1] = h1.z;
        n.c[0] = c0;
        n.c[1] = c1;
        grow(lo, hi, l0, h0);
        grow(lo, hi, l1, h1);
        return (int32_t)node_at;
    }
};

// The binary tree below `ref` four children wide (BvhNode4): a node's children are its two children, and while there is
// room the inner child with the largest box is replaced by ITS two children - boxes and leaf references are the binary
// tree's.  Returns the reference in the wide tree (a leaf reference stays what it is).
int32_t widen(const std::vector<BvhNode> &bin, int32_t ref, std::vector<BvhNode4> &wide) {
    if (ref < 0) return ref;
    struct Kid {
        int32_t ref;
        float lo[3], hi[3];
    };
    auto kid_of = [&](const BvhNode &n, int h) {
        Kid k;
        k.ref = n.c[h];
        k.lo[0] = n.lox[h], k.lo[1] = n.loy[h], k.lo[2] = n.loz[h];
        k.hi[0] = n.hix[h], k.hi[1] = n.hiy[h], k.hi[2] = n.hiz[h];
        return k;
    };
    auto area = [](const Kid &k) {
        const float dx = k.hi[0] - k.lo[0], dy = k.hi[1] - k.lo[1], dz = k.hi[2] - k.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<Kid> kids = {kid_of(bin[(size_t)ref], 0), kid_of(bin[(size_t)ref], 1)};
    while (kids.size() < 4u) {
        int best = -1;
        for (int i = 0; i < (int)kids.size(); ++i)
            if (kids[(size_t)i].ref >= 0 && (best < 0 || area(kids[(size_t)i]) > area(kids[(size_t)best]))) best = i;
        if (best < 0) break;
        const BvhNode &n = bin[(size_t)kids[(size_t)best].ref];
        kids[(size_t)best] = kid_of(n, 0);  // (the first child takes the parent's place: the order stays a function of the scene)
        kids.insert(kids.begin() + best + 1, kid_of(n, 1));
    }
    const size_t at = wide.size();
    wide.push_back(BvhNode4{});
    const float nan = std::numeric_limits<float>::quiet_NaN();
    int32_t refs[4];
    for (size_t j = 0; j < 4u; ++j) refs[j] = j < kids.size() ? widen(bin, kids[j].ref, wide) : 0;
    BvhNode4 &w = wide[at];
    for (size_t j = 0; j < 4u; ++j) {
        const bool has = j < kids.size();
        w.lox[j] = has ? kids[j].lo[0] : nan, w.loy[j] = has ? kids[j].lo[1] : nan, w.loz[j] = has ? kids[j].lo[2] : nan;
        w.hix[j] = has ? kids[j].hi[0] : nan, w.hiy[j] = has ? kids[j].hi[1] : nan, w.hiz[j] = has ? kids[j].hi[2] : nan;
        w.c[j] = has ? refs[j] : refs[0];
    }
    return (int32_t)at;
}

}  // namespace

// The per-wave walk queue (bvh_closest_queue) and the leaf list (bvh_closest_postponed) pack `owner lane | reference << 6`
// into 32 bits: a node index, or a leaf code first_record << kBvhLeafBits | count - 1, has 26 bits.
bool bvh_refs_fit(uint64_t n_nodes, uint64_t n_pair_records) {
    return n_nodes < (1ull << 26) && (n_pair_records << kBvhLeafBits) < (1ull << 26);
}

bool flatten_scene(const pt_camera &cam, const pt_object *objs, uint32_t n_objs, const pt_triangle *tris,
                   uint32_t n_tris, FlatScene &out, std::string &err) {
    if (n_objs >= (1u << 30) || n_tris >= (1u << 30)) {
        err = "scene too large";
        return false;
    }
    out.objs.assign(n_objs, ObjRec{});
    out.mats.assign(n_objs, MatRec{});
    out.tri_pairs.clear();
    out.bvh_nodes.clear();
    out.tri_shade.assign(n_tris, TriShade{});
    // R: bound on |ray origin - any vertex|: ray origins are the lens centre or points on objects
    const float finf = std::numeric_limits<float>::infinity();
    vec3 slo = mk(finf, finf, finf), shi = mk(-finf, -finf, -finf);
    {
        float lens[3], su[3], sv[3];
        camera_basis(cam, lens, su, sv);
        BvhBuilder::grow(slo, shi, ld(lens), ld(lens));
        for (uint32_t i = 0; i < n_objs; ++i) {
            const vec3 pos = ld(objs[i].position);
            if (objs[i].kind == PT_SPHERE) {
                const float r = f_abs(objs[i].radius);
                BvhBuilder::grow(slo, shi, pos - mk(r, r, r), pos + mk(r, r, r));
            } else if (objs[i].kind == PT_MESH && (uint64_t)objs[i].tri_offset + objs[i].tri_count <= n_tris) {
                for (uint32_t k = objs[i].tri_offset; k < objs[i].tri_offset + objs[i].tri_count; ++k) {
                    BvhBuilder::grow(slo, shi, ld(tris[k].a) + pos, ld(tris[k].a) + pos);
                    BvhBuilder::grow(slo, shi, ld(tris[k].b) + pos, ld(tris[k].b) + pos);
                    BvhBuilder::grow(slo, shi, ld(tris[k].c) + pos, ld(tris[k].c) + pos);
                }
            }
        }
    }
    const float scene_R = n_objs ? length(shi - slo) : 0.0f;
    std::vector<uint8_t> claimed(n_tris, 0);
    bool have_bvh = false;
    out.bvh_pair_base = 0;
    out.bvh_stack = 0;
    out.bvh_pair_span = 0;
    for (uint32_t i = 0; i < n_objs; ++i) {
        const pt_object &o = objs[i];
        if (o.kind != PT_SPHERE && o.kind != PT_MESH) {
            err = "object " + std::to_string(i) + ": unknown kind";
            return false;
        }
        if (o.reflect_type > PT_REFRACT) {
            err = "object " + std::to_string(i) + ": unknown reflect_type";
            return false;
        }
        const vec3 position = ld(o.position);
        ObjRec &r = out.objs[i];
        MatRec &m = out.mats[i];
        r.kind = o.kind;
        if (o.kind == PT_SPHERE) {
            r.cx = position.x;
            r.cy = position.y;
            r.cz = position.z;
            r.rr = o.radius * o.radius;  // radius.powi(2), mod.rs:416
            r.rr_in = -1.0f;
            r.tri_begin = 0;
            r.tri_count = 0;
            r.pair_begin = 0;
            r.pair_count = 0;
            r.bvh_root = kNoBvh;
        } else {
            if ((uint64_t)o.tri_offset + o.tri_count > n_tris) {
                err = "object " + std::to_string(i) + ": triangle range outside the triangle array";
                return false;
            }
            const vec3 gate = ld(o.bs_center) + position;  // mod.rs:268
            r.cx = gate.x;
            r.cy = gate.y;
            r.cz = gate.z;
            r.rr = o.bs_radius * o.bs_radius;
            {
                // rr_in: see intersect_scene_dev.  A point of the ray, ahead of the origin, within (1 - eta) r of the
                // centre with eta = 1e-3: the chord through it is >= 0.09 r long, so the exact discriminant is
                // >= 2e-3 r^2 and the far root lies >= 1e-3 r ahead of the origin.  With every origin within 4 r of the
                // centre (origins lie in the scene's bounding box) the f32 discriminant is off by <= 4 e (16+16+1) r^2
                // ~ 8e-6 r^2, its root by <= 9e-5 r, so the computed far root is >= 9e-4 r >= 1e-4 for r >= 0.2: the
                // gate passes.  The factor 0.998 and the absolute term cover the device's own o + d*t and distance.
                const float rad = f_abs(o.bs_radius);
                float far2 = 0.0f;  // squared distance from the centre to the farthest corner of the scene's box
                for (int k = 0; k < 8; ++k) {
                    const vec3 corner = mk((k & 1) ? shi.x : slo.x, (k & 2) ? shi.y : slo.y, (k & 4) ? shi.z : slo.z);
                    far2 = f_max(far2, dot(corner - gate, corner - gate));
                }
                const bool offer = std::isfinite(rad) && rad >= 0.2f && far2 <= 16.0f * rad * rad;
                r.rr_in = offer ? r.rr * 0.998f - 1e-4f * (1.0f + rad) : -1.0f;
            }
            r.tri_begin = o.tri_offset;
            r.tri_count = o.tri_count;
            r.pair_begin = (uint32_t)out.tri_pairs.size();
            r.bvh_root = kNoBvh;
            std::vector<BuildTri> bt;
            bt.reserve(o.tri_count);
            for (uint32_t k = o.tri_offset; k < o.tri_offset + o.tri_count; ++k) {
                if (claimed[k]) {
                    err = "triangle " + std::to_string(k) + " belongs to two objects";
                    return false;
                }
                claimed[k] = 1;
                const vec3 a = ld(tris[k].a) + position;  // Triangle::transformed, mod.rs:546-552
                const vec3 b = ld(tris[k].b) + position;
                const vec3 c = ld(tris[k].c) + position;
                const vec3 e1 = b - a, e2 = c - a;  // mod.rs:560-561
                const vec3 nrm = normalize(cross(e1, e2));  // mod.rs:605
                TriShade &s = out.tri_shade[k];
                s.nx = nrm.x;
                s.ny = nrm.y;
                s.nz = nrm.z;
                s.owner = i;
                BuildTri q;
                q.a = a, q.e1 = e1, q.e2 = e2, q.id = k;
                q.lo = mk(std::fmin(a.x, std::fmin(b.x, c.x)), std::fmin(a.y, std::fmin(b.y, c.y)),
                          std::fmin(a.z, std::fmin(b.z, c.z)));
                q.hi = mk(std::fmax(a.x, std::fmax(b.x, c.x)), std::fmax(a.y, std::fmax(b.y, c.y)),
                          std::fmax(a.z, std::fmax(b.z, c.z)));
                q.mid = (q.lo + q.hi) * 0.5f;
                {
                    // Padding = bound on how far from the exact triangle a hit accepted by the f32 Moller-Trumbore
                    // arithmetic can lie.  With |det| >= 1e-4 (mod.rs:571), |tvec| <= R (scene diagonal), this
                    // triangle's edges <= L and unit roundoff e = 2^-24, forward error analysis of mod.rs:560-589
                    // gives |du|,|dv| <= e L (7L + 8R) / 1e-4 (the hit point moves by that times L) and
                    // |dt| <= e L^2 (7 t + 8R) / 1e-4 with t <= R; 16 e L^2 (R+L) / 1e-4 covers each of the three,
                    // so three times that (the 16 already holds a factor 2 of slack), plus the slab test's own roundoff.
                    const float L = std::fmax(length(e1), std::fmax(length(e2), length(c - b)));
                    const float e = 5.9604645e-8f;
                    const float pad = 3.0f * (16.0f * e * L * L * (scene_R + L) / 1e-4f) + 16.0f * e * (scene_R + L) + 1e-6f;
                    q.lo = q.lo - mk(pad, pad, pad);
                    q.hi = q.hi + mk(pad, pad, pad);
                }
                bt.push_back(q);
            }
            if (o.tri_count >= kBvhMinTris) {
                const size_t pairs_mark = out.tri_pairs.size(), nodes_mark = out.bvh_nodes.size();
                const std::vector<BuildTri> keep = bt;
                for (int attempt = 0; attempt < 2; ++attempt) {
                    BvhBuilder bb{out, bt};
                    bb.use_sah = attempt == 0;
                    vec3 blo, bhi;
                    r.bvh_root = bb.build(0, bt.size(), blo, bhi, 0);
                    if (bb.depth_max + 2 < kBvhStack) {
                        out.bvh_stack = std::max(out.bvh_stack, (uint32_t)bb.depth_max + 2u);
                        break;
                    }
                    if (attempt == 1) {
                        err = "object " + std::to_string(i) + ": BVH deeper than the traversal stack";
                        return false;
                    }
                    out.tri_pairs.resize(pairs_mark);  // SAH tree too deep for the stack: rebuild balanced
                    out.bvh_nodes.resize(nodes_mark);
                    bt = keep;
                }
                if (!have_bvh) out.bvh_pair_base = (uint32_t)pairs_mark;
                have_bvh = true;
                out.bvh_pair_span = (uint32_t)out.tri_pairs.size() - out.bvh_pair_base;
                if (!bvh_refs_fit(out.bvh_nodes.size(), out.tri_pairs.size())) {
                    err = "mesh too large for the BVH walkers: node indices and leaf codes are packed into 26 bits of a queue entry "
                          "(2^26 nodes, 2^25 pair records with leaves of two records)";
                    return false;
                }
            } else {
                for (size_t k = 0; k < bt.size(); k += 2) {  // list order, two triangles per record
                    TriPairRec rec{};
                    rec.id[0] = rec.id[1] = kNoTri;
                    for (size_t hf = 0; hf < 2 && k + hf < bt.size(); ++hf) {
                        const BuildTri &q = bt[k + hf];
                        rec.ax[hf] = q.a.x, rec.ay[hf] = q.a.y, rec.az[hf] = q.a.z;
                        rec.e1x[hf] = q.e1.x, rec.e1y[hf] = q.e1.y, rec.e1z[hf] = q.e1.z;
                        rec.e2x[hf] = q.e2.x, rec.e2y[hf] = q.e2.y, rec.e2z[hf] = q.e2.z;
                        rec.id[hf] = q.id;
                    }
                    out.tri_pairs.push_back(rec);
                }
            }
            r.pair_count = (uint32_t)out.tri_pairs.size() - r.pair_begin;
        }
        m.cr = o.color[0];
        m.cg = o.color[1];
        m.cb = o.color[2];
        m.max_refl = f_max(o.color[0], f_max(o.color[1], o.color[2]));  // mod.rs:668
        m.er = o.emission[0];
        m.eg = o.emission[1];
        m.eb = o.emission[2];
        m.inv_max_refl = 1.0f / m.max_refl;  // mod.rs:679
        m.px = position.x;
        m.py = position.y;
        m.pz = position.z;
        m.reflect = o.reflect_type;
    }
    // pairs in visiting order (mod.rs:637: highest index first)
    out.obj_pairs.assign((n_objs + 1u) / 2u, ObjPairRec{});
    for (uint32_t v = 0; v < 2u * (uint32_t)out.obj_pairs.size(); ++v) {
        ObjPairRec &pr = out.obj_pairs[v / 2u];
        const uint32_t hf = v & 1u;
        if (v < n_objs) {
            const ObjRec &r = out.objs[n_objs - 1u - v];
            pr.cx[hf] = r.cx, pr.cy[hf] = r.cy, pr.cz[hf] = r.cz, pr.rr[hf] = r.rr;
            pr.kind[hf] = r.kind;
            pr.pair_begin[hf] = r.pair_begin;
            pr.pair_count[hf] = r.pair_count;
            pr.bvh_root[hf] = r.bvh_root;
            pr.obj[hf] = n_objs - 1u - v;
            // A gate pays when a whole wave of 64 unrelated rays misses the sphere, i.e. when a single ray hits it
            // with probability well under 1/64: roughly (radius / distance)^2 / 4 with distances of the order of the
            // scene.  Spheres above an eighth of the scene diagonal are not worth their arithmetic.
            // (a mesh with a BVH keeps its gate: there it saves a whole walk, and the deferred walks need its result)
            pr.admit[hf] = (r.kind == kKindMesh && r.bvh_root == kNoBvh && r.rr > (scene_R * 0.125f) * (scene_R * 0.125f)) ? 1u : 0u;
            // a mesh scanned pair by pair whose triangles all lie in one axis-aligned plane: which axis (bits 1-2)
            if (r.kind == kKindMesh && r.bvh_root == kNoBvh) {
                for (uint32_t ax = 0; ax < 3; ++ax) {
                    bool flat = r.pair_count != 0u;
                    for (uint32_t pp = r.pair_begin; flat && pp < r.pair_begin + r.pair_count; ++pp) {
                        const TriPairRec &tp = out.tri_pairs[pp];
                        const float *e1 = ax == 0 ? tp.e1x : (ax == 1 ? tp.e1y : tp.e1z);
                        const float *e2 = ax == 0 ? tp.e2x : (ax == 1 ? tp.e2y : tp.e2z);
                        flat = e1[0] == 0.0f && e1[1] == 0.0f && e2[0] == 0.0f && e2[1] == 0.0f;  // fillers are all zero
                    }
                    if (flat) {
                        pr.admit[hf] |= (ax + 1u) << 1;
                        break;
                    }
                }
            }
        } else {  // filler: a sphere whose discriminant is -inf for every finite ray
            pr.cx[hf] = pr.cy[hf] = pr.cz[hf] = 0.0f;
            pr.rr[hf] = -std::numeric_limits<float>::infinity();
            pr.kind[hf] = kKindSphere;
            pr.bvh_root[hf] = kNoBvh;
            pr.obj[hf] = 0;
            pr.admit[hf] = 0u;
        }
    }
    // ---- tables of the candidate scan (intersect_cand) -----------------------------------------------------------
    // ranks: the reference's visiting sequence - objects from the last to the first (mod.rs:637), a mesh's triangles in
    // list order (mod.rs:558)
    std::vector<uint32_t> tri_rank(n_tris ? n_tris : 1u, 0u);
    out.rank_id.assign((size_t)n_objs + n_tris + 1u, 0u);
    std::vector<uint32_t> obj_rank(n_objs, 0u);
    {
        uint32_t next = 0;
        for (uint32_t v = 0; v < n_objs; ++v) {
            const uint32_t i = n_objs - 1u - v;
            obj_rank[i] = next;
            if (objs[i].kind == PT_SPHERE) {
                out.rank_id[next++] = i;
            } else {
                for (uint32_t k = 0; k < objs[i].tri_count; ++k) {
                    tri_rank[objs[i].tri_offset + k] = next;
                    out.rank_id[next++] = n_objs + objs[i].tri_offset + k;
                }
            }
        }
    }
    // shading records by rank
    out.surf.assign(out.rank_id.size(), SurfRec{});
    for (size_t rk = 0; rk + 1 < out.rank_id.size(); ++rk) {
        const uint32_t id = out.rank_id[rk];
        SurfRec &sr = out.surf[rk];
        uint32_t owner = id;
        if (id >= n_objs) {
            const TriShade &ts = out.tri_shade[id - n_objs];
            sr.vx = ts.nx, sr.vy = ts.ny, sr.vz = ts.nz;
            owner = ts.owner;
            sr.kind = 0x100u;
        }
        const MatRec &mm = out.mats[owner];
        if (id < n_objs) sr.vx = mm.px, sr.vy = mm.py, sr.vz = mm.pz;
        sr.kind |= mm.reflect & 3u;
        sr.cr = mm.cr, sr.cg = mm.cg, sr.cb = mm.cb, sr.max_refl = mm.max_refl;
        sr.er = mm.er, sr.eg = mm.eg, sr.eb = mm.eb, sr.inv_max_refl = mm.inv_max_refl;
    }
    out.sph_pairs.clear();
    out.flat_pairs.clear();
    out.cand_pairs.clear();
    std::vector<CandPairRec> filtered;  // records that have a filter (appended after the unfiltered ones below)
    // the exact-test record of pair record pp of object i
    auto cand_rec = [&](uint32_t i, uint32_t pp) {
        const TriPairRec &tp = out.tri_pairs[pp];
        CandPairRec c{};
        for (int hf = 0; hf < 2; ++hf) {
            c.ax[hf] = tp.ax[hf], c.ay[hf] = tp.ay[hf], c.az[hf] = tp.az[hf];
            c.e1x[hf] = tp.e1x[hf], c.e1y[hf] = tp.e1y[hf], c.e1z[hf] = tp.e1z[hf];
            c.e2x[hf] = tp.e2x[hf], c.e2y[hf] = tp.e2y[hf], c.e2z[hf] = tp.e2z[hf];
            c.id[hf] = tp.id[hf] == kNoTri ? kNoTri : tri_rank[tp.id[hf]];
        }
        c.gx = out.objs[i].cx, c.gy = out.objs[i].cy, c.gz = out.objs[i].cz;
        c.grr = out.objs[i].rr;
        c.grr_in = out.objs[i].rr_in;
        return c;
    };
    {
        const float ninf = -std::numeric_limits<float>::infinity();
        uint32_t n_sph = 0;
        for (uint32_t v = 0; v < n_objs; ++v) {
            const uint32_t i = n_objs - 1u - v;
            if (objs[i].kind != PT_SPHERE) continue;
            if ((n_sph & 1u) == 0u) {
                SphPairRec f{};
                f.rr[0] = f.rr[1] = ninf;
                out.sph_pairs.push_back(f);
            }
            SphPairRec &sp = out.sph_pairs.back();
            const uint32_t hf = n_sph & 1u;
            sp.cx[hf] = out.objs[i].cx, sp.cy[hf] = out.objs[i].cy, sp.cz[hf] = out.objs[i].cz, sp.rr[hf] = out.objs[i].rr;
            sp.rank[hf] = obj_rank[i];
            ++n_sph;
        }
        // pair records of meshes without a BVH: flat ones (both triangles in one axis-aligned plane) get a filter,
        // grouped by axis so that two of them share a record; the rest are candidates for every ray
        std::vector<FlatPairRec> by_axis[3];
        uint32_t fill[3] = {0, 0, 0};
        for (uint32_t i = 0; i < n_objs; ++i) {
            const ObjRec &r = out.objs[i];
            if (r.kind != kKindMesh || r.bvh_root != kNoBvh) continue;
            for (uint32_t pp = r.pair_begin; pp < r.pair_begin + r.pair_count; ++pp) {
                const TriPairRec &tp = out.tri_pairs[pp];
                int axis = -1;
                for (int ax = 0; ax < 3 && axis < 0; ++ax) {
                    const float *e1 = ax == 0 ? tp.e1x : (ax == 1 ? tp.e1y : tp.e1z);
                    const float *e2 = ax == 0 ? tp.e2x : (ax == 1 ? tp.e2y : tp.e2z);
                    const float *aa = ax == 0 ? tp.ax : (ax == 1 ? tp.ay : tp.az);
                    const bool two = tp.id[1] != kNoTri;
                    if (e1[0] == 0.0f && e2[0] == 0.0f && (!two || (e1[1] == 0.0f && e2[1] == 0.0f && aa[1] == aa[0]))) axis = ax;
                }
                // |N| of the record's triangles (the smaller one), their longest edge, their bounds in the plane
                float n_min = std::numeric_limits<float>::infinity(), L = 0.0f;
                vec3 lo = mk(finf, finf, finf), hi = mk(-finf, -finf, -finf);
                for (int hf = 0; hf < 2; ++hf) {
                    if (tp.id[hf] == kNoTri) continue;
                    const vec3 a = mk(tp.ax[hf], tp.ay[hf], tp.az[hf]);
                    const vec3 e1 = mk(tp.e1x[hf], tp.e1y[hf], tp.e1z[hf]), e2 = mk(tp.e2x[hf], tp.e2y[hf], tp.e2z[hf]);
                    n_min = std::fmin(n_min, length(cross(e1, e2)));
                    L = std::fmax(L, std::fmax(length(e1), std::fmax(length(e2), length(e2 - e1))));
                    BvhBuilder::grow(lo, hi, a, a);
                    BvhBuilder::grow(lo, hi, a + e1, a + e1);
                    BvhBuilder::grow(lo, hi, a + e2, a + e2);
                }
                // The filter only judges rays with |d_a| >= kGrazing, for which |determinant| = |d_a| |N| >= |N| / 64
                // (and >= 1e-4, mod.rs:571): the forward-error bound of the BVH boxes above with that determinant in
                // place of 1e-4, plus the filter's own arithmetic (an approximate reciprocal and two fmas on
                // distances up to R / kGrazing).
                const float e = 5.9604645e-8f;
                const float det_min = std::fmax(1e-4f, n_min * kGrazing);
                const float pad = 3.0f * (16.0f * e * L * L * (scene_R + L) / det_min) + (32.0f / kGrazing) * e * (scene_R + L) + 1e-5f;
                const bool usable = axis >= 0 && std::isfinite(pad) && std::isfinite(n_min) && n_min > 0.0f;
                if (!usable) {
                    out.cand_pairs.push_back(cand_rec(i, pp));
                    continue;
                }
                const int a = axis, b = (axis + 1) % 3, c = (axis + 2) % 3;
                const float lov[3] = {lo.x, lo.y, lo.z}, hiv[3] = {hi.x, hi.y, hi.z};
                if ((fill[a] & 1u) == 0u) {
                    FlatPairRec f{};
                    f.pair[0] = f.pair[1] = kNoPair;
                    f.hb[0] = f.hb[1] = f.hc[0] = f.hc[1] = -1.0f;  // empty rectangle: |x - c| <= -1 never holds
                    f.tpad[0] = f.tpad[1] = 0.0f;
                    f.axis = (uint32_t)a;
                    f.sign_exact = 1u;  // (cleared by the first half that does not qualify; a filler half qualifies)
                    by_axis[a].push_back(f);
                }
                FlatPairRec &f = by_axis[a].back();
                const uint32_t hf = fill[a] & 1u;
                f.pc[hf] = lov[a];
                f.cb[hf] = 0.5f * (lov[b] + hiv[b]);
                f.hb[hf] = 0.5f * (hiv[b] - lov[b]) + pad + 4.0f * e * (f_abs(lov[b]) + f_abs(hiv[b]));
                f.cc[hf] = 0.5f * (lov[c] + hiv[c]);
                f.hc[hf] = 0.5f * (hiv[c] - lov[c]) + pad + 4.0f * e * (f_abs(lov[c]) + f_abs(hiv[c]));
                f.tpad[hf] = pad;
                // filter_flat's sign rule: the two products whose difference is the plane normal's component along the axis
                // must not nearly cancel, for both triangles of the record, in Triangle::intersect's `distance` numerator and
                // in its determinant alike (the same two products of edge components, mod.rs:563-564, 583-589)
                for (int h2 = 0; h2 < 2; ++h2) {
                    if (tp.id[h2] == kNoTri) continue;
                    const float e1v[3] = {tp.e1x[h2], tp.e1y[h2], tp.e1z[h2]}, e2v[3] = {tp.e2x[h2], tp.e2y[h2], tp.e2z[h2]};
                    const double pa = (double)e1v[b] * e2v[c], pb = (double)e1v[c] * e2v[b];
                    if (!(std::fabs(pa - pb) >= 1e-3 * (std::fabs(pa) + std::fabs(pb))) || !std::isfinite(pa) || !std::isfinite(pb))
                        f.sign_exact = 0u;
                }
                f.pair[hf] = (uint32_t)filtered.size();  // + n_other_pairs below
                filtered.push_back(cand_rec(i, pp));
                ++fill[a];
            }
        }
        for (int a = 0; a < 3; ++a) out.flat_pairs.insert(out.flat_pairs.end(), by_axis[a].begin(), by_axis[a].end());
        // records with the exact sign rule first: the kernels run them in a loop of their own (no branch on the rule per record)
        out.n_flat_exact = (uint32_t)(std::stable_partition(out.flat_pairs.begin(), out.flat_pairs.end(),
                                                            [](const FlatPairRec &f) { return f.sign_exact != 0u; }) -
                                      out.flat_pairs.begin());
        out.n_other_pairs = (uint32_t)out.cand_pairs.size();
        for (FlatPairRec &f : out.flat_pairs)
            for (int hf = 0; hf < 2; ++hf)
                if (f.pair[hf] != kNoPair) f.pair[hf] += out.n_other_pairs;
        out.cand_pairs.insert(out.cand_pairs.end(), filtered.begin(), filtered.end());
        out.cand_ok = out.cand_pairs.size() <= kCandMaxPairs;
    }
    out.tri_rank = tri_rank;
    out.bvh_meshes.clear();
    for (uint32_t v = 0; v < n_objs; ++v) {
        const ObjRec &r = out.objs[n_objs - 1u - v];
        if (r.kind != kKindMesh || r.bvh_root == kNoBvh) continue;
        BvhMeshRec bm{};
        bm.cx = r.cx, bm.cy = r.cy, bm.cz = r.cz, bm.rr = r.rr;
        bm.root = r.bvh_root;
        bm.root4 = widen(out.bvh_nodes, r.bvh_root, out.bvh_nodes4);
        out.bvh_meshes.push_back(bm);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// Passes and streams of a wavefront frame (see render_wavefront for how the plan is used, and for the retry on a failed
// allocation).
int plan_pass(const PassPlanIn &in, PassPlan &out, uint64_t *want_next) {
    const uint64_t npix = in.npix;
    uint32_t spp_pass = (uint32_t)std::min<uint64_t>(in.want / (npix ? npix : 1u), 0xffffffffull);
    if (spp_pass == 0) spp_pass = 1;
    if (spp_pass > in.spp) spp_pass = in.spp;
    if (spp_pass > kMaxPassSpp) spp_pass = kMaxPassSpp;  // sample-in-pass field of the stream bookkeeping word
    if (in.stack_form && spp_pass < in.spp) {
        // passes of equal length (4096 samples in passes of 682 at most would end with one of 4 samples: short streams,
        // a launch that cannot fill the chip); a pass may be up to a twentieth longer than asked for that - k_pass_cand's
        // memory does not grow with the pass
        const uint32_t stretch = spp_pass + spp_pass / 20u;
        const uint32_t n_eq = (in.spp + stretch - 1u) / stretch;
        const uint32_t eq = (in.spp + n_eq - 1u) / n_eq;
        if (eq <= kMaxPassSpp) spp_pass = eq;
    }
    // Streams: many more than the 2048 workgroups the chip holds at once, so that the dispatcher keeps every CU busy
    // until a launch ends, but each still a few launches' worth of work for its workgroup - about 2048 primary rays
    // per stream and pass (measured on cornell 1024x768: 2048 streams 22.0, 8192 24.3, 16384 24.7, 65536 23.2 G
    // bounces/s).  A stream owns at most kMaxStreamPixels pixels (their accumulators live in LDS inside k_shade).
    // (scenes with a BVH stage its nodes into LDS once per workgroup: twice the work per stream; mesh.json 2048 streams
    // 7.3, 8192 7.6, 16384 7.0)
    // (candidate scan, four waves per SIMD: 12288 streams 35.8, 16384 35.4, 8192 32.2, 24576 33.7 G bounces/s)
    // (candidate scan with walks, mesh.json: 24576 streams 19.8, 26624 20.5, 28672 20.1, 30720 20.3, 32768 19.9 G bounces/s)
    // (k_pass_cand's waves run without levels: a wave's first and last trips - the stack fills, the last rays die - are
    // the only ones that are not full, so its streams are long: 12 Ki primaries 43.4, 24 Ki 44.3, 48 Ki 43.8.  Fewer, much
    // longer streams - 2 048 of 384 pixels, two rounds of resident workgroups - lose: 42.0 against 46.4; 4 096: 43.4; 8 192:
    // 45.1 - same instruction count, but a fifth of the launch with few workgroups left (PMC: busy cycles per bounce 1.97
    // against 1.64 at fewer wave-cycles): streams are not equally long and two rounds cannot average that out)
    // (round 4, with the walk queue over the four-wide tree: scenes with walks a little shorter - mesh.json 1024x768 @1024,
    // 18 Ki primaries per stream 29.0, 20 Ki 29.1, 21-22 Ki 29.2, 24 Ki 28.7, 32 Ki 28.5 G bounces/s; cornell.json is flat from 12 Ki to
    // 48 Ki: 47.2-47.7)
    const uint64_t per_stream = in.stack_form ? (uint64_t)(in.per_stream ? in.per_stream : (in.has_bvh ? 22528u : 16384u))
                                              : (in.has_bvh ? 4096u : 2048u);
    uint64_t k_target = (npix * spp_pass + per_stream - 1u) / per_stream;
    if (k_target < 2048u) k_target = 2048u;
    if (in.streams) k_target = in.streams;
    uint32_t m = (uint32_t)((npix + k_target - 1) / k_target);
    if (m == 0) m = 1;
    if (m > kMaxStreamPixels) m = kMaxStreamPixels;
    // k_pass_cand: frames of few samples get more, shorter streams rather than a handful of workgroups per CU slot, each
    // with hundreds of pixels' accumulators and tables (36 B per pixel) in LDS (1024x768 @128: 12 288 streams of 64
    // pixels, not 4 096 of 192)
    // (round 4, final kernels: with walks up to 128 - 3000x2000 @100, parts of 2^20 pixels: 64 pixels 25.2, 96: 27.1, 128: 27.7, 192: 24.2 G
    // bounces/s on mesh.json, 1024x768 @128: 26.4 / 27.0 / 26.9 / 23.4; without walks the two frames disagree - 49.0 / 50.4 / 50.6 / 48.5
    // and 50.4 / 49.5 / 49.0 / 47.9 - and 64 stays)
    const uint32_t m_few = in.has_bvh ? 128u : 64u;
    if (in.stack_form && !in.streams && m > m_few) m = m_few;
    // A launch runs its workgroups in rounds of as many as the chip holds (four per CU); a stream's work grows with its m
    // pixels, so a launch takes about ceil(K / resident) x m: among the m within -15 % / +20 % of the tuned size take the
    // one for which that is smallest (cornell 1024x768: m = 21 -> 24, 37 450 streams in 36.6 rounds -> 32 768 in 32.0,
    // 37.3 -> 37.7 G bounces/s).  Not for scenes with walks, whose streams differ too much in length for rounds to show
    // (mesh.json: 24.0 rounds are slower than 25.6).
    // SMALL FRAMES (round 4: the reference's own sizes - its launch configuration is 450x300 @500, .vscode/launch.json).  When
    // the tuned stream size gives the launch fewer than eight rounds of resident workgroups, what a launch takes is
    // ceil(K / resident) x m to a good approximation, and a last round that is mostly empty costs a whole round: 450x300
    // @500 at the tuned size is 2 756 streams of 49 pixels = 2.7 rounds, mesh.json 21.3 G bounces/s; 4 120 streams of 33
    // pixels = 4.02 rounds: 26.2; 5 493 of 25 = 5.4 rounds: 23.6; 8 240 of 17 = 8.05: 25.0 (cornell.json: 42.1 / 43.0 / 43.2 /
    // 43.8 - the same order, flatter).  So among the stream sizes from a quarter of the tuned one up to it, the one with the
    // least ceil(K / resident) x m; of equals the longer streams where rays walk (their streams differ most in length), the
    // shorter ones otherwise.
    bool small_frame = false;
    if (in.stack_form && !in.streams && in.n_cus != 0u && m >= 4u) {
        const uint64_t resident = (uint64_t)in.n_cus * (in.groups_per_cu ? in.groups_per_cu : 4u);
        const uint64_t k_tuned = (npix + m - 1u) / m;
        // (fewer than EIGHT rounds - sixteen until the final kernels of round 4: at 12.8 rounds, 3000x2000 @100 in parts of 2^20
        // pixels, the rule took streams of 20 pixels for the whole rounds' sake and lost 6 % to the 64-pixel streams it replaced)
        if ((k_tuned + resident - 1u) / resident < 8u) {
            small_frame = true;
            uint32_t best_m = m;
            uint64_t best_cost = ~0ull;
            for (uint32_t mm = m / 4u ? m / 4u : 1u; mm <= m; ++mm) {
                const uint64_t kk = (npix + mm - 1u) / mm;
                const uint64_t cost = ((kk + resident - 1u) / resident) * mm;
                if (cost < best_cost || (cost == best_cost && in.has_bvh)) {
                    best_cost = cost;
                    best_m = mm;
                }
            }
            m = best_m;
        }
    }
    if (!small_frame && in.cand_scan && !in.has_bvh && !in.streams && in.n_cus != 0u && m >= 8u) {
        const uint64_t resident = (uint64_t)in.n_cus * (in.groups_per_cu ? in.groups_per_cu : 4u);
        uint32_t best_m = m;
        uint64_t best_cost = ~0ull;
        for (uint32_t mm = m - m * 15u / 100u; mm <= m + m / 5u && mm <= (in.stack_form ? 72u : kMaxStreamPixels); ++mm) {
            const uint64_t kk = (npix + mm - 1u) / mm;
            const uint64_t cost = ((kk + resident - 1u) / resident) * mm;
            if (cost < best_cost || (cost == best_cost && (mm > m ? mm - m : m - mm) < (best_m > m ? best_m - m : m - best_m))) {
                best_cost = cost;
                best_m = mm;
            }
        }
        m = best_m;
    }
    const uint32_t K = (uint32_t)((npix + m - 1) / m);
    // a primary ray has at most 4 descendants alive at one depth (two refract splits, mod.rs:760); k_pass_cand gives each
    // of its four waves a quarter of the slice and ceil(n / 4) of the stream's n primaries: 4 * ceil(n / 4) <= n + 3
    uint64_t cap64 = (4ull * m * spp_pass + 16u + kBlock - 1) / kBlock * kBlock;
    // k_pass_cand keeps a wave's waiting rays on a stack of at most kWaveStackMax slots (a quarter of the stream's slice
    // per wave, a power of two of at least 128 slots; a pass whose waves' whole quarters fit a smaller stack gets that)
    if (in.stack_form) {
        // (a wave gets every fourth chunk of 64 of the stream's m * spp_pass primaries: at most ceil(chunks / 4) * 64 of them, each
        // with at most four descendants waiting at a time - k_pass_cand's `room_for_all`)
        const uint64_t n_prim = (uint64_t)m * spp_pass, most = ((n_prim + 63u) / 64u + 3u) / 4u * 64u;
        const uint64_t need_w = 4u * (most < n_prim ? most : n_prim) + 3u;
        uint64_t cap_w = 128u;
        const uint64_t stack_max = in.wave_stack ? in.wave_stack : kWaveStackMax;
        while (cap_w < need_w && cap_w < stack_max) cap_w *= 2u;
        cap64 = 4u * cap_w;
        const size_t need = queue_bytes(K, (uint32_t)cap64) + (in.stack_park ? (size_t)K * 4u * kWaveParkBytes : 0u);
        if (in.stack_budget && need > in.stack_budget && spp_pass > 1u) {
            // the default pass does not fit the budget: smaller passes have fewer streams or smaller stacks
            *want_next = npix * (spp_pass / 2u);
            return kPlanRetry;
        }
    }
    // (slot indices are 32-bit over the whole queue, byte offsets 32-bit inside a stream's slice of cap * 40 bytes)
    if (cap64 * K > 0xffffffffull / 2 || cap64 * kRayBytes > 0xffffffffull) {
        if (spp_pass > 1u && in.want_is_default) {  // (a default this large only on a device with > 680 GB)
            *want_next = in.want / 2;
            return kPlanRetry;
        }
        return kPlanTooLarge;
    }
    out.spp_pass = spp_pass;
    out.m = m;
    out.K = K;
    out.cap = (uint32_t)cap64;
    out.bytes0 = queue_bytes(K, out.cap);
    out.bytes1 = in.stack_form ? (in.stack_park ? (size_t)K * 4u * kWaveParkBytes : 0u) : out.bytes0;
    return kPlanOk;
}

uint32_t next_pass_samples(double rate, double target_ms, uint64_t npix, uint64_t probe, uint32_t s_prev, uint32_t left,
                           uint32_t max_pass) {
    if (npix == 0u) npix = 1u;
    uint64_t fit = rate > 0.0 ? (uint64_t)(rate * target_ms / (double)npix) : probe / npix;
    if (s_prev && fit > 16ull * s_prev) fit = 16ull * s_prev;
    if (fit < 1u) fit = 1u;
    if (fit > max_pass) fit = max_pass;
    uint64_t stretch = fit + fit / 5u;
    if (stretch > max_pass) stretch = max_pass;
    const uint32_t n_left = (uint32_t)((left + stretch - 1u) / stretch);
    return (left + n_left - 1u) / n_left;
}

}  // namespace host
}  // namespace pt
