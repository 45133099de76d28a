// pt_host.cpp — once-per-frame host arithmetic feeding the kernels.  Built with -ffp-contract=off: the
// values produced here must be the f32 values the reference computes (per frame for the camera, per ray
// for the triangle edges — see pt_device.h).
#include "pt_host.h"

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

bool flatten_scene(const pt_object *objs, uint32_t n_objs, const pt_triangle *tris, uint32_t n_tris, FlatScene &out,
                   std::string &err) {
    if (n_objs >= (1u << 30) || n_tris >= (1u << 30)) {
        err = "scene too large";
        return false;
    }
    out.objs.assign(n_objs, ObjRec{});
    out.mats.assign(n_objs, MatRec{});
    out.tri_pairs.clear();
    out.tri_shade.assign(n_tris, TriShade{});
    std::vector<uint8_t> claimed(n_tris, 0);
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
            r.tri_begin = 0;
            r.tri_count = 0;
            r.pair_begin = 0;
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
            r.tri_begin = o.tri_offset;
            r.tri_count = o.tri_count;
            r.pair_begin = (uint32_t)out.tri_pairs.size();
            out.tri_pairs.resize(out.tri_pairs.size() + (o.tri_count + 1u) / 2u, TriPairRec{});
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
                const uint32_t local = k - o.tri_offset;
                TriPairRec &t = out.tri_pairs[r.pair_begin + local / 2u];
                const uint32_t hf = local & 1u;
                t.ax[hf] = a.x;
                t.ay[hf] = a.y;
                t.az[hf] = a.z;
                t.e1x[hf] = e1.x;
                t.e1y[hf] = e1.y;
                t.e1z[hf] = e1.z;
                t.e2x[hf] = e2.x;
                t.e2y[hf] = e2.y;
                t.e2z[hf] = e2.z;
                const vec3 nrm = normalize(cross(e1, e2));  // mod.rs:605
                TriShade &s = out.tri_shade[k];
                s.nx = nrm.x;
                s.ny = nrm.y;
                s.nz = nrm.z;
                s.owner = i;
            }
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
    return true;
}

}  // namespace host
}  // namespace pt
