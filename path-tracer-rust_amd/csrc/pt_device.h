// pt_device.h — device data layout and the per-ray device functions of the radiance() path.
//
// Hot-path functions restated for CDNA4 (citations relative to /root/reference):
//   intersect_scene_dev  : intersect_scene (mod.rs:631-659) + SceneObjectData::intersect (mod.rs:261-280)
//                          + intersect_sphere (mod.rs:412-438) + Triangle::intersect (mod.rs:554-615)
//   shade_hit            : body of radiance() after the intersection (mod.rs:665-789), iterative:
//                          throughput goes down the path instead of radiance coming back up
//   primary_ray          : the per-sample part of render_pixel (mod.rs:812-843)
//
// Data layout (HBM): the scene is flattened once per frame into small read-only tables.
//   - the object loop of intersect_scene is wave-uniform, so object/triangle records are read with
//     wave-uniform indices: hipcc turns those into scalar loads (s_load_dwordx4/x8 -> SGPRs) and the
//     VALU reads them as scalar operands: no VGPRs, no LDS traffic, one scalar-cache line per record;
//   - per-triangle A, e1 = B-A, e2 = C-A and the normal are precomputed on the host with the same f32
//     operations the reference performs per ray (Triangle::transformed then the two subtractions), so
//     the values are bit-identical to what the reference recomputes for every ray;
//   - material / normal records are gathered per lane (divergent index) by the shade step.
#pragma once

#include "pt_math.h"

namespace pt {

constexpr uint32_t kKindSphere = 0u, kKindMesh = 1u;
constexpr uint32_t kDiffuse = 0u, kSpecular = 1u, kRefract = 2u;
constexpr int kMaxDepth = 12;  // MAX_DEPTH, mod.rs:661

// one record per scene object, read with wave-uniform index by the intersect loop
struct alignas(16) ObjRec {
    float cx, cy, cz;    // sphere: position; mesh: bounding_sphere.position + position (mod.rs:268)
    float rr;            // radius.powi(2) of that sphere (mod.rs:416)
    uint32_t kind;       // kKindSphere / kKindMesh
    uint32_t tri_begin;   // first triangle of the mesh in the flattened triangle numbering
    uint32_t tri_count;
    uint32_t pair_begin;  // first TriPairRec of the mesh; it has (tri_count+1)/2 of them
};

// Two consecutive triangles of one mesh (world space), component by component, read with a wave-uniform
// index.  Each component pair sits in two adjacent dwords, so after the scalar load it is an aligned SGPR
// pair and can be the scalar operand of a packed VALU instruction (v_pk_mul_f32 / v_pk_add_f32): one ray is
// tested against both triangles with one instruction per arithmetic step.  On gfx950 a VALU instruction
// occupies its SIMD for 4 cycles whether it is packed or not (measured: profiles/README.md), so packing
// halves the issue slots of the Moller-Trumbore arithmetic; every half is still an IEEE mul/add/sub.
// A mesh with an odd triangle count gets an all-zero second triangle: its determinant is 0, which the
// reference's first test rejects (mod.rs:571).
struct alignas(16) TriPairRec {
    float ax[2], ay[2], az[2];     // tri.a + offset                            (mod.rs:548)
    float e1x[2], e1y[2], e1z[2];  // va_vb = (tri.b+offset) - (tri.a+offset)   (mod.rs:560)
    float e2x[2], e2y[2], e2z[2];  // va_vc                                      (mod.rs:561)
    float pad[2];
};

// per-object material record, gathered per lane in shade
struct alignas(16) MatRec {
    float cr, cg, cb, max_refl;      // color, max(color)                       (mod.rs:667-668)
    float er, eg, eb, inv_max_refl;  // emmission, 1.0/max_reflection            (mod.rs:679)
    float px, py, pz;                // object position (sphere centre for the normal, mod.rs:431)
    uint32_t reflect;                // kDiffuse / kSpecular / kRefract
};

// per-triangle shading record, gathered per lane in shade
struct alignas(16) TriShade {
    float nx, ny, nz;  // va_vb.cross(va_vc).normalize()  (mod.rs:605)
    uint32_t owner;    // object index
};

struct DevScene {
    const ObjRec *objs;
    const TriPairRec *tri_pairs;
    const MatRec *mats;
    const TriShade *tri_shade;
    uint32_t n_objs;
    uint32_t n_tris;
};

// per-frame constants
struct FrameParams {
    uint32_t width, height, spp;
    uint32_t idx_begin;  // first framebuffer index of the band
    uint32_t npix;       // pixels in the band
    uint32_t seed_lo, seed_hi;
    float cam_px, cam_py, cam_pz;  // camera.position (sensor origin)
    float lens_x, lens_y, lens_z;  // lens_center()
    float su_x, su_y, su_z;        // orthogonals().0
    float sv_x, sv_y, sv_z;        // orthogonals().1
};

// ray meta word: sample index (24 bits) | depth (4 bits) | branch id (3 bits)
PT_HD uint32_t pack_meta(uint32_t sample, uint32_t depth, uint32_t branch) {
    return (sample & 0xFFFFFFu) | (depth << 24) | (branch << 28);
}
PT_HD uint32_t meta_sample(uint32_t m) { return m & 0xFFFFFFu; }
PT_HD uint32_t meta_depth(uint32_t m) { return (m >> 24) & 0xFu; }
PT_HD uint32_t meta_branch(uint32_t m) { return (m >> 28) & 0x7u; }

struct HitRec {
    float t;
    int32_t id;  // -1 miss; [0,n_objs) sphere object; n_objs + k = flattened triangle k
};

#if defined(__HIPCC__)

// two f32 lanes per register pair: arithmetic on it compiles to v_pk_mul_f32 / v_pk_add_f32 (each half an
// IEEE operation, no fusion under -ffp-contract=off)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) {
    f32x2 r = {v, v};
    return r;
}
__device__ __forceinline__ f32x2 ld2(const float (&p)[2]) {
    f32x2 r = {p[0], p[1]};
    return r;
}

// ---------------------------------------------------------------------------------------------
// closest hit.  Objects in reverse index order, strict '<' (ties keep the higher object index,
// mod.rs:637,649); inside a mesh the first triangle in list order wins ties (mod.rs:598).
// `best` starts at +inf instead of Option::None: identical for every finite, non-NaN distance.
__device__ __forceinline__ HitRec intersect_scene_dev(const DevScene &S, vec3 o, vec3 d) {
    float best_t = __builtin_inff();
    int32_t best_id = -1;
    const float eps = 1e-4f;
    for (int i = (int)S.n_objs - 1; i >= 0; --i) {
        const ObjRec ob = S.objs[i];  // wave-uniform -> scalar loads
        const vec3 op = mk(ob.cx, ob.cy, ob.cz) - o;
        const float b = dot(op, d);
        const float det = b * b - dot(op, op) + ob.rr;
        const float sq = f_sqrt(det);  // NaN when det < 0: both comparisons below are then false
        const float t0 = b - sq, t1 = b + sq;
        const bool near_ok = t0 >= eps, far_ok = t1 >= eps;
        const bool sph_hit = !(det < 0.0f) && (near_ok || far_ok);
        if (ob.kind == kKindSphere) {
            const float t = near_ok ? t0 : t1;
            if (sph_hit && t < best_t) {
                best_t = t;
                best_id = i;
            }
        } else {
            // bounding-sphere gate (mod.rs:267-273): skip the triangle list when no lane passes
            if (__builtin_amdgcn_ballot_w64(sph_hit) != 0ull) {
                float mt = __builtin_inff();
                int32_t mid = -1;
                const f32x2 ox2 = splat2(o.x), oy2 = splat2(o.y), oz2 = splat2(o.z);
                const f32x2 dx2 = splat2(d.x), dy2 = splat2(d.y), dz2 = splat2(d.z);
                const uint32_t n_pairs = (ob.tri_count + 1u) >> 1;
                for (uint32_t p = 0; p < n_pairs; ++p) {
                    const TriPairRec tr = S.tri_pairs[ob.pair_begin + p];  // wave-uniform -> scalar loads
                    const f32x2 e1x = ld2(tr.e1x), e1y = ld2(tr.e1y), e1z = ld2(tr.e1z);
                    const f32x2 e2x = ld2(tr.e2x), e2y = ld2(tr.e2y), e2z = ld2(tr.e2z);
                    // pvec = ray.direction.cross(va_vc)                          (mod.rs:563)
                    const f32x2 px = dy2 * e2z - e2y * dz2, py = dz2 * e2x - e2z * dx2, pz = dx2 * e2y - e2x * dy2;
                    const f32x2 determinant = (e1x * px + e1y * py) + e1z * pz;  // mod.rs:564
                    const f32x2 inv_det = 1.0f / determinant;                    // mod.rs:576
                    const f32x2 tx = ox2 - ld2(tr.ax), ty = oy2 - ld2(tr.ay), tz = oz2 - ld2(tr.az);  // mod.rs:577
                    const f32x2 u = ((tx * px + ty * py) + tz * pz) * inv_det;                        // mod.rs:578
                    // qvec = tvec.cross(va_vb)                                   (mod.rs:583)
                    const f32x2 qx = ty * e1z - e1y * tz, qy = tz * e1x - e1z * tx, qz = tx * e1y - e1x * ty;
                    const f32x2 v = ((dx2 * qx + dy2 * qy) + dz2 * qz) * inv_det;     // mod.rs:584
                    const f32x2 dist = ((e2x * qx + e2y * qy) + e2z * qz) * inv_det;  // mod.rs:589
                    const f32x2 uv = u + v;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {  // list order: the earlier triangle keeps ties (mod.rs:598)
                        // the reference's `continue` tests, negated one by one (NaN behaves the same)
                        const bool keep = !(f_abs(determinant[hf]) < 1e-4f) && !(u[hf] < 0.0f || u[hf] > 1.0f) &&
                                          !(v[hf] < 0.0f || uv[hf] > 1.0f) && !(dist[hf] <= 0.0f);
                        if (keep && dist[hf] < mt) {
                            mt = dist[hf];
                            mid = (int32_t)(ob.tri_begin + 2u * p) + hf;
                        }
                    }
                }
                if (sph_hit && mid >= 0 && mt < best_t) {
                    best_t = mt;
                    best_id = (int32_t)S.n_objs + mid;
                }
            }
        }
    }
    HitRec h;
    h.t = best_t;
    h.id = best_id;
    return h;
}

// ---------------------------------------------------------------------------------------------
struct PathRay {
    vec3 o, d;
    vec3 thr;       // product of the colours (and split / roulette weights) above this ray
    uint32_t pix;   // framebuffer index (global, also the RNG counter)
    uint32_t meta;  // pack_meta(sample, depth, branch)
};

struct ShadeOut {
    int n_rays;          // 0, 1 or 2 continuation rays, all starting at x
    vec3 x;              // hit point
    vec3 d0, thr0;       // first continuation (on a refract split: the reflected ray, branch 2b)
    vec3 d1, thr1;       // second continuation (refract split only: the transmitted ray, branch 2b+1)
    uint32_t meta0, meta1;
    vec3 contrib;        // throughput * emission of the hit object (zero when it does not emit)
    bool emits;
};

// hit point, normal and material of a hit, as intersect_sphere / Triangle::intersect return them
struct Surface {
    vec3 x, n;
    vec3 color, emission;
    float max_refl, inv_max_refl;
    uint32_t reflect;
};

__device__ __forceinline__ Surface fetch_surface(const DevScene &S, vec3 o, vec3 d, HitRec h) {
    Surface s;
    const bool is_tri = h.id >= (int32_t)S.n_objs;
    uint32_t obj = (uint32_t)h.id;
    vec3 tn = mk(0.0f, 0.0f, 0.0f);
    if (is_tri) {
        const TriShade ts = S.tri_shade[h.id - (int32_t)S.n_objs];
        tn = mk(ts.nx, ts.ny, ts.nz);
        obj = ts.owner;
    }
    const MatRec m = S.mats[obj];
    s.x = o + d * h.t;  // mod.rs:430 / mod.rs:604
    s.n = is_tri ? tn : normalize(s.x - mk(m.px, m.py, m.pz));
    s.color = mk(m.cr, m.cg, m.cb);
    s.emission = mk(m.er, m.eg, m.eb);
    s.max_refl = m.max_refl;
    s.inv_max_refl = m.inv_max_refl;
    s.reflect = m.reflect;
    return s;
}

// One radiance() invocation after its intersect_scene call returned Some (mod.rs:665-789).
__device__ __forceinline__ void shade_hit(const DevScene &S, const FrameParams &F, const PathRay &in, HitRec h,
                                          ShadeOut &out) {
    const Surface sf = fetch_surface(S, in.o, in.d, h);
    const vec3 d = in.d;
    const vec3 n = sf.n;
    const vec3 nl = dot(n, d) < 0.0f ? n : n * -1.0f;  // normal_towards_ray
    const uint32_t sample = meta_sample(in.meta), depth = meta_depth(in.meta), branch = meta_branch(in.meta);
    const uint32_t new_depth = depth + 1u;
    const u32x4 rnd = draw_block(((uint64_t)F.seed_hi << 32) | F.seed_lo, in.pix, sample, (branch << 8) | new_depth);

    out.emits = (sf.emission.x != 0.0f) || (sf.emission.y != 0.0f) || (sf.emission.z != 0.0f);
    out.contrib = in.thr * sf.emission;
    out.x = sf.x;
    out.meta0 = out.meta1 = pack_meta(sample, new_depth, branch);

    // Russian roulette, mod.rs:677-683 (the draw is taken first: `rand01() < max_reflection && ...`)
    vec3 color = sf.color;
    bool alive = true;
    if (new_depth > 5u) {
        if (unit_f32(rnd.a) < sf.max_refl && new_depth < (uint32_t)kMaxDepth)
            color = color * sf.inv_max_refl;
        else
            alive = false;
    }
    const vec3 thr = in.thr * color;
    int n_rays = alive ? 1 : 0;
    vec3 d0 = d, thr0 = thr, d1 = d, thr1 = thr;

    if (sf.reflect == kDiffuse) {  // mod.rs:687-715
        const float r1 = (2.0f * 3.141592653589793f) * unit_f32(rnd.b);
        const float r2 = unit_f32(rnd.c);
        const float r2s = f_sqrt(r2);
        const vec3 w = nl;
        const vec3 uu = normalize(cross(f_abs(w.x) > 0.1f ? mk(0.0f, 1.0f, 0.0f) : mk(1.0f, 0.0f, 0.0f), w));
        const vec3 vv = cross(w, uu);
        float sn, cs;
        sincos_f32(r1, &sn, &cs);
        d0 = normalize(uu * cs * r2s + vv * sn * r2s + w * f_sqrt(1.0f - r2));
    } else {
        const vec3 refl = d - n * 2.0f * dot(n, d);  // mod.rs:722-723 / 733-734
        d0 = refl;
        if (sf.reflect == kRefract) {  // mod.rs:729-788
            const bool into = dot(n, nl) > 0.0f;
            const float nc = 1.0f, nt = 1.5f;
            const float nnt = into ? nc / nt : nt / nc;
            const float ddn = dot(d, nl);
            const float cos2t = 1.0f - (nnt * nnt) * (1.0f - ddn * ddn);
            if (!(cos2t < 0.0f)) {  // otherwise total internal reflection: the reflected ray alone
                const vec3 tdir = normalize(d * nnt - n * ((into ? 1.0f : -1.0f) * (ddn * nnt + f_sqrt(cos2t))));
                const float a = nt - nc, bb = nt + nc;
                const float r0 = a * a / (bb * bb);
                const float c = 1.0f - (into ? -ddn : dot(tdir, n));
                const float c5 = c * ((c * c) * (c * c));  // powi(5)
                const float re = r0 + (1.0f - r0) * c5;
                const float tr = 1.0f - re;
                const float p = 0.25f + 0.5f * re;
                const float rp = re / p;
                const float tp = tr / (1.0f - p);
                if (new_depth > 2u) {  // mod.rs:760-774: one of the two, chosen with probability p
                    const bool pick_refl = unit_f32(rnd.b) < p;
                    d0 = pick_refl ? refl : tdir;
                    thr0 = thr * (pick_refl ? rp : tp);
                } else {  // mod.rs:775-786: both subtrees
                    thr0 = thr * re;
                    d1 = tdir;
                    thr1 = thr * tr;
                    out.meta0 = pack_meta(sample, new_depth, 2u * branch);
                    out.meta1 = pack_meta(sample, new_depth, 2u * branch + 1u);
                    n_rays = alive ? 2 : 0;
                }
            }
        }
    }
    out.d0 = d0;
    out.thr0 = thr0;
    out.d1 = d1;
    out.thr1 = thr1;
    out.n_rays = n_rays;
}

// render_pixel's per-sample ray (mod.rs:805-843) for framebuffer index `pix`, sample `s`
__device__ __forceinline__ PathRay primary_ray(const FrameParams &F, uint32_t pix, uint32_t s) {
    const uint32_t y = F.height - 1u - pix / F.width;
    const uint32_t x = pix % F.width;
    const float ysub = (float)((s / 2u) % 2u);
    const float xsub = (float)(s % 2u);
    const u32x4 rnd = draw_block(((uint64_t)F.seed_hi << 32) | F.seed_lo, pix, s, 0u);
    const float r1 = 2.0f * unit_f32(rnd.a);
    const float r2 = 2.0f * unit_f32(rnd.b);
    const float xfilter = tent(r1);
    const float yfilter = tent(r2);
    const float sx = ((float)x + 0.5f * (0.5f + xsub + xfilter)) / (float)F.width - 0.5f;
    const float sy = ((float)y + 0.5f * (0.5f + ysub + yfilter)) / (float)F.height - 0.5f;
    const vec3 lens = mk(F.lens_x, F.lens_y, F.lens_z);
    const vec3 sensor_pos = mk(F.cam_px, F.cam_py, F.cam_pz) + mk(F.su_x, F.su_y, F.su_z) * sx + mk(F.sv_x, F.sv_y, F.sv_z) * sy;
    PathRay r;
    r.o = lens;
    r.d = normalize(lens - sensor_pos);
    r.thr = mk(1.0f, 1.0f, 1.0f);
    r.pix = pix;
    r.meta = pack_meta(s, 0u, 1u);
    return r;
}

// Radiance is summed per pixel in unsigned 32.32 fixed point: integer adds commute, so the image does
// not depend on the order in which paths finish (any queue order, any number of GPUs), and the sum is
// exact to 2^-32 per contribution.  v is finite and >= 0 (throughput and emission are non-negative).
__device__ __forceinline__ uint64_t to_fixed(float v) {
    const float c = v < 4294967040.0f ? v : 4294967040.0f;  // saturate; NaN falls through to cvt -> 0
    const uint32_t hi = (uint32_t)c;                         // truncation, exact
    const float frac = c - (float)hi;                        // exact
    const uint32_t lo = (uint32_t)(frac * 4294967296.0f);    // exact scaling, truncation
    return ((uint64_t)hi << 32) | lo;
}

#endif  // __HIPCC__

}  // namespace pt
