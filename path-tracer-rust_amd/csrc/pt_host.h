// pt_host.h — host-side setup arithmetic of the path (camera basis, Mesh::new bounds, scene flattening).
#pragma once

#include <string>
#include <vector>

#include "../../include/ptrace.h"
#include "pt_device.h"

namespace pt {

void set_error(const std::string &m);

namespace host {

struct FlatScene {
    std::vector<ObjRec> objs;
    std::vector<ObjPairRec> obj_pairs;
    std::vector<TriPairRec> tri_pairs;
    std::vector<MatRec> mats;
    std::vector<TriShade> tri_shade;
    std::vector<BvhNode> bvh_nodes;
    std::vector<BvhNode4> bvh_nodes4;  // the same trees four children wide (two levels folded into one)
    std::vector<SphPairRec> sph_pairs;
    std::vector<FlatPairRec> flat_pairs;
    std::vector<CandPairRec> cand_pairs;  // [0, n_other_pairs): records without a filter
    std::vector<uint32_t> rank_id;
    std::vector<SurfRec> surf;  // by rank
    std::vector<uint32_t> tri_rank;  // rank of triangle k (the inverse of rank_id over the triangles)
    std::vector<BvhMeshRec> bvh_meshes;  // the meshes that have a BVH, in visiting order (last object first)
    uint32_t n_other_pairs = 0;
    uint32_t n_flat_exact = 0;  // flat_pairs [0, n_flat_exact) have sign_exact set (they come first)
    bool cand_ok = false;  // the scene can use the candidate scan (the records of its meshes without a BVH are numbered in
                           // 9 bits; meshes with a BVH are walked: k_pass_cand<.., BVH>)
    uint32_t bvh_stack = 0;      // traversal-stack entries the deepest tree needs (<= kBvhStack)
    uint32_t bvh_pair_base = 0;  // first TriPairRec that is a BVH leaf
    uint32_t bvh_pair_span = 0;  // leaves lie in [bvh_pair_base, bvh_pair_base + bvh_pair_span)
};

// CameraData::{lens_center, orthogonals} — src/render/mod.rs:211-232
void camera_basis(const pt_camera &cam, float lens_center[3], float su[3], float sv[3]);
// Mesh::new bounding sphere — src/render/mod.rs:450-499
void mesh_bounding_sphere(const pt_triangle *tris, uint32_t n, float center[3], float *radius);
// Mesh::new's bounding_box: bounding_box_to_triangles over the AABB of the (object-local) triangles — src/render/mod.rs:452-476,501-536
void mesh_bounding_box(const pt_triangle *tris, uint32_t n, pt_triangle out[12]);
// the 12 triangles of one object's bounding box as 6 pair records in list order (world space: Triangle::transformed,
// mod.rs:546-552, then the edge subtractions of mod.rs:560-561), ids 0..11 — what SceneObjectData::intersect_bounds scans
void box_pair_records(const pt_triangle box[12], const float position[3], TriPairRec out[6]);
// validate + flatten (see pt_device.h for the record layouts); false + message on malformed input
// `cam` only widens the distance bound that sizes the BVH box padding (ray origins include the lens centre)
bool bvh_refs_fit(uint64_t n_nodes, uint64_t n_pair_records);

// How a wavefront frame is cut into passes and streams (render_wavefront, pt_api.hip): pure arithmetic, tested on the CPU.
struct PassPlanIn {
    uint64_t npix = 0;          // pixels of the call (or of its part)
    uint32_t spp = 0;           // samples per pixel of the frame
    uint64_t want = 0;          // primary rays per pass that are asked for
    bool want_is_default = true;  // `want` is the library's choice (it may be halved to fit index ranges)
    bool stack_form = false;    // k_pass_cand: a stack of waiting rays per wave, passes sized by time
    bool stack_park = false;    // ... with walks: a parking area per wave in the second container
    bool cand_scan = false, has_bvh = false;
    uint64_t streams = 0;       // PT_STREAMS (0: derived)
    uint32_t per_stream = 0;    // PT_PER_STREAM (0: default)
    uint32_t wave_stack = 0;    // PT_WAVE_STACK (0: kWaveStackMax)
    uint32_t n_cus = 0;         // compute units of the device (0: unknown - no whole-rounds nudge)
    uint32_t groups_per_cu = 4; // workgroups of the pass kernel a compute unit holds at a time (its waves per SIMD)
    size_t stack_budget = 0;    // stack_form, default pass: bytes the streams' stacks may take (0: unbounded)
};
struct PassPlan {
    uint32_t spp_pass = 0;  // samples of a pixel per pass
    uint32_t m = 0;         // pixels per stream
    uint32_t K = 0;         // streams (workgroups per launch)
    uint32_t cap = 0;       // slots of a stream's slice of a queue container (k_pass_cand: 4 x the waves' stack)
    size_t bytes0 = 0, bytes1 = 0;  // the two containers
};
enum { kPlanOk = 0, kPlanRetry = 1, kPlanTooLarge = 2 };
// kPlanRetry: the plan does not fit (the stacks' budget, or 32-bit slot indices) - try again with in.want = *want_next
int plan_pass(const PassPlanIn &in, PassPlan &out, uint64_t *want_next);
// Samples of a pixel in the next pass (wavefront) / round (megakernel) of a frame whose passes follow the scene (pt_api.hip:
// render_wavefront, render_mega).  `rate`: primary samples per millisecond the last timed pass went through (0: nothing
// measured yet - the pass is `probe` samples in all); `s_prev`: samples per pixel of the pass before (0: none); `left`: samples
// per pixel still to be issued (> 0); `max_pass`: what one pass may hold.  As many samples as `rate` fits into `target_ms`, at
// most sixteen times the pass before, the rest of the frame in equal passes - each up to a fifth longer than the target
// rather than one pass more.
uint32_t next_pass_samples(double rate, double target_ms, uint64_t npix, uint64_t probe, uint32_t s_prev, uint32_t left,
                           uint32_t max_pass);
bool flatten_scene(const pt_camera &cam, const pt_object *objs, uint32_t n_objs, const pt_triangle *tris,
                   uint32_t n_tris, FlatScene &out, std::string &err);

}  // namespace host
}  // namespace pt
