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
    uint32_t pair_begin;  // first TriPairRec of the mesh
    uint32_t pair_count;  // number of TriPairRec of the mesh (>= (tri_count+1)/2; BVH leaves may be half full)
    int32_t bvh_root;     // kNoBvh, or the root reference of the mesh's BVH (>= 0 node index, < 0: ~pair index)
    float rr_in;          // mesh: a point of the ray within sqrt(rr_in) of the sphere's centre proves the gate passes
                          // (intersect_scene_dev); negative when no such shortcut is offered
    uint32_t pad1;
};

// Two consecutive objects of the scene in the order intersect_scene visits them (reverse index order: half 0
// is the higher index), component-interleaved like TriPairRec: the sphere / bounding-sphere arithmetic of both
// runs on packed instructions.  A scene with an odd object count gets a filler half that can never be hit
// (rr = -inf makes the discriminant negative).
struct alignas(16) ObjPairRec {
    float cx[2], cy[2], cz[2];  // sphere: position; mesh: bounding_sphere.position + position (mod.rs:268)
    float rr[2];                // radius.powi(2) (mod.rs:416)
    uint32_t kind[2];
    uint32_t pair_begin[2];     // first TriPairRec of a mesh
    uint32_t pair_count[2];
    int32_t bvh_root[2];
    uint32_t obj[2];            // object index
    uint32_t admit[2];          // bit 0: mesh whose bounding sphere is so large that a wave of rays practically never misses
                                // it as a whole: the speculative scan skips the gate arithmetic (see scan_scene);
                                // bits 1-2: 1 + the axis both edge vectors of every triangle of the mesh are exactly
                                // zero along (a mesh in an axis-aligned plane: test_pair_planar), 0 = none
};

// Two consecutive triangles of one mesh (world space), component by component, read with a wave-uniform
// index.  Each component pair sits in two adjacent dwords, so after the scalar load it is an aligned SGPR
// pair and can be the scalar operand of a packed VALU instruction (v_pk_mul_f32 / v_pk_add_f32): one ray is
// tested against both triangles with one instruction per arithmetic step.  On gfx950 a packed f32 instruction
// occupies its SIMD for 4 cycles, a scalar one for 2 with VGPR sources but for 4 with an SGPR source (measured:
// profiles/r02_valu_issue_costs.json) - and the scene operands ARE SGPRs here, so the packed form does two
// triangles in the 4 cycles one would take; every half is still an IEEE mul/add/sub.
// A mesh with an odd triangle count gets an all-zero second triangle: its determinant is 0, which the
// reference's first test rejects (mod.rs:571).
struct alignas(16) TriPairRec {
    float ax[2], ay[2], az[2];     // tri.a + offset                            (mod.rs:548)
    float e1x[2], e1y[2], e1z[2];  // va_vb = (tri.b+offset) - (tri.a+offset)   (mod.rs:560)
    float e2x[2], e2y[2], e2z[2];  // va_vc                                      (mod.rs:561)
    uint32_t id[2];                // flattened triangle index of each half (kNoTri for a filler half)
};

// BVH over the triangles of one large mesh: a node stores the (padded) boxes of its two children, so one
// 64-byte fetch decides both.  Child reference >= 0: node index; < 0: ~index of a TriPairRec leaf.
// Nodes are staged in LDS by every workgroup (per-lane, divergent indexing); leaves stay in global memory.
// The two boxes are stored component-interleaved so that both are tested by one packed instruction per step.
struct alignas(16) BvhNode {
    float lox[2], loy[2], loz[2];
    float hix[2], hiy[2], hiz[2];
    int32_t c[2];
    uint32_t pad[2];
};
// The same tree FOUR children wide, for the walk queue of k_pass_cand / k_mega (bvh_closest_queue): the binary tree with
// two levels folded into one wherever a child is an inner node (the child with the larger box first), so a walk visits
// half the nodes - what a (ray, node) item costs is mostly its trip through the queue (pop, the owner's ray constants,
// pushes), not its box arithmetic.  The boxes are the binary tree's (padded the same way), a child reference is a
// node4 index (>= 0) or the binary tree's leaf reference (< 0).  A node with fewer than four children fills up with boxes
// of NaN: min / max ignore a NaN, so such a box has tin = 0, tout = NaN, and `tin <= tout * (1 + e)` is false - never hit;
// their reference repeats child 0's (harmless if it ever were).  Component-interleaved in pairs (0,1) and (2,3) like BvhNode.
struct alignas(16) BvhNode4 {
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    int32_t c[4];
    uint32_t pad[4];
};
static_assert(sizeof(BvhNode4) == 128, "BvhNode4 layout");
constexpr int32_t kNoBvh = 0x7fffffff;
constexpr uint32_t kNoTri = 0x7fffffffu;
constexpr uint32_t kBvhStack = 24;          // per-lane traversal stack entries (LDS; u16 each when nodes are staged)
constexpr uint32_t kBvhMinTris = 16;        // meshes with fewer triangles are scanned linearly
// A BVH leaf is a run of up to kBvhLeafPairs consecutive pair records.  A leaf reference is ~(first record << kBvhLeafBits | count - 1).
// Dense leaf batches make a pair test cheaper than a depth-first node step, so the last levels of the tree are better
// spent as tests (k_pass_bvh, depth-first walks: 1 record per leaf 14.4, 2: 15.4, 4: 15.5, 8: 15.2 G bounces/s on
// mesh.json).  With the box tests dense too (k_pass_cand's walk queue) the balance moves back a little:
// 1: 19.5, 2: 20.0, 3: 20.2, 4: 19.8; and without levels (round 3) 2: 26.5, 3: 26.8, 4: 26.2 (640 000 triangles: 25.9 / 25.6 / 25.3).
// With the four-wide tree (round 4: half the node visits) smaller leaves pay: 1: 27.7, 2: 28.4, 3: 26.0, 4: 25.7 (binary tree,
// same build: 2: 26.3, 3: 26.5).
#ifndef PT_BVH_LEAF_PAIRS
#define PT_BVH_LEAF_PAIRS 2
#endif
constexpr uint32_t kBvhLeafPairs = PT_BVH_LEAF_PAIRS;
constexpr uint32_t kBvhLeafBits = kBvhLeafPairs <= 2 ? 1 : (kBvhLeafPairs <= 4 ? 2 : (kBvhLeafPairs <= 8 ? 3 : 4));
PT_HD uint32_t leaf_first(uint32_t code) { return code >> kBvhLeafBits; }
PT_HD uint32_t leaf_count(uint32_t code) { return (code & ((1u << kBvhLeafBits) - 1u)) + 1u; }
constexpr uint32_t kBvhMaxLdsNodes = 512;   // 32 KiB of nodes at most are staged in LDS (+ 24 KiB of stacks < 64 KiB)

// ---- candidate scan (k_pass): see intersect_cand -----------------------------------------------------------------
// Spheres of the scene only, two per record in the order intersect_scene visits them; `rank` = position of the object in
// the reference's visiting sequence (see DevScene.rank_id).  An odd count gets a filler half (rr = -inf: never hit).
struct alignas(16) SphPairRec {
    float cx[2], cy[2], cz[2], rr[2];
    uint32_t rank[2];
    uint32_t pad[2];
};
// Conservative filter of two pair records (TriPairRec) whose triangles lie in a plane perpendicular to axis `axis`
// (a = axis, b = a+1, c = a+2 mod 3): the plane coordinate, and the rectangle that bounds the record's two triangles in
// the other two coordinates as centre / half extent, the half extent already widened by `pad` - the bound on how far from
// the exact triangle the f32 Moller-Trumbore arithmetic can still accept a ray that is not grazing (|d_a| >= 1/64), plus
// this filter's own roundoff (pt_host.cpp).  A filler half has pair = kNoPair and an empty rectangle.
struct alignas(16) FlatPairRec {
    float pc[2];          // plane coordinate along a
    float cb[2], hb[2];   // centre, padded half extent along b
    float cc[2], hc[2];   // centre, padded half extent along c
    float tpad[2];        // slack on the distance along the ray
    uint32_t pair[2];     // TriPairRec index
    uint32_t axis;        // the same for both halves
    uint32_t sign_exact;  // 1: the SIGN of the distance Triangle::intersect computes for these triangles is that of
                          // (plane - origin) * direction along the axis, exactly (see filter_flat; pt_host.cpp checks the
                          // condition) - the filter then drops every ray that does not move towards the plane
};
constexpr uint32_t kNoPair = 0xffffffffu;
#ifndef PT_GRAZING_INV
#define PT_GRAZING_INV 64.0f
#endif
constexpr float kGrazing = 1.0f / PT_GRAZING_INV;  // |d_a| below this: the filter does not judge, the exact test does
constexpr uint32_t kCandQueueCap = 256;   // per-wave candidate ring (u16 entries): 63 left over + 2 x 64 pushed by one filter step
                                          // = 191 at most; a power of two, so that positions wrap with one v_and instead of
                                          // two compare / select pairs per push (4-cycle instructions, eight pushes per trip)
static_assert((kCandQueueCap & (kCandQueueCap - 1u)) == 0u && kCandQueueCap >= 192u, "ring positions wrap by masking");
constexpr uint32_t kCandMaxPairs = 512;   // candidate records are numbered in 9 bits of a queue entry
// What the exact test of a candidate needs, in one record (a copy of the TriPairRec with ids replaced by visiting ranks,
// plus the bounding-sphere gate of the mesh the two triangles belong to): 7 rows of 16 bytes, gathered per lane from
// LDS (staged by the workgroup when the scene's records fit) or from global memory.
struct alignas(16) CandPairRec {
    float ax[2], ay[2], az[2];
    float e1x[2], e1y[2], e1z[2];
    float e2x[2], e2y[2], e2z[2];
    uint32_t id[2];               // visiting rank of each triangle (kNoTri for a filler half)
    float gx, gy, gz, grr;        // gate: bounding_sphere.position + position, radius^2 (mod.rs:267-273)
    float grr_in;                 // see ObjRec.rr_in
    uint32_t pad[3];
};
static_assert(sizeof(CandPairRec) == 112, "7 rows of 16 bytes");

// per-object material record, gathered per lane in shade
struct alignas(16) MatRec {
    float cr, cg, cb, max_refl;      // color, max(color)                       (mod.rs:667-668)
    float er, eg, eb, inv_max_refl;  // emmission, 1.0/max_reflection            (mod.rs:679)
    float px, py, pz;                // object position (sphere centre for the normal, mod.rs:431)
    uint32_t reflect;                // kDiffuse / kSpecular / kRefract
};

// Everything the shading step needs about the primitive of one visiting rank (candidate scan: the key of a hit holds
// the rank), in one 48-byte gather instead of rank -> id -> TriShade -> MatRec: for a triangle its normal, for a sphere
// its centre; then the object's material as in MatRec.
struct alignas(16) SurfRec {
    float vx, vy, vz;  // triangle: va_vb.cross(va_vc).normalize() (mod.rs:605); sphere: position (mod.rs:431)
    uint32_t kind;     // bits 0-1: reflect type, bit 8: triangle
    float cr, cg, cb, max_refl;
    float er, eg, eb, inv_max_refl;
};

// per-triangle shading record, gathered per lane in shade
struct alignas(16) TriShade {
    float nx, ny, nz;  // va_vb.cross(va_vc).normalize()  (mod.rs:605)
    uint32_t owner;    // object index
};

// a mesh that has a BVH, in intersect_scene's visiting order (k_pass_cand: gate + walk per entry)
struct alignas(16) BvhMeshRec {
    float cx, cy, cz, rr;  // bounding sphere in world space, radius squared
    int32_t root;          // node index (>= 0) or a leaf reference (< 0: the whole mesh is one leaf)
    int32_t root4;         // the same in the four-wide tree (DevScene.bvh_nodes4)
    uint32_t pad[2];
};
static_assert(sizeof(BvhMeshRec) == 32, "BvhMeshRec layout");

struct DevScene {
    const ObjRec *objs;
    const ObjPairRec *obj_pairs;  // ceil(n_objs / 2) records
    const TriPairRec *tri_pairs;
    const MatRec *mats;
    const TriShade *tri_shade;
    const BvhNode *bvh_nodes;
    const BvhNode4 *bvh_nodes4;  // the four-wide form of the same trees (walk queue)
    uint32_t n_bvh_nodes4;
    uint32_t n_objs;
    uint32_t n_tris;
    uint32_t n_bvh_nodes;  // 0: no mesh of the scene has a BVH (or BVH use is switched off for this frame)
    uint32_t bvh_in_lds;   // bit 0: nodes are staged in LDS; bit 1: child references fit 16 bits (u16 traversal stacks);
                           // bit 2 (set by launch_pass for k_pass_cand with walks): that kernel stages the nodes in LDS and its
                           // walk queues hold kWalkQueueBytesStaged
    uint32_t bvh_pair_base;  // first TriPairRec that is a BVH leaf (leaf references on a u16 stack are relative to it)
    uint32_t bvh_stack;      // traversal-stack entries the deepest tree of the scene needs (<= kBvhStack): k_pass_cand sizes its LDS by it
    uint32_t leaf_quorum;    // BVH walk: lanes on a leaf that send the wave to the triangle code (see bvh_closest)
    uint32_t planar;         // 0: every triangle through the general test_pair (PT_FLAG_NO_BVH)
    // candidate scan (scenes without BVH meshes): spheres, filters of the flat pair records, the other pair records
    // (always candidates), and the visiting ranks: rank of a sphere / triangle = its position in the sequence in which
    // intersect_scene + Triangle::intersect visit primitives (objects from the last to the first, triangles in list
    // order), so that "closest, earliest visited among equals" (mod.rs:598,649) is one integer minimum over
    // (distance bits << 32 | rank).  rank_id maps a rank back to the hit id of HitRec.
    const SphPairRec *sph_pairs;
    const FlatPairRec *flat_pairs;  // FlatPairRec.pair = index into cand_pairs
    const CandPairRec *cand_pairs;  // the pair records of every mesh without a BVH
    const uint32_t *rank_id;        // [n_objs + n_tris]
    const SurfRec *surf;            // [n_objs + n_tris], by rank
    const uint32_t *tri_rank;       // [n_tris]: rank of a triangle (BVH walks report triangle indices)
    const BvhMeshRec *bvh_meshes;   // [n_bvh_meshes], visiting order
    uint32_t n_bvh_meshes;
    uint32_t n_sph_pairs, n_flat_pairs, n_cand_pairs;
    uint32_t n_other_pairs;         // cand_pairs [0, n_other_pairs) have no filter: candidates for every ray
    uint32_t n_flat_exact;          // flat_pairs [0, n_flat_exact) have sign_exact set
    uint32_t glass_defer_ok;        // k_pass_cand may defer glass hits to dense batches (PT_GLASS_DEFER=1; default: shaded in place)
    uint32_t nodes_in_lds_ok;       // k_pass_cand with walks may stage the BVH nodes in LDS when they fit (PT_NODES_LDS=0: never)
    uint32_t cand_scan;             // 1: k_pass uses the candidate scan
    uint32_t cand_staged;           // 1: the workgroup holds cand_pairs in LDS
    uint32_t surf_staged;           // 1: ... and surf (set per launch: launch_pass)
    uint32_t surf_head;             // else: this many leading ranks of surf (the objects visited first) are staged
    uint32_t walk_queue_cap;        // 0, or a smaller capacity for the walk queue than its LDS area holds (tests: >= 128)
#ifdef PT_WALK_STATS
    unsigned long long *stats;      // [16] counters of a -DPT_WALK_STATS build (tools/walk_stats.py): never in the shipped library
#endif
#ifdef PT_PHASE_STATS
    unsigned long long *phase_stats;  // [kPhCount][3] + [2] (wave lifetimes: s_memtime, s_memrealtime) of a -DPT_PHASE_STATS build
#endif
};
#ifdef PT_PHASE_STATS
// Diagnostic build only (tools/phase_budget.py -> profiles/r03_*_phase_budget.json; never in the shipped library): the
// wave's lifetime inside k_pass_cand split by PHASE.  PT_PHASE(id) stamps s_memtime (shader cycles) and books the cycles
// since the last stamp to the phase the wave was in, per wave in LDS (one active lane does the bookkeeping); the phase
// being entered gets one entry and popcount(EXEC) lanes.  The totals go to DevScene.phase_stats when the workgroup ends,
// together with the wave's whole lifetime in s_memtime and s_memrealtime ticks (100 MHz): the clock the chip held.
enum : uint32_t {
    kPhOther = 0, kPhLoad, kPhSpheres, kPhFilter, kPhBatch, kPhFinish, kPhSurface, kPhRng, kPhDiffuse, kPhSpecular, kPhGlass,
    kPhAppend, kPhDefer, kPhBarrier, kPhWants, kPhWalkGate, kPhWalkBox, kPhWalkLeaf, kPhPrimary, kPhEmit,
    // round 4, the EXEC budget: the divergent blocks INSIDE the phases above, stamped on their own (lanes at entry = the lanes
    // that take the branch): a candidate's push to the ring, the second ray of a refract split, the roulette's rescale, the
    // fixed-point adds of a hit on an emitter
    kPhPush, kPhAppend2, kPhRoulette, kPhEmitAdd, kPhCount
};
struct PhaseLds {
    unsigned long long state[4];              // per wave of the workgroup: current phase << 32 | last stamp
    unsigned long long cyc[4][kPhCount];      // cycles
    unsigned long long cnt[4][kPhCount];      // entries << 32 | lanes at entry
};
#define PT_PHASE(id) ::pt::phase_to(id)
#define PT_PHASE_N(id, n) ::pt::phase_to(id, n)
// keeps a value's computation in front of the next stamp (an empty asm that reads it)
#define PT_PHASE_PIN(v) asm volatile("" ::"v"(v))
#else
#define PT_PHASE(id) do { } while (0)
#define PT_PHASE_N(id, n) do { } while (0)
#define PT_PHASE_PIN(v) do { } while (0)
#endif

#ifdef PT_WALK_STATS
// v is wave-uniform; one lane of the active ones adds it
#define PT_WSTAT(S, i, v)                                                                               \
    do {                                                                                                \
        const unsigned long long v_ = (unsigned long long)(v);                                          \
        const uint64_t m_ = __builtin_amdgcn_ballot_w64(true);                                          \
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m_)) atomicAdd((S).stats + (i), v_);       \
    } while (0)
#else
#define PT_WSTAT(S, i, v) do { } while (0)
#endif

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
    uint32_t debug;                // ablation switches for profiling builds (PT_DEBUG env): 0 in production
    uint32_t chunk_pixels, chunk_first, chunk_step;  // interleaved partition (chunk_step <= 1: contiguous band)
    uint32_t n_streams;            // wavefront: K ray streams; stream b owns the call's pixels b, b+K, b+2K, ...
    uint32_t k_begin;              // first call-local pixel of this launch's part of the call (large calls are rendered in
                                   // parts of about a million pixels: pt_ctx_render); npix counts the part's pixels
    uint32_t depth0;               // `depth` argument of the radiance() calls the primary rays stand for: 0 for a frame (mod.rs:844)
    // pt_ctx_radiance: every "primary ray" of the call is this one ray (radiance(&ray, depth0, scene), test.rs:146-183); the
    // call has one pixel, whose index idx_begin is only the RNG counter, and spp samples
    uint32_t probe;
    float probe_ox, probe_oy, probe_oz, probe_dx, probe_dy, probe_dz;
};

// framebuffer index of the k-th pixel of this call (identity + idx_begin for a contiguous band)
template <class Params>
PT_HD uint32_t global_pixel(const Params &F, uint32_t k) {
    k += F.k_begin;
    if (F.chunk_step <= 1u) return F.idx_begin + k;
    const uint32_t c = k / F.chunk_pixels, w = k - c * F.chunk_pixels;
    return F.idx_begin + (F.chunk_first + c * F.chunk_step) * F.chunk_pixels + w;
}

// the part of FrameParams the shading step needs (keeps k_shade's kernel arguments - SGPRs - small)
struct ShadeParams {
    uint32_t idx_begin, npix;
    uint32_t seed_lo, seed_hi;
    uint32_t debug;
    uint32_t s0;  // first sample index of the pass
    uint32_t chunk_pixels, chunk_first, chunk_step;
    uint32_t n_streams;
    uint32_t k_begin;
};

// Pixels of a call are dealt to the K streams round-robin: stream b owns the call-local pixels b, b+K, b+2K, ...
// (its j-th pixel is j*K + b).  Every stream then samples the whole band instead of one short run of a row, so the
// streams of a launch carry nearly the same number of rays whatever the picture shows and however small the band
// of this rank is (a stream of consecutive pixels that lies on the glass sphere carries 3-4x the rays of one on a
// wall, and a launch ends with its slowest stream).  Accumulators are kept stream-major (slot b*m + j) so that a
// stream's flush stays one contiguous run; k_resolve undoes the permutation.
// (K is usually a multiple of the image width - stream counts are nudged to whole rounds of 1024 resident workgroups, the bench
// frame is 1024 pixels wide - so a stream's pixels are ONE COLUMN of the image, every K / width-th row of it.  Rotating every
// block of K pixels by a pseudo-random amount, so that a stream's pixels are spread over the columns as well, was measured:
// cornell 45.0 against 46.3 G bounces/s, mesh.json 27.1 against 26.7 - the primary rays of a column agree about the walls
// they can reach, and whole filter pushes are skipped for their waves.  Not kept.  Runs of consecutive pixels, the most
// coherent form, bring the imbalance of round 1 back: 32.2 and 21.6.)
PT_HD uint32_t stream_pixel(uint32_t n_streams, uint32_t b, uint32_t j) { return j * n_streams + b; }
PT_HD uint32_t stream_pixel_count(uint32_t npix, uint32_t n_streams, uint32_t b) {
    return b < npix ? (npix - b + n_streams - 1u) / n_streams : 0u;
}

// Per-ray bookkeeping word as stored in a stream (4 B, the w lane of the throughput packet): a stream owns at
// most 1024 pixels and a pass holds at most 32767 samples of a pixel, so both are stored relative to the stream /
// pass: pixel-in-stream (10 bits) | sample-in-pass (15) | depth (4) | branch (3).
constexpr uint32_t kMaxPassSpp = 32767u;
constexpr uint32_t kBlock = 256;             // 4 waves of 64: the workgroup of the stream kernels
constexpr uint32_t kMaxStreamPixels = 1024;  // pixels owned by one stream (24 KiB of LDS accumulators at most)
constexpr uint32_t kRayBytes = 40;
constexpr uint32_t kWaveParkCap = 128;    // parked rays of a wave (k_pass_cand with walks: 63 left over + 64 new at most)
constexpr uint32_t kWaveParkBytes = kWaveParkCap * (kRayBytes + 8u);  // the ray (40 B) and its key so far (8 B)
constexpr uint32_t kWaveStackMax = 1024;  // slots of a wave's ray stack (k_pass_cand: a quarter of the stream's slice).  What may
                                          // ever wait is bounded by phi (k_pass_cand); with 1024 slots a wave that is about to
                                          // start primaries (fewer than 64 rays waiting, fewer than 64 parked) is never held back
PT_HD size_t queue_bytes(size_t K, uint32_t cap) { return K * (size_t)cap * kRayBytes; }
PT_HD uint32_t pack_word(uint32_t pix_in_stream, uint32_t sample_in_pass, uint32_t depth, uint32_t branch) {
    return (pix_in_stream & 1023u) | ((sample_in_pass & 32767u) << 10) | ((depth & 15u) << 25) | (branch << 29);
}
PT_HD uint32_t word_pix(uint32_t w) { return w & 1023u; }
PT_HD uint32_t word_sample(uint32_t w) { return (w >> 10) & 32767u; }
PT_HD uint32_t word_depth(uint32_t w) { return (w >> 25) & 15u; }
PT_HD uint32_t word_branch(uint32_t w) { return w >> 29; }

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

#ifdef PT_PHASE_STATS
__device__ __forceinline__ PhaseLds &phase_lds() {
    __shared__ PhaseLds s;
    return s;
}
// s_memtime through volatile asm: the builtin may be hoisted above a divergent branch whose other side stamps too, and the
// per-wave state word in LDS is written by whichever lane is the first active one - every access to it is volatile.
// One stamp = s_memtime, one LDS read, two LDS adds without return, one LDS write by one lane: about a dozen instructions
// (charged to the phase being entered).
__device__ __forceinline__ uint32_t phase_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return (uint32_t)t;
}
// `units`: what the phase's "lanes" column counts when that is not the active lanes (a batch's entries); ~0u = the lanes
__device__ __forceinline__ void phase_to(uint32_t id, uint32_t units = ~0u) {
    PhaseLds &P = phase_lds();
    const uint32_t w = threadIdx.x >> 6;
    const uint64_t m = __builtin_amdgcn_ballot_w64(true);
    const uint32_t now = phase_clock();
    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) {
        const unsigned long long st = *(volatile unsigned long long *)&P.state[w];
        atomicAdd(&P.cyc[w][(uint32_t)(st >> 32)], (unsigned long long)(now - (uint32_t)st));
        atomicAdd(&P.cnt[w][id], (1ull << 32) | (unsigned long long)(units != ~0u ? units : (uint32_t)__builtin_popcountll(m)));
        *(volatile unsigned long long *)&P.state[w] = ((unsigned long long)id << 32) | now;
    }
}
// first / last thing a workgroup does
__device__ __forceinline__ void phase_begin(unsigned long long *t0, unsigned long long *r0) {
    PhaseLds &P = phase_lds();
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t k = lane; k < kPhCount; k += 64u) {
        *(volatile unsigned long long *)&P.cyc[w][k] = 0ull;
        *(volatile unsigned long long *)&P.cnt[w][k] = 0ull;
    }
    *t0 = __builtin_amdgcn_s_memtime();
    *r0 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0u) *(volatile unsigned long long *)&P.state[w] = ((unsigned long long)kPhOther << 32) | (uint32_t)*t0;
}
__device__ __forceinline__ void phase_end(unsigned long long *out, unsigned long long t0, unsigned long long r0) {
    phase_to(kPhOther);
    PhaseLds &P = phase_lds();
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t k = lane; k < kPhCount; k += 64u) {
        const unsigned long long c = *(volatile unsigned long long *)&P.cyc[w][k], n = *(volatile unsigned long long *)&P.cnt[w][k];
        if (c) atomicAdd(out + 3u * k, c);
        if (n) {
            atomicAdd(out + 3u * k + 1u, n >> 32);
            atomicAdd(out + 3u * k + 2u, n & 0xffffffffull);
        }
    }
    if (lane == 0u) {
        atomicAdd(out + kPhCount * 3u, (unsigned long long)(__builtin_amdgcn_s_memtime() - t0));
        atomicAdd(out + kPhCount * 3u + 1u, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - r0));
    }
}
#endif

// two f32 lanes per register pair: arithmetic on it compiles to v_pk_mul_f32 / v_pk_add_f32 (each half an
// IEEE operation, no fusion under -ffp-contract=off)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) {
    f32x2 r = {v, v};
    return r;
}
// Load of a scene-table record whose index is the same for the whole wave.  The scene tables are written once by
// pt_ctx_set_scene and never by a kernel, so the load goes through the constant address space: the backend then
// always issues scalar loads (s_load_dwordx16/x4 into SGPRs), whatever it can or cannot prove about the kernel's own
// stores.  With a plain global load it falls back to per-lane loads of the same address as soon as a store through
// a pointer it cannot tell apart appears in the kernel (k_pass appends rays): 2x slower, the packed pair tests want
// their operands in SGPR pairs.
template <class T>
__device__ __forceinline__ T ld_uniform(const T *p) {
    static_assert(sizeof(T) % 16 == 0 && alignof(T) >= 16, "records are whole 16-byte rows");
    typedef uint32_t row_t __attribute__((ext_vector_type(4)));
    union {
        T v;
        row_t w[sizeof(T) / 16];
    } u;
    const __attribute__((address_space(4))) row_t *q =
        (const __attribute__((address_space(4))) row_t *)(unsigned long long)p;
#pragma unroll
    for (uint32_t i = 0; i < sizeof(T) / 16; ++i) u.w[i] = q[i];
    return u.v;
}
__device__ __forceinline__ f32x2 ld2(const float (&p)[2]) {
    f32x2 r = {p[0], p[1]};
    return r;
}
// f_rcp of both halves: the native reciprocals are per half, the Newton step runs packed (v_pk_fma_f32); every half goes
// through exactly f_rcp's sequence of IEEE operations, so the results are f_rcp's.
__device__ __forceinline__ f32x2 f_rcp2(f32x2 d) {
    const f32x2 one = splat2(1.0f);
    const f32x2 r0 = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2 e0 = __builtin_elementwise_fma(-d, r0, one);
    return __builtin_elementwise_fma(e0, r0, r0);
}

// ---------------------------------------------------------------------------------------------
// One ray against the two triangles of a TriPairRec: Triangle::intersect's arithmetic (mod.rs:559-594), both
// triangles per packed instruction.  Updates the mesh-local closest hit (mt, mid).
//   ORDERED: records are visited in list order, so strict '<' keeps the first of equal distances (mod.rs:598);
//   otherwise (BVH order) the tie is broken explicitly towards the smaller triangle index: same result.
template <bool ORDERED, class Rec = TriPairRec>
__device__ __forceinline__ void test_pair(const Rec &tr, f32x2 ox2, f32x2 oy2, f32x2 oz2, f32x2 dx2, f32x2 dy2,
                                          f32x2 dz2, float &mt, int32_t &mid) {
    const f32x2 e1x = ld2(tr.e1x), e1y = ld2(tr.e1y), e1z = ld2(tr.e1z);
    const f32x2 e2x = ld2(tr.e2x), e2y = ld2(tr.e2y), e2z = ld2(tr.e2z);
    // pvec = ray.direction.cross(va_vc)                                       (mod.rs:563)
    const f32x2 px = dy2 * e2z - e2y * dz2, py = dz2 * e2x - e2z * dx2, pz = dx2 * e2y - e2x * dy2;
    const f32x2 determinant = (e1x * px + e1y * py) + e1z * pz;                // mod.rs:564
    const f32x2 inv_det = f_rcp2(determinant);                                // mod.rs:576
    const f32x2 tx = ox2 - ld2(tr.ax), ty = oy2 - ld2(tr.ay), tz = oz2 - ld2(tr.az);  // mod.rs:577
    const f32x2 u = ((tx * px + ty * py) + tz * pz) * inv_det;                 // mod.rs:578
    // qvec = tvec.cross(va_vb)                                                (mod.rs:583)
    const f32x2 qx = ty * e1z - e1y * tz, qy = tz * e1x - e1z * tx, qz = tx * e1y - e1x * ty;
    const f32x2 v = ((dx2 * qx + dy2 * qy) + dz2 * qz) * inv_det;              // mod.rs:584
    const f32x2 dist = ((e2x * qx + e2y * qy) + e2z * qz) * inv_det;           // mod.rs:589
    const f32x2 uv = u + v;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        // The reference's five `continue` tests (mod.rs:571,579,585,592), written as "stay" conditions.  The two
        // forms differ only when an operand is NaN, and for a finite ray u, v, dist can only be NaN/inf when
        // determinant == 0, which the first condition already rejects - so plain ordered compares are exact here.
        // `u > 1.0` (mod.rs:579) needs no instruction of its own: with v >= 0, u <= RN(u+v) by monotonicity of
        // rounding, so u > 1 implies (u+v) > 1, which is tested.
        const bool keep = (f_abs(determinant[hf]) >= 1e-4f) & (u[hf] >= 0.0f) & (v[hf] >= 0.0f) & (uv[hf] <= 1.0f) &
                          (dist[hf] > 0.0f);
        const int32_t id = (int32_t)tr.id[hf];
        const bool closer = ORDERED ? (dist[hf] < mt) : (dist[hf] < mt || (dist[hf] == mt && id < mid));
        if (keep & closer) {
            mt = dist[hf];
            mid = id;
        }
    }
}

// test_pair for two triangles that lie in a plane perpendicular to axis AXIS: both edge vectors are exactly zero along
// it (the walls of the reference's rooms, scenes.rs:322-367).  The reference's arithmetic then multiplies by those
// zeros and adds the +-0 products; leaving them out gives the same numbers - x*0 = +-0, y + (+-0) = y for y != 0 - except
// that a result which is itself zero may come out with the other sign, and no decision of mod.rs:571-594 and no
// output depends on the sign of a zero (|det| < 1e-4, u < 0, v < 0, u + v > 1, t <= 0 treat -0 as +0; a hit has
// t != 0).  12 of the 53 packed instructions go.  With (i, j, k) = (AXIS, AXIS+1, AXIS+2) mod 3:
//   p = d x e2:  p_i = d_j e2_k - e2_j d_k,   p_j = -(e2_k d_i),   p_k = d_i e2_j          (mod.rs:563)
//   det = e1_j p_j + e1_k p_k   (the two products the reference adds to a +-0, in either order)   (mod.rs:564)
//   q = tv x e1: q_i = tv_j e1_k - e1_j tv_k, q_j = -(e1_k tv_i),  q_k = tv_i e1_j          (mod.rs:583)
//   t = (e2_j q_j + e2_k q_k) / det                                                        (mod.rs:589)
// u and v keep the reference's three-term sums in x, y, z order.
template <int AXIS, bool ORDERED>
__device__ __forceinline__ void test_pair_planar(const TriPairRec &tr, f32x2 ox2, f32x2 oy2, f32x2 oz2, f32x2 dx2,
                                                 f32x2 dy2, f32x2 dz2, float &mt, int32_t &mid) {
    constexpr int I = AXIS, J = (AXIS + 1) % 3, K = (AXIS + 2) % 3;
    const f32x2 e1[3] = {ld2(tr.e1x), ld2(tr.e1y), ld2(tr.e1z)};
    const f32x2 e2[3] = {ld2(tr.e2x), ld2(tr.e2y), ld2(tr.e2z)};
    const f32x2 dd[3] = {dx2, dy2, dz2};
    f32x2 p[3], q[3];
    p[I] = dd[J] * e2[K] - e2[J] * dd[K];
    p[J] = -(e2[K] * dd[I]);
    p[K] = dd[I] * e2[J];
    const f32x2 determinant = e1[J] * p[J] + e1[K] * p[K];
    const f32x2 inv_det = f_rcp2(determinant);
    const f32x2 tv[3] = {ox2 - ld2(tr.ax), oy2 - ld2(tr.ay), oz2 - ld2(tr.az)};
    const f32x2 u = ((tv[0] * p[0] + tv[1] * p[1]) + tv[2] * p[2]) * inv_det;
    q[I] = tv[J] * e1[K] - e1[J] * tv[K];
    q[J] = -(e1[K] * tv[I]);
    q[K] = tv[I] * e1[J];
    const f32x2 v = ((dx2 * q[0] + dy2 * q[1]) + dz2 * q[2]) * inv_det;
    const f32x2 dist = (e2[J] * q[J] + e2[K] * q[K]) * inv_det;
    const f32x2 uv = u + v;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const bool keep = (f_abs(determinant[hf]) >= 1e-4f) & (u[hf] >= 0.0f) & (v[hf] >= 0.0f) & (uv[hf] <= 1.0f) &
                          (dist[hf] > 0.0f);
        const int32_t id = (int32_t)tr.id[hf];
        const bool closer = ORDERED ? (dist[hf] < mt) : (dist[hf] < mt || (dist[hf] == mt && id < mid));
        if (keep & closer) {
            mt = dist[hf];
            mid = id;
        }
    }
}

// Triangle::intersect (mod.rs:559-592) of one ray against the two triangles of a pair record, with the outcome of each
// as an integer key instead of a decision: key = (k << 32) | tr.id[half] where k = bits(distance) - 1 for an accepted hit
// (order and ties of the distances are kept) and >= 0x7fffffff for a rejected one - so that "closest, smallest id among
// equals" over any number of triangles is one integer minimum (an LDS atomic), with no compare or select instruction.
//   The five rejections of mod.rs:571-592 are ONE sign bit, built with two-cycle integer instructions instead of
//   compares and selects (4 cycles each, profiles/r02_valu_issue_costs.json):
//     u < 0, v < 0, u + v > 1  <=> the sign bit of (u + 0) | (v + 0) | (1 - (u + v)): adding +0 turns a -0 into +0 (the
//                                  reference keeps u = -0: `u < 0.0` is false), 1 - s is negative exactly when s > 1 (s and
//                                  1 are floats: the rounded difference has the sign of the real one), and u > 1 is implied
//                                  by u + v > 1 (test_pair).  NaN cannot occur with |det| >= 1e-4;
//     |det| < 1e-4             <=> (bits(det) & 0x7fffffff) - bits(1e-4) is negative as an integer.
//   The bit is OR-ed into the bits of the distance; k = bits - 1 then maps distance > 0 to [0, 0x7f7ffffe], +0 to
//   0xffffffff and anything negative, -0 or rejected to >= 0x7fffffff.
template <class Rec>
__device__ __forceinline__ void pair_test_keys(const Rec &tr, vec3 ro, vec3 rd, unsigned long long keys2[2], float t_hit[2]) {
    const f32x2 ox2 = splat2(ro.x), oy2 = splat2(ro.y), oz2 = splat2(ro.z);
    const f32x2 dx2 = splat2(rd.x), dy2 = splat2(rd.y), dz2 = splat2(rd.z);
    const f32x2 e1x = ld2(tr.e1x), e1y = ld2(tr.e1y), e1z = ld2(tr.e1z);
    const f32x2 e2x = ld2(tr.e2x), e2y = ld2(tr.e2y), e2z = ld2(tr.e2z);
    const f32x2 px = dy2 * e2z - e2y * dz2, py = dz2 * e2x - e2z * dx2, pz = dx2 * e2y - e2x * dy2;
    const f32x2 determinant = (e1x * px + e1y * py) + e1z * pz;
    const f32x2 inv_det = f_rcp2(determinant);
    const f32x2 tx = ox2 - ld2(tr.ax), ty = oy2 - ld2(tr.ay), tz = oz2 - ld2(tr.az);
    const f32x2 u = ((tx * px + ty * py) + tz * pz) * inv_det;
    const f32x2 qx = ty * e1z - e1y * tz, qy = tz * e1x - e1z * tx, qz = tx * e1y - e1x * ty;
    const f32x2 v = ((dx2 * qx + dy2 * qy) + dz2 * qz) * inv_det;
    const f32x2 dist = ((e2x * qx + e2y * qy) + e2z * qz) * inv_det;
    const f32x2 zero = splat2(0.0f);
    const f32x2 s_uv = (u + zero) + (v + zero);
    const f32x2 w = splat2(1.0f) - s_uv;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const uint32_t ub = __float_as_uint(u[hf] + 0.0f), vb = __float_as_uint(v[hf] + 0.0f), wb = __float_as_uint(w[hf]);
        const uint32_t db = (__float_as_uint(determinant[hf]) & 0x7fffffffu) - 0x38d1b717u;  // bits(1e-4f)
        const uint32_t bad = (ub | vb | wb | db) & 0x80000000u;
        const uint32_t k = (__float_as_uint(dist[hf]) | bad) - 1u;
        keys2[hf] = ((unsigned long long)k << 32) | tr.id[hf];
        t_hit[hf] = dist[hf];
    }
}

// Slab test of the two (padded) child boxes of a node at once: packed fma/min/max, NaN-ignoring min/max (a
// slab whose bounds come out NaN - ray parallel to it and 0*inf - simply does not constrain).  This test only
// has to be conservative, not bit-identical to anything: the boxes are padded by the worst-case error of the
// f32 Moller-Trumbore hit plus the roundoff of this test (pt_host.cpp), so a box reported as missed cannot hold
// a triangle that the reference's arithmetic would accept closer than `bound`.
__device__ __forceinline__ void hit_boxes(const BvhNode &n, f32x2 ivx, f32x2 ivy, f32x2 ivz, f32x2 oix, f32x2 oiy,
                                          f32x2 oiz, float bound, bool *h0, bool *h1, f32x2 *t_in) {
    const f32x2 ax = __builtin_elementwise_fma(ld2(n.lox), ivx, -oix), bx = __builtin_elementwise_fma(ld2(n.hix), ivx, -oix);
    const f32x2 ay = __builtin_elementwise_fma(ld2(n.loy), ivy, -oiy), by = __builtin_elementwise_fma(ld2(n.hiy), ivy, -oiy);
    const f32x2 az = __builtin_elementwise_fma(ld2(n.loz), ivz, -oiz), bz = __builtin_elementwise_fma(ld2(n.hiz), ivz, -oiz);
    const f32x2 zero = splat2(0.0f);
    const f32x2 tin = __builtin_elementwise_max(
        __builtin_elementwise_max(__builtin_elementwise_min(ax, bx), __builtin_elementwise_min(ay, by)),
        __builtin_elementwise_max(__builtin_elementwise_min(az, bz), zero));
    const f32x2 tout = __builtin_elementwise_min(
        __builtin_elementwise_min(__builtin_elementwise_max(ax, bx), __builtin_elementwise_max(ay, by)),
        __builtin_elementwise_max(az, bz));
    const f32x2 lim = tout * splat2(1.0000005f);
    *h0 = tin[0] <= lim[0] && tin[0] <= bound;
    *h1 = tin[1] <= lim[1] && tin[1] <= bound;
    *t_in = tin;
}
// the same with the verdicts as WAVE MASKS (every lane of the wave calls it): ballots of the single comparisons combined by
// scalar and - a ballot of a boolean that is itself a combination of comparisons costs a v_cndmask and a v_cmp on top
__device__ __forceinline__ void hit_boxes_masks(const BvhNode &n, f32x2 ivx, f32x2 ivy, f32x2 ivz, f32x2 oix, f32x2 oiy,
                                                f32x2 oiz, float bound, uint64_t *m0, uint64_t *m1, f32x2 *t_in) {
    const f32x2 ax = __builtin_elementwise_fma(ld2(n.lox), ivx, -oix), bx = __builtin_elementwise_fma(ld2(n.hix), ivx, -oix);
    const f32x2 ay = __builtin_elementwise_fma(ld2(n.loy), ivy, -oiy), by = __builtin_elementwise_fma(ld2(n.hiy), ivy, -oiy);
    const f32x2 az = __builtin_elementwise_fma(ld2(n.loz), ivz, -oiz), bz = __builtin_elementwise_fma(ld2(n.hiz), ivz, -oiz);
    const f32x2 zero = splat2(0.0f);
    const f32x2 tin = __builtin_elementwise_max(
        __builtin_elementwise_max(__builtin_elementwise_min(ax, bx), __builtin_elementwise_min(ay, by)),
        __builtin_elementwise_max(__builtin_elementwise_min(az, bz), zero));
    const f32x2 tout = __builtin_elementwise_min(
        __builtin_elementwise_min(__builtin_elementwise_max(ax, bx), __builtin_elementwise_max(ay, by)),
        __builtin_elementwise_max(az, bz));
    const f32x2 lim = tout * splat2(1.0000005f);
    *m0 = __builtin_amdgcn_ballot_w64(tin[0] <= lim[0]) & __builtin_amdgcn_ballot_w64(tin[0] <= bound);
    *m1 = __builtin_amdgcn_ballot_w64(tin[1] <= lim[1]) & __builtin_amdgcn_ballot_w64(tin[1] <= bound);
    *t_in = tin;
}

// Traversal-stack entry codecs.  With the nodes staged in LDS the scene has < 2^15 nodes and BVH leaves, so a
// child reference fits 16 bits (halves the LDS the stacks take, which is what bounds occupancy here).
struct Stack16 {
    typedef uint16_t T;
    uint32_t pair_base;
    __device__ __forceinline__ T enc(int32_t r) const {
        return (T)(r >= 0 ? (uint32_t)r : (0x8000u | ((uint32_t)(~r) - (pair_base << kBvhLeafBits))));
    }
    __device__ __forceinline__ int32_t dec(T e) const {
        return (e & 0x8000u) ? ~(int32_t)((uint32_t)(e & 0x7fffu) + (pair_base << kBvhLeafBits)) : (int32_t)e;
    }
};
struct Stack32 {
    typedef uint32_t T;
    __device__ __forceinline__ T enc(int32_t r) const { return (T)r; }
    __device__ __forceinline__ int32_t dec(T e) const { return (int32_t)e; }
};
// Either width, chosen at run time by a wave-uniform flag (DevScene.bvh_in_lds bit 1): one copy of the walk in the kernel
// instead of one per width.  `at` is a byte address in the lane's stack column.
struct StackDyn {
    bool narrow;
    Stack16 c16;
    __device__ __forceinline__ uint32_t entry_bytes() const { return narrow ? 2u : 4u; }
    __device__ __forceinline__ void store(char *at, int32_t r) const {
        if (narrow)
            *reinterpret_cast<uint16_t *>(at) = c16.enc(r);
        else
            *reinterpret_cast<int32_t *>(at) = r;
    }
    __device__ __forceinline__ int32_t load(const char *at) const {
        return narrow ? c16.dec(*reinterpret_cast<const uint16_t *>(at)) : *reinterpret_cast<const int32_t *>(at);
    }
};

// Closest triangle of one BVH mesh for one lane.  `nodes` is LDS (staged) or global memory; `stack` is this
// lane's column of the workgroup's LDS stack (stride `stride`).  Only triangles that could beat `best_t`
// matter to the caller (mod.rs:649 replaces on strict '<'), so boxes farther than min(mt, best_t) are skipped.
template <class NodePtr, class Codec>
__device__ __forceinline__ void bvh_closest(const DevScene &S, NodePtr nodes, Codec codec, typename Codec::T *stack,
                                            uint32_t stride, vec3 o, vec3 d, int32_t root, float best_t, float &mt,
                                            int32_t &mid) {
    // 1/d clamped to +-1e18: a direction component that is exactly 0 (diffuse draw r2 == 0 off an axis-aligned
    // wall, mirror rays) must give finite slab bounds whose sign still says on which side of the origin the
    // plane lies; with +-inf the fma form below would turn "inside the slab" into NaN/-inf.
    const float big = 1e18f;
    const vec3 inv = mk(__builtin_fmaxf(__builtin_fminf(1.0f / d.x, big), -big),
                        __builtin_fmaxf(__builtin_fminf(1.0f / d.y, big), -big),
                        __builtin_fmaxf(__builtin_fminf(1.0f / d.z, big), -big));
    const f32x2 ivx = splat2(inv.x), ivy = splat2(inv.y), ivz = splat2(inv.z);
    const f32x2 oix = splat2(o.x * inv.x), oiy = splat2(o.y * inv.y), oiz = splat2(o.z * inv.z);
    const f32x2 ox2 = splat2(o.x), oy2 = splat2(o.y), oz2 = splat2(o.z);
    const f32x2 dx2 = splat2(d.x), dy2 = splat2(d.y), dz2 = splat2(d.z);
    constexpr int32_t kDone = (int32_t)0x80000000;  // not a valid leaf reference (~0x7fffffff)
    uint32_t sp = 0;
    int32_t cur = root;
    for (;;) {
        // 1. walk down through inner nodes until this lane stands on a leaf (or has nothing left).  Lanes that
        //    arrive early wait here, so the (longer) triangle code below runs once per round for the whole wave.
        for (;;) {  // wave-uniform loop: the lanes still on an inner node take one step per iteration
            // leave for the triangle code when every lane stands on a leaf (or is done), or as soon as S.leaf_quorum
            // lanes do: waiting for the deepest descent of 64 lanes leaves most of them idle, going at the first leaf
            // runs the (twice as long) triangle code for one lane
            if (__builtin_amdgcn_ballot_w64(cur >= 0) == 0ull) break;
            if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(cur < 0 && cur != kDone)) >= S.leaf_quorum) break;
            if (cur >= 0) {
                const BvhNode n = nodes[cur];
                const float bound = __builtin_fminf(mt, best_t);
                bool h0, h1;
                f32x2 tin;
                hit_boxes(n, ivx, ivy, ivz, oix, oiy, oiz, bound, &h0, &h1, &tin);
                if (h0 && h1) {
                    const bool first0 = tin[0] <= tin[1];
                    if (sp < kBvhStack) stack[sp * stride] = codec.enc(first0 ? n.c[1] : n.c[0]);
                    ++sp;  // (the host guarantees tree depth < kBvhStack)
                    cur = first0 ? n.c[0] : n.c[1];
                } else if (h0 || h1) {
                    cur = h0 ? n.c[0] : n.c[1];
                } else if (sp != 0u) {
                    --sp;
                    cur = codec.dec(stack[sp * stride]);
                } else {
                    cur = kDone;
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(cur != kDone) == 0ull) break;  // every lane of the wave is done
        // 2. leaf: two triangles (lanes still descending - early exit by quorum - and finished lanes sit out)
        if (cur < 0 && cur != kDone) {
            const uint32_t code = (uint32_t)~cur;
            for (uint32_t j = 0; j < leaf_count(code); ++j)
                test_pair<false>(S.tri_pairs[leaf_first(code) + j], ox2, oy2, oz2, dx2, dy2, dz2, mt, mid);
            if (sp == 0u) {
                cur = kDone;
            } else {
                --sp;
                cur = codec.dec(stack[sp * stride]);
            }
        }
    }
}

// bvh_closest with the LEAVES POSTPONED.  In bvh_closest a lane that reaches a leaf waits until enough others have (the
// quorum) and the triangle code - by far the longest part of a step - then runs for those lanes only: the walks ran at
// 17 of 64 lanes.  Here a lane that reaches a leaf only appends (lane, leaf) to a per-wave list in LDS and moves on to the
// next entry of its stack; whenever the list holds as many entries as the wave has walking lanes, the wave runs one
// DENSE batch of leaf tests: lane e fetches entry e's ray from its owner's registers (ds_bpermute), gathers the leaf's pair
// record, evaluates both triangles (pair_test_keys: the reference's arithmetic, outcome as integer keys) and folds them
// into the owner's key with two LDS atomic minima - key = (distance bits - 1) << 32 | triangle index: the closest, the
// smaller index among equal distances, exactly test_pair<false>'s rule, whatever the order.  After a batch every lane
// refreshes its pruning bound from its key.  Called by the walking lanes of a wave together (any subset of the wave).
// A key is written by other lanes of the wave (LDS atomic minima) and read back by its owner: the read has to be an atomic
// load - a plain one may be satisfied from the value the owner itself stored last (the compiler sees no other writer).
__device__ __forceinline__ unsigned long long load_key(const unsigned long long *p) {
    return __atomic_load_n(p, __ATOMIC_RELAXED);
}

struct LeafLds {
    unsigned long long *keys;  // [64] of this wave
    uint32_t *list;            // [kLeafListCap] of this wave: lane | leaf code << 6 (leaf_first / leaf_count)
};
constexpr uint32_t kLeafListCap = 128u;  // < 64 left over + at most 64 leaves appended by one step

template <class NodePtr>
__device__ __forceinline__ void bvh_closest_postponed(const DevScene &S, NodePtr nodes, StackDyn codec, char *stack,
                                                      uint32_t stride, const LeafLds &L, vec3 o, vec3 d, int32_t root,
                                                      float best_t, float &mt, int32_t &mid) {
    // `stack`: this lane's column of the workgroup's stacks (byte address of entry 0), `stride`: bytes between entries
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t act = __builtin_amdgcn_ballot_w64(true);
    const uint32_t n_act = (uint32_t)__builtin_popcountll(act);
    const uint32_t my = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
    L.keys[lane] = ~0ull;
    const float big = 1e18f;  // see bvh_closest
    const vec3 inv = mk(__builtin_fmaxf(__builtin_fminf(1.0f / d.x, big), -big),
                        __builtin_fmaxf(__builtin_fminf(1.0f / d.y, big), -big),
                        __builtin_fmaxf(__builtin_fminf(1.0f / d.z, big), -big));
    const f32x2 ivx = splat2(inv.x), ivy = splat2(inv.y), ivz = splat2(inv.z);
    const f32x2 oix = splat2(o.x * inv.x), oiy = splat2(o.y * inv.y), oiz = splat2(o.z * inv.z);
    constexpr int32_t kDone = (int32_t)0x80000000;
    uint32_t n_leaf = 0;   // wave-uniform
    float bound = best_t;  // min(best of the other objects, this lane's best triangle so far)
    auto leaf_batch = [&](uint32_t base, uint32_t count) {  // list entries [base, base + count), count <= n_act
        const bool valid = my < count;
        const uint32_t ent = valid ? L.list[base + my] : lane;
        const int sel = (int)((ent & 63u) << 2);
        vec3 ro, rd;
        ro.x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.x)));
        ro.y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.y)));
        ro.z = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.z)));
        rd.x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.x)));
        rd.y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.y)));
        rd.z = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.z)));
        // an entry is a whole leaf: its records one after the other (nearly every leaf of a built tree is full)
        const uint32_t code = ent >> 6;
        const uint32_t first = leaf_first(code), cnt = valid ? leaf_count(code) : 0u;
        for (uint32_t r = 0; r < kBvhLeafPairs; ++r) {
            const bool has = r < cnt;
            if (r != 0u && __builtin_amdgcn_ballot_w64(has) == 0ull) break;
            if (has) {
                const TriPairRec tr = S.tri_pairs[first + r];
                unsigned long long k2[2];
                float th[2];
                pair_test_keys(tr, ro, rd, k2, th);
                atomicMin(&L.keys[ent & 63u], k2[0]);
                atomicMin(&L.keys[ent & 63u], k2[1]);
            }
        }
    };
    uint32_t sp = 0;
    char *top = stack;  // entry sp of this lane's column
    int32_t cur = root;
    PT_WSTAT(S, 0, n_act);  // walks
    PT_WSTAT(S, 10, 1);     // wave-walks
    for (;;) {  // wave-uniform loop: every lane that still has work takes one step per trip
        const bool walking = __builtin_amdgcn_ballot_w64(cur != kDone) != 0ull;
        if (!walking && n_leaf == 0u) break;
        if (walking) {
            PT_WSTAT(S, 1, 1);                                                                              // trips
            PT_WSTAT(S, 2, __builtin_popcountll(__builtin_amdgcn_ballot_w64(cur != kDone)));                // lanes with work
            PT_WSTAT(S, 3, __builtin_popcountll(__builtin_amdgcn_ballot_w64(cur >= 0)));                    // node steps
            bool at_leaf = false, pop = false;
            uint32_t leaf = 0;
            if (cur >= 0) {
                const BvhNode n = nodes[cur];
                bool h0, h1;
                f32x2 tin;
                hit_boxes(n, ivx, ivy, ivz, oix, oiy, oiz, bound, &h0, &h1, &tin);
                const bool first0 = tin[0] <= tin[1];
                if (h0 && h1) {
                    if (sp < S.bvh_stack) codec.store(top, first0 ? n.c[1] : n.c[0]);
                    ++sp;  // (the host guarantees tree depth < DevScene.bvh_stack <= kBvhStack)
                    top += stride;
                    cur = first0 ? n.c[0] : n.c[1];
                } else if (h0 || h1) {
                    cur = h0 ? n.c[0] : n.c[1];
                } else {
                    pop = true;
                }
            } else if (cur != kDone) {
                at_leaf = true;
                leaf = (uint32_t)~cur;
                pop = true;
            }
            if (pop) {  // one place for both: nothing below this node / the leaf is on the list
                if (sp == 0u) {
                    cur = kDone;
                } else {
                    --sp;
                    top -= stride;
                    cur = codec.load(top);
                }
            }
            const uint64_t ml = __builtin_amdgcn_ballot_w64(at_leaf);
            if (ml != 0ull) {
                if (at_leaf)
                    L.list[n_leaf + __builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0u))] =
                        lane | (leaf << 6);
                n_leaf += (uint32_t)__builtin_popcountll(ml);
            }
        }
        // a dense batch whenever as many leaves wait as the wave has walking lanes; what is left when the walks are over
        // (at most n_act - 1 entries are ever left over: the list never holds more than 127)
        while (n_leaf >= n_act || (!walking && n_leaf != 0u)) {
            const uint32_t cnt = n_leaf < n_act ? n_leaf : n_act;
            n_leaf -= cnt;
            PT_WSTAT(S, 5, 1);
            PT_WSTAT(S, 6, cnt);
            leaf_batch(n_leaf, cnt);
            const uint32_t k = (uint32_t)(load_key(&L.keys[lane]) >> 32);
            if (k < 0x7f800000u) bound = __builtin_fminf(best_t, __uint_as_float(k + 1u));
        }
    }
    const unsigned long long key = load_key(&L.keys[lane]);
    const uint32_t k = (uint32_t)(key >> 32);
    if (k < 0x7f800000u) {
        mt = __uint_as_float(k + 1u);
        mid = (int32_t)(uint32_t)key;
    }
}

// The walks of one BVH mesh for a whole wave at once, as a QUEUE of (ray, node) items.  In bvh_closest_postponed every lane
// walks its own ray depth-first: the wave takes as many trips as its longest walk and a third of the lanes has work per
// trip (mesh.json: 37 trips, 22 lanes).  Here the pending box tests of all rays of the wave wait in one last-in-first-out
// queue in LDS - (owner lane, child reference, distance at which the ray enters the child's box) per item - and the wave
// tests the top 64 items at a time: lane e fetches the owner's ray constants from the owner's registers (ds_bpermute) and
// the owner's bound from its key, drops the item if its box now lies beyond the bound, else gathers the node, tests both
// boxes and pushes the children that are hit, the nearer node last; children that are leaves go to the leaf list, which
// grows down from the other end of the same area and is tested in dense batches as in bvh_closest_postponed (most recent
// leaves first; a leaf whose box lies beyond its owner's bound by then is skipped).  A batch is dense whatever the rays'
// depths, and a ray's subtrees are tested side by side: 22 rounds of dependent loads instead of 37.
// Any order gives the same result: a key only ever takes the minimum over (distance, triangle index) of the triangles
// tested, and a box is only skipped when it lies beyond the owner's key at the time (pruning on '>').
// A ray's pending items are not bounded by the tree depth as a depth-first stack is; when a batch's pushes would not fit
// they are dropped and the owners flagged, and flagged rays are walked again depth-first afterwards
// (bvh_closest_postponed, from the root, with the bound the queue found: stacks and leaf list live where the queue was).
struct WalkQueue {
    uint32_t *redo;  // [2]: flags of the wave's lanes (+ 8 B of padding), then
    uint2 *ent;      // [cap]: .x = owner lane | child << 6 (node index or leaf code), .y = bits of the entry distance;
                     // box tests queue up from ent[0], leaves down from ent[cap - 1]
    uint32_t cap;
};
constexpr uint32_t kWalkQueueHeader = 16;  // bytes in front of the entries

template <class NodePtr>
__device__ __forceinline__ void bvh_closest_queue(const DevScene &S, NodePtr nodes, const WalkQueue &Q, unsigned long long *keys,
                                                  bool walk, vec3 o, vec3 d, int32_t root, float best_t, float &mt, int32_t &mid) {
    // called by every active lane of the wave (the workers); `walk`: this lane's ray takes part.  keys: [64] of the wave
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t act = __builtin_amdgcn_ballot_w64(true);
    const uint32_t n_act = (uint32_t)__builtin_popcountll(act);
    const uint32_t my = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
    auto prefix = [](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    auto count = [](uint64_t m) { return (uint32_t)__builtin_popcountll(m); };
    // the key starts at "just beyond best_t, no triangle": the bound of a box test is always (key >> 32) + 1
    if (walk) keys[lane] = ((unsigned long long)(__float_as_uint(best_t) - 1u) << 32) | 0xffffffffull;
    if (my == 0u) Q.redo[0] = Q.redo[1] = 0u;
    const float big = 1e18f;  // see bvh_closest
    const vec3 inv = mk(__builtin_fmaxf(__builtin_fminf(1.0f / d.x, big), -big),
                        __builtin_fmaxf(__builtin_fminf(1.0f / d.y, big), -big),
                        __builtin_fmaxf(__builtin_fminf(1.0f / d.z, big), -big));
    const vec3 oi = mk(o.x * inv.x, o.y * inv.y, o.z * inv.z);
    uint32_t q_count, n_leaf;  // wave-uniform: box tests in ent[0, q_count), leaves in ent[cap - n_leaf, cap)
    {
        const uint64_t m_node = __builtin_amdgcn_ballot_w64(walk && root >= 0), m_leaf = __builtin_amdgcn_ballot_w64(walk && root < 0);
        q_count = count(m_node);
        n_leaf = count(m_leaf);
        if (walk && root >= 0) Q.ent[prefix(m_node)] = make_uint2(lane | ((uint32_t)root << 6), 0u);
        if (walk && root < 0) Q.ent[Q.cap - 1u - prefix(m_leaf)] = make_uint2(lane | ((uint32_t)~root << 6), 0u);
    }
    PT_WSTAT(S, 0, q_count + n_leaf);  // walks
    PT_WSTAT(S, 10, 1);                // wave-walks
    for (;;) {
        if (q_count == 0u && n_leaf == 0u) break;
        if (q_count != 0u) {
            PT_PHASE(kPhWalkBox);
            // (smaller batches - closer to depth-first order, fewer boxes tested - lose: 48 items 18.5, 32 items 17.1 against
            // 19.3 G bounces/s on mesh.json; so do leaf batches started at 32 waiting leaves: 18.7)
            const uint32_t c = q_count < n_act ? q_count : n_act;
            q_count -= c;
            const bool has = my < c;
            uint2 e = make_uint2(lane, 0u);
            if (has) e = Q.ent[q_count + my];
            const uint32_t owner = e.x & 63u;
            const int sel = (int)(owner << 2);
            const float r_ivx = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(inv.x)));
            const float r_ivy = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(inv.y)));
            const float r_ivz = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(inv.z)));
            const float r_oix = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(oi.x)));
            const float r_oiy = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(oi.y)));
            const float r_oiz = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(oi.z)));
            const float bound = __uint_as_float((uint32_t)(load_key(&keys[owner]) >> 32) + 1u);
            // the item is still worth testing unless its owner has found something nearer since it was pushed
            const uint64_t m_valid = __builtin_amdgcn_ballot_w64(has) & ~__builtin_amdgcn_ballot_w64(__uint_as_float(e.y) > bound);
            PT_WSTAT(S, 1, 1);                                   // batches of box tests
            PT_WSTAT(S, 2, c);                                   // items
            PT_WSTAT(S, 3, __builtin_popcountll(m_valid));       // ... still worth testing
            // Every lane tests a node - the ones without an item node 0 with their own ray, a harmless read - and the verdicts
            // are wave masks that scalar and / or combine: a divergent branch around the test and ballots of combined
            // booleans cost four v_cndmask / v_cmp pairs per batch.
            uint64_t m_h0, m_h1;
            f32x2 tin;
            const BvhNode n = nodes[e.x >> 6];
            hit_boxes_masks(n, splat2(r_ivx), splat2(r_ivy), splat2(r_ivz), splat2(r_oix), splat2(r_oiy), splat2(r_oiz), bound, &m_h0, &m_h1,
                            &tin);
            m_h0 &= m_valid;
            m_h1 &= m_valid;
            const bool first0 = tin[0] <= tin[1];
            const int32_t c0 = n.c[0], c1 = n.c[1];
            // children that are leaves -> leaf list; children that are nodes -> queue, the nearer of two on top
            // (sending the farther of two hit children, when it is a leaf, through the queue once more - to be dropped if
            // the nearer subtree has found something in front of it by then - does not pay: the nearer leaf has rarely
            // been tested when the item comes up again; 5.12 instead of 5.18 leaves per walk, 20.3 instead of 18.2 batches)
            const uint64_t m_neg0 = __builtin_amdgcn_ballot_w64(c0 < 0), m_neg1 = __builtin_amdgcn_ballot_w64(c1 < 0);
            const uint64_t m_l0 = m_h0 & m_neg0, m_l1 = m_h1 & m_neg1, m_n0 = m_h0 & ~m_neg0, m_n1 = m_h1 & ~m_neg1;
            const uint64_t m_far = m_n0 & m_n1, m_near = m_n0 | m_n1;
            const uint32_t add_q = count(m_far) + count(m_near), add_l = count(m_l0) + count(m_l1);
            if (q_count + n_leaf + add_q + add_l > Q.cap) {  // wave-uniform: no room - these rays are walked again afterwards
                if (__builtin_amdgcn_inverse_ballot_w64(m_h0 | m_h1)) atomicOr(&Q.redo[owner >> 5], 1u << (owner & 31u));
                PT_WSTAT(S, 11, 1);
            } else {
                const bool n0 = __builtin_amdgcn_inverse_ballot_w64(m_n0);
                const bool near0 = __builtin_amdgcn_inverse_ballot_w64(m_far) ? first0 : n0;  // which child is the (nearer) node pushed last
                if (__builtin_amdgcn_inverse_ballot_w64(m_far))
                    Q.ent[q_count + prefix(m_far)] = make_uint2(owner | ((uint32_t)(near0 ? c1 : c0) << 6), __float_as_uint(near0 ? tin[1] : tin[0]));
                if (__builtin_amdgcn_inverse_ballot_w64(m_near))
                    Q.ent[q_count + count(m_far) + prefix(m_near)] = make_uint2(owner | ((uint32_t)(near0 ? c0 : c1) << 6), __float_as_uint(near0 ? tin[0] : tin[1]));
                q_count += add_q;
                if (__builtin_amdgcn_inverse_ballot_w64(m_l0))
                    Q.ent[Q.cap - 1u - (n_leaf + prefix(m_l0))] = make_uint2(owner | ((uint32_t)~c0 << 6), __float_as_uint(tin[0]));
                if (__builtin_amdgcn_inverse_ballot_w64(m_l1))
                    Q.ent[Q.cap - 1u - (n_leaf + count(m_l0) + prefix(m_l1))] = make_uint2(owner | ((uint32_t)~c1 << 6), __float_as_uint(tin[1]));
                n_leaf += add_l;
            }
        }
        // a dense batch of leaf tests whenever as many leaves wait as the wave has workers (before the next box tests: a hit
        // tightens its owner's bound); what is left when the queue is empty.  The most recent leaves first.
        while (n_leaf >= n_act || (q_count == 0u && n_leaf != 0u)) {
            PT_PHASE(kPhWalkLeaf);
            const uint32_t cnt = n_leaf < n_act ? n_leaf : n_act;
            bool valid = my < cnt;
            uint2 e = make_uint2(lane, 0u);
            if (valid) e = Q.ent[Q.cap - n_leaf + my];
            n_leaf -= cnt;
            const uint32_t owner = e.x & 63u;
            const int sel = (int)(owner << 2);
            vec3 ro, rd;
            ro.x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.x)));
            ro.y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.y)));
            ro.z = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.z)));
            rd.x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.x)));
            rd.y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.y)));
            rd.z = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.z)));
            const float bound = __uint_as_float((uint32_t)(load_key(&keys[owner]) >> 32) + 1u);
            valid = valid && !(__uint_as_float(e.y) > bound);
            PT_WSTAT(S, 5, 1);
            PT_WSTAT(S, 6, cnt);
            PT_WSTAT(S, 12, __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid)));  // leaves still worth testing
            const uint32_t code = e.x >> 6;
            const uint32_t first = leaf_first(code), lc = valid ? leaf_count(code) : 0u;
            for (uint32_t r = 0; r < kBvhLeafPairs; ++r) {
                const bool has = r < lc;
                if (__builtin_amdgcn_ballot_w64(has) == 0ull) break;
                if (has) {
                    const TriPairRec tr = S.tri_pairs[first + r];
                    unsigned long long k2[2];
                    float th[2];
                    pair_test_keys(tr, ro, rd, k2, th);
                    atomicMin(&keys[owner], k2[0]);
                    atomicMin(&keys[owner], k2[1]);
                }
            }
        }
    }
    PT_PHASE(kPhWalkGate);
    if (walk) {
        const unsigned long long key = load_key(&keys[lane]);
        if ((uint32_t)key != 0xffffffffu) {
            mt = __uint_as_float((uint32_t)(key >> 32) + 1u);
            mid = (int32_t)(uint32_t)key;
        }
    }
    // rays whose pushes were dropped: depth-first from the root, with the bound found so far
    const uint32_t redo = __atomic_load_n(&Q.redo[lane >> 5], __ATOMIC_RELAXED);
    const bool again = walk && ((redo >> (lane & 31u)) & 1u) != 0u;
    if (__builtin_amdgcn_ballot_w64(again) != 0ull) {
        if (again) {
            StackDyn codec;
            codec.narrow = (S.bvh_in_lds & 2u) != 0u;
            codec.c16.pair_base = S.bvh_pair_base;
            const uint32_t eb = codec.entry_bytes();
            LeafLds L;  // [stacks: bvh_stack x 64 x u16|u32][leaf list: kLeafListCap x u32] in the queue's area
            L.keys = keys;
            L.list = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(Q.ent) + S.bvh_stack * 64u * eb);
            float mt2 = __builtin_inff();
            int32_t mid2 = -1;
            bvh_closest_postponed(S, nodes, codec, reinterpret_cast<char *>(Q.ent) + lane * eb, 64u * eb, L, o, d, root,
                                  __builtin_fminf(best_t, mt), mt2, mid2);
            if (mid2 >= 0 && (mid < 0 || mt2 < mt || (mt2 == mt && (uint32_t)mid2 < (uint32_t)mid))) {
                mt = mt2;
                mid = mid2;
            }
        }
    }
}

// Slab test of the four child boxes of a BvhNode4 (two packed pairs), verdicts as wave masks (see hit_boxes_masks)
__device__ __forceinline__ void hit_boxes4_masks(const BvhNode4 &n, f32x2 ivx, f32x2 ivy, f32x2 ivz, f32x2 oix, f32x2 oiy, f32x2 oiz,
                                                 float bound, uint64_t m[4], float tin_out[4]) {
    const f32x2 zero = splat2(0.0f);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2 lx = {n.lox[2 * p], n.lox[2 * p + 1]}, ly = {n.loy[2 * p], n.loy[2 * p + 1]}, lz = {n.loz[2 * p], n.loz[2 * p + 1]};
        const f32x2 hx = {n.hix[2 * p], n.hix[2 * p + 1]}, hy = {n.hiy[2 * p], n.hiy[2 * p + 1]}, hz = {n.hiz[2 * p], n.hiz[2 * p + 1]};
        const f32x2 ax = __builtin_elementwise_fma(lx, ivx, -oix), bx = __builtin_elementwise_fma(hx, ivx, -oix);
        const f32x2 ay = __builtin_elementwise_fma(ly, ivy, -oiy), by = __builtin_elementwise_fma(hy, ivy, -oiy);
        const f32x2 az = __builtin_elementwise_fma(lz, ivz, -oiz), bz = __builtin_elementwise_fma(hz, ivz, -oiz);
        const f32x2 tin = __builtin_elementwise_max(
            __builtin_elementwise_max(__builtin_elementwise_min(ax, bx), __builtin_elementwise_min(ay, by)),
            __builtin_elementwise_max(__builtin_elementwise_min(az, bz), zero));
        const f32x2 tout = __builtin_elementwise_min(
            __builtin_elementwise_min(__builtin_elementwise_max(ax, bx), __builtin_elementwise_max(ay, by)),
            __builtin_elementwise_max(az, bz));
        const f32x2 lim = tout * splat2(1.0000005f);
        m[2 * p] = __builtin_amdgcn_ballot_w64(tin[0] <= lim[0]) & __builtin_amdgcn_ballot_w64(tin[0] <= bound);
        m[2 * p + 1] = __builtin_amdgcn_ballot_w64(tin[1] <= lim[1]) & __builtin_amdgcn_ballot_w64(tin[1] <= bound);
        tin_out[2 * p] = tin[0];
        tin_out[2 * p + 1] = tin[1];
    }
}

// bvh_closest_queue over the FOUR-WIDE tree (round 4; DevScene.bvh_nodes4, BvhMeshRec.root4): the same queue of (ray, node)
// items and the same leaf list, but an item tests four boxes and pushes up to four children, so a walk sends half as many
// items through the queue - and what an item costs is mostly that trip (pop, the owner's ray constants by ds_bpermute, the
// pushes), not its box arithmetic.  No near-before-far order among the pushes: measured on mesh.json (walk statistics, and
// an offline replay of the frame's rays), ordering buys nothing - the boxes of a thin surface mesh overlap, 99 % of the
// popped items are still inside their owner's bound - and its compare / select instructions are not free.  `nodes4` is
// LDS (staged) or global memory; `nodes2` (the binary tree, global memory) serves the depth-first second walk of rays whose
// pushes did not fit.
template <class NodePtr4>
__device__ __forceinline__ void bvh_closest_queue4(const DevScene &S, NodePtr4 nodes4, const WalkQueue &Q, unsigned long long *keys,
                                                   bool walk, vec3 o, vec3 d, int32_t root4, int32_t root2, float best_t, float &mt,
                                                   int32_t &mid) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t act = __builtin_amdgcn_ballot_w64(true);
    const uint32_t n_act = (uint32_t)__builtin_popcountll(act);
    const uint32_t my = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
    auto prefix = [](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    auto count = [](uint64_t m) { return (uint32_t)__builtin_popcountll(m); };
    if (walk) keys[lane] = ((unsigned long long)(__float_as_uint(best_t) - 1u) << 32) | 0xffffffffull;
    if (my == 0u) Q.redo[0] = Q.redo[1] = 0u;
    const float big = 1e18f;  // see bvh_closest
    const vec3 inv = mk(__builtin_fmaxf(__builtin_fminf(1.0f / d.x, big), -big),
                        __builtin_fmaxf(__builtin_fminf(1.0f / d.y, big), -big),
                        __builtin_fmaxf(__builtin_fminf(1.0f / d.z, big), -big));
    const vec3 oi = mk(o.x * inv.x, o.y * inv.y, o.z * inv.z);
    uint32_t q_count, n_leaf;  // wave-uniform: box tests in ent[0, q_count), leaves in ent[cap - n_leaf, cap)
    {
        const uint64_t m_node = __builtin_amdgcn_ballot_w64(walk && root4 >= 0), m_leaf = __builtin_amdgcn_ballot_w64(walk && root4 < 0);
        q_count = count(m_node);
        n_leaf = count(m_leaf);
        if (walk && root4 >= 0) Q.ent[prefix(m_node)] = make_uint2(lane | ((uint32_t)root4 << 6), 0u);
        if (walk && root4 < 0) Q.ent[Q.cap - 1u - prefix(m_leaf)] = make_uint2(lane | ((uint32_t)~root4 << 6), 0u);
    }
    PT_WSTAT(S, 0, q_count + n_leaf);  // walks
    PT_WSTAT(S, 10, 1);                // wave-walks
    for (;;) {
        if (q_count == 0u && n_leaf == 0u) break;
        if (q_count != 0u) {
            PT_PHASE(kPhWalkBox);
            const uint32_t c = q_count < n_act ? q_count : n_act;
            q_count -= c;
            const bool has = my < c;
            uint2 e = make_uint2(lane, 0u);
            if (has) e = Q.ent[q_count + my];
            const uint32_t owner = e.x & 63u;
            const int sel = (int)(owner << 2);
            const float r_ivx = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(inv.x)));
            const float r_ivy = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(inv.y)));
            const float r_ivz = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(inv.z)));
            const float r_oix = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(oi.x)));
            const float r_oiy = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(oi.y)));
            const float r_oiz = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(oi.z)));
            const float bound = __uint_as_float((uint32_t)(load_key(&keys[owner]) >> 32) + 1u);
            const uint64_t m_valid = __builtin_amdgcn_ballot_w64(has) & ~__builtin_amdgcn_ballot_w64(__uint_as_float(e.y) > bound);
            PT_WSTAT(S, 1, 1);                                   // batches of box tests
            PT_WSTAT(S, 2, c);                                   // items
            PT_WSTAT(S, 3, __builtin_popcountll(m_valid));       // ... still worth testing
            uint64_t m_h[4];
            float tin[4];
            const BvhNode4 n = nodes4[e.x >> 6];  // (lanes without an item: node 0 with their own ray, a harmless read)
            hit_boxes4_masks(n, splat2(r_ivx), splat2(r_ivy), splat2(r_ivz), splat2(r_oix), splat2(r_oiy), splat2(r_oiz), bound, m_h, tin);
            uint64_t m_n[4], m_l[4];
            uint32_t add_q = 0u, add_l = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t neg = __builtin_amdgcn_ballot_w64(n.c[j] < 0);
                m_h[j] &= m_valid;
                m_n[j] = m_h[j] & ~neg;
                m_l[j] = m_h[j] & neg;
                add_q += count(m_n[j]);
                add_l += count(m_l[j]);
            }
            if (q_count + n_leaf + add_q + add_l > Q.cap) {  // wave-uniform: no room - these rays are walked again afterwards
                if (__builtin_amdgcn_inverse_ballot_w64(m_h[0] | m_h[1] | m_h[2] | m_h[3])) atomicOr(&Q.redo[owner >> 5], 1u << (owner & 31u));
                PT_WSTAT(S, 11, 1);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (__builtin_amdgcn_inverse_ballot_w64(m_n[j]))
                        Q.ent[q_count + prefix(m_n[j])] = make_uint2(owner | ((uint32_t)n.c[j] << 6), __float_as_uint(tin[j]));
                    q_count += count(m_n[j]);
                    if (__builtin_amdgcn_inverse_ballot_w64(m_l[j]))
                        Q.ent[Q.cap - 1u - (n_leaf + prefix(m_l[j]))] = make_uint2(owner | ((uint32_t)~n.c[j] << 6), __float_as_uint(tin[j]));
                    n_leaf += count(m_l[j]);
                }
            }
        }
        while (n_leaf >= n_act || (q_count == 0u && n_leaf != 0u)) {
            PT_PHASE(kPhWalkLeaf);
            const uint32_t cnt = n_leaf < n_act ? n_leaf : n_act;
            bool valid = my < cnt;
            uint2 e = make_uint2(lane, 0u);
            if (valid) e = Q.ent[Q.cap - n_leaf + my];
            n_leaf -= cnt;
            const uint32_t owner = e.x & 63u;
            const int sel = (int)(owner << 2);
            vec3 ro, rd;
            ro.x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.x)));
            ro.y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.y)));
            ro.z = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.z)));
            rd.x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.x)));
            rd.y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.y)));
            rd.z = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.z)));
            const float bound = __uint_as_float((uint32_t)(load_key(&keys[owner]) >> 32) + 1u);
            valid = valid && !(__uint_as_float(e.y) > bound);
            PT_WSTAT(S, 5, 1);
            PT_WSTAT(S, 6, cnt);
            PT_WSTAT(S, 12, __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid)));  // leaves still worth testing
            const uint32_t code = e.x >> 6;
            const uint32_t first = leaf_first(code), lc = valid ? leaf_count(code) : 0u;
            for (uint32_t r = 0; r < kBvhLeafPairs; ++r) {
                const bool has = r < lc;
                if (__builtin_amdgcn_ballot_w64(has) == 0ull) break;
                if (has) {
                    const TriPairRec tr = S.tri_pairs[first + r];
                    unsigned long long k2[2];
                    float th[2];
                    pair_test_keys(tr, ro, rd, k2, th);
                    atomicMin(&keys[owner], k2[0]);
                    atomicMin(&keys[owner], k2[1]);
                }
            }
        }
    }
    PT_PHASE(kPhWalkGate);
    if (walk) {
        const unsigned long long key = load_key(&keys[lane]);
        if ((uint32_t)key != 0xffffffffu) {
            mt = __uint_as_float((uint32_t)(key >> 32) + 1u);
            mid = (int32_t)(uint32_t)key;
        }
    }
    // rays whose pushes were dropped: depth-first from the root of the binary tree, with the bound found so far
    const uint32_t redo = __atomic_load_n(&Q.redo[lane >> 5], __ATOMIC_RELAXED);
    const bool again = walk && ((redo >> (lane & 31u)) & 1u) != 0u;
    if (__builtin_amdgcn_ballot_w64(again) != 0ull) {
        if (again) {
            StackDyn codec;
            codec.narrow = (S.bvh_in_lds & 2u) != 0u;
            codec.c16.pair_base = S.bvh_pair_base;
            const uint32_t eb = codec.entry_bytes();
            LeafLds L;  // [stacks: bvh_stack x 64 x u16|u32][leaf list: kLeafListCap x u32] in the queue's area
            L.keys = keys;
            L.list = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(Q.ent) + S.bvh_stack * 64u * eb);
            float mt2 = __builtin_inff();
            int32_t mid2 = -1;
            bvh_closest_postponed(S, S.bvh_nodes, codec, reinterpret_cast<char *>(Q.ent) + lane * eb, 64u * eb, L, o, d, root2,
                                  __builtin_fminf(best_t, mt), mt2, mid2);
            if (mid2 >= 0 && (mid < 0 || mt2 < mt || (mt2 == mt && (uint32_t)mid2 < (uint32_t)mid))) {
                mt = mt2;
                mid = mid2;
            }
        }
    }
}

// Which walker k_pass_cand / k_mega's candidate forms use (A/B builds; the tests hold both to the same bits):
//   2 (default)  bvh_closest_queue4: the walk queue over the four-wide tree
//   0            bvh_closest_queue: the walk queue over the binary tree (rounds 2-3)
// (Round 4 also built a third form - lanes that keep an item while they DESCEND, going on with the nearer of two children
// that are hit and pushing only the farther one onto a pool that idle lanes refill from, sixteen at a time: the order of a
// depth-first walk per ray at the density of the queue.  Bit-identical, and no faster: 25.8 against 26.4 G bounces/s on
// mesh.json - 14.3 node visits and 5.2 leaves per walk as before, 89 % of the far subtrees still inside their owner's
// bound when popped: on a thin surface mesh the boxes along a ray overlap, there is little for best-t pruning to prune,
// and an offline replay of the frame's rays against the same tree says so too: 10.2 / 2.7 with ideal depth-first order
// against 12.9 / 3.9 with none.)
#ifndef PT_WALK_FORM
#define PT_WALK_FORM 2
#endif


// LDS carve-up of the kernels that intersect: [BvhNode x n_bvh_nodes][u16 stack: kBvhStack x blockDim] when the
// nodes are staged, else [u32 stack: kBvhStack x blockDim] alone (nodes read from global memory)
__device__ __forceinline__ void stage_bvh(const DevScene &S, uint4 *lds) {
    if (S.n_bvh_nodes != 0u && (S.bvh_in_lds & 1u) != 0u) {
        const uint4 *src = reinterpret_cast<const uint4 *>(S.bvh_nodes);
        for (uint32_t i = threadIdx.x; i < S.n_bvh_nodes * 4u; i += blockDim.x) lds[i] = src[i];
        __syncthreads();
    }
}
__host__ __device__ inline size_t bvh_lds_bytes(const DevScene &S, uint32_t block) {
    if (S.n_bvh_nodes == 0u) return 0;
    return ((S.bvh_in_lds & 1u) ? (size_t)S.n_bvh_nodes * sizeof(BvhNode) : 0u) +
           (size_t)kBvhStack * block * ((S.bvh_in_lds & 2u) ? sizeof(uint16_t) : sizeof(uint32_t));
}

#if defined(__HIPCC__)
// one lane's walk of one mesh, with the node storage and stack width the scene was set up for
__device__ __forceinline__ void bvh_walk(const DevScene &S, uint4 *lds, vec3 o, vec3 d, int32_t root, float best_t,
                                         float &mt, int32_t &mid, const LeafLds *leaves = nullptr) {
    uint4 *const stacks = (S.bvh_in_lds & 1u) ? lds + S.n_bvh_nodes * 4u : lds;
    if (leaves) {  // k_pass_bvh, k_pass_cand: nodes in global memory, leaves postponed
        StackDyn codec;
        codec.narrow = (S.bvh_in_lds & 2u) != 0u;
        codec.c16.pair_base = S.bvh_pair_base;
        const uint32_t eb = codec.entry_bytes();
        bvh_closest_postponed(S, S.bvh_nodes, codec, reinterpret_cast<char *>(stacks) + threadIdx.x * eb, blockDim.x * eb, *leaves,
                              o, d, root, best_t, mt, mid);
        return;
    }
    if (S.bvh_in_lds & 2u) {
        uint16_t *stack = reinterpret_cast<uint16_t *>(stacks) + threadIdx.x;
        Stack16 codec;
        codec.pair_base = S.bvh_pair_base;
        if (S.bvh_in_lds & 1u)
            bvh_closest(S, reinterpret_cast<const BvhNode *>(lds), codec, stack, blockDim.x, o, d, root, best_t, mt, mid);
        else
            bvh_closest(S, S.bvh_nodes, codec, stack, blockDim.x, o, d, root, best_t, mt, mid);
    } else {
        uint32_t *stack = reinterpret_cast<uint32_t *>(stacks) + threadIdx.x;
        bvh_closest(S, S.bvh_nodes, Stack32(), stack, blockDim.x, o, d, root, best_t, mt, mid);
    }
}
#endif

// One object of a pair: the tail of SceneObjectData::intersect (mod.rs:261-280) + intersect_scene's replace
// rule (mod.rs:649), given the sphere discriminant arithmetic (b, det) that was done packed for both halves.
//   EXACT_GATES = false is the speculative pass: a mesh is admitted when its bounding-sphere discriminant is
//   non-negative - a necessary condition of the gate that needs no square root - instead of the full
//   intersect_sphere(...).is_some() (mod.rs:267-273).  intersect_scene_dev verifies the winner afterwards.
//   DEFER_WALK: a mesh with a BVH is not walked here; its gate is evaluated exactly and *want_walk reports whether
//   the ray has to walk it (k_intersect<true> collects such rays and walks them in dense waves: walk_deferred).
template <bool BVH, bool EXACT_GATES, bool DEFER_WALK = false>
__device__ __forceinline__ void consider_object(const DevScene &S, const ObjPairRec &ob, int hf, float b, float det,
                                                vec3 o, vec3 d, uint4 *lds, float &best_t, int32_t &best_id,
                                                bool admit = false, bool *want_walk = nullptr) {
    const float eps = 1e-4f;
    bool sph_hit;
    if (ob.kind[hf] == kKindSphere) {
        const float sq = f_sqrt(det);  // NaN when det < 0: both comparisons below are then false
        const float t0 = b - sq, t1 = b + sq;
        const bool near_ok = t0 >= eps, far_ok = t1 >= eps;
        sph_hit = !(det < 0.0f) && (near_ok || far_ok);
        const float t = near_ok ? t0 : t1;
        if (sph_hit && t < best_t) {
            best_t = t;
            best_id = (int32_t)ob.obj[hf];
        }
        return;
    }
    if (BVH && DEFER_WALK && ob.bvh_root[hf] != kNoBvh && S.n_bvh_nodes != 0u) {  // wave-uniform
        const float sq = f_sqrt(det);
        bool pass = !(det < 0.0f) && ((b - sq) >= eps || (b + sq) >= eps);
        // The first step of the walk, taken here with the root node in SGPRs: a ray that passes the (loose) sphere
        // but misses both boxes of the root has nothing to find and is not parked.  best_t is the best of the objects
        // visited so far, an upper bound of the bound the walk itself would use.
        const int32_t root = ob.bvh_root[hf];
        if (root >= 0 && __builtin_amdgcn_ballot_w64(pass) != 0ull) {
            const BvhNode rn = ld_uniform(S.bvh_nodes + root);
            const float big = 1e18f;
            const vec3 inv = mk(__builtin_fmaxf(__builtin_fminf(1.0f / d.x, big), -big),
                                __builtin_fmaxf(__builtin_fminf(1.0f / d.y, big), -big),
                                __builtin_fmaxf(__builtin_fminf(1.0f / d.z, big), -big));
            bool h0, h1;
            f32x2 tin;
            hit_boxes(rn, splat2(inv.x), splat2(inv.y), splat2(inv.z), splat2(o.x * inv.x), splat2(o.y * inv.y),
                      splat2(o.z * inv.z), best_t, &h0, &h1, &tin);
            pass = pass && (h0 || h1);
        }
        if (pass) *want_walk = true;
        return;
    }
    if (EXACT_GATES) {
        const float sq = f_sqrt(det);
        sph_hit = !(det < 0.0f) && ((b - sq) >= eps || (b + sq) >= eps);
    } else {
        sph_hit = admit || !(det < 0.0f);
    }
    // bounding-sphere gate (mod.rs:267-273): skip the triangle list when no lane passes
    if (!admit && __builtin_amdgcn_ballot_w64(sph_hit) == 0ull) return;
    float mt = __builtin_inff();
    int32_t mid = -1;
    const int32_t root = ob.bvh_root[hf];
    if (BVH && root != kNoBvh && S.n_bvh_nodes != 0u) {
        if (sph_hit) {  // per lane: divergent traversal
            bvh_walk(S, lds, o, d, root, best_t, mt, mid);
        }
    } else {
        const f32x2 ox2 = splat2(o.x), oy2 = splat2(o.y), oz2 = splat2(o.z);
        const f32x2 dx2 = splat2(d.x), dy2 = splat2(d.y), dz2 = splat2(d.z);
        const uint32_t pb = ob.pair_begin[hf], pc = ob.pair_count[hf];
        const uint32_t plane = S.planar ? (ob.admit[hf] >> 1) & 3u : 0u;  // wave-uniform
        if (root == kNoBvh && plane != 0u) {
            for (uint32_t p = 0; p < pc; ++p) {
                const TriPairRec tr = ld_uniform(S.tri_pairs + (pb + p));
                if (plane == 1u)
                    test_pair_planar<0, true>(tr, ox2, oy2, oz2, dx2, dy2, dz2, mt, mid);
                else if (plane == 2u)
                    test_pair_planar<1, true>(tr, ox2, oy2, oz2, dx2, dy2, dz2, mt, mid);
                else
                    test_pair_planar<2, true>(tr, ox2, oy2, oz2, dx2, dy2, dz2, mt, mid);
            }
        } else if (root == kNoBvh) {
            for (uint32_t p = 0; p < pc; ++p)  // wave-uniform index -> scalar loads
                test_pair<true>(ld_uniform(S.tri_pairs + (pb + p)), ox2, oy2, oz2, dx2, dy2, dz2, mt, mid);
        } else {  // BVH switched off for this frame: the reference's full scan over BVH-ordered records
            for (uint32_t p = 0; p < pc; ++p) test_pair<false>(ld_uniform(S.tri_pairs + (pb + p)), ox2, oy2, oz2, dx2, dy2, dz2, mt, mid);
        }
    }
    if (sph_hit && mid >= 0 && mt < best_t) {
        best_t = mt;
        best_id = (int32_t)S.n_objs + mid;
    }
}

// ---------------------------------------------------------------------------------------------
// closest hit.  Objects in reverse index order, strict '<' (ties keep the higher object index,
// mod.rs:637,649); inside a mesh the first triangle in list order wins ties (mod.rs:598).
// `best` starts at +inf instead of Option::None: identical for every finite, non-NaN distance.
// BVH = false compiles the traversal out (scenes without a BVH mesh keep the small register footprint).
template <bool BVH, bool EXACT_GATES, bool DEFER_WALK = false>
__device__ __forceinline__ HitRec scan_scene(const DevScene &S, vec3 o, vec3 d, uint4 *lds, bool *want_walk = nullptr) {
    float best_t = __builtin_inff();
    int32_t best_id = -1;
    const f32x2 ox2 = splat2(o.x), oy2 = splat2(o.y), oz2 = splat2(o.z);
    const f32x2 dx2 = splat2(d.x), dy2 = splat2(d.y), dz2 = splat2(d.z);
    const uint32_t n_pairs = (S.n_objs + 1u) >> 1;
    for (uint32_t p = 0; p < n_pairs; ++p) {
        const ObjPairRec ob = ld_uniform(S.obj_pairs + p);  // wave-uniform -> scalar loads
        // Speculative pass: a mesh flagged `admit` goes straight to its triangles.  Admitting more meshes than the
        // reference keeps the argument of intersect_scene_dev intact (winner over a superset, verified afterwards);
        // what it saves is the gate arithmetic of spheres that no wave ever misses (the walls of a room).
        const bool a0 = !EXACT_GATES && (ob.admit[0] & 1u) != 0u, a1 = !EXACT_GATES && (ob.admit[1] & 1u) != 0u;
        if (a0 && a1) {  // wave-uniform
            consider_object<BVH, EXACT_GATES, DEFER_WALK>(S, ob, 0, 0.0f, 0.0f, o, d, lds, best_t, best_id, true, want_walk);
            consider_object<BVH, EXACT_GATES, DEFER_WALK>(S, ob, 1, 0.0f, 0.0f, o, d, lds, best_t, best_id, true, want_walk);
            continue;
        }
        // intersect_sphere's discriminant (mod.rs:413-416) for both objects of the pair
        const f32x2 opx = ld2(ob.cx) - ox2, opy = ld2(ob.cy) - oy2, opz = ld2(ob.cz) - oz2;
        const f32x2 b = (opx * dx2 + opy * dy2) + opz * dz2;
        const f32x2 det = (b * b - ((opx * opx + opy * opy) + opz * opz)) + ld2(ob.rr);
        consider_object<BVH, EXACT_GATES, DEFER_WALK>(S, ob, 0, b[0], det[0], o, d, lds, best_t, best_id, a0, want_walk);
        if (2u * p + 1u < S.n_objs)  // the second half of the last pair of an odd scene is a filler (wave-uniform)
            consider_object<BVH, EXACT_GATES, DEFER_WALK>(S, ob, 1, b[1], det[1], o, d, lds, best_t, best_id, a1, want_walk);
    }
    HitRec h;
    h.t = best_t;
    h.id = best_id;
    return h;
}

// Speculate on the gates, verify the winner.  The speculative scan admits a superset of the meshes the reference
// admits (discriminant >= 0 is necessary for intersect_sphere to return Some), so its winner is the minimum over a
// superset of the reference's candidates under the same order and tie rules: if that winner is itself a reference
// candidate - a sphere, or a triangle of a mesh whose exact gate passes - it IS the reference's winner.  Only the
// winner's own gate has to be evaluated exactly (one square root per ray instead of one per mesh); when it fails for
// some lane (rounding at the rim of a bounding sphere) the wave repeats the scan with exact gates.
template <bool BVH, bool DEFER_WALK = false>
__device__ __forceinline__ HitRec intersect_scene_dev(const DevScene &S, vec3 o, vec3 d, uint4 *lds,
                                                      bool *want_walk = nullptr) {
    HitRec h = scan_scene<BVH, false, DEFER_WALK>(S, o, d, lds, want_walk);
    bool suspect = false;
    if (h.id >= (int32_t)S.n_objs) {
        const uint32_t owner = S.tri_shade[h.id - (int32_t)S.n_objs].owner;
        const ObjRec g = S.objs[owner];  // per-lane gather: the winner's bounding sphere
        // Usually the hit point lies well inside that sphere (it is a point of a triangle the sphere encloses): a point
        // of the ray, ahead of the origin, deeper than 0.1 % of the radius inside the sphere means a chord long enough
        // that the f32 gate - discriminant >= 0, far root >= 1e-4 - cannot fail (pt_host.cpp sets rr_in with the
        // margins, or -1).  Only when some lane cannot show that is the gate evaluated as the reference does.
        const vec3 pc = (o + d * h.t) - mk(g.cx, g.cy, g.cz);
        const bool deep = dot(pc, pc) <= g.rr_in;  // false on NaN
        if (__builtin_amdgcn_ballot_w64(!deep) != 0ull) {
            const vec3 op = mk(g.cx, g.cy, g.cz) - o;
            const float b = dot(op, d);
            const float det = b * b - dot(op, op) + g.rr;
            const float sq = f_sqrt(det);
            suspect = !(!(det < 0.0f) && ((b - sq) >= 1e-4f || (b + sq) >= 1e-4f));
        }
    }
    if (__builtin_amdgcn_ballot_w64(suspect) != 0ull) h = scan_scene<BVH, true, DEFER_WALK>(S, o, d, lds, want_walk);
    return h;
}

// ---------------------------------------------------------------------------------------------
// Candidate scan (scenes whose meshes are all scanned pair by pair - no BVH - i.e. k_pass).
//
// intersect_scene tests every ray against every triangle; on the bench scene that is 14 Moller-Trumbore tests per ray of
// which one or two can succeed, and the loop over objects is wave-uniform: 64 unrelated rays always need all of them.
// Here the exact test runs only on (ray, pair record) CANDIDATES, and in full waves:
//   1. every lane tests its ray against the spheres (exactly, two per packed instruction) - the best of them is the
//      ray's first key - and against a cheap conservative FILTER of each pair record (FlatPairRec: distance to the plane,
//      point in the padded rectangle, not farther than the best sphere; records without a filter are always candidates);
//   2. candidates go to a per-wave ring in LDS as (ray slot, record); whenever 64 are queued the wave runs one dense
//      batch: lane e fetches the ray of entry e's owner (ds_bpermute from the owner's registers), gathers the record per
//      lane and runs the reference's arithmetic (test_pair: identical bits to the scan's) and the mesh's bounding-sphere
//      gate (mod.rs:267-273: a mesh whose gate fails contributes nothing), then folds the result into the owner's key
//      with one LDS atomic minimum.  What is left in the ring (< 64 entries) waits for the next 64 rays of the wave:
//      k_pass finishes a ray one chunk after it started it, so batches run full;
//   3. key = (distance bits << 32) | visiting rank: the minimum is the closest hit, the earliest visited among equal
//      distances - intersect_scene's and Triangle::intersect's strict '<' (mod.rs:598,649) - whatever order the
//      candidates were processed in.
// Exactness: the filter may only ever reject a (ray, record) whose exact test would fail or lose: its pads come from the
// forward-error analysis that sizes the BVH boxes (pt_host.cpp), grazing rays are not judged at all, and a hit farther
// than the best sphere cannot win.
struct CandLds {
    unsigned long long *keys;   // [2][64] of this wave: slot = (chunk parity << 6) | lane
    float4 *ray_a;              // [2][64]: origin xyz, direction x of the slot's ray
    float2 *ray_b;              // [2][64]: direction yz
    uint16_t *queue;            // [kCandQueueCap] ring of this wave: slot | record << 7
    const CandPairRec *staged;  // the workgroup's LDS copy of DevScene.cand_pairs (STAGED), else unused
    const SurfRec *surf;        // the workgroup's LDS copy of DevScene.surf (STAGED), else DevScene.surf
};
constexpr unsigned long long kKeyMiss = 0x7f800000ffffffffull;  // +inf, last rank

// wave-uniform state of the ring
struct CandRing {
    uint32_t head, count;
};

// The verdicts are WAVE MASKS, not per-lane booleans: every compare lands in an SGPR pair anyway, and combining the pairs
// with scalar ands / ors keeps the VALU out of it - a ballot of a boolean that is itself a combination of compares costs a
// v_cndmask and a v_cmp to get back to the mask the compares already were (16 issue cycles per record).
// THE LOWER END OF THE DISTANCE TEST.  A ray that leaves a wall starts ON that wall's plane - the reference's triangles have no
// self-intersection epsilon, so Triangle::intersect is evaluated against the wall the ray came from, and its `distance <= 0`
// (mod.rs:592) decides by rounding.  With the slack a conservative filter needs on distances (`t >= -tpad`) that wall is a
// candidate of EVERY such ray: 0.8 of the 2.0 candidate records per ray on cornell.  But for triangles in a plane
// perpendicular to axis a (e1_a = e2_a = 0) the sign of the computed distance is known without computing it:
//   distance = ((0 + e2_b * (-(e1_c * t_a))) + e2_c * (t_a * e1_b)) * (1 / det),  t_a = o_a - A_a              (mod.rs:577-589)
//   det      =  (0 + e1_b * (-(e2_c * d_a))) + e1_c * (d_a * e2_b)                                               (mod.rs:563-564)
// - every term carries t_a (or d_a) as a factor, rounding keeps the sign of a product and of a sum whose terms do not nearly
// cancel (pt_host.cpp: |e1_b e2_c - e1_c e2_b| >= 0.001 (|e1_b e2_c| + |e1_c e2_b|) for both triangles), so
// sign(distance) = sign(A_a - o_a) * sign(d_a), and t_a = 0 gives distance = +-0: `distance <= 0` rejects exactly the rays
// with (A_a - o_a) * sign(d_a) <= 0 (an underflow to zero only rejects more).  For such records the filter keeps a ray only
// if (plane - origin) * sign(d_a) > 0: about two thirds of the self-candidates go (the origin is on the inner side of the plane,
// or exactly on it).
template <int AXIS, bool EXACT>
__device__ __forceinline__ void filter_flat(const FlatPairRec &f, vec3 o, vec3 d, vec3 inv, float bound, uint64_t valid_m,
                                            uint64_t graze, uint64_t *m0, uint64_t *m1) {
    const float oa = AXIS == 0 ? o.x : (AXIS == 1 ? o.y : o.z), ob = AXIS == 0 ? o.y : (AXIS == 1 ? o.z : o.x),
                oc = AXIS == 0 ? o.z : (AXIS == 1 ? o.x : o.y);
    const float db = AXIS == 0 ? d.y : (AXIS == 1 ? d.z : d.x), dc = AXIS == 0 ? d.z : (AXIS == 1 ? d.x : d.y);
    const float ia = AXIS == 0 ? inv.x : (AXIS == 1 ? inv.y : inv.z);
    const f32x2 tv = ld2(f.pc) - splat2(oa);  // -(tvec_a) of mod.rs:577: exact sign, zero iff the origin is on the plane
    const f32x2 t2 = tv * splat2(ia);         // distance to the plane (approximate reciprocal)
    const f32x2 yb = __builtin_elementwise_fma(splat2(db), t2, splat2(ob)) - ld2(f.cb);
    const f32x2 zc = __builtin_elementwise_fma(splat2(dc), t2, splat2(oc)) - ld2(f.cc);
    const f32x2 lim = splat2(bound) + ld2(f.tpad);
    uint64_t in[2];
    if (EXACT) {  // (records with FlatPairRec.sign_exact: a loop of their own, cand_filter_and_drain)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            in[hf] = __builtin_amdgcn_ballot_w64(f_abs(yb[hf]) <= f.hb[hf]) & __builtin_amdgcn_ballot_w64(f_abs(zc[hf]) <= f.hc[hf]) &
                     __builtin_amdgcn_ballot_w64(t2[hf] <= lim[hf]);
        // the sign test also holds for grazing rays (it does not use the reciprocal): it is applied to them too
        // "the ray moves towards the plane", (plane - origin) * sign(d_a) > 0, read off the approximate distance: v_rcp_f32 keeps
        // the sign of d_a (+-0 -> +-inf) and its magnitude is at least 1 - 2^-23 (a unit direction), so tv * ia has the sign of
        // tv * sign(d_a), is never rounded to zero for tv != 0, and is NaN (0 * inf) or +-0 - not > 0 - for tv = 0
        const uint64_t fwd0 = __builtin_amdgcn_ballot_w64(t2[0] > 0.0f), fwd1 = __builtin_amdgcn_ballot_w64(t2[1] > 0.0f);
        *m0 = f.pair[0] != kNoPair ? ((graze | in[0]) & fwd0 & valid_m) : 0ull;
        *m1 = f.pair[1] != kNoPair ? ((graze | in[1]) & fwd1 & valid_m) : 0ull;
        return;
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
        in[hf] = __builtin_amdgcn_ballot_w64(f_abs(yb[hf]) <= f.hb[hf]) & __builtin_amdgcn_ballot_w64(f_abs(zc[hf]) <= f.hc[hf]) &
                 __builtin_amdgcn_ballot_w64(t2[hf] >= -f.tpad[hf]) & __builtin_amdgcn_ballot_w64(t2[hf] <= lim[hf]);
    *m0 = f.pair[0] != kNoPair ? ((graze | in[0]) & valid_m) : 0ull;
    *m1 = f.pair[1] != kNoPair ? ((graze | in[1]) & valid_m) : 0ull;
}

// intersect_sphere against the scene's spheres (mod.rs:412-427), two per record, in visiting order (strict '<' keeps
// the first): the ray's first key
__device__ __forceinline__ unsigned long long cand_spheres(const DevScene &S, vec3 o, vec3 d, float *best_out) {
    float best_t = __builtin_inff();
    uint32_t best_rank = 0xffffffffu;
    const f32x2 ox2 = splat2(o.x), oy2 = splat2(o.y), oz2 = splat2(o.z);
    const f32x2 dx2 = splat2(d.x), dy2 = splat2(d.y), dz2 = splat2(d.z);
    for (uint32_t p = 0; p < S.n_sph_pairs; ++p) {
        const SphPairRec sp = ld_uniform(S.sph_pairs + p);
        const f32x2 opx = ld2(sp.cx) - ox2, opy = ld2(sp.cy) - oy2, opz = ld2(sp.cz) - oz2;
        const f32x2 b = (opx * dx2 + opy * dy2) + opz * dz2;
        const f32x2 det = (b * b - ((opx * opx + opy * opy) + opz * opz)) + ld2(sp.rr);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (__builtin_amdgcn_ballot_w64(!(det[hf] < 0.0f)) == 0ull) continue;  // no lane of the wave hits this one
            const float sq = f_sqrt(det[hf]);  // NaN when det < 0: both comparisons below are then false
            const float t0 = b[hf] - sq, t1 = b[hf] + sq;
            const bool near_ok = t0 >= 1e-4f, far_ok = t1 >= 1e-4f;
            const float t = near_ok ? t0 : t1;
            if (!(det[hf] < 0.0f) && (near_ok || far_ok) && t < best_t) {
                best_t = t;
                best_rank = sp.rank[hf];
            }
        }
    }
    *best_out = best_t;
    return ((unsigned long long)__float_as_uint(best_t) << 32) | best_rank;
}
// (Round 4 tried to run the tail - square root, two roots, the 1e-4 tests - once per trip instead of once per sphere: a lane
// NOTES the sphere whose discriminant is non-negative (b, det, rank) and the tail runs after the last record, and in between
// only when a lane that already holds one meets a second.  Same keys, and no fewer instructions: 847 against 848 per bounce
// (PMC) - some lane of 64 unrelated rays points at two of cornell's four spheres in nearly every trip, also when spheres
// behind the origin are left out (852: the two extra compares per sphere).)

// One dense batch: the `count` (<= 64) oldest ring entries.  Every lane of the wave takes part.  An entry's ray is read
// from its slot in LDS (k_pass_cand keeps the rays of the two chunks in flight there).
template <bool STAGED>
__device__ __forceinline__ void cand_batch(const DevScene &S, const CandLds &L, CandRing &R, uint32_t lane, uint32_t count) {
    PT_PHASE_N(kPhBatch, count);  // (the budget's "lanes" of this phase: entries in the batch)
    const bool valid = lane < count;
    const uint32_t at = (R.head + lane) & (kCandQueueCap - 1u);
    R.head = (R.head + count) & (kCandQueueCap - 1u);
    R.count -= count;
    if (valid) {
        const uint32_t ent = (uint32_t)L.queue[at];
        const uint32_t slot = ent & 127u, q = ent >> 7;
        const float4 ra = L.ray_a[slot];
        const float2 rb = L.ray_b[slot];
        const vec3 ro = mk(ra.x, ra.y, ra.z), rd = mk(ra.w, rb.x, rb.y);
        CandPairRec tr;
        if (STAGED)
            tr = L.staged[q];  // per-lane gather from LDS
        else
            tr = S.cand_pairs[q];
        unsigned long long keys2[2];
        float t_hit[2];
        pair_test_keys(tr, ro, rd, keys2, t_hit);
        // the mesh's gate (mod.rs:267-273), for the nearer accepted triangle of the record (a gate that fails rejects
        // both: it depends on the ray and the mesh only).  Usually the hit point lies well inside the bounding sphere
        // (ObjRec.rr_in: then the f32 gate cannot fail); otherwise intersect_sphere(...).is_some() as the reference does.
        const bool first = keys2[0] <= keys2[1];
        const unsigned long long kmin = first ? keys2[0] : keys2[1];
        const float mt = first ? t_hit[0] : t_hit[1];
        bool pass = (uint32_t)(kmin >> 32) < 0x7f800000u;
        const vec3 g = mk(tr.gx, tr.gy, tr.gz);
        const vec3 pc = (ro + rd * mt) - g;
        const bool deep = dot(pc, pc) <= tr.grr_in;  // false on NaN / when no shortcut is offered
        if (pass && !deep) {
            const vec3 og = g - ro;
            const float b = dot(og, rd);
            const float det = (b * b - dot(og, og)) + tr.grr;
            const float sq = f_sqrt(det);
            pass = !(det < 0.0f) && ((b - sq) >= 1e-4f || (b + sq) >= 1e-4f);
        }
        // key of a hit = (bits(distance) << 32) | rank; k holds bits - 1: adding 1 << 32 restores it
        if (pass) atomicMin(&L.keys[slot], kmin + (1ull << 32));
    }
}

// Filters of the current chunk's ray (o, d; `valid` lanes) -> ring; full batches are run as they fill up.  `par` is the
// chunk's parity (its slots).
template <bool STAGED>
__device__ __forceinline__ void cand_filter_and_drain(const DevScene &S, const CandLds &L, CandRing &R, uint32_t lane,
                                                      uint32_t par, bool valid, vec3 o, vec3 d, float bound) {
    // both halves of a filter record in one step: entries of half 0 first, then half 1.  (Collecting the filters' verdicts as
    // bits per lane and writing the set bits to the ring one per lane at a time - fewer ballot / prefix rounds when every
    // lane has one or two candidates - was tried: the rounds follow the lane with the most candidates; cornell 36.8
    // against 37.5 G bounces/s.  One position per lane from the joint mask when no lane pushes both halves - opposite walls
    // under the sign rule: nearly always - saves four instructions per record and loses: 45.7 against 46.65.)
    auto push2 = [&](uint64_t m0, uint32_t q0, uint64_t m1, uint32_t q1) {  // the lanes of m0 push record q0, those of m1 q1
        if ((m0 | m1) == 0ull) return;
        const uint32_t n0 = (uint32_t)__builtin_popcountll(m0);
        const uint32_t base = R.head + R.count;
        const uint32_t me = lane | (par << 6);
        if (__builtin_amdgcn_inverse_ballot_w64(m0)) {
            PT_PHASE(kPhPush);
            const uint32_t at = (base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))) & (kCandQueueCap - 1u);
            L.queue[at] = (uint16_t)(me | (q0 << 7));
        }
        if (__builtin_amdgcn_inverse_ballot_w64(m1)) {
            PT_PHASE(kPhPush);
            const uint32_t at = (base + n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))) & (kCandQueueCap - 1u);
            L.queue[at] = (uint16_t)(me | (q1 << 7));
        }
        PT_PHASE(kPhFilter);
        R.count += n0 + (uint32_t)__builtin_popcountll(m1);
    };
    auto drain = [&]() {
#ifdef PT_PHASE_STATS
        const bool any = R.count >= 64u;
#endif
        while (R.count >= 64u) cand_batch<STAGED>(S, L, R, lane, 64u);
#ifdef PT_PHASE_STATS
        if (any) PT_PHASE(kPhFilter);
#endif
    };
    PT_PHASE(kPhFilter);
    const vec3 inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    // (Noting a lane's first two candidate records in registers while the filters run and pushing twice at the end of the
    // loop instead of twice per record - a third candidate pushed on the spot - was built and measured: some lane of a wave
    // nearly always grazes a plane (|d_a| < 1/64 makes it a candidate of both halves of that axis' record), so the "rare"
    // third-candidate push runs in most trips on top of the two at the end: cornell 36.3 against 39.6 G bounces/s.)
    const uint64_t valid_m = __builtin_amdgcn_ballot_w64(valid);
    // rays that graze the planes of an axis (|d_a| < kGrazing, or NaN): not judged by the filters of that axis
    const uint64_t gz_x = __builtin_amdgcn_ballot_w64(!(f_abs(d.x) >= kGrazing)), gz_y = __builtin_amdgcn_ballot_w64(!(f_abs(d.y) >= kGrazing)),
                   gz_z = __builtin_amdgcn_ballot_w64(!(f_abs(d.z) >= kGrazing));
    // (Issuing the next record's scalar load before this record's arithmetic - a second sixteen-register tuple - was measured in
    // round 4: 44.3 against 47.2 G bounces/s; the kernel is short of scalar registers, not of time to wait for them.)
    for (uint32_t p = 0; p < S.n_flat_exact; ++p) {  // the records with the exact sign rule (the host puts them first)
        const FlatPairRec f = ld_uniform(S.flat_pairs + p);
        uint64_t m0, m1;
        if (f.axis == 0u)
            filter_flat<0, true>(f, o, d, inv, bound, valid_m, gz_x, &m0, &m1);
        else if (f.axis == 1u)
            filter_flat<1, true>(f, o, d, inv, bound, valid_m, gz_y, &m0, &m1);
        else
            filter_flat<2, true>(f, o, d, inv, bound, valid_m, gz_z, &m0, &m1);
        push2(m0, f.pair[0], m1, f.pair[1]);
        drain();
    }
    for (uint32_t p = S.n_flat_exact; p < S.n_flat_pairs; ++p) {
        const FlatPairRec f = ld_uniform(S.flat_pairs + p);
        uint64_t m0, m1;
        if (f.axis == 0u)
            filter_flat<0, false>(f, o, d, inv, bound, valid_m, gz_x, &m0, &m1);
        else if (f.axis == 1u)
            filter_flat<1, false>(f, o, d, inv, bound, valid_m, gz_y, &m0, &m1);
        else
            filter_flat<2, false>(f, o, d, inv, bound, valid_m, gz_z, &m0, &m1);
        push2(m0, f.pair[0], m1, f.pair[1]);
        drain();
    }
    for (uint32_t q = 0; q < S.n_other_pairs; ++q) {  // records without a filter: a candidate for every ray
        push2(valid_m, q, 0ull, 0u);
        drain();
    }
}

// The BVH walks of a ray whose other objects are done (best_t / best_id hold their winner): every mesh with a BVH, in
// visiting order, exact gate first.  A triangle strictly closer than the best so far wins; one exactly as far would
// need the reference's visiting order to decide - reported, and the caller repeats that ray with the in-order scan.
__device__ __forceinline__ bool walk_deferred(const DevScene &S, vec3 o, vec3 d, uint4 *lds, float &best_t,
                                              int32_t &best_id, const LeafLds *leaves = nullptr) {
    bool tie = false;
    const uint32_t n_pairs = (S.n_objs + 1u) >> 1;
    for (uint32_t p = 0; p < n_pairs; ++p) {
        const ObjPairRec ob = ld_uniform(S.obj_pairs + p);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int32_t root = ob.bvh_root[hf];
            if (ob.kind[hf] != kKindMesh || root == kNoBvh) continue;  // wave-uniform
            const vec3 op = mk(ob.cx[hf], ob.cy[hf], ob.cz[hf]) - o;   // intersect_sphere, mod.rs:413-427
            const float b = dot(op, d);
            const float det = (b * b - dot(op, op)) + ob.rr[hf];
            const float sq = f_sqrt(det);
            const bool pass = !(det < 0.0f) && ((b - sq) >= 1e-4f || (b + sq) >= 1e-4f);
            if (__builtin_amdgcn_ballot_w64(pass) == 0ull) continue;
            float mt = __builtin_inff();
            int32_t mid = -1;
            if (pass) {
                bvh_walk(S, lds, o, d, root, best_t, mt, mid, leaves);
            }
            if (pass && mid >= 0) {
                if (mt < best_t) {
                    best_t = mt;
                    best_id = (int32_t)S.n_objs + mid;
                } else if (mt == best_t) {
                    tie = true;
                }
            }
        }
    }
    return tie;
}

// The candidate scan's form of the two steps above (k_pass_cand<.., BVH>): a ray's state is its key - (distance bits << 32)
// | visiting rank of the best primitive so far.
// 1. does the ray have to walk a BVH mesh at all?  Exact gate (mod.rs:267-273), then the first step of the walk with the
//    root node in SGPRs: boxes farther than the best hit so far (the spheres and the candidate records are done) cannot hold
//    a winner - a triangle exactly as far is still looked at (hit_boxes prunes on '>' only), ranks decide among equals.
__device__ __forceinline__ bool bvh_wants(const DevScene &S, vec3 o, vec3 d, float best_t) {
    // Only a superset of the rays with something to find is needed here (the walk evaluates the gate exactly), so:
    //  * of the gate only the discriminant's sign (no square root): a sphere behind the origin is left to the box test;
    //  * the slab test with approximate reciprocals (v_rcp_f32: 1 ulp) instead of correctly rounded divisions; each slab
    //    distance is then off by a relative 2^-22 at most, which the comparisons allow for (entry distance scaled down,
    //    exit distance and bound scaled up by 4e-7 each - on top of the pads the boxes carry for the exact test).
    PT_PHASE(kPhWants);
    bool want = false;
    const float big = 1e18f;
    const vec3 inv = mk(__builtin_fmaxf(__builtin_fminf(__builtin_amdgcn_rcpf(d.x), big), -big),
                        __builtin_fmaxf(__builtin_fminf(__builtin_amdgcn_rcpf(d.y), big), -big),
                        __builtin_fmaxf(__builtin_fminf(__builtin_amdgcn_rcpf(d.z), big), -big));
    const f32x2 ivx = splat2(inv.x), ivy = splat2(inv.y), ivz = splat2(inv.z);
    const f32x2 oix = splat2(o.x * inv.x), oiy = splat2(o.y * inv.y), oiz = splat2(o.z * inv.z);
    const float bound_up = best_t * 1.0000004f;
    for (uint32_t q = 0; q < S.n_bvh_meshes; ++q) {
        const BvhMeshRec bm = ld_uniform(S.bvh_meshes + q);
        const vec3 op = mk(bm.cx, bm.cy, bm.cz) - o;  // intersect_sphere's discriminant, mod.rs:413-416
        const float b = dot(op, d);
        const float det = (b * b - dot(op, op)) + bm.rr;
        bool pass = !(det < 0.0f);
        if (bm.root >= 0 && __builtin_amdgcn_ballot_w64(pass) != 0ull) {
            const BvhNode rn = ld_uniform(S.bvh_nodes + bm.root);
            const f32x2 ax = __builtin_elementwise_fma(ld2(rn.lox), ivx, -oix), bx = __builtin_elementwise_fma(ld2(rn.hix), ivx, -oix);
            const f32x2 ay = __builtin_elementwise_fma(ld2(rn.loy), ivy, -oiy), by = __builtin_elementwise_fma(ld2(rn.hiy), ivy, -oiy);
            const f32x2 az = __builtin_elementwise_fma(ld2(rn.loz), ivz, -oiz), bz = __builtin_elementwise_fma(ld2(rn.hiz), ivz, -oiz);
            const f32x2 zero = splat2(0.0f);
            const f32x2 tin = __builtin_elementwise_max(
                __builtin_elementwise_max(__builtin_elementwise_min(ax, bx), __builtin_elementwise_min(ay, by)),
                __builtin_elementwise_max(__builtin_elementwise_min(az, bz), zero));
            const f32x2 tout = __builtin_elementwise_min(
                __builtin_elementwise_min(__builtin_elementwise_max(ax, bx), __builtin_elementwise_max(ay, by)),
                __builtin_elementwise_max(az, bz));
            const f32x2 lo = tin * splat2(0.9999996f), lim = tout * splat2(1.0000009f);
            const bool h0 = lo[0] <= lim[0] && lo[0] <= bound_up, h1 = lo[1] <= lim[1] && lo[1] <= bound_up;
            pass = pass && (h0 || h1);
        }
        want = want || pass;
    }
    return want;
}
// 2. the walks of a parked ray: every BVH mesh in visiting order, exact gate first; the closest triangle of a mesh (the
//    first in list order among equals: bvh_closest_postponed) enters the key with its rank - one integer minimum is
//    intersect_scene's strict '<' over the whole visiting sequence (mod.rs:598,649), no tie needs a second look.
template <class NodePtr>
__device__ __forceinline__ unsigned long long walk_deferred_keys(const DevScene &S, NodePtr nodes, vec3 o, vec3 d, const WalkQueue &Q,
                                                                 unsigned long long key, unsigned long long *wave_keys,
                                                                 bool want = true) {  // want: this lane's ray may walk at all
    for (uint32_t q = 0; q < S.n_bvh_meshes; ++q) {
        PT_PHASE(kPhWalkGate);
        const BvhMeshRec bm = ld_uniform(S.bvh_meshes + q);
        const vec3 op = mk(bm.cx, bm.cy, bm.cz) - o;
        const float b = dot(op, d);
        const float det = (b * b - dot(op, op)) + bm.rr;
        const float sq = f_sqrt(det);
        const bool pass = want && !(det < 0.0f) && ((b - sq) >= 1e-4f || (b + sq) >= 1e-4f);
        if (__builtin_amdgcn_ballot_w64(pass) == 0ull) continue;
        float mt = __builtin_inff();
        int32_t mid = -1;
#if PT_WALK_FORM == 2
        bvh_closest_queue4(S, nodes, Q, wave_keys, pass, o, d, bm.root4, bm.root, __uint_as_float((uint32_t)(key >> 32)), mt, mid);
#else
        bvh_closest_queue(S, nodes, Q, wave_keys, pass, o, d, bm.root, __uint_as_float((uint32_t)(key >> 32)), mt, mid);
#endif
        bool won = false;
        if (pass && mid >= 0) {
            const unsigned long long k2 = ((unsigned long long)__float_as_uint(mt) << 32) | S.tri_rank[mid];
            won = k2 < key;
            key = k2 < key ? k2 : key;
        }
        PT_WSTAT(S, 13, __builtin_popcountll(__builtin_amdgcn_ballot_w64(won)));  // walks that found the ray's hit
        (void)won;
    }
    return key;
}

// the nodes the walker of the candidate forms reads (walk_deferred_keys' `nodes`: global memory, or the workgroup's LDS copy)
#if PT_WALK_FORM == 2
typedef BvhNode4 WalkNode;
__device__ __forceinline__ const WalkNode *walk_nodes(const DevScene &S) { return S.bvh_nodes4; }
__host__ __device__ inline uint32_t walk_node_count(const DevScene &S) { return S.n_bvh_nodes4; }
#else
typedef BvhNode WalkNode;
__device__ __forceinline__ const WalkNode *walk_nodes(const DevScene &S) { return S.bvh_nodes; }
__host__ __device__ inline uint32_t walk_node_count(const DevScene &S) { return S.n_bvh_nodes; }
#endif

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
    bool deferred;       // kShadeDeferRefract only: the hit is on glass and was not shaded
};

// shade_hit variants.  A wave almost always holds at least one ray that hit the glass sphere, so the whole wave pays
// for the refraction body (the longest of the three materials) on every chunk.  k_pass therefore shades with
// kShadeDeferRefract - glass hits are only reported - collects them per wave in LDS and shades them 64 at a time
// with kShadeRefractOnly: the same arithmetic per ray, executed by full waves.
constexpr int kShadeAll = 0, kShadeDeferRefract = 1, kShadeRefractOnly = 2;

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

// the same from the record of a visiting rank (candidate scan)
// `head` leading ranks may have a copy in LDS (`surf_lds`: k_pass_cand with walks, where the whole table does not fit)
__device__ __forceinline__ Surface fetch_surface_rank(const SurfRec *surf, vec3 o, vec3 d, float t, uint32_t rank,
                                                      const SurfRec *surf_lds = nullptr, uint32_t head = 0u) {
    SurfRec r;
    if (rank < head)
        r = surf_lds[rank];
    else
        r = surf[rank];
    Surface s;
    s.x = o + d * t;  // mod.rs:430 / mod.rs:604
    s.n = (r.kind & 0x100u) ? mk(r.vx, r.vy, r.vz) : normalize(s.x - mk(r.vx, r.vy, r.vz));
    s.color = mk(r.cr, r.cg, r.cb);
    s.emission = mk(r.er, r.eg, r.eb);
    s.max_refl = r.max_refl;
    s.inv_max_refl = r.inv_max_refl;
    s.reflect = r.kind & 3u;
    return s;
}

// One radiance() invocation after its intersect_scene call returned Some (mod.rs:665-789), given the surface hit.
template <int MODE = kShadeAll, class Params>
__device__ __forceinline__ void shade_surface(const Params &F, const PathRay &in, const Surface &sf, ShadeOut &out);

template <int MODE = kShadeAll, class Params>
__device__ __forceinline__ void shade_hit(const DevScene &S, const Params &F, const PathRay &in, HitRec h,
                                          ShadeOut &out) {
    const Surface sf = fetch_surface(S, in.o, in.d, h);
    shade_surface<MODE>(F, in, sf, out);
}

template <int MODE, class Params>
__device__ __forceinline__ void shade_surface(const Params &F, const PathRay &in, const Surface &sf, ShadeOut &out) {
    out.deferred = false;
    if (MODE == kShadeDeferRefract && sf.reflect == kRefract) {
        out.deferred = true;
        out.n_rays = 0;
        out.emits = false;
        return;
    }
    PT_PHASE(kPhRng);
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
        PT_PHASE(kPhRoulette);
        if (unit_f32(rnd.a) < sf.max_refl && new_depth < (uint32_t)kMaxDepth)
            color = color * sf.inv_max_refl;
        else
            alive = false;
    }
    PT_PHASE(kPhRng);
    const vec3 thr = in.thr * color;
    int n_rays = alive ? 1 : 0;
    vec3 d0 = d, thr0 = thr, d1 = d, thr1 = thr;

    PT_PHASE_PIN(rnd.a);
    PT_PHASE_PIN(rnd.b);
    PT_PHASE_PIN(rnd.c);
    PT_PHASE_PIN(thr.x);
    if (MODE != kShadeRefractOnly && sf.reflect == kDiffuse) {  // mod.rs:687-715
        PT_PHASE(kPhDiffuse);
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
        PT_PHASE(kPhSpecular);
        const vec3 refl = d - n * 2.0f * dot(n, d);  // mod.rs:722-723 / 733-734
        d0 = refl;
        if (MODE == kShadeRefractOnly || sf.reflect == kRefract) {  // mod.rs:729-788
            PT_PHASE(kPhGlass);
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
                if (new_depth > 2u) {  // mod.rs:760-774: one of the two, chosen with probability p
                    // (the reference computes RP = Re / P and TP = Tr / (1 - P) and uses one: the operands are chosen first here,
                    // and the one division that is needed is the same division)
                    const bool pick_refl = unit_f32(rnd.b) < p;
                    d0 = pick_refl ? refl : tdir;
                    thr0 = thr * ((pick_refl ? re : tr) / (pick_refl ? p : 1.0f - p));
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
    PT_PHASE_PIN(d0.x);
    PT_PHASE_PIN(d0.y);
    PT_PHASE_PIN(d0.z);
    PT_PHASE(kPhFinish);
    out.d0 = d0;
    out.thr0 = thr0;
    out.d1 = d1;
    out.thr1 = thr1;
    out.n_rays = n_rays;
}

// render_pixel's per-sample ray (mod.rs:805-843) for framebuffer index `pix`, sample `s`
// PROBE: the kernel instance pt_ctx_radiance launches - every primary ray is FrameParams' fixed ray at depth depth0.  A
// template parameter, not a branch on F.probe: the branch alone cost the frame kernels 1.3 % (registers around the two ray
// makers; A/B on one GPU), and a frame never probes.
__device__ __forceinline__ PathRay primary_ray_at(const FrameParams &F, uint32_t pix, uint32_t x, uint32_t y, uint32_t s);
template <bool PROBE = false>
__device__ __forceinline__ PathRay primary_ray(const FrameParams &F, uint32_t pix, uint32_t s) {
    if (PROBE) {  // pt_ctx_radiance's fixed ray, sample s of "pixel" pix
        PathRay r;
        r.o = mk(F.probe_ox, F.probe_oy, F.probe_oz);
        r.d = mk(F.probe_dx, F.probe_dy, F.probe_dz);
        r.thr = mk(1.0f, 1.0f, 1.0f);
        r.pix = pix;
        r.meta = pack_meta(s, F.depth0, 1u);
        return r;
    }
    return primary_ray_at(F, pix, pix % F.width, F.height - 1u - pix / F.width, s);
}
// the same with the pixel's column x and row y (from the bottom: mod.rs:805-806) already known: k_pass_cand keeps them per
// stream pixel in LDS instead of dividing by the frame width for every primary ray
__device__ __forceinline__ PathRay primary_ray_at(const FrameParams &F, uint32_t pix, uint32_t x, uint32_t y, uint32_t s) {
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
    r.meta = pack_meta(s, 0u, 1u);  // radiance(&ray, 0, ..), mod.rs:844
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
