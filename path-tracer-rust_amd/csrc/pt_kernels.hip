// pt_kernels.hip — gfx950 kernels of the wavefront radiance() pipeline and the persistent megakernel.
//
// Wavefront pipeline (one pass = P primary rays):
//   k_pass_cand                        THE DEFAULT: the whole pass in one launch, candidate scan instead of testing every
//                                      triangle (pt_device.h), and no levels - every wave of a workgroup (= ray stream: a set
//                                      of pixels, all samples of the pass) keeps its waiting rays on a stack of its own in
//                                      global memory, pops 64 or starts 64 primary rays (consecutive samples of one pixel,
//                                      the stream's chunks of 64 dealt to the four waves in turn); BVH meshes: walks parked per wave
// TWO TRANSLATION UNITS.  This file is compiled twice: as it is (every kernel but the instances of k_pass_cand WITHOUT walks and
// k_intersect_cand), and through pt_kernels_flat.hip with PT_TU_FLAT defined (those instances alone, five waves per SIMD): the
// two want different instruction scheduling from the back end, and -mllvm options are per compile (Makefile: MLLVM, MLLVM_FLAT).
// the level-by-level forms (PT_CAND_SCAN=0 / PT_FLAG_NO_BVH / PT_CAND_BVH=0), scenes without BVH meshes:
//   k_pass                             the whole pass in one launch: per workgroup (= ray stream) the primary rays
//                                      (render_pixel, mod.rs:812-843), then level by level closest hit
//                                      (intersect_scene, mod.rs:631-659), roulette / emission / BRDF sample / refract
//                                      split (radiance body, mod.rs:665-789) and stream compaction
//   k_pass_bvh                         the same for scenes with BVH meshes (walks parked per wave, done 64 at a time)
// PT_FLAG_SEPARATE_KERNELS / PT_PASS_KERNEL=0: the same steps as separate kernels
//   k_generate; for depth = 0..11: k_intersect, k_shade
//   k_resolve (once per frame)         /spp, clamp                      (mod.rs:849-856)
//
// Ray streams.  The ray queue is cut into K workgroup-private streams: workgroup b reads stream b of the current
// level and appends survivors to stream b of the next level.  A stream is a contiguous slice
// [b*cap, b*cap+count) of each queue array, so
//   * loads/stores are fully coalesced 16-/8-/4-byte per lane accesses,
//   * compaction needs no global atomic at all: wave ballot + mbcnt prefix gives the slot inside the
//     wave, one LDS add per wave orders the waves of the workgroup, and the stream length is a plain
//     store at the end (a single global tail counter would saturate at ~90 M atomics/s on this chip),
//   * stream b owns the pixels b, b+K, b+2K, ... of the call (all samples of the pass): every contribution to
//     those pixels is produced by workgroup b, so radiance is summed with LDS atomics and flushed to the frame
//     accumulator by plain read-modify-write stores - no global atomic in the pipeline (global 64-bit atomics cost
//     27 % of k_shade when they were used: profiles/README.md) - and every stream samples the whole picture, so
//     the streams of a launch carry the same load,
//   * no ray ever crosses streams, so a pass needs no grid-wide barrier: k_pass,
//   * K (16 K on the bench frame) >> the 1536-2048 workgroups the chip holds: the dispatcher keeps every CU busy.
// Queue arrays (per ray): od0 = (ox,oy,oz,dx) 16 B, od1 = (dy,dz) 8 B, tp = (throughput rgb, bookkeeping word) 16 B;
// hit = (t, id) 8 B (three-kernel form only).  k_pass moves 40 B out + 40 B in per ray of depth >= 1; the separate
// kernels 32 B/ray (intersect) and 48 B in + 40 B out per survivor (shade).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "pt_kernels.h"

namespace pt {

void set_error(const std::string &m);  // pt_api.hip

// dynamic LDS of the kernels that intersect: staged BVH nodes + per-lane traversal stacks (0 bytes for scenes
// without a BVH mesh).  No static __shared__ object precedes it in those kernels, so its base is 16-byte aligned.
extern __shared__ uint4 dyn_lds[];

constexpr uint32_t kDeferCap = 128;  // k_pass: deferred glass hits per wave (63 left over + 64 new at most)
// k_pass LDS: [u64 acc: 3*m][kPassTailWords x u32: counters, camera][u32 pixel index, column, row: 3*m][pad to 16][float4 deferred hits: waves x 3 x kDeferCap]
// the words between the accumulators and the pixel tables: [0..3] counters, [4..17] the camera for k_pass_cand's primary rays
constexpr uint32_t kPassTailWords = 20;
// u64 slots of the accumulator area: 3*m rounded up to even, so that the words behind it start on a 16-byte boundary whatever m
// is (k_pass_cand reads the camera from there as three float4: with an odd m - small frames have m = 1 - those were
// 8-byte-aligned ds_read_b128, which only the hardware's unaligned-DS mode forgives)
__host__ __device__ constexpr uint32_t pass_acc_slots(uint32_t m) { return (3u * m + 1u) & ~1u; }
__host__ __device__ constexpr size_t pass_lds_defer_offset(uint32_t m) {
    return ((size_t)pass_acc_slots(m) * sizeof(unsigned long long) + kPassTailWords * 4u + (size_t)3 * m * sizeof(uint32_t) + 15) & ~(size_t)15;
}
static_assert(pass_acc_slots(1) == 4u && pass_acc_slots(2) == 6u && (pass_acc_slots(7) * 8u) % 16u == 0u, "tails are 16-byte aligned");

// k_pass_cand LDS: [accumulators, tails, pixel indices as k_pass][per wave: float4 ray_a [128] | u64 key [128] |
// float2 ray_b [128] | u16 ring [kCandQueueCap]][staged candidate records]
// Glass deferral (DEFER): the glass hits of a wave wait in the wave's PARKING AREA in global memory - the ray (40 B) and its hit
// (distance, rank: 8 B), where scenes with BVH meshes park the rays that have to walk - until 64 of them make a dense wave.
// (Rounds 2-3 kept them in LDS: 18 KB per workgroup for 96 entries per wave, flushed at 32 - half-full batches, and with
// the levels gone the pushes and flushes cost what the batches saved.)
constexpr uint32_t kCandDeferFlush = 64;
__host__ __device__ constexpr size_t pass_lds_cand_offset(uint32_t m, bool /*defer*/) { return pass_lds_defer_offset(m); }
constexpr size_t kCandWaveBytes = 128u * 16u + 128u * 8u + 128u * 8u + kCandQueueCap * 2u;  // 4480
__host__ __device__ constexpr size_t pass_lds_cand_bytes() { return (size_t)(kBlock / 64u) * kCandWaveBytes; }
static_assert(kCandWaveBytes % 16u == 0u, "per-wave areas stay 16-byte aligned");

// a wave-uniform value, said so to the compiler (loop-carried counters of the pass kernels otherwise end up in VGPRs)
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_prefix(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// One stream's slice of a queue: ONE scalar base pointer and two 32-bit offsets (the starts of the throughput and the
// direction-yz arrays inside the slice).  Every access is base (an SGPR pair) + a 32-bit VGPR byte offset - `slot * 16 +
// off_tp` is one v_lshl_add with the offset as its scalar operand.  (Rounds 1-2 kept three arrays per queue and three 64-bit
// slice bases per level and direction: twelve SGPRs that the pass kernels, short of scalar registers, parked in VGPR lanes
// and fetched back with two v_readlane per access.)
struct StreamSlice {
    char *base;
    uint32_t off_tp, off_od1;
};
__device__ __forceinline__ StreamSlice slice_of(const RayQueue &q, uint32_t b, uint32_t cap) {
    StreamSlice s;
    s.base = q.buf + (size_t)b * cap * kRayBytes;
    s.off_tp = cap * 16u;
    s.off_od1 = cap * 32u;
    return s;
}
__device__ __forceinline__ float4 ld_od0(const StreamSlice &q, uint32_t i) { return *reinterpret_cast<const float4 *>(q.base + i * 16u); }
__device__ __forceinline__ float4 ld_tp(const StreamSlice &q, uint32_t i) { return *reinterpret_cast<const float4 *>(q.base + (i * 16u + q.off_tp)); }
__device__ __forceinline__ float2 ld_od1(const StreamSlice &q, uint32_t i) { return *reinterpret_cast<const float2 *>(q.base + (i * 8u + q.off_od1)); }
__device__ __forceinline__ void store_ray(const StreamSlice &q, uint32_t slot, vec3 o, vec3 d, vec3 thr, uint32_t word) {
    *reinterpret_cast<float4 *>(q.base + slot * 16u) = make_float4(o.x, o.y, o.z, d.x);
    *reinterpret_cast<float2 *>(q.base + (slot * 8u + q.off_od1)) = make_float2(d.y, d.z);
    *reinterpret_cast<float4 *>(q.base + (slot * 16u + q.off_tp)) = make_float4(thr.x, thr.y, thr.z, __uint_as_float(word));
}
// a ray of a stream's slice (read once per level; non-temporal loads / stores were tried: loads +0.2 %, stores -6 % - the
// next level reads what this one wrote from L2)
__device__ __forceinline__ void load_ray_slice(const StreamSlice &q, uint32_t i, vec3 &o, vec3 &d, vec3 &thr, uint32_t &word) {
    const float4 a = ld_od0(q, i);
    const float4 tp = ld_tp(q, i);
    const float2 c = ld_od1(q, i);
    o = mk(a.x, a.y, a.z);
    d = mk(a.w, c.x, c.y);
    thr = mk(tp.x, tp.y, tp.z);
    word = __float_as_uint(tp.w);
}

#ifndef PT_TU_FLAT  // (the translation unit of k_pass_cand without walks - pt_kernels_flat.hip - holds nothing else)
// ------------------------------------------------------------------------------------------------
template <bool PROBE>
__global__ __launch_bounds__(kBlock) void k_generate(FrameParams F, RayQueue q, uint32_t *__restrict__ cnt0,
                                                     uint32_t cap, uint32_t s0, uint32_t s_here, uint32_t m) {
    // stream b = its pixels (stream_pixel) x samples [s0, s0+s_here); lane order: pixel fastest
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const StreamSlice qs = slice_of(q, b, cap);
    const uint32_t mb = stream_pixel_count(F.npix, F.n_streams, b);  // <= m
    const uint32_t n = mb * s_here;
    for (uint32_t g = tid; g < n; g += kBlock) {
        const uint32_t pl = stream_pixel(F.n_streams, b, g % mb);
        const uint32_t s = s0 + g / mb;
        const PathRay r = primary_ray<PROBE>(F, global_pixel(F, pl), s);
        store_ray(qs, g, r.o, r.d, r.thr, pack_word(g % mb, g / mb, PROBE ? F.depth0 : 0u, 1u));
    }
    if (tid == 0) cnt0[b] = n;
}

// ------------------------------------------------------------------------------------------------
// The BVH variant.  Only the rays that pass the bounding-sphere gate of a BVH mesh walk it - a fifth of the lanes on
// mesh.json - and a walk is hundreds of instructions, so walking inside the object loop ran the kernel's VALU at 21 %
// lane occupancy (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = 13.8 of 64).  Here the scan skips the walks
// (intersect_scene_dev<true, true>: gates evaluated exactly, want_walk reported), rays that need one are parked per
// wave in LDS (ray, best hit so far, ray index: 36 B) and walked 64 at a time (walk_deferred): full waves.
constexpr uint32_t kParkCap = 128;  // 63 left over + 64 new at most
__host__ __device__ inline size_t bvh_park_offset(const DevScene &S, uint32_t block) {
    return (bvh_lds_bytes(S, block) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t bvh_park_bytes(uint32_t block) { return (size_t)(block / 64u) * kParkCap * 36u; }

template <bool BVH>
__global__ __launch_bounds__(BVH ? kBlockBvh : kBlock, BVH ? 5 : 1) void k_intersect(DevScene S, RayQueue q, float2 *__restrict__ hit,
                                                      const uint32_t *__restrict__ cnt, uint32_t cap,
                                                      unsigned long long *__restrict__ blk_rays) {
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = cnt[b];
    const size_t base = (size_t)b * cap;  // (of the hit records)
    const StreamSlice qs = slice_of(q, b, cap);
    if (!BVH) {
        for (uint32_t i = tid; i < n; i += blockDim.x) {
            const float4 a = ld_od0(qs, i);
            const float2 c = ld_od1(qs, i);
            const HitRec h = intersect_scene_dev<false>(S, mk(a.x, a.y, a.z), mk(a.w, c.x, c.y), dyn_lds);
            hit[base + i] = make_float2(h.t, __int_as_float(h.id));
        }
    } else {
        if (n != 0u) stage_bvh(S, dyn_lds);
        // this wave's parking area: float4 (o, d.x) | float4 (d.y, d.z, t, id) | u32 ray index
        char *park = reinterpret_cast<char *>(dyn_lds) + bvh_park_offset(S, blockDim.x) + (size_t)(tid >> 6) * kParkCap * 36u;
        float4 *const pa = reinterpret_cast<float4 *>(park), *const pb = pa + kParkCap;
        uint32_t *const pi = reinterpret_cast<uint32_t *>(pb + kParkCap);
        const uint32_t lane = tid & 63u;
        uint32_t n_park = 0;  // wave-uniform
        auto walk = [&](uint32_t e, bool valid) {
            if (valid) {
                const float4 a = pa[e], bq = pb[e];
                const uint32_t idx = pi[e];
                const vec3 o = mk(a.x, a.y, a.z), d = mk(a.w, bq.x, bq.y);
                float best_t = bq.z;
                int32_t best_id = __float_as_int(bq.w);
                if (walk_deferred(S, o, d, dyn_lds, best_t, best_id)) {  // a tie in distance: the in-order scan decides
                    const HitRec h = scan_scene<true, true>(S, o, d, dyn_lds);
                    best_t = h.t;
                    best_id = h.id;
                }
                hit[base + idx] = make_float2(best_t, __int_as_float(best_id));
            }
        };
        for (uint32_t j0 = 0; j0 < n; j0 += blockDim.x) {  // wave-uniform trip count
            const uint32_t i = j0 + tid;
            bool want = false;
            vec3 o = mk(0, 0, 0), d = o;
            HitRec h;
            h.t = 0.0f;
            h.id = -1;
            if (i < n) {
                const float4 a = ld_od0(qs, i);
                const float2 c = ld_od1(qs, i);
                o = mk(a.x, a.y, a.z);
                d = mk(a.w, c.x, c.y);
                h = intersect_scene_dev<true, true>(S, o, d, dyn_lds, &want);
                if (!want) hit[base + i] = make_float2(h.t, __int_as_float(h.id));
            }
            const uint64_t mw = __builtin_amdgcn_ballot_w64(want);
            if (mw != 0ull) {
                if (want) {
                    const uint32_t e = n_park + lane_prefix(mw);
                    pa[e] = make_float4(o.x, o.y, o.z, d.x);
                    pb[e] = make_float4(d.y, d.z, h.t, __int_as_float(h.id));
                    pi[e] = i;
                }
                n_park += (uint32_t)__builtin_popcountll(mw);
            }
            if (n_park >= 64u) {
                n_park -= 64u;
                walk(n_park + lane, true);
            }
        }
        if (n_park != 0u) walk(lane, lane < n_park);
    }
    if (tid == 0) blk_rays[b] += n;
}

#endif  // PT_TU_FLAT (k_intersect_cand below is the flat unit's too)
// ------------------------------------------------------------------------------------------------
// The stand-alone intersect step with the CANDIDATE SCAN of k_pass_cand (pt_device.h: "Candidate scan"; scenes without BVH
// meshes): intersect_scene (mod.rs:631-659) for every ray of a stream, as its own kernel - 24 B of ray in, 8 B of hit
// record out, nothing else: the kernel the north star's "HBM roofline of the intersect kernel" speaks of.  Per chunk of
// 256 rays: origin and direction into the chunk's LDS slots, spheres exactly (first key), filters of the flat pair records,
// candidates to the wave's ring, full batches of 64 exact tests as the ring fills; a ray's hit is written one chunk after
// it was started, when every candidate of its chunk has been through a batch (the ring is first-in first-out).  The hit
// record is HitRec's: (t, id) with id = DevScene.rank_id[rank of the key], -1 and t = +inf for a miss - bit for bit what
// k_intersect<false> writes (test_pass_kernel_equals_separate_kernels).
__host__ __device__ constexpr size_t intersect_cand_lds_bytes() { return (size_t)(kBlock / 64u) * kCandWaveBytes; }

template <bool STAGED>
__global__ __launch_bounds__(kBlock, PT_ISECT_WAVES) void k_intersect_cand(DevScene S, RayQueue q, float2 *__restrict__ hit,
                                                              const uint32_t *__restrict__ cnt, uint32_t cap,
                                                              unsigned long long *__restrict__ blk_rays) {
    const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t n = cnt[b];
    if (n == 0u) return;  // (the whole workgroup)
    CandLds cand;
    {
        char *wbase = reinterpret_cast<char *>(dyn_lds) + (size_t)(tid >> 6) * kCandWaveBytes;
        cand.ray_a = reinterpret_cast<float4 *>(wbase);
        cand.keys = reinterpret_cast<unsigned long long *>(wbase + 128u * 16u);
        cand.ray_b = reinterpret_cast<float2 *>(wbase + 128u * 24u);
        cand.queue = reinterpret_cast<uint16_t *>(wbase + 128u * 32u);
        char *sbase = reinterpret_cast<char *>(dyn_lds) + intersect_cand_lds_bytes();
        cand.staged = reinterpret_cast<const CandPairRec *>(sbase);
        cand.surf = S.surf;
        if (STAGED) {
            const uint4 *src = reinterpret_cast<const uint4 *>(S.cand_pairs);
            uint4 *dst = reinterpret_cast<uint4 *>(sbase);
            const uint32_t n_rows = S.n_cand_pairs * (uint32_t)(sizeof(CandPairRec) / 16u);
            for (uint32_t k = tid; k < n_rows; k += kBlock) dst[k] = src[k];
            __syncthreads();
        }
    }
    const StreamSlice qin = slice_of(q, b, cap);
    float2 *const hit_b = hit + (size_t)b * cap;
    CandRing ring;
    ring.head = 0u;
    ring.count = 0u;
    const uint32_t n_chunks = (n + kBlock - 1u) / kBlock;
    bool prev_valid = false;
#if PT_ISECT_PREFETCH
    // the next chunk's ray is loaded a trip ahead: the load's latency passes under this trip's arithmetic
    float4 na = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float2 nc = make_float2(0.0f, 0.0f);
    if (tid < n) {
        na = ld_od0(qin, tid);
        nc = ld_od1(qin, tid);
    }
#endif
    for (uint32_t it = 0; it <= n_chunks; ++it) {  // uniform trip count; the last trip only finishes chunk n_chunks - 1
        const uint32_t par = it & 1u;
        const uint32_t i = it * kBlock + tid;
        const bool cur_valid = it < n_chunks && i < n;
        const uint32_t pending = ring.count;  // entries of chunk it - 1 still queued (< 64)
        bool ran_batch = false;
        if (it < n_chunks) {
            vec3 o = mk(0.0f, 0.0f, 0.0f), d = o;
#if PT_ISECT_PREFETCH
            if (cur_valid) {
                o = mk(na.x, na.y, na.z);
                d = mk(na.w, nc.x, nc.y);
            }
            if (i + kBlock < n) {
                na = ld_od0(qin, i + kBlock);
                nc = ld_od1(qin, i + kBlock);
            }
#else
            if (cur_valid) {
                const float4 a = ld_od0(qin, i);
                const float2 c = ld_od1(qin, i);
                o = mk(a.x, a.y, a.z);
                d = mk(a.w, c.x, c.y);
            }
#endif
            const uint32_t slot = (par << 6) | lane;
            cand.ray_a[slot] = make_float4(o.x, o.y, o.z, d.x);
            cand.ray_b[slot] = make_float2(d.y, d.z);
            float bound;
            const unsigned long long key0 = cand_spheres(S, o, d, &bound);
            cand.keys[slot] = cur_valid ? key0 : kKeyMiss;
            const uint32_t before = ring.head;
            cand_filter_and_drain<STAGED>(S, cand, ring, lane, par, cur_valid, o, d, bound);
            ran_batch = ring.head != before;
        }
        if (it > 0u) {
            if (pending != 0u && !ran_batch) cand_batch<STAGED>(S, cand, ring, lane, ring.count);
            if (prev_valid) {
                const unsigned long long key = load_key(&cand.keys[((par ^ 1u) << 6) | lane]);
                const uint32_t rank = (uint32_t)key;
                const int32_t id = rank != 0xffffffffu ? (int32_t)S.rank_id[rank] : -1;
                hit_b[(it - 1u) * kBlock + tid] = make_float2(__uint_as_float((uint32_t)(key >> 32)), __int_as_float(id));
            }
        }
        prev_valid = cur_valid;
    }
    if (tid == 0) blk_rays[b] += n;
}

// ------------------------------------------------------------------------------------------------
// radiance of the stream's own pixels, summed in LDS (u64 32.32 fixed point, [3][m])
__device__ __forceinline__ void add_radiance_lds(unsigned long long *lds_acc, uint32_t m, uint32_t slot, vec3 v) {
    const uint64_t r = to_fixed(v.x), g = to_fixed(v.y), bl = to_fixed(v.z);
    if (r) atomicAdd(&lds_acc[slot], (unsigned long long)r);
    if (g) atomicAdd(&lds_acc[m + slot], (unsigned long long)g);
    if (bl) atomicAdd(&lds_acc[2u * m + slot], (unsigned long long)bl);
}

#ifndef PT_TU_FLAT
__global__ __launch_bounds__(kBlock) void k_shade(DevScene S, ShadeParams F, RayQueue qin, RayQueue qout,
                                                  const float2 *__restrict__ hit,
                                                  const uint32_t *__restrict__ cnt_in,
                                                  uint32_t *__restrict__ cnt_out, uint32_t cap,
                                                  unsigned long long *__restrict__ acc,
                                                  uint32_t *__restrict__ flags, uint32_t m) {
    // dynamic LDS: [u64 acc: 3*m][u32 tail]
    unsigned long long *lds_acc = reinterpret_cast<unsigned long long *>(dyn_lds);
    uint32_t *s_tail_p = reinterpret_cast<uint32_t *>(lds_acc + pass_acc_slots(m));
#define s_tail (*s_tail_p)
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = cnt_in[b];
    if (n == 0u) {  // an empty stream leaves an empty stream
        if (tid == 0) cnt_out[b] = 0u;
        return;
    }
    for (uint32_t k = tid; k < 3u * m; k += kBlock) lds_acc[k] = 0ull;
    if (tid == 0) s_tail = 0;
    __syncthreads();
    const size_t base = (size_t)b * cap;
    bool overflow = false;
    // Software pipeline over the stream: the five loads of chunk j+1 are issued (unconditionally, whatever the
    // hit record says - a miss costs 44 unused bytes on < 5 % of the rays) before chunk j is shaded, so a wave
    // exposes one memory latency per chunk instead of two dependent ones.
    float2 n_hr = make_float2(0.0f, __int_as_float(-1));
    float4 n_a = make_float4(0, 0, 0, 0), n_tp = n_a;
    float2 n_c = make_float2(0, 0);
    const StreamSlice sin = slice_of(qin, b, cap), sout = slice_of(qout, b, cap);
    if (tid < n) {
        n_hr = hit[base + tid];
        n_a = ld_od0(sin, tid);
        n_c = ld_od1(sin, tid);
        n_tp = ld_tp(sin, tid);
    }
    for (uint32_t j0 = 0; j0 < n; j0 += kBlock) {  // uniform trip count: every lane reaches the ballots
        const uint32_t i = j0 + tid;
        const float2 hr = n_hr;
        const float4 a = n_a, tp = n_tp;
        const float2 c = n_c;
        const uint32_t i_next = i + kBlock;
        if (i_next < n) {
            n_hr = hit[base + i_next];
            n_a = ld_od0(sin, i_next);
            n_c = ld_od1(sin, i_next);
            n_tp = ld_tp(sin, i_next);
        }
        ShadeOut so;
        so.n_rays = 0;
        so.emits = false;
        uint32_t word = 0;
        if (i < n) {
            HitRec h;
            h.t = hr.x;
            h.id = __float_as_int(hr.y);
            if (h.id >= 0) {
                PathRay in;
                in.o = mk(a.x, a.y, a.z);
                in.d = mk(a.w, c.x, c.y);
                in.thr = mk(tp.x, tp.y, tp.z);
                word = __float_as_uint(tp.w);
                in.pix = global_pixel(F, stream_pixel(F.n_streams, b, word_pix(word)));
                in.meta = pack_meta(F.s0 + word_sample(word), word_depth(word), word_branch(word));
                shade_hit(S, F, in, h, so);
                if (so.emits && !(F.debug & 1u)) add_radiance_lds(lds_acc, m, word_pix(word), so.contrib);
                if (F.debug & 1u) asm volatile("" ::"v"(so.contrib.x), "v"(so.contrib.y), "v"(so.contrib.z));
            }
        }
        // stream compaction: survivors first (path order kept inside the wave), split children after them
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(so.n_rays >= 1);
        const uint64_t m2 = __builtin_amdgcn_ballot_w64(so.n_rays == 2);
        const uint32_t c1 = (uint32_t)__builtin_popcountll(m1), c2 = (uint32_t)__builtin_popcountll(m2);
        uint32_t wbase = 0;
        if ((tid & 63u) == 0u && (c1 + c2) != 0u) wbase = atomicAdd(&s_tail, c1 + c2);
        wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
        if (so.n_rays >= 1) {
            const uint32_t slot = wbase + lane_prefix(m1);
            if (slot < cap)
                store_ray(sout, slot, so.x, so.d0, so.thr0,
                          pack_word(word_pix(word), word_sample(word), meta_depth(so.meta0), meta_branch(so.meta0)));
            else
                overflow = true;
        }
        if (so.n_rays == 2) {
            const uint32_t slot = wbase + c1 + lane_prefix(m2);
            if (slot < cap)
                store_ray(sout, slot, so.x, so.d1, so.thr1,
                          pack_word(word_pix(word), word_sample(word), meta_depth(so.meta1), meta_branch(so.meta1)));
            else
                overflow = true;
        }
    }
    if (overflow) atomicOr(flags, 1u);
    __syncthreads();
    if (tid == 0) cnt_out[b] = s_tail < cap ? s_tail : cap;
    // flush: this workgroup is the only writer of its pixels, launches on the stream are ordered
    const uint32_t mb = stream_pixel_count(F.npix, F.n_streams, b);
    const size_t plane = (size_t)F.n_streams * m;  // accumulator slots per colour channel (stream-major)
    for (uint32_t k = tid; k < 3u * mb; k += kBlock) {
        const uint32_t c = k / mb, p = k - c * mb;
        const unsigned long long v = lds_acc[c * m + p];
        if (v) acc[(size_t)c * plane + (size_t)b * m + p] += v;
    }
#undef s_tail
}

// ------------------------------------------------------------------------------------------------
// k_pass: one launch renders one whole pass.  Streams never exchange rays - workgroup b reads level d of stream b
// and appends to level d+1 of stream b - so nothing in a pass needs a grid-wide barrier: workgroup b generates the
// primary rays of its pixels in registers, finds their hits, shades them, appends the survivors to its slice of the
// queue, and then walks that slice level by level (workgroup barriers only) until the stream is empty.  Against the
// three-kernel form this removes 25 launches (and 24 drains of the whole chip) per pass, the hit records and the
// second read of every ray (intersect and shade of a ray happen back to back, 40 B in + 40 B out per bounce), the
// write and re-read of the primary rays, and it lets streams that are intersecting (VALU-bound) share a CU with
// streams that are waiting on their queue loads.  The accumulators stay in LDS for the whole pass and are flushed
// once.  Scenes with a BVH keep the three-kernel form (their intersect step wants 512-thread workgroups).
template <bool DEFER, bool PROBE>
__global__ __launch_bounds__(kBlock, 6) void k_pass(DevScene S, FrameParams F, RayQueue q0, RayQueue q1, uint32_t cap,
                                                 uint32_t s0, uint32_t s_here, uint32_t m,
                                                 unsigned long long *__restrict__ acc,
                                                 unsigned long long *__restrict__ blk_rays,
                                                 uint32_t *__restrict__ flags) {
    unsigned long long *lds_acc = reinterpret_cast<unsigned long long *>(dyn_lds);
    uint32_t *s_tail_p = reinterpret_cast<uint32_t *>(lds_acc + pass_acc_slots(m));
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t mb = stream_pixel_count(F.npix, F.n_streams, b);  // <= m
    if (mb == 0u) return;
    for (uint32_t k = tid; k < 3u * m; k += kBlock) lds_acc[k] = 0ull;
    if (tid == 0) s_tail_p[0] = s_tail_p[1] = 0u;  // appends of level d are counted in s_tail_p[d & 1]
    // framebuffer index (= RNG key) of each of the stream's pixels: the interleaved partition of a multi-GPU call
    // makes it a division per lookup, and it is needed once per bounce
    uint32_t *lds_pix = s_tail_p + kPassTailWords;
    for (uint32_t j = tid; j < mb; j += kBlock) lds_pix[j] = global_pixel(F, stream_pixel(F.n_streams, b, j));
    ShadeParams P;
    P.idx_begin = F.idx_begin;
    P.npix = F.npix;
    P.n_streams = F.n_streams;
    P.seed_lo = F.seed_lo;
    P.seed_hi = F.seed_hi;
    P.debug = F.debug;
    P.s0 = s0;
    P.chunk_pixels = F.chunk_pixels;
    P.chunk_first = F.chunk_first;
    P.chunk_step = F.chunk_step;
    P.k_begin = F.k_begin;
    bool overflow = false;
    unsigned long long total = 0ull;
    uint32_t n = mb * s_here;  // rays of the current level
    // per-wave buffer of deferred glass hits (ray, throughput, word, hit): at most 63 left over + 64 new entries
    float4 *const dbuf = reinterpret_cast<float4 *>(reinterpret_cast<char *>(dyn_lds) + pass_lds_defer_offset(m)) +
                         (size_t)(tid >> 6) * (3u * kDeferCap);
    uint32_t n_defer = 0;  // wave-uniform
    StreamSlice qout{};  // this stream's slice of the level's output container
    uint32_t *tail_p = nullptr;
    // stream compaction: survivors first (path order kept inside the wave), split children after them
    auto append = [&](const ShadeOut &so, uint32_t word) {
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(so.n_rays >= 1);
        const uint64_t m2 = __builtin_amdgcn_ballot_w64(so.n_rays == 2);
        const uint32_t c1 = (uint32_t)__builtin_popcountll(m1), c2 = (uint32_t)__builtin_popcountll(m2);
        if ((c1 + c2) == 0u) return;  // wave-uniform
        uint32_t wbase = 0;
        if ((tid & 63u) == 0u) wbase = atomicAdd(tail_p, c1 + c2);
        wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
        if (so.n_rays >= 1) {
            const uint32_t slot = wbase + lane_prefix(m1);
            if (slot < cap)
                store_ray(qout, slot, so.x, so.d0, so.thr0,
                          pack_word(word_pix(word), word_sample(word), meta_depth(so.meta0), meta_branch(so.meta0)));
            else
                overflow = true;
        }
        if (so.n_rays == 2) {
            const uint32_t slot = wbase + c1 + lane_prefix(m2);
            if (slot < cap)
                store_ray(qout, slot, so.x, so.d1, so.thr1,
                          pack_word(word_pix(word), word_sample(word), meta_depth(so.meta1), meta_branch(so.meta1)));
            else
                overflow = true;
        }
    };
    // one dense wave of glass hits out of this wave's buffer (entry `e` for the lanes with `valid`)
    auto shade_deferred = [&](uint32_t e, bool valid) {
        ShadeOut so;
        so.n_rays = 0;
        so.emits = false;
        uint32_t word = 0;
        if (valid) {
            const float4 a = dbuf[e], bq = dbuf[kDeferCap + e], cq = dbuf[2u * kDeferCap + e];
            PathRay in;
            in.o = mk(a.x, a.y, a.z);
            in.d = mk(a.w, bq.x, bq.y);
            in.thr = mk(bq.z, bq.w, cq.x);
            word = __float_as_uint(cq.y);
            HitRec h;
            h.t = cq.z;
            h.id = __float_as_int(cq.w);
            in.pix = lds_pix[word_pix(word)];
            in.meta = pack_meta(s0 + word_sample(word), word_depth(word), word_branch(word));
            shade_hit<kShadeRefractOnly>(S, P, in, h, so);
            if (so.emits) add_radiance_lds(lds_acc, m, word_pix(word), so.contrib);
        }
        append(so, word);
    };
    for (uint32_t depth = 0; depth < (uint32_t)kMaxDepth && n != 0u; ++depth) {
        const StreamSlice qin = slice_of((depth & 1u) ? q1 : q0, b, cap);  // level 0 is never stored
        qout = slice_of((depth & 1u) ? q0 : q1, b, cap);
        __syncthreads();  // level `depth` of the stream is complete and visible to the whole workgroup
        // the other counter was last read before this barrier (end of the level before) and is next added to after
        // the next one
        if (tid == 0) s_tail_p[(depth + 1u) & 1u] = 0u;
        tail_p = s_tail_p + (depth & 1u);
        total += n;
        for (uint32_t j0 = 0; j0 < n; j0 += kBlock) {  // uniform trip count: every lane reaches the ballots
            const uint32_t i = j0 + tid;
            PathRay in;
            uint32_t word = 0;
            if (i < n) {
                if (depth == 0u) {  // render_pixel's ray for (pixel i % mb of the stream, sample s0 + i / mb)
                    const uint32_t pj = i % mb, sj = i / mb;
                    in = primary_ray<PROBE>(F, lds_pix[pj], s0 + sj);
                    word = pack_word(pj, sj, PROBE ? F.depth0 : 0u, 1u);
                } else {
                    const float4 a = ld_od0(qin, i);
                    const float4 tp = ld_tp(qin, i);
                    const float2 c = ld_od1(qin, i);
                    in.o = mk(a.x, a.y, a.z);
                    in.d = mk(a.w, c.x, c.y);
                    in.thr = mk(tp.x, tp.y, tp.z);
                    word = __float_as_uint(tp.w);
                }
            }
            ShadeOut so;
            so.n_rays = 0;
            so.emits = false;
            so.deferred = false;
            HitRec h;
            h.t = 0.0f;
            h.id = -1;
            if (i < n) {
                h = intersect_scene_dev<false>(S, in.o, in.d, nullptr);
                if (h.id >= 0) {
                    in.pix = lds_pix[word_pix(word)];
                    in.meta = pack_meta(s0 + word_sample(word), word_depth(word), word_branch(word));
                    shade_hit<DEFER ? kShadeDeferRefract : kShadeAll>(S, P, in, h, so);
                    if (so.emits) add_radiance_lds(lds_acc, m, word_pix(word), so.contrib);
                }
            }
            append(so, word);
            // glass hits wait in this wave's LDS buffer until there are 64 of them
            const uint64_t md = DEFER ? __builtin_amdgcn_ballot_w64(so.deferred) : 0ull;
            if (DEFER && md != 0ull) {
                if (so.deferred) {
                    const uint32_t e = n_defer + lane_prefix(md);
                    dbuf[e] = make_float4(in.o.x, in.o.y, in.o.z, in.d.x);
                    dbuf[kDeferCap + e] = make_float4(in.d.y, in.d.z, in.thr.x, in.thr.y);
                    dbuf[2u * kDeferCap + e] = make_float4(in.thr.z, __uint_as_float(word), h.t, __int_as_float(h.id));
                }
                n_defer = rfl(n_defer + (uint32_t)__builtin_popcountll(md));
            }
            if (DEFER && n_defer >= 64u) {  // wave-uniform
                n_defer -= 64u;
                shade_deferred(n_defer + (tid & 63u), true);
            }
        }
        if (DEFER && n_defer != 0u) {  // the rest of this wave's glass hits of the level (carrying them over would let a few
                              // late rays stretch the stream by many nearly empty levels: measured 23.4 against 29.3)
            shade_deferred(tid & 63u, (tid & 63u) < n_defer);
            n_defer = 0u;
        }
        __syncthreads();  // every append of this level is counted
        const uint32_t tail = *tail_p;
        n = tail < cap ? tail : cap;
    }
    if (overflow) atomicOr(flags, 1u);
    __syncthreads();
    if (tid == 0) blk_rays[b] += total;
    // flush once per pass: this workgroup is the only writer of its pixels, launches on the stream are ordered
    const size_t plane = (size_t)F.n_streams * m;
    for (uint32_t k = tid; k < 3u * mb; k += kBlock) {
        const uint32_t c = k / mb, p = k - c * mb;
        const unsigned long long v = lds_acc[c * m + p];
        if (v) acc[(size_t)c * plane + (size_t)b * m + p] += v;
    }
}

#endif  // PT_TU_FLAT
// ------------------------------------------------------------------------------------------------
// k_pass with the candidate scan (pt_device.h: "Candidate scan") and WITHOUT LEVELS: a ray is FINISHED one chunk after it was
// started, and the rays that wait for their next bounce live on a stack of the wave (see "THE WAVE'S RAY STACK" in the body).
// A trip of a wave's loop starts a chunk of 64 rays - popped from the stack, or new primary rays when fewer than 64 wait - puts
// origin and direction into the chunk's slots in LDS, tests the spheres, runs the filters and queues the candidates; full
// batches of 64 candidates are run as the ring fills; then every candidate of the chunk started in the trip BEFORE has been
// through a batch (they were the oldest entries of the ring), so those rays take their hit from their key, are shaded and
// their continuations pushed.  Accumulators and the final flush are k_pass's; glass hits are shaded in place (DEFER, the per-wave
// deferral of k_pass, is an A/B switch here: launch_pass).
// BVH = true (scenes with BVH meshes): when a ray is finished - its key holds the best of the spheres and of the candidate
// records - bvh_wants decides whether it has to walk a BVH mesh; such a ray is not shaded but PARKED per wave (the ray in the
// wave's parking area in global memory, its key in LDS), and 64 parked rays at a time are walked (bvh_closest_queue: the
// wave's pending box tests as one queue), the closest triangle folded into the key by rank, shaded and pushed.
// LDS of that form, between the per-wave candidate areas and the staged records:
//   [per wave: walk queue (pass_cand_queue_bytes)][per wave: u64 key x 64][BVH nodes (NLDS: a small tree's nodes, staged)]
// (the parked rays' keys travel with the rays in the wave's parking area)
constexpr uint32_t kCandParkCap = kWaveParkCap;  // 63 left over + 64 new at most
// per wave: the walk queue (header + 8-byte entries: box tests from one end, leaves from the other), which is also where
// the depth-first stacks (DevScene.bvh_stack entries x 64 lanes x u16, or u32 when a tree has 32 768 nodes or leaves) and
// the leaf list of the rare second walk live
// (448 entries.  With sample-major primary rays the walkers of a session are alike and their items crowd the queue together: at
// 320 entries one wave-walk in fifty dropped pushes - those rays walk again depth-first - at 448 mesh.json gains 1.4 %; 512: the same)
#ifndef PT_WALK_QUEUE_BYTES
#define PT_WALK_QUEUE_BYTES 3584
#endif
constexpr uint32_t kWalkQueueBytes = PT_WALK_QUEUE_BYTES;
constexpr uint32_t kWalkQueueBytesStaged = 2048;  // 256 entries: beside the workgroup's copy of the nodes (bvh_in_lds bit 2)
__host__ __device__ inline size_t pass_cand_queue_bytes(const DevScene &S) {
    const size_t again = (size_t)S.bvh_stack * 64u * ((S.bvh_in_lds & 2u) ? 2u : 4u) + kLeafListCap * 4u;
    const size_t q = (S.bvh_in_lds & 4u) ? kWalkQueueBytesStaged : kWalkQueueBytes;
    return kWalkQueueHeader + (((again > q ? again : q) + 15) & ~(size_t)15);
}
__host__ __device__ inline size_t pass_cand_queues_bytes(const DevScene &S) { return (size_t)(kBlock / 64u) * pass_cand_queue_bytes(S); }
constexpr size_t kCandWalkKeyBytes = 64u * 8u;                         // per wave: the walkers' keys
// [the waves' walk queues][the waves' walk keys][the workgroup's copy of the BVH nodes, when they fit (bvh_in_lds bit 2)]
__host__ __device__ inline size_t pass_cand_nodes_offset(const DevScene &S) {
    return pass_cand_queues_bytes(S) + (size_t)(kBlock / 64u) * kCandWalkKeyBytes;
}
__host__ __device__ inline size_t pass_cand_bvh_bytes(const DevScene &S) {
    return pass_cand_nodes_offset(S) + ((S.bvh_in_lds & 4u) ? (size_t)walk_node_count(S) * sizeof(WalkNode) : 0u);
}
static_assert(kCandWalkKeyBytes % 16u == 0u && sizeof(WalkNode) % 16u == 0u, "per-wave areas stay 16-byte aligned");

// (The workgroup's own copy of the nodes in LDS, in front of the stacks, was tried: mesh.json's 141 nodes are 9 KB, which
// leaves room for three workgroups per CU instead of four - 16.1 against 17.8 G bounces/s.)
template <bool STAGED, bool DEFER, bool BVH, bool PROBE, bool NLDS = false>
__global__ __launch_bounds__(kBlock, BVH ? PT_CAND_BVH_WAVES : PT_CAND_WAVES) void k_pass_cand(DevScene S, FrameParams F, RayQueue q0, RayQueue q1, uint32_t cap,
                                                         uint32_t s0, uint32_t s_here, uint32_t m,
                                                         unsigned long long *__restrict__ acc,
                                                         unsigned long long *__restrict__ blk_rays,
                                                         uint32_t *__restrict__ flags) {
    unsigned long long *lds_acc = reinterpret_cast<unsigned long long *>(dyn_lds);
    uint32_t *s_tail_p = reinterpret_cast<uint32_t *>(lds_acc + pass_acc_slots(m));
    const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t mb = stream_pixel_count(F.npix, F.n_streams, b);  // <= m
    if (mb == 0u) return;
#ifdef PT_PHASE_STATS
    unsigned long long ph_t0, ph_r0;
    phase_begin(&ph_t0, &ph_r0);
#endif
    for (uint32_t k = tid; k < 3u * m; k += kBlock) lds_acc[k] = 0ull;
    if (tid == 0) s_tail_p[0] = s_tail_p[1] = 0u;  // (here: the workgroup's ray count, u64)
    // The camera in LDS: a primary trip (one in nine) reads it from there.  As fields of FrameParams it was a sixteen-register
    // tuple that the register allocator parked in VGPR lanes for the length of the loop and fetched back with 16 v_readlane per
    // use (SGPRs are what this kernel is shortest of): 46.6 -> 47.3 G bounces/s on cornell.
    if (tid == 0) {
        float *cw = reinterpret_cast<float *>(s_tail_p + 4);
        cw[0] = F.lens_x, cw[1] = F.lens_y, cw[2] = F.lens_z, cw[3] = F.cam_px, cw[4] = F.cam_py, cw[5] = F.cam_pz;
        cw[6] = F.su_x, cw[7] = F.su_y, cw[8] = F.su_z, cw[9] = F.sv_x, cw[10] = F.sv_y, cw[11] = F.sv_z;
        s_tail_p[16] = F.width, s_tail_p[17] = F.height;
    }
    // per stream pixel: framebuffer index (the RNG counter), and its column and row from the bottom (render_pixel's x, y,
    // mod.rs:805-806): two divisions here instead of two per primary ray
    uint32_t *lds_pix = s_tail_p + kPassTailWords;
    uint32_t *lds_px = lds_pix + m, *lds_py = lds_px + m;
    for (uint32_t j = tid; j < mb; j += kBlock) {
        const uint32_t pix = global_pixel(F, stream_pixel(F.n_streams, b, j));
        lds_pix[j] = pix;
        lds_px[j] = pix % F.width;
        lds_py[j] = F.height - 1u - pix / F.width;
    }
    // THE WAVES OF A WORKGROUP DO NOT WAIT FOR EACH OTHER: every wave traces a quarter of the stream's primary rays
    // on its own, its waiting rays on a stack of its own (below) - no workgroup barrier between the first one (LDS tables in
    // place) and the last (accumulators complete).  The workgroup shares the stream's pixel accumulators (LDS atomics) and
    // the staged scene records.
    // The stream's chunks of 64 primary rays are dealt to the four waves in turn (chunk c to wave c % 4), so that every wave gets
    // its share of every pixel: with a contiguous quarter each and SAMPLE-MAJOR order (below) a wave would own whole pixels, and
    // the waves of a workgroup, whose pixels' paths differ in length, would finish far apart.  A lane's primaries are 256 apart,
    // so its (pixel, sample) advance by a fixed step with a carry - no division per trip.
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t cap_w = cap >> 2;
    const uint32_t n0 = mb * s_here;                        // primary rays of the stream
    const uint32_t n_chunks = (n0 + 63u) >> 6;
    const uint32_t my_chunks = (n_chunks + 3u - wv) >> 2;   // chunks wv, wv + 4, ...
    const uint32_t quarter = (((n_chunks + 3u) >> 2) << 6) < n0 ? (((n_chunks + 3u) >> 2) << 6) : n0;  // upper bound of a wave's primaries
    const uint32_t base0 = wv << 6;
    uint32_t gen_left = 0u;                                 // primary rays this wave has still to start
    if (my_chunks != 0u) {
        const uint32_t last = wv + ((my_chunks - 1u) << 2);  // the wave's last chunk (the stream's last one may be partial)
        gen_left = ((my_chunks - 1u) << 6) + (n0 - (last << 6) < 64u ? n0 - (last << 6) : 64u);
    }
#if PT_SAMPLE_MAJOR
    // SAMPLE-MAJOR: primary ray g is (pixel g / s_here, sample g % s_here) - the 64 rays a trip starts are consecutive samples of
    // one pixel (or of a few, when a pass has fewer than 64 samples per pixel).  They leave the camera nearly alike, hit the same
    // object, and - the stack being last in, first out - their descendants are popped together: the blocks only some materials
    // need (the refraction body, the second ray of a split, mirror reflection, emitter adds) run in few trips with many lanes
    // instead of in nearly every trip with four.
    const uint32_t step_q = 256u / s_here, step_r = 256u % s_here;
    uint32_t gen_pj = (base0 + lane) / s_here, gen_sj = (base0 + lane) % s_here;
#else  // primary ray g is (pixel g % mb, sample g / mb): a trip starts one sample of 64 pixels (rounds 1-3)
    const uint32_t step_q = 256u / mb, step_r = 256u % mb;
    uint32_t gen_pj = (base0 + lane) % mb, gen_sj = (base0 + lane) / mb;
#endif
    CandLds cand;
    const SurfRec *surf_lds = nullptr;
    uint32_t surf_head = 0u;
    {
        char *cbase = reinterpret_cast<char *>(dyn_lds) + pass_lds_cand_offset(m, DEFER);
        char *wbase = cbase + (size_t)(tid >> 6) * kCandWaveBytes;
        cand.ray_a = reinterpret_cast<float4 *>(wbase);
        cand.keys = reinterpret_cast<unsigned long long *>(wbase + 128u * 16u);
        cand.ray_b = reinterpret_cast<float2 *>(wbase + 128u * 24u);
        cand.queue = reinterpret_cast<uint16_t *>(wbase + 128u * 32u);
        char *sbase = cbase + pass_lds_cand_bytes() + (BVH ? pass_cand_bvh_bytes(S) : 0u);
        cand.staged = reinterpret_cast<const CandPairRec *>(sbase);
        cand.surf = S.surf;
        if (STAGED) {  // the workgroup's own copies of the candidate records and (while they fit) of the shading records
            const uint4 *src = reinterpret_cast<const uint4 *>(S.cand_pairs);
            uint4 *dst = reinterpret_cast<uint4 *>(sbase);
            const uint32_t n_rows = S.n_cand_pairs * (uint32_t)(sizeof(CandPairRec) / 16u);
            for (uint32_t k = tid; k < n_rows; k += kBlock) dst[k] = src[k];
            if (!BVH || S.surf_staged) {  // wave-uniform
                const uint4 *src2 = reinterpret_cast<const uint4 *>(S.surf);
                uint4 *dst2 = dst + n_rows;
                const uint32_t n_rows2 = (S.n_objs + S.n_tris) * (uint32_t)(sizeof(SurfRec) / 16u);
                for (uint32_t k = tid; k < n_rows2; k += kBlock) dst2[k] = src2[k];
                cand.surf = reinterpret_cast<const SurfRec *>(dst2);
            } else if (S.surf_head != 0u) {  // the table is too large: only the ranks of the objects visited first
                const uint4 *src2 = reinterpret_cast<const uint4 *>(S.surf);
                uint4 *dst2 = dst + n_rows;
                const uint32_t n_rows2 = S.surf_head * (uint32_t)(sizeof(SurfRec) / 16u);
                for (uint32_t k = tid; k < n_rows2; k += kBlock) dst2[k] = src2[k];
                surf_lds = reinterpret_cast<const SurfRec *>(dst2);
                surf_head = S.surf_head;
            }
        }
    }
    // BVH: the waves' walk queues, then the waves' walk keys, then (NLDS) the workgroup's copy of the BVH nodes
    char *const walk_lds = (reinterpret_cast<char *>(dyn_lds) + pass_lds_cand_offset(m, DEFER) + pass_lds_cand_bytes());
    unsigned long long *p_key = nullptr;
    unsigned long long *walk_keys = nullptr;
    WalkQueue wq{};
    const WalkNode *const nodes_lds = reinterpret_cast<const WalkNode *>(walk_lds + pass_cand_nodes_offset(S));
    if (BVH && NLDS) {  // a box test of the walk queue then waits for an LDS read instead of a 64-byte gather from L2
        const uint4 *src = reinterpret_cast<const uint4 *>(walk_nodes(S));
        uint4 *dst = reinterpret_cast<uint4 *>(walk_lds + pass_cand_nodes_offset(S));
        for (uint32_t k = tid; k < walk_node_count(S) * (uint32_t)(sizeof(WalkNode) / 16u); k += kBlock) dst[k] = src[k];
    }
    if (BVH) {
        walk_keys = reinterpret_cast<unsigned long long *>(walk_lds + pass_cand_queues_bytes(S) + (size_t)(tid >> 6) * kCandWalkKeyBytes);
        char *qb = walk_lds + (size_t)(tid >> 6) * pass_cand_queue_bytes(S);
        wq.redo = reinterpret_cast<uint32_t *>(qb);
        wq.ent = reinterpret_cast<uint2 *>(qb + kWalkQueueHeader);
        wq.cap = (uint32_t)((pass_cand_queue_bytes(S) - kWalkQueueHeader) / 8u);
        if (S.walk_queue_cap >= 128u && S.walk_queue_cap < wq.cap) wq.cap = S.walk_queue_cap;
    }
    ShadeParams P;
    P.idx_begin = F.idx_begin;
    P.npix = F.npix;
    P.n_streams = F.n_streams;
    {  // the key as two scalars of their own: left inside FrameParams' sixteen-register tuple, every draw of the loop
       // fetched the whole tuple back from the VGPR lanes the register allocator had parked it in (16 v_readlane per site)
        uint32_t k_lo = F.seed_lo, k_hi = F.seed_hi;
        asm volatile("" : "+s"(k_lo), "+s"(k_hi));
        P.seed_lo = k_lo;
        P.seed_hi = k_hi;
    }
    P.debug = F.debug;
    P.s0 = s0;
    P.chunk_pixels = F.chunk_pixels;
    P.chunk_first = F.chunk_first;
    P.chunk_step = F.chunk_step;
    P.k_begin = F.k_begin;
    bool overflow = false;  // wave-uniform
    unsigned long long total = 0ull;
    // THE WAVE'S RAY STACK.  Rays that wait for their next bounce live on a stack of the wave in global memory (the wave's
    // quarter of the stream's slice of container 0): the wave pops the 64 most recent ones whenever at least 64 wait and
    // starts 64 new primary rays otherwise - there are no levels, rays of every depth share a chunk (each carries its own
    // depth in its bookkeeping word), so that until a wave's very last rays every chunk, every exact batch, every batch of
    // glass hits and every walk session of 64 parked rays runs full (with levels every level ended in a partial chunk, a
    // flushing trip and - with walks - a partial session: half of mesh.json's sessions), and the live rays of a wave are a
    // few hundred (depth first), which stay in L2.  A slot that is popped may be pushed over in the same trip: the vector
    // memory operations of one wave are performed in order.  (A ring - first in, first out, so that a pop never reads what the
    // trip before has just stored - was measured: the ring's positions wander through all its slots and the waves' live rays
    // no longer stay in L2; cornell 40.8 against 42.6 G bounces/s, mesh.json 23.0 against 25.8.)
    // Room: a waiting ray of depth 1 can have two descendants waiting at a time, deeper ones one, a primary ray four (two
    // refract splits, mod.rs:760).  `phi` bounds what the rays the wave holds anywhere (stack, chunk in flight, parked,
    // deferred) can ever put on the stack at once; new primaries (at most 4 x 64 more) are started only while that fits.
    const uint32_t room = cap_w < kWaveStackMax ? cap_w : kWaveStackMax;
    const bool room_for_all = 4u * quarter + 3u <= room;  // everything this wave will ever trace fits: nothing to watch
    StreamSlice qs;
    qs.base = q0.buf + (size_t)(b * 4u + wv) * cap_w * kRayBytes;
    qs.off_tp = room * 16u;
    qs.off_od1 = room * 32u;
    StreamSlice qpark{};  // BVH: the parked rays of the wave (container 1); DEFER: its glass hits
    if (BVH || DEFER) {
        qpark.base = q1.buf + (size_t)(b * 4u + wv) * kWaveParkBytes;
        qpark.off_tp = kCandParkCap * 16u;
        qpark.off_od1 = kCandParkCap * 32u;
        p_key = reinterpret_cast<unsigned long long *>(qpark.base + kCandParkCap * kRayBytes);  // (the keys follow the rays)
    }
    if (!room_for_all && room < 512u) {  // (the host sizes the slices; never a hang)
        if (tid == 0) atomicOr(flags, 2u);
        return;
    }
    uint32_t top = 0u;  // wave-uniform: rays on the stack
    auto append = [&](const ShadeOut &so, uint32_t word) {
        PT_PHASE(kPhAppend);
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(so.n_rays >= 1);
        const uint64_t m2 = __builtin_amdgcn_ballot_w64(so.n_rays == 2);
        const uint32_t c1 = (uint32_t)__builtin_popcountll(m1), c2 = (uint32_t)__builtin_popcountll(m2);
        if ((c1 + c2) == 0u) return;  // wave-uniform
        const uint32_t wbase = top;
        if (wbase + c1 + c2 > room) {  // (cannot happen, see above)
            overflow = true;
            return;
        }
        top = rfl(top + c1 + c2);
        if (so.n_rays >= 1)
            store_ray(qs, wbase + lane_prefix(m1), so.x, so.d0, so.thr0,
                      pack_word(word_pix(word), word_sample(word), meta_depth(so.meta0), meta_branch(so.meta0)));
        if (so.n_rays == 2) {
            PT_PHASE(kPhAppend2);
            store_ray(qs, wbase + c1 + lane_prefix(m2), so.x, so.d1, so.thr1,
                      pack_word(word_pix(word), word_sample(word), meta_depth(so.meta1), meta_branch(so.meta1)));
        }
    };
    // DEFER: glass hits of this wave wait in its parking area until kCandDeferFlush of them make a dense wave
    uint32_t n_defer = 0;  // wave-uniform
    auto shade_deferred = [&](uint32_t e, bool valid) {
        ShadeOut so;
        so.n_rays = 0;
        so.emits = false;
        uint32_t word = 0;
        PT_PHASE(kPhDefer);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the wave's own stores, read back by other lanes)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (valid) {
            PathRay in;
            load_ray_slice(qpark, e, in.o, in.d, in.thr, word);
            const unsigned long long hk = p_key[e];  // (distance bits << 32) | rank, as parked
            in.pix = lds_pix[word_pix(word)];
            in.meta = pack_meta(s0 + word_sample(word), word_depth(word), word_branch(word));
            const Surface sf = fetch_surface_rank(cand.surf, in.o, in.d, __uint_as_float((uint32_t)(hk >> 32)), (uint32_t)hk);
            shade_surface<kShadeRefractOnly>(P, in, sf, so);
            PT_PHASE(kPhEmit);
            if (so.emits) add_radiance_lds(lds_acc, m, word_pix(word), so.contrib);
        }
        append(so, word);
    };
    // BVH: 64 parked rays (or the wave's last ones): the ray again (from the parking area), its walks, shading, appending
    uint32_t n_park = 0;  // wave-uniform
    auto walk_batch = [&](uint32_t e, bool valid) {
        ShadeOut so;
        so.n_rays = 0;
        so.emits = false;
        uint32_t word = 0;
        PT_PHASE(kPhLoad);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the wave's own stores, read back by other lanes)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (valid) {
            PathRay in;
            load_ray_slice(qpark, e, in.o, in.d, in.thr, word);
            unsigned long long key;
            if constexpr (NLDS)
                key = walk_deferred_keys(S, nodes_lds, in.o, in.d, wq, p_key[e], walk_keys);
            else
                key = walk_deferred_keys(S, walk_nodes(S), in.o, in.d, wq, p_key[e], walk_keys);
            const uint32_t rank = (uint32_t)key;
            if (rank != 0xffffffffu) {
                in.pix = lds_pix[word_pix(word)];
                in.meta = pack_meta(s0 + word_sample(word), word_depth(word), word_branch(word));
                PT_PHASE(kPhSurface);
                const Surface sf = fetch_surface_rank(cand.surf, in.o, in.d, __uint_as_float((uint32_t)(key >> 32)), rank, surf_lds, surf_head);
                shade_surface<kShadeAll>(P, in, sf, so);
                PT_PHASE(kPhEmit);
                if (so.emits) add_radiance_lds(lds_acc, m, word_pix(word), so.contrib);
            }
        }
        append(so, word);
    };
    CandRing ring;
    ring.head = 0u;
    ring.count = 0u;
    PT_PHASE(kPhBarrier);
    __syncthreads();  // accumulators, pixel tables and staged records are in place
    PT_PHASE(kPhOther);
    uint32_t par = 0u;                      // the slots (LDS) of the chunk started in this trip
    bool pending = false;                   // wave-uniform: the chunk started in the trip before waits to be finished
    vec3 prev_thr = mk(0.0f, 0.0f, 0.0f);   // what the ray started in the trip before still needs from registers
    uint32_t prev_word = 0;
    bool prev_valid = false;
    for (;;) {  // one trip: start a chunk of up to 64 rays (filters, candidates), finish the chunk started in the trip before
        const uint32_t phi = 2u * top + 4u * (n_park + n_defer + (pending ? 64u : 0u));
        const bool may_start = gen_left != 0u && (room_for_all || phi + 256u <= room);
        uint32_t src = 0u, cnt = 0u;  // 1: pop from the stack, 2: primary rays
        if (top >= 64u) {
            src = 1u;
            cnt = 64u;
        } else if (may_start) {
            src = 2u;
            cnt = gen_left < 64u ? gen_left : 64u;
        } else if (top != 0u) {
            src = 1u;
            cnt = top;
        }
        src = rfl(src);
        cnt = rfl(cnt);
        // nothing to start and nothing to finish: what still waits in the wave's side buffers (below), then the end
        // (A wave's last rays - its primaries used up, fewer than 64 waiting - take some 30 trips of ever fewer lanes.  Leaving
        // them to ONE wave of the workgroup - waves 1-3 stop at that point, wave 0 moves their rays onto its own stack and
        // traces the workgroup's last rays alone, a few full trips and one fading tail instead of four - was built and
        // measured: the three idle wave slots cost more than the partial trips, cornell 43.7 against 45.5 G bounces/s,
        // mesh.json 22.5 against 26.6.)
        const bool idle = src == 0u && !pending;
        if (idle && n_defer == 0u && n_park == 0u) break;  // (gen_left is 0: with nothing held phi is 0 and primaries may start)
        const bool cur_valid = lane < cnt;
        vec3 cur_thr = mk(0.0f, 0.0f, 0.0f);
        uint32_t word = 0;
        const uint32_t pend_entries = ring.count;  // candidate entries of the chunk to finish that are still queued (< 64)
        bool ran_batch = false;
        if (src != 0u) {
            PathRay in;
            in.o = in.d = in.thr = mk(0.0f, 0.0f, 0.0f);
            if (src == 2u) {  // render_pixel's rays for (pixel, sample) = (g % mb, s0 + g / mb), g = this wave's next 64 indices
                PT_PHASE(kPhPrimary);
                if (cur_valid) {
                    const uint32_t pj = gen_pj, sj = gen_sj;
                    FrameParams Fl;  // (the fields primary_ray_at reads)
                    {
                        const float4 c0 = *reinterpret_cast<const float4 *>(s_tail_p + 4), c1 = *reinterpret_cast<const float4 *>(s_tail_p + 8),
                                     c2 = *reinterpret_cast<const float4 *>(s_tail_p + 12);
                        Fl.lens_x = c0.x, Fl.lens_y = c0.y, Fl.lens_z = c0.z, Fl.cam_px = c0.w, Fl.cam_py = c1.x, Fl.cam_pz = c1.y;
                        Fl.su_x = c1.z, Fl.su_y = c1.w, Fl.su_z = c2.x, Fl.sv_x = c2.y, Fl.sv_y = c2.z, Fl.sv_z = c2.w;
                        Fl.width = s_tail_p[16], Fl.height = s_tail_p[17];
                        Fl.seed_lo = P.seed_lo, Fl.seed_hi = P.seed_hi;
                    }
                    in = PROBE ? primary_ray<PROBE>(F, lds_pix[pj], s0 + sj) : primary_ray_at(Fl, lds_pix[pj], lds_px[pj], lds_py[pj], s0 + sj);
                    word = pack_word(pj, sj, PROBE ? F.depth0 : 0u, 1u);
                }
#if PT_SAMPLE_MAJOR
                gen_sj += step_r;
                gen_pj += step_q;
                if (gen_sj >= s_here) {
                    gen_sj -= s_here;
                    gen_pj += 1u;
                }
#else
                gen_pj += step_r;
                gen_sj += step_q;
                if (gen_pj >= mb) {
                    gen_pj -= mb;
                    gen_sj += 1u;
                }
#endif
                gen_left = rfl(gen_left - cnt);
            } else {
                PT_PHASE(kPhLoad);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the wave's own stores, read back by other lanes)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                top = rfl(top - cnt);
                if (cur_valid) load_ray_slice(qs, top + lane, in.o, in.d, in.thr, word);
            }
            total += cnt;
            cur_thr = in.thr;
            const uint32_t slot = (par << 6) | lane;
            cand.ray_a[slot] = make_float4(in.o.x, in.o.y, in.o.z, in.d.x);
            cand.ray_b[slot] = make_float2(in.d.y, in.d.z);
            float bound;
            PT_PHASE(kPhSpheres);
            const unsigned long long key0 = cand_spheres(S, in.o, in.d, &bound);
            cand.keys[slot] = cur_valid ? key0 : kKeyMiss;
            const uint32_t before = ring.head;
            cand_filter_and_drain<STAGED>(S, cand, ring, lane, par, cur_valid, in.o, in.d, bound);
            ran_batch = ring.head != before;
        }
        if (pending) {
            PT_PHASE(kPhFinish);
            // the chunk started in the trip before: its candidates were the oldest entries of the ring; if no full batch ran in
            // this trip (few candidates, or nothing was started) run what is queued now
            if (pend_entries != 0u && !ran_batch) cand_batch<STAGED>(S, cand, ring, lane, ring.count);
            PT_PHASE(kPhFinish);
            ShadeOut so;
            so.n_rays = 0;
            so.emits = false;
            so.deferred = false;
            float hit_t = 0.0f;
            uint32_t hit_rank = 0xffffffffu;
            PathRay pr;
            pr.o = pr.d = pr.thr = mk(0.0f, 0.0f, 0.0f);
            bool park = false;
            unsigned long long park_key = 0ull;
            if (prev_valid) {
                const uint32_t slot = ((par ^ 1u) << 6) | lane;
                const unsigned long long key = load_key(&cand.keys[slot]);
                const uint32_t rank = (uint32_t)key;
                if (BVH) {  // (a miss so far may still hit a BVH mesh)
                    const float4 ra = cand.ray_a[slot];
                    const float2 rb = cand.ray_b[slot];
                    pr.o = mk(ra.x, ra.y, ra.z);
                    pr.d = mk(ra.w, rb.x, rb.y);
                    park = bvh_wants(S, pr.o, pr.d, __uint_as_float((uint32_t)(key >> 32)));
                    park_key = key;
                }
                if (rank != 0xffffffffu && !park) {
                    hit_t = __uint_as_float((uint32_t)(key >> 32));
                    hit_rank = rank;
                    if (!BVH) {
                        const float4 ra = cand.ray_a[slot];
                        const float2 rb = cand.ray_b[slot];
                        pr.o = mk(ra.x, ra.y, ra.z);
                        pr.d = mk(ra.w, rb.x, rb.y);
                    }
                    pr.thr = prev_thr;
                    pr.pix = lds_pix[word_pix(prev_word)];
                    pr.meta = pack_meta(s0 + word_sample(prev_word), word_depth(prev_word), word_branch(prev_word));
                    PT_PHASE(kPhSurface);
                    const Surface sf = fetch_surface_rank(cand.surf, pr.o, pr.d, hit_t, rank, surf_lds, surf_head);
                    PT_PHASE_PIN(sf.n.x);
                    PT_PHASE_PIN(sf.n.y);
                    PT_PHASE_PIN(sf.n.z);
                    PT_PHASE_PIN(sf.x.x);
                    shade_surface<DEFER ? kShadeDeferRefract : kShadeAll>(P, pr, sf, so);
                    PT_PHASE(kPhEmit);
                    // (Collecting the hits on emitters per wave and adding them 64 at a time - 16 B per entry, dense fixed-point
                    // conversions and LDS atomics - was measured once the glass deferral had left its LDS free: +0.45 %, not kept.)
                    if (so.emits) {
                        PT_PHASE(kPhEmitAdd);
                        add_radiance_lds(lds_acc, m, word_pix(prev_word), so.contrib);
                    }
                }
            }
            append(so, prev_word);
            if (BVH) {
                const uint64_t mw = __builtin_amdgcn_ballot_w64(park);
                PT_WSTAT(S, 7, __builtin_popcountll(__builtin_amdgcn_ballot_w64(prev_valid)));  // rays
                PT_WSTAT(S, 8, __builtin_popcountll(mw));                                       // parked
                PT_WSTAT(S, 9, 1);
                if (mw != 0ull) {
                    if (park) {  // the whole ray goes to the parking area (its slot in LDS is the next chunk's in two trips)
                        const uint32_t e = n_park + lane_prefix(mw);
                        store_ray(qpark, e, pr.o, pr.d, prev_thr, prev_word);
                        p_key[e] = park_key;
                    }
                    n_park = rfl(n_park + (uint32_t)__builtin_popcountll(mw));
                }
            }
            PT_PHASE(kPhDefer);
            const uint64_t md = DEFER ? __builtin_amdgcn_ballot_w64(so.deferred) : 0ull;
            if (DEFER && md != 0ull) {
                if (so.deferred) {
                    const uint32_t e = n_defer + lane_prefix(md);
                    store_ray(qpark, e, pr.o, pr.d, pr.thr, prev_word);
                    p_key[e] = ((unsigned long long)__float_as_uint(hit_t) << 32) | hit_rank;
                }
                n_defer = rfl(n_defer + (uint32_t)__builtin_popcountll(md));
            }
        }
        // 64 parked rays make a session; an idle wave walks what is left.  (Dealing the four waves' leftovers out again as
        // batches of 64 behind a barrier was tried in the level-by-level form: fuller batches, but every wave then waits for
        // the slowest scan of the level before any leftover is walked - 19.2 against 20.2 G bounces/s on mesh.json.)
        while (BVH && (n_park >= 64u || (idle && n_park != 0u))) {  // wave-uniform
            const uint32_t c = n_park < 64u ? n_park : 64u;
            n_park = rfl(n_park - c);
            walk_batch(n_park + lane, lane < c);
        }
        if (DEFER && (n_defer >= kCandDeferFlush || (idle && n_defer != 0u))) {  // wave-uniform
            const uint32_t c = n_defer < 64u ? n_defer : 64u;
            n_defer = rfl(n_defer - c);
            shade_deferred(n_defer + lane, lane < c);
        }
        PT_PHASE(kPhOther);
        prev_thr = cur_thr;
        prev_word = word;
        prev_valid = cur_valid;
        pending = src != 0u;
        par = rfl(par ^ 1u);
    }
    if (overflow) atomicOr(flags, 1u);
    if (lane == 0u) atomicAdd(reinterpret_cast<unsigned long long *>(s_tail_p), total);
    PT_PHASE(kPhBarrier);
    __syncthreads();  // every wave of the workgroup is done: the accumulators are complete
    PT_PHASE(kPhOther);
    if (tid == 0) blk_rays[b] += *reinterpret_cast<unsigned long long *>(s_tail_p);
    const size_t plane = (size_t)F.n_streams * m;
    for (uint32_t k = tid; k < 3u * mb; k += kBlock) {
        const uint32_t c = k / mb, p = k - c * mb;
        const unsigned long long v = lds_acc[c * m + p];
        if (v) acc[(size_t)c * plane + (size_t)b * m + p] += v;
    }
#ifdef PT_PHASE_STATS
    phase_end(S.phase_stats, ph_t0, ph_r0);
#endif
}

#ifndef PT_TU_FLAT
// ------------------------------------------------------------------------------------------------
// k_pass for scenes with a BVH: the same one-launch-per-pass walk of a stream, with the BVH walks of k_intersect<true>:
// the scan skips them (gates exact, first step of the walk taken from SGPRs), a ray that needs one is parked per wave
// in LDS as (ray index, best hit so far) - the ray itself stays in the queue slice - and walked, shaded and appended
// 64 at a time.  Level 0 goes through the queue too, so that every parked ray can be read back by index.
constexpr uint32_t kPassParkCap = 128;  // 63 left over + 64 new at most
__host__ __device__ inline size_t pass_bvh_stack_offset(uint32_t m) { return pass_lds_defer_offset(m); }
__host__ __device__ inline size_t pass_bvh_park_offset(const DevScene &S, uint32_t m) {
    return (pass_bvh_stack_offset(m) + (size_t)kBvhStack * kBlock * ((S.bvh_in_lds & 2u) ? sizeof(uint16_t) : sizeof(uint32_t)) + 15) &
           ~(size_t)15;
}
// after the parking areas: per wave [u64 key: 64][u32 leaf list: kLeafListCap] of the postponed leaf tests
__host__ __device__ inline size_t pass_bvh_leaf_offset(const DevScene &S, uint32_t m) {
    return (pass_bvh_park_offset(S, m) + (size_t)(kBlock / 64u) * kPassParkCap * 12u + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t pass_bvh_lds_bytes(const DevScene &S, uint32_t m) {
    return pass_bvh_leaf_offset(S, m) + (size_t)(kBlock / 64u) * (64u * 8u + kLeafListCap * 4u);
}

template <bool PROBE>
__global__ __launch_bounds__(kBlock, 4) void k_pass_bvh(DevScene S, FrameParams F, RayQueue q0, RayQueue q1, uint32_t cap,
                                                        uint32_t s0, uint32_t s_here, uint32_t m,
                                                        unsigned long long *__restrict__ acc,
                                                        unsigned long long *__restrict__ blk_rays,
                                                        uint32_t *__restrict__ flags) {
    unsigned long long *lds_acc = reinterpret_cast<unsigned long long *>(dyn_lds);
    uint32_t *s_tail_p = reinterpret_cast<uint32_t *>(lds_acc + pass_acc_slots(m));
    const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t mb = stream_pixel_count(F.npix, F.n_streams, b);  // <= m
    if (mb == 0u) return;
    for (uint32_t k = tid; k < 3u * m; k += kBlock) lds_acc[k] = 0ull;
    if (tid == 0) s_tail_p[0] = s_tail_p[1] = 0u;
    uint32_t *lds_pix = s_tail_p + kPassTailWords;
    for (uint32_t j = tid; j < mb; j += kBlock) lds_pix[j] = global_pixel(F, stream_pixel(F.n_streams, b, j));
    // traversal stacks (one column per thread), then this wave's parking area: u32 ray index | f32 t | i32 id
    uint4 *const stacks = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(dyn_lds) + pass_bvh_stack_offset(m));
    char *park = reinterpret_cast<char *>(dyn_lds) + pass_bvh_park_offset(S, m) + (size_t)(tid >> 6) * kPassParkCap * 12u;
    uint32_t *const p_idx = reinterpret_cast<uint32_t *>(park);
    float *const p_t = reinterpret_cast<float *>(p_idx + kPassParkCap);
    int32_t *const p_id = reinterpret_cast<int32_t *>(p_t + kPassParkCap);
    LeafLds leaves;
    {
        char *lbase = reinterpret_cast<char *>(dyn_lds) + pass_bvh_leaf_offset(S, m) + (size_t)(tid >> 6) * (64u * 8u + kLeafListCap * 4u);
        leaves.keys = reinterpret_cast<unsigned long long *>(lbase);
        leaves.list = reinterpret_cast<uint32_t *>(lbase + 64u * 8u);
    }
    ShadeParams P;
    P.idx_begin = F.idx_begin;
    P.npix = F.npix;
    P.n_streams = F.n_streams;
    P.seed_lo = F.seed_lo;
    P.seed_hi = F.seed_hi;
    P.debug = F.debug;
    P.s0 = s0;
    P.chunk_pixels = F.chunk_pixels;
    P.chunk_first = F.chunk_first;
    P.chunk_step = F.chunk_step;
    P.k_begin = F.k_begin;
    bool overflow = false;
    unsigned long long total = 0ull;
    uint32_t n = mb * s_here;  // rays of the current level
    __syncthreads();           // lds_pix
    for (uint32_t g = tid; g < n; g += kBlock) {  // level 0: render_pixel's rays (pixel g % mb, sample s0 + g / mb)
        const uint32_t pj = g % mb, sj = g / mb;
        const PathRay r = primary_ray<PROBE>(F, lds_pix[pj], s0 + sj);
        store_ray(slice_of(q0, b, cap), g, r.o, r.d, r.thr, pack_word(pj, sj, PROBE ? F.depth0 : 0u, 1u));
    }
    uint32_t n_park = 0;  // wave-uniform
    StreamSlice qin{}, qout{};
    uint32_t *tail_p = nullptr;
    auto append = [&](const ShadeOut &so, uint32_t word) {
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(so.n_rays >= 1);
        const uint64_t m2 = __builtin_amdgcn_ballot_w64(so.n_rays == 2);
        const uint32_t c1 = (uint32_t)__builtin_popcountll(m1), c2 = (uint32_t)__builtin_popcountll(m2);
        if ((c1 + c2) == 0u) return;  // wave-uniform
        uint32_t wbase = 0;
        if (lane == 0u) wbase = atomicAdd(tail_p, c1 + c2);
        wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
        if (so.n_rays >= 1) {
            const uint32_t slot = wbase + lane_prefix(m1);
            if (slot < cap)
                store_ray(qout, slot, so.x, so.d0, so.thr0,
                          pack_word(word_pix(word), word_sample(word), meta_depth(so.meta0), meta_branch(so.meta0)));
            else
                overflow = true;
        }
        if (so.n_rays == 2) {
            const uint32_t slot = wbase + c1 + lane_prefix(m2);
            if (slot < cap)
                store_ray(qout, slot, so.x, so.d1, so.thr1,
                          pack_word(word_pix(word), word_sample(word), meta_depth(so.meta1), meta_branch(so.meta1)));
            else
                overflow = true;
        }
    };
    auto load_ray = [&](uint32_t i, PathRay &in, uint32_t &word) {
        const float4 a = ld_od0(qin, i);
        const float4 tp = ld_tp(qin, i);
        const float2 c = ld_od1(qin, i);
        in.o = mk(a.x, a.y, a.z);
        in.d = mk(a.w, c.x, c.y);
        in.thr = mk(tp.x, tp.y, tp.z);
        word = __float_as_uint(tp.w);
    };
    auto shade_and_append = [&](bool valid, PathRay &in, uint32_t word, HitRec h) {
        ShadeOut so;
        so.n_rays = 0;
        so.emits = false;
        if (valid && h.id >= 0) {
            in.pix = lds_pix[word_pix(word)];
            in.meta = pack_meta(s0 + word_sample(word), word_depth(word), word_branch(word));
            shade_hit(S, P, in, h, so);
            if (so.emits) add_radiance_lds(lds_acc, m, word_pix(word), so.contrib);
        }
        append(so, word);
    };
    // 64 parked rays (or the rest): walk, shade, append
    auto walk_batch = [&](uint32_t e, bool valid) {
        PathRay in;
        uint32_t word = 0;
        HitRec h;
        h.t = 0.0f;
        h.id = -1;
        if (valid) {
            load_ray(p_idx[e], in, word);
            h.t = p_t[e];
            h.id = p_id[e];
            if (walk_deferred(S, in.o, in.d, stacks, h.t, h.id, &leaves))
                h = scan_scene<true, true>(S, in.o, in.d, stacks);
        }
        shade_and_append(valid, in, word, h);
    };
    for (uint32_t depth = 0; depth < (uint32_t)kMaxDepth && n != 0u; ++depth) {
        qin = slice_of((depth & 1u) ? q1 : q0, b, cap);
        qout = slice_of((depth & 1u) ? q0 : q1, b, cap);
        __syncthreads();  // level `depth` of the stream is complete and visible to the whole workgroup
        if (tid == 0) s_tail_p[(depth + 1u) & 1u] = 0u;  // two-counter protocol of k_pass
        tail_p = s_tail_p + (depth & 1u);
        total += n;
        for (uint32_t j0 = 0; j0 < n; j0 += kBlock) {  // uniform trip count: every lane reaches the ballots
            const uint32_t i = j0 + tid;
            PathRay in;
            uint32_t word = 0;
            HitRec h;
            h.t = 0.0f;
            h.id = -1;
            bool want = false;
            if (i < n) {
                load_ray(i, in, word);
                h = intersect_scene_dev<true, true>(S, in.o, in.d, stacks, &want);
            }
            shade_and_append(i < n && !want, in, word, h);
            const uint64_t mw = __builtin_amdgcn_ballot_w64(want);
            PT_WSTAT(S, 7, __builtin_popcountll(__builtin_amdgcn_ballot_w64(i < n)));  // rays
            PT_WSTAT(S, 8, __builtin_popcountll(mw));                                  // parked
            PT_WSTAT(S, 9, 1);                                                         // scan trips of a wave
            if (mw != 0ull) {
                if (want) {
                    const uint32_t e = n_park + lane_prefix(mw);
                    p_idx[e] = i;
                    p_t[e] = h.t;
                    p_id[e] = h.id;
                }
                n_park += (uint32_t)__builtin_popcountll(mw);
            }
            if (n_park >= 64u) {
                n_park -= 64u;
                walk_batch(n_park + lane, true);
            }
        }
        if (n_park != 0u) {
            walk_batch(lane, lane < n_park);
            n_park = 0u;
        }
        __syncthreads();  // every append of this level is counted
        const uint32_t tail = *tail_p;
        n = tail < cap ? tail : cap;
    }
    if (overflow) atomicOr(flags, 1u);
    __syncthreads();
    if (tid == 0) blk_rays[b] += total;
    const size_t plane = (size_t)F.n_streams * m;
    for (uint32_t k = tid; k < 3u * mb; k += kBlock) {
        const uint32_t c = k / mb, p = k - c * mb;
        const unsigned long long v = lds_acc[c * m + p];
        if (v) acc[(size_t)c * plane + (size_t)b * m + p] += v;
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_resolve(const unsigned long long *__restrict__ acc,
                                                    float *__restrict__ out, uint32_t npix, uint32_t spp,
                                                    uint32_t n_streams, uint32_t m, uint32_t clamp) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= npix) return;
    // pixel p is pixel j = p / K of stream b = p % K; its accumulator is slot b*m + j (megakernel: K = 1, m = npix)
    const size_t plane = (size_t)n_streams * m;
    const size_t slot = (size_t)(p % n_streams) * m + p / n_streams;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double sum = (double)acc[(size_t)c * plane + slot] * (1.0 / 4294967296.0);
        const float v = (float)sum / (float)spp;  // radiance_v / samples_per_pixel, mod.rs:849
        out[(size_t)p * 3 + c] = clamp ? clamp01(v) : v;  // mod.rs:852-856 (pt_ctx_radiance: the mean itself)
    }
}

// pipeline j of n wrote its chunks back to back; its k-th chunk is the call's (k*n + j)-th chunk
__global__ __launch_bounds__(kBlock) void k_scatter_chunks(const float *__restrict__ src, float *__restrict__ dst,
                                                           uint32_t npix, uint32_t C, uint32_t n, uint32_t j) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= npix) return;
    const uint32_t k = p / C, w = p - k * C;
    const size_t at = ((size_t)k * n + j) * C + w;
    dst[at * 3 + 0] = src[(size_t)p * 3 + 0];
    dst[at * 3 + 1] = src[(size_t)p * 3 + 1];
    dst[at * 3 + 2] = src[(size_t)p * 3 + 2];
}

// ------------------------------------------------------------------------------------------------
// Persistent megakernel: one lane = one (pixel, sample chunk); the lane walks its samples one after the
// other and each loop trip advances every live path of the wave by one bounce, so the intersect and
// shade code is executed by (nearly) full waves whatever the depths of the individual paths.
// The refract split (mod.rs:775-786) pushes the transmitted ray on a two-entry stack in registers.
// One launch = one ROUND: the samples [s_begin, s_end) of every pixel of the call; item = (pixel, part of the round):
// part k of `n_split` walks the samples s_begin + k*lane_spp ... (< s_end).
// (Scenes whose candidate records fit LDS run k_mega_cand below instead: the candidate scan, two paths per lane.  This form -
// every object and triangle per lane - remains for PT_CAND_SCAN=0 / PT_FLAG_NO_BVH and for scenes with too many records.)
template <bool BVH, bool PROBE>
__global__ __launch_bounds__(kBlock) void k_mega(DevScene S, FrameParams F, unsigned long long *__restrict__ acc,
                                                 uint32_t s_begin, uint32_t s_end, uint32_t lane_spp, uint32_t n_split,
                                                 unsigned long long *__restrict__ total_rays) {
    const uint64_t items = (uint64_t)F.npix * n_split;
    unsigned long long rays = 0;
    if (BVH) stage_bvh(S, dyn_lds);
    for (uint64_t first = (uint64_t)blockIdx.x * kBlock; first < items; first += (uint64_t)gridDim.x * kBlock) {
        const uint64_t item = first + threadIdx.x;
        const bool lane_valid = item < items;
        const uint32_t pl = lane_valid ? (uint32_t)(item % F.npix) : 0u;
        const uint32_t part = lane_valid ? (uint32_t)(item / F.npix) : 0u;
        uint32_t s = s_begin + part * lane_spp;
        const uint32_t s_lim = s + lane_spp;
        const uint32_t s_stop = lane_valid ? (s_lim < s_end ? s_lim : s_end) : s;
        if (s > s_stop) s = s_stop;
        uint64_t ar = 0, ag = 0, ab = 0;
        PathRay cur, st0, st1;
        cur.o = cur.d = cur.thr = mk(0.0f, 0.0f, 0.0f);
        cur.pix = 0;
        cur.meta = 0;
        st0 = cur;
        st1 = cur;
        int sp = 0;
        bool active = false;
        for (;;) {
            if (!active) {
                if (sp == 2) {
                    cur = st1;
                    sp = 1;
                    active = true;
                } else if (sp == 1) {
                    cur = st0;
                    sp = 0;
                    active = true;
                } else if (s < s_stop) {
                    cur = primary_ray<PROBE>(F, global_pixel(F, pl), s);
                    ++s;
                    active = true;
                }
            }
            if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
            HitRec h;
            h.t = 0.0f;
            h.id = -1;
            if (active) {
                h = intersect_scene_dev<BVH>(S, cur.o, cur.d, dyn_lds);
                ++rays;
                if (h.id < 0) {
                    active = false;
                } else {
                    ShadeOut so;
                    shade_hit(S, F, cur, h, so);
                    if (so.emits) {
                        ar += to_fixed(so.contrib.x);
                        ag += to_fixed(so.contrib.y);
                        ab += to_fixed(so.contrib.z);
                    }
                    if (so.n_rays == 0) {
                        active = false;
                    } else {
                        cur.o = so.x;
                        cur.d = so.d0;
                        cur.thr = so.thr0;
                        cur.meta = so.meta0;
                        if (so.n_rays == 2) {
                            PathRay child = cur;
                            child.d = so.d1;
                            child.thr = so.thr1;
                            child.meta = so.meta1;
                            if (sp == 0)
                                st0 = child;
                            else
                                st1 = child;
                            ++sp;
                        }
                    }
                }
            }
        }
        if (lane_valid) {
            if (ar) atomicAdd(&acc[pl], (unsigned long long)ar);
            if (ag) atomicAdd(&acc[(size_t)F.npix + pl], (unsigned long long)ag);
            if (ab) atomicAdd(&acc[2 * (size_t)F.npix + pl], (unsigned long long)ab);
        }
    }
    // one counter update per wave
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63u) == 0u && rays) atomicAdd(total_rays, rays);
}

// ------------------------------------------------------------------------------------------------
// The megakernel with the candidate scan, round 4: TWO PATHS PER LANE, one trip apart.
// k_mega<.., CAND> (round 3) finished every ray in the trip that started it: whatever the wave's ring held at the end of a
// trip was tested as a partial batch (two batches per trip, the second nearly empty), and a lane whose path ended made the
// whole wave run render_pixel's ray maker for its handful of lanes in EVERY trip.  Here a lane walks its item's samples as
// two interleaved paths: a trip STARTS the ray of one path (spheres, filters, candidates to the ring, full batches - the ray
// and its key wait in the lane's LDS slot of that parity) and FINISHES the ray the other path started a trip earlier (what
// the ring still held of it has been tested by this trip's batches, or is now), shades it, and the continuation is the ray
// the next trip starts.  That is k_pass_cand's trip with the rays bound to their lanes: no ray ever goes to memory, radiance
// is summed in the lane's registers, and only the transmitted child of a refract split (mod.rs:775-786: at most two per path
// are waiting) is put aside - on a four-entry stack per lane in global memory, touched when a split happens.
// mesh.json: a lane whose finished ray may hit a BVH mesh (bvh_wants) walks it in that trip, all lanes of the wave working
// the walk queue (as round 3).
// PRIMARY RAYS IN DENSE ROUNDS.  A path ends after 8.7 bounces on average, so in every trip some seven of the wave's 64
// lanes need render_pixel's next primary ray - and the ray maker (Philox, two tent filters, two divisions, a normalisation:
// 150 instructions) ran in every trip for those seven.  Each lane therefore keeps up to `n_spare_max` (2 or 4) primary rays of its own
// pixel's next samples in LDS (the direction: 16 B each; the origin is the lens), made in ROUNDS in which every lane with
// room takes part, started when some lane that needs a ray has none: one round in four to five trips at about half the
// lanes instead of one in every trip at a ninth.  (pt_ctx_radiance's fixed probe ray needs no maker and no spares.)
// (two per lane, four where the workgroup's LDS has room for them: 8 / 16 KB)
__host__ __device__ constexpr size_t mega_spare_bytes(uint32_t depth) { return (size_t)depth * kBlock * sizeof(float4); }
constexpr size_t kMegaOwnerBytes = kBlock * sizeof(uint32_t);  // the rounds' owner tables
struct MegaStack {
    float4 *a;  // [4][lanes] origin xyz, direction x
    float4 *b;  // [4][lanes] throughput rgb, meta
    float2 *c;  // [4][lanes] direction yz
    uint32_t lanes;
};
constexpr uint32_t kMegaStackEntries = 4;
// items a wave takes from the global counter at a time.  Small: what a wave holds of its share when the counter runs out is
// work the other waves cannot take over (cornell.json 1024x768 @1024, items of 43 samples: 256 per atomic 34.9, 128: 37.2, 64:
// 41.0, 32: 41.3, 16: 41.1 G bounces/s) - and 100 000 atomics per launch are nothing.
constexpr uint32_t kMegaItemChunk = 32;
__host__ __device__ inline size_t mega_stack_bytes(uint32_t lanes) { return (size_t)lanes * kMegaStackEntries * 40u; }

template <bool BVH, bool PROBE>
__global__ __launch_bounds__(kBlock, BVH ? 4 : PT_MEGA_WAVES) void k_mega_cand(DevScene S, FrameParams F, unsigned long long *__restrict__ acc,
                                                         uint32_t s_begin, uint32_t s_end, uint32_t lane_spp, uint32_t n_split,
                                                         unsigned long long *__restrict__ total_rays, MegaStack stk, uint32_t spare_off,
                                                         uint32_t n_spare_max, uint32_t surf_off) {
    const uint64_t items = (uint64_t)F.npix * n_split;
    unsigned long long rays = 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gl = blockIdx.x * kBlock + threadIdx.x;  // this lane's column of the split stacks
    CandLds cand{};
    CandRing ring;
    ring.head = 0u;
    ring.count = 0u;
    WalkQueue wq{};
    unsigned long long *walk_keys = nullptr;
    if (BVH) {
        char *wl = reinterpret_cast<char *>(dyn_lds) + intersect_cand_lds_bytes() + (((size_t)S.n_cand_pairs * sizeof(CandPairRec) + 15) & ~(size_t)15);
        char *qb = wl + (size_t)(threadIdx.x >> 6) * pass_cand_queue_bytes(S);
        wq.redo = reinterpret_cast<uint32_t *>(qb);
        wq.ent = reinterpret_cast<uint2 *>(qb + kWalkQueueHeader);
        wq.cap = (uint32_t)((pass_cand_queue_bytes(S) - kWalkQueueHeader) / 8u);
        if (S.walk_queue_cap >= 128u && S.walk_queue_cap < wq.cap) wq.cap = S.walk_queue_cap;
        walk_keys = reinterpret_cast<unsigned long long *>(wl + pass_cand_queues_bytes(S) + (size_t)(threadIdx.x >> 6) * kCandWalkKeyBytes);
    }
    {
        char *wbase = reinterpret_cast<char *>(dyn_lds) + (size_t)(threadIdx.x >> 6) * kCandWaveBytes;
        cand.ray_a = reinterpret_cast<float4 *>(wbase);
        cand.keys = reinterpret_cast<unsigned long long *>(wbase + 128u * 16u);
        cand.ray_b = reinterpret_cast<float2 *>(wbase + 128u * 24u);
        cand.queue = reinterpret_cast<uint16_t *>(wbase + 128u * 32u);
        char *sbase = reinterpret_cast<char *>(dyn_lds) + intersect_cand_lds_bytes();
        cand.staged = reinterpret_cast<const CandPairRec *>(sbase);
        cand.surf = S.surf;
        const uint4 *src = reinterpret_cast<const uint4 *>(S.cand_pairs);
        uint4 *dst = reinterpret_cast<uint4 *>(sbase);
        const uint32_t n_rows = S.n_cand_pairs * (uint32_t)(sizeof(CandPairRec) / 16u);
        for (uint32_t k = threadIdx.x; k < n_rows; k += kBlock) dst[k] = src[k];
        if (surf_off != 0u) {  // the shading records by rank too, when they fit (as k_pass_cand)
            const uint4 *src2 = reinterpret_cast<const uint4 *>(S.surf);
            uint4 *dst2 = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(dyn_lds) + surf_off);
            const uint32_t n_rows2 = (S.n_objs + S.n_tris) * (uint32_t)(sizeof(SurfRec) / 16u);
            for (uint32_t k = threadIdx.x; k < n_rows2; k += kBlock) dst2[k] = src2[k];
            cand.surf = reinterpret_cast<const SurfRec *>(dst2);
        }
        __syncthreads();
    }
    ShadeParams P{};  // (the RNG key as two scalars of its own: see k_pass_cand)
    {
        uint32_t k_lo = F.seed_lo, k_hi = F.seed_hi;
        asm volatile("" : "+s"(k_lo), "+s"(k_hi));
        P.seed_lo = k_lo;
        P.seed_hi = k_hi;
    }
    // [n_spare_max][kBlock] float4 at the end of the workgroup's LDS: the lanes' spare primary rays (entry k of lane t at
    // k * kBlock + t)
    float4 *const spare = reinterpret_cast<float4 *>(reinterpret_cast<char *>(dyn_lds) + spare_off);
    // ... and behind them, per wave, [64] u32: which lane owns the w-th ray of a round of the ray maker
    uint32_t *const owner_of = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(dyn_lds) + spare_off + mega_spare_bytes(n_spare_max)) + (threadIdx.x & ~63u);
    // ITEMS ARE HANDED OUT, NOT DEALT.  An item is (pixel, part of the round's samples); a lane takes its next one when its
    // item is exhausted - from its wave's share of a counter in global memory (kMegaItemChunk items per atomic; the lanes that
    // ask in a trip are numbered by a prefix count).  Dealt in advance - item k of a lane = first + k * stride - the lanes of
    // a wave finish far apart: a pixel on the glass sphere costs three to four times the rays of one on a wall, and a wave
    // whose 64 neighbouring pixels straddle its silhouette ran a third of its trips for its slowest lanes alone.  (One atomic
    // per asking wave and trip instead of per chunk: at 6 M items per launch the ONE counter word saturates - 63 M atomics/s -
    // and the waves wait half their lifetime for it: 25 G bounces/s against 39.)  The host zeroes the counter before a launch.
    unsigned long long *const item_ctr = total_rays + 7;
    unsigned long long pool_next = 0ull;  // wave-uniform: the wave's share of the counter
    uint32_t pool_left = 0u;
    bool have_item = false, no_more = false;
    uint32_t pl = 0u, gpix = 0u, s = 0u, s_stop = 0u, n_spare = 0u;
    uint64_t ar = 0, ag = 0, ab = 0;
    {
        // the ray the next trip starts (this lane's path of that parity), and what the ray started a trip ago still needs
        PathRay cur;
        cur.o = cur.d = cur.thr = mk(0.0f, 0.0f, 0.0f);
        cur.pix = gpix;
        cur.meta = 0;
        bool cur_active = false;
        vec3 prev_thr = mk(0.0f, 0.0f, 0.0f);
        uint32_t prev_meta = 0;
        bool prev_valid = false;
        uint32_t sp = 0;   // entries on this lane's split stack
        uint32_t par = 0;  // the slots (LDS) of the ray started in this trip
        uint32_t w_room = 0u;  // wave-uniform: spare rays taken since the last round of the ray maker
        for (;;) {
            // 0. an item whose samples are all traced (nothing in flight, nothing waiting): its radiance goes to the pixel, the
            //    lane asks for its next item
            const bool ask = !cur_active && !prev_valid && sp == 0u && n_spare == 0u && s >= s_stop && !no_more;
            const uint64_t m_ask = __builtin_amdgcn_ballot_w64(ask);
            if (m_ask != 0ull) {  // wave-uniform
                if (pool_left == 0u) {  // the wave's share of the counter is used up: kMegaItemChunk more
                    unsigned long long got = 0ull;
                    const uint32_t src = (uint32_t)__builtin_ctzll(m_ask);
                    if (lane == src) got = atomicAdd(item_ctr, (unsigned long long)kMegaItemChunk);
                    pool_next = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(got >> 32), (int)src) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)got, (int)src);
                    pool_left = kMegaItemChunk;
                }
                const uint32_t n_ask = (uint32_t)__builtin_popcountll(m_ask);
                const uint32_t give = n_ask < pool_left ? n_ask : pool_left;  // (the others ask again in the next trip)
                const uint32_t rank = lane_prefix(m_ask);
                if (ask && rank < give) {
                    if (have_item) {
                        if (ar) atomicAdd(&acc[pl], (unsigned long long)ar);
                        if (ag) atomicAdd(&acc[(size_t)F.npix + pl], (unsigned long long)ag);
                        if (ab) atomicAdd(&acc[2 * (size_t)F.npix + pl], (unsigned long long)ab);
                        ar = ag = ab = 0;
                        have_item = false;
                    }
                    const unsigned long long item = pool_next + rank;
                    if (item < items) {
                        pl = (uint32_t)(item % F.npix);
                        const uint32_t part = (uint32_t)(item / F.npix);
                        s = s_begin + part * lane_spp;
                        const uint32_t s_lim = s + lane_spp;
                        s_stop = s_lim < s_end ? s_lim : s_end;
                        if (s > s_stop) s = s_stop;
                        gpix = global_pixel(F, pl);
                        have_item = true;
                    } else {
                        no_more = true;
                    }
                }
                pool_next += give;
                pool_left -= give;
            }
            // 1. a path without a ray takes the lane's most recent split child, or the item's next sample
            if (!cur_active) {
                if (sp != 0u) {
                    --sp;
                    const size_t at = (size_t)sp * stk.lanes + gl;
                    const float4 a = stk.a[at], b = stk.b[at];
                    const float2 c2 = stk.c[at];
                    cur.o = mk(a.x, a.y, a.z);
                    cur.d = mk(a.w, c2.x, c2.y);
                    cur.thr = mk(b.x, b.y, b.z);
                    cur.meta = __float_as_uint(b.w);
                    cur_active = true;
                } else if (PROBE && s < s_stop) {
                    cur = primary_ray<true>(F, gpix, s);
                    ++s;
                    cur_active = true;
                }
            }
            if (!PROBE) {
                // A round of the ray maker when the wave's lanes have room for 64 rays between them (w_room counts the spares
                // taken since the last round), or a lane that needs a primary ray has no spare one.  THE WAVE MAKES THE RAYS
                // TOGETHER: lane L has room for d_L rays (0..4: its pixel's next samples); the d_L are numbered through by a
                // prefix count, and worker w makes the w-th ray of the round - whichever lane's pixel and sample that is
                // (owner, pixel and sample index by ds_bpermute) - and writes it into the OWNER's spare slots.  A round is then 64
                // rays by 64 lanes whatever the single lanes' deficits are; made by their owners, rounds ran for the few lanes
                // with room (a lane that has just taken a new item starves four times in a row): 0.7 per trip at ten lanes.
                const bool starved = !cur_active && n_spare == 0u && s < s_stop;
                if (__builtin_amdgcn_ballot_w64(starved) != 0ull || w_room >= 64u) {
                    w_room = 0u;
                    for (;;) {
                        const uint32_t left = s_stop - s, free_slots = n_spare_max - n_spare;
                        const uint32_t d = left < free_slots ? left : free_slots;  // 0..4
                        const uint64_t b0 = __builtin_amdgcn_ballot_w64((d & 1u) != 0u), b1 = __builtin_amdgcn_ballot_w64((d & 2u) != 0u),
                                       b2 = __builtin_amdgcn_ballot_w64((d & 4u) != 0u);
                        const uint32_t total = (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(b1) +
                                               4u * (uint32_t)__builtin_popcountll(b2);
                        if (total == 0u) break;
                        const uint32_t before = lane_prefix(b0) + 2u * lane_prefix(b1) + 4u * lane_prefix(b2);  // rays of the lanes below
                        const uint32_t served = before >= 64u ? 0u : (before + d > 64u ? 64u - before : d);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        for (uint32_t k = 0; k < 4u; ++k)
                            if (k < served) owner_of[before + k] = lane;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const uint32_t n_make = total < 64u ? total : 64u;
                        const uint32_t owner = lane < n_make ? owner_of[lane] : lane;
                        const int sel = (int)(owner << 2);
                        const uint32_t o_before = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)before);
                        const uint32_t o_pix = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)gpix);
                        const uint32_t o_s = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)s);
                        if (lane < n_make) {
                            const uint32_t smp = o_s + (lane - o_before);
                            const PathRay r = primary_ray<false>(F, o_pix, smp);
                            spare[(smp & (n_spare_max - 1u)) * kBlock + ((threadIdx.x & ~63u) | owner)] = make_float4(r.d.x, r.d.y, r.d.z, 0.0f);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        s += served;
                        n_spare += served;
#ifdef PT_MEGA_STATS
                        if (lane == 0u) {
                            atomicAdd(total_rays + 5, 1ull);
                            atomicAdd(total_rays + 6, (unsigned long long)n_make);
                        }
#endif
                        if (total <= 64u + 31u) break;  // (what is left would be a round of fewer than half the lanes)
                    }
                }
                w_room += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(!cur_active && n_spare != 0u));
                if (!cur_active && n_spare != 0u) {  // (the lane's own LDS entries: written and read by this lane only)
                    const uint32_t smp = s - n_spare;
                    const float4 dd = spare[(smp & (n_spare_max - 1u)) * kBlock + threadIdx.x];
                    --n_spare;
                    cur.o = mk(F.lens_x, F.lens_y, F.lens_z);
                    cur.pix = gpix;
                    cur.d = mk(dd.x, dd.y, dd.z);
                    cur.thr = mk(1.0f, 1.0f, 1.0f);
                    cur.meta = pack_meta(smp, 0u, 1u);  // radiance(&ray, 0, ..), mod.rs:844
                    cur_active = true;
                }
            }
            const uint64_t m_cur = __builtin_amdgcn_ballot_w64(cur_active), m_prev = __builtin_amdgcn_ballot_w64(prev_valid);
            if (m_cur == 0ull && m_prev == 0ull) break;  // every path of the wave's items has ended
#ifdef PT_MEGA_STATS
            {
                const unsigned long long n_item = (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(have_item));
                const unsigned long long n_dry = (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(have_item && !cur_active && s >= s_stop && n_spare == 0u));
                const unsigned long long n_nomore = (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(no_more));
                if (lane == 0u) {
                    atomicAdd(total_rays + 8, n_item);
                    atomicAdd(total_rays + 9, n_dry);
                    atomicAdd(total_rays + 10, n_nomore);
                }
            }
            if (lane == 0u) {
                atomicAdd(total_rays + 2, 1ull);
                atomicAdd(total_rays + 3, (unsigned long long)__builtin_popcountll(m_cur));
                atomicAdd(total_rays + 4, (unsigned long long)__builtin_popcountll(m_prev));
            }
#endif
            // 2. start this trip's rays (every lane takes part; the lanes without one have no candidates)
            const uint32_t pend_entries = ring.count;  // entries of the rays to finish that are still queued (< 64)
            bool ran_batch = false;
            if (m_cur != 0ull) {
                const uint32_t slot = (par << 6) | lane;
                cand.ray_a[slot] = make_float4(cur.o.x, cur.o.y, cur.o.z, cur.d.x);
                cand.ray_b[slot] = make_float2(cur.d.y, cur.d.z);
                float bound;
                const unsigned long long key0 = cand_spheres(S, cur.o, cur.d, &bound);
                cand.keys[slot] = cur_active ? key0 : kKeyMiss;
                const uint32_t before = ring.head;
                cand_filter_and_drain<true>(S, cand, ring, lane, par, cur_active, cur.o, cur.d, bound);
                ran_batch = ring.head != before;
            }
            // 3. finish the rays started a trip ago
            PathRay next;
            next.o = next.d = next.thr = mk(0.0f, 0.0f, 0.0f);
            next.pix = gpix;
            next.meta = 0;
            bool next_active = false;
            if (m_prev != 0ull) {
                if (pend_entries != 0u && !ran_batch) cand_batch<true>(S, cand, ring, lane, ring.count);
                const uint32_t slot = ((par ^ 1u) << 6) | lane;
                unsigned long long key = kKeyMiss;
                PathRay pr;
                pr.o = pr.d = mk(0.0f, 0.0f, 0.0f);
                if (prev_valid) {
                    key = load_key(&cand.keys[slot]);
                    const float4 ra = cand.ray_a[slot];
                    const float2 rb = cand.ray_b[slot];
                    pr.o = mk(ra.x, ra.y, ra.z);
                    pr.d = mk(ra.w, rb.x, rb.y);
                }
                if (BVH) {
                    const bool want = prev_valid && bvh_wants(S, pr.o, pr.d, __uint_as_float((uint32_t)(key >> 32)));
                    if (__builtin_amdgcn_ballot_w64(want) != 0ull)  // wave-uniform
                        key = walk_deferred_keys(S, walk_nodes(S), pr.o, pr.d, wq, key, walk_keys, want);
                }
                if (prev_valid) {
                    ++rays;
                    const uint32_t rank = (uint32_t)key;
                    if (rank != 0xffffffffu) {
                        pr.thr = prev_thr;
                        pr.pix = gpix;
                        pr.meta = prev_meta;
                        const Surface sf = fetch_surface_rank(cand.surf, pr.o, pr.d, __uint_as_float((uint32_t)(key >> 32)), rank);
                        ShadeOut so;
                        shade_surface<kShadeAll>(P, pr, sf, so);
                        if (so.emits) {
                            ar += to_fixed(so.contrib.x);
                            ag += to_fixed(so.contrib.y);
                            ab += to_fixed(so.contrib.z);
                        }
                        if (so.n_rays >= 1) {
                            next.o = so.x;
                            next.d = so.d0;
                            next.thr = so.thr0;
                            next.meta = so.meta0;
                            next_active = true;
                        }
                        if (so.n_rays == 2 && sp >= kMegaStackEntries) {
                            atomicOr(total_rays + 1, 1ull);  // (cannot happen: two paths, at most two waiting children each)
                        } else if (so.n_rays == 2) {  // the transmitted child waits on the lane's stack
                            const size_t at = (size_t)sp * stk.lanes + gl;
                            stk.a[at] = make_float4(so.x.x, so.x.y, so.x.z, so.d1.x);
                            stk.b[at] = make_float4(so.thr1.x, so.thr1.y, so.thr1.z, __uint_as_float(so.meta1));
                            stk.c[at] = make_float2(so.d1.y, so.d1.z);
                            ++sp;
                        }
                    }
                }
            }
            // 4. the ray started in this trip waits; the continuation of the finished one is what the next trip starts
            prev_thr = cur.thr;
            prev_meta = cur.meta;
            prev_valid = cur_active;
            cur = next;
            cur_active = next_active;
            par ^= 1u;
        }
    }
    // (every lane's last item was flushed in step 0 of the trip after its last ray: the loop only ends when no lane holds an
    // item with anything left - a lane with have_item set at the exit has an exhausted item whose flush is still due)
    if (have_item) {
        if (ar) atomicAdd(&acc[pl], (unsigned long long)ar);
        if (ag) atomicAdd(&acc[(size_t)F.npix + pl], (unsigned long long)ag);
        if (ab) atomicAdd(&acc[2 * (size_t)F.npix + pl], (unsigned long long)ab);
    }
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63u) == 0u && rays) atomicAdd(total_rays, rays);
}

// ------------------------------------------------------------------------------------------------
// single-ray queries (picking / click-debug callers of intersect_scene)
__global__ __launch_bounds__(kBlock) void k_query(DevScene S, const float *__restrict__ o,
                                                  const float *__restrict__ d, uint32_t n, float *__restrict__ t,
                                                  int32_t *__restrict__ object_id, int32_t *__restrict__ tri_id,
                                                  float *__restrict__ x, float *__restrict__ nrm) {
    stage_bvh(S, dyn_lds);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const vec3 ro = mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        const vec3 rd = mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        const HitRec h = intersect_scene_dev<true>(S, ro, rd, dyn_lds);
        int32_t oid = -1, tid = -1;
        vec3 hx = mk(0.0f, 0.0f, 0.0f), hn = hx;
        float ht = 0.0f;
        if (h.id >= 0) {
            const Surface sf = fetch_surface(S, ro, rd, h);
            hx = sf.x;
            hn = sf.n;
            ht = h.t;
            if (h.id >= (int32_t)S.n_objs) {
                const uint32_t k = (uint32_t)(h.id - (int32_t)S.n_objs);
                oid = (int32_t)S.tri_shade[k].owner;
                tid = (int32_t)(k - S.objs[oid].tri_begin);
            } else {
                oid = h.id;
            }
        }
        t[i] = ht;
        object_id[i] = oid;
        tri_id[i] = tid;
        x[3 * i] = hx.x;
        x[3 * i + 1] = hx.y;
        x[3 * i + 2] = hx.z;
        nrm[3 * i] = hn.x;
        nrm[3 * i + 1] = hn.y;
        nrm[3 * i + 2] = hn.z;
    }
}

// intersect_bounds / get_orbit_point (picking callers next to the path: mod.rs:282-290, viewport_tab.rs:401-431).
// One lane = one ray, plain per-lane loops: these queries come one ray at a time.  `boxes` holds 6 pair records per
// object (box_pair_records; ignored for spheres).
__device__ __forceinline__ bool sphere_hit(const ObjRec &g, vec3 o, vec3 d, float *t_out) {  // intersect_sphere, mod.rs:412-427
    const vec3 op = mk(g.cx, g.cy, g.cz) - o;
    const float b = dot(op, d);
    const float det = (b * b - dot(op, op)) + g.rr;
    if (det < 0.0f) return false;
    const float sq = f_sqrt(det);
    const float t0 = b - sq, t1 = b + sq;
    if (t0 >= 1e-4f) {
        *t_out = t0;
        return true;
    }
    if (t1 >= 1e-4f) {
        *t_out = t1;
        return true;
    }
    return false;
}
// Triangle::intersect over `count` pair records in list order (first of equal distances wins, mod.rs:598)
__device__ __forceinline__ bool scan_pairs(const TriPairRec *recs, uint32_t count, vec3 o, vec3 d, float *t_out, int32_t *id_out) {
    const f32x2 ox2 = splat2(o.x), oy2 = splat2(o.y), oz2 = splat2(o.z);
    const f32x2 dx2 = splat2(d.x), dy2 = splat2(d.y), dz2 = splat2(d.z);
    float mt = __builtin_inff();
    int32_t mid = -1;
    for (uint32_t p = 0; p < count; ++p) test_pair<false>(recs[p], ox2, oy2, oz2, dx2, dy2, dz2, mt, mid);
    *t_out = mt;
    *id_out = mid;
    return mid >= 0;
}
// normal of the triangle `id` found by scan_pairs: va_vb.cross(va_vc).normalize() (mod.rs:605)
__device__ __forceinline__ vec3 pair_normal(const TriPairRec *recs, uint32_t count, int32_t id) {
    for (uint32_t p = 0; p < count; ++p)
        for (int hf = 0; hf < 2; ++hf)
            if ((int32_t)recs[p].id[hf] == id)
                return normalize(cross(mk(recs[p].e1x[hf], recs[p].e1y[hf], recs[p].e1z[hf]),
                                       mk(recs[p].e2x[hf], recs[p].e2y[hf], recs[p].e2z[hf])));
    return mk(0.0f, 0.0f, 0.0f);
}

// mode 0: SceneObjectData::intersect_bounds of object `object` -> hit, t, x, normal
// mode 1: get_orbit_point over the whole scene -> hit (found), t, x (the point), object_id
__global__ __launch_bounds__(kBlock) void k_bounds(DevScene S, const TriPairRec *__restrict__ boxes, uint32_t mode,
                                                   uint32_t object, const float *__restrict__ o,
                                                   const float *__restrict__ d, uint32_t n, int32_t *__restrict__ hit,
                                                   float *__restrict__ t, float *__restrict__ x,
                                                   float *__restrict__ nrm, int32_t *__restrict__ object_id) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const vec3 ro = mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        const vec3 rd = mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        bool have = false;
        float best_t = 0.0f;
        vec3 best_n = mk(0.0f, 0.0f, 0.0f);
        int32_t best_obj = -1;
        const int32_t k_first = mode == 0u ? (int32_t)object : (int32_t)S.n_objs - 1;
        const int32_t k_last = mode == 0u ? (int32_t)object : 0;
        for (int32_t k = k_first; k >= k_last; --k) {  // reverse order, strict '<' (viewport_tab.rs:404,419)
            const ObjRec g = S.objs[k];
            const MatRec m = S.mats[k];
            float tb = 0.0f;
            vec3 nb = mk(0.0f, 0.0f, 0.0f);
            bool hb;
            if (g.kind == kKindSphere) {
                hb = sphere_hit(g, ro, rd, &tb);
                if (hb) nb = normalize((ro + rd * tb) - mk(m.px, m.py, m.pz));
            } else {
                int32_t bid;
                hb = scan_pairs(boxes + 6u * (uint32_t)k, 6u, ro, rd, &tb, &bid);
                if (hb) nb = pair_normal(boxes + 6u * (uint32_t)k, 6u, bid);
            }
            if (!hb) continue;
            float tn = tb;
            vec3 nn = nb;
            if (mode == 1u && g.kind == kKindMesh) {  // the object's own hit if it has one, else the bounds hit
                float tg, to;
                int32_t tid;
                if (sphere_hit(g, ro, rd, &tg) && scan_pairs(S.tri_pairs + g.pair_begin, g.pair_count, ro, rd, &to, &tid)) {
                    tn = to;
                    const TriShade ts = S.tri_shade[tid];
                    nn = mk(ts.nx, ts.ny, ts.nz);
                }
            }
            if (!have || tn < best_t) {
                have = true;
                best_t = tn;
                best_n = nn;
                best_obj = k;
            }
        }
        const vec3 bx = have ? ro + rd * best_t : mk(0.0f, 0.0f, 0.0f);
        if (hit) hit[i] = have ? 1 : 0;
        if (t) t[i] = have ? best_t : 0.0f;
        if (object_id) object_id[i] = best_obj;
        if (x) {
            x[3 * i] = bx.x;
            x[3 * i + 1] = bx.y;
            x[3 * i + 2] = bx.z;
        }
        if (nrm) {
            nrm[3 * i] = best_n.x;
            nrm[3 * i + 1] = best_n.y;
            nrm[3 * i + 2] = best_n.z;
        }
    }
}

// numerics self-check kernel: the device evaluates the contract functions on given inputs so that the
// tests can compare them bit for bit with the oracle's host evaluation.
__global__ void k_numerics(const float *__restrict__ in, uint32_t n, float *__restrict__ out_sin,
                           float *__restrict__ out_cos, float *__restrict__ out_sqrt, float *__restrict__ out_rcp,
                           uint32_t *__restrict__ out_philox) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = in[i];
    float s, c;
    sincos_f32(v, &s, &c);
    out_sin[i] = s;
    out_cos[i] = c;
    out_sqrt[i] = f_sqrt(v);
    out_rcp[i] = f_rcp(v);
    const u32x4 r = draw_block(0x0123456789abcdefull, i, __float_as_uint(v), (i << 8) | (i & 15u));
    out_philox[4 * i + 0] = r.a;
    out_philox[4 * i + 1] = r.b;
    out_philox[4 * i + 2] = r.c;
    out_philox[4 * i + 3] = r.d;
}

// Exhaustive form of the same check for the two short sequences: f_sqrt against the compiler's IEEE square root on ALL
// 2^32 bit patterns, f_rcp against the compiler's IEEE division 1/d on every normal d with 2^-126 <= |d| <= 2^126 (its
// domain).  out[0] / out[1]: inputs whose results differ in bits (NaN == NaN); out[2] / out[3]: inputs compared.
__global__ __launch_bounds__(256) void k_numerics_sweep(unsigned long long *__restrict__ out) {
    unsigned long long bad_s = 0, bad_r = 0, n_s = 0, n_r = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const uint32_t a = (uint32_t)i & 0x7fffffffu;
        const uint32_t gs = __float_as_uint(f_sqrt(x)), ws = __float_as_uint(__builtin_sqrtf(x));
        const bool nan_both = (gs & 0x7fffffffu) > 0x7f800000u && (ws & 0x7fffffffu) > 0x7f800000u;
        bad_s += (gs != ws && !nan_both) ? 1u : 0u;
        ++n_s;
        if (a >= 0x00800000u && a <= 0x7e800000u) {  // 2^-126 <= |d| <= 2^126
            bad_r += (__float_as_uint(f_rcp(x)) != __float_as_uint(1.0f / x)) ? 1u : 0u;
            ++n_r;
        }
    }
    if (bad_s) atomicAdd(out + 0, bad_s);
    if (bad_r) atomicAdd(out + 1, bad_r);
    atomicAdd(out + 2, n_s);
    atomicAdd(out + 3, n_r);
}

// The one transcendental of the path (mod.rs:703: cos / sin of r1 = 2 pi rand01()) on EVERY argument it can see: rand01()
// returns k * 2^-24, k < 2^24 (rand 0.8.5's f32 mapping), so r1 = (2 pi as f32) * (k * 2^-24) takes 2^24 values.  The device's
// sincos_f32 against a table of the host instantiation of the same source (which tests/test_abi.py holds to the platform
// libm on the same 2^24 arguments): out[0] = arguments whose sine or cosine differs in bits, out[1] = arguments compared.
__global__ __launch_bounds__(256) void k_sincos_sweep(const uint32_t *__restrict__ want_sin, const uint32_t *__restrict__ want_cos,
                                                      unsigned long long *__restrict__ out) {
    unsigned long long bad = 0, n = 0;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < (1u << 24); k += gridDim.x * blockDim.x) {
        const float r1 = (2.0f * 3.141592653589793f) * unit_f32(k << 8);  // shade_surface's expression
        float s, c;
        sincos_f32(r1, &s, &c);
        bad += (__float_as_uint(s) != want_sin[k] || __float_as_uint(c) != want_cos[k]) ? 1u : 0u;
        ++n;
    }
    if (bad) atomicAdd(out + 0, bad);
    atomicAdd(out + 1, n);
}

// render_pixel's per-sample ray (mod.rs:805-843) for given (framebuffer index, sample) pairs, as the frame kernels make it:
// form 0 = primary_ray (k_generate, k_mega, k_pass: column and row by division), form 1 = primary_ray_at with the column
// and row precomputed (k_pass_cand keeps them per stream pixel in LDS).  For tests/kats_camera.py.
__global__ __launch_bounds__(256) void k_primary_rays(FrameParams F, const uint32_t *__restrict__ pixel, const uint32_t *__restrict__ sample,
                                                      uint32_t n, uint32_t form, float *__restrict__ o, float *__restrict__ d) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pix = pixel[i], s = sample[i];
    const PathRay r = form == 0u ? primary_ray<false>(F, pix, s) : primary_ray_at(F, pix, pix % F.width, F.height - 1u - pix / F.width, s);
    o[3 * i + 0] = r.o.x, o[3 * i + 1] = r.o.y, o[3 * i + 2] = r.o.z;
    d[3 * i + 0] = r.d.x, d[3 * i + 1] = r.d.y, d[3 * i + 2] = r.d.z;
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_sincos_sweep(hipStream_t st, const uint32_t *want_sin, const uint32_t *want_cos, unsigned long long *out2) {
    hipLaunchKernelGGL(k_sincos_sweep, dim3(2048), dim3(256), 0, st, want_sin, want_cos, out2);
}
void launch_primary_rays(hipStream_t st, const FrameParams &F, const uint32_t *pixel, const uint32_t *sample, uint32_t n, uint32_t form,
                         float *o, float *d) {
    hipLaunchKernelGGL(k_primary_rays, dim3((n + 255u) / 256u), dim3(256), 0, st, F, pixel, sample, n, form, o, d);
}
void launch_generate(hipStream_t st, uint32_t K, const FrameParams &F, const RayQueue &q, uint32_t *cnt0,
                     uint32_t cap, uint32_t s0, uint32_t s_here, uint32_t m) {
    if (F.probe)
        hipLaunchKernelGGL(k_generate<true>, dim3(K), dim3(kBlock), 0, st, F, q, cnt0, cap, s0, s_here, m);
    else
        hipLaunchKernelGGL(k_generate<false>, dim3(K), dim3(kBlock), 0, st, F, q, cnt0, cap, s0, s_here, m);
}
void launch_intersect(hipStream_t st, uint32_t K, const DevScene &S, const RayQueue &q, float2 *hit,
                      const uint32_t *cnt, uint32_t cap, unsigned long long *blk_rays) {
    if (S.n_bvh_nodes != 0u) {
        hipLaunchKernelGGL(k_intersect<true>, dim3(K), dim3(kBlockBvh),
                           bvh_park_offset(S, kBlockBvh) + bvh_park_bytes(kBlockBvh), st, S, q, hit, cnt, cap, blk_rays);
    } else if (S.cand_scan) {  // the candidate scan (PT_CAND_SCAN=0 / PT_FLAG_NO_BVH: the every-triangle scan below)
        launch_intersect_cand(st, K, S, q, hit, cnt, cap, blk_rays);
    } else {
        hipLaunchKernelGGL(k_intersect<false>, dim3(K), dim3(kBlock), 0, st, S, q, hit, cnt, cap, blk_rays);
    }
}
void launch_shade(hipStream_t st, uint32_t K, const DevScene &S, const FrameParams &F, const RayQueue &qin,
                  const RayQueue &qout, const float2 *hit, const uint32_t *cnt_in, uint32_t *cnt_out, uint32_t cap,
                  unsigned long long *acc, uint32_t *flags, uint32_t m, uint32_t s0) {
    const size_t lds = (size_t)pass_acc_slots(m) * sizeof(unsigned long long) + 16;
    ShadeParams P;
    P.idx_begin = F.idx_begin;
    P.npix = F.npix;
    P.n_streams = F.n_streams;
    P.seed_lo = F.seed_lo;
    P.seed_hi = F.seed_hi;
    P.debug = F.debug;
    P.s0 = s0;
    P.chunk_pixels = F.chunk_pixels;
    P.chunk_first = F.chunk_first;
    P.chunk_step = F.chunk_step;
    P.k_begin = F.k_begin;
    hipLaunchKernelGGL(k_shade, dim3(K), dim3(kBlock), lds, st, S, P, qin, qout, hit, cnt_in, cnt_out, cap, acc,
                       flags, m);
}
#endif  // PT_TU_FLAT
// k_pass_cand for one pass: the workgroup's LDS (`lds` bytes) is laid out by launch_pass (below); the instances WITHOUT walks
// are compiled in a translation unit of their own (pt_kernels_flat.hip: PT_CAND_WAVES waves per SIMD and the instruction
// scheduling that fits them - the Makefile says which and why), the instances with walks here.
#define PT_LAUNCH_CAND(ST, DF, BV, NL)                                                                                 \
    do {                                                                                                               \
        if (F.probe)                                                                                                   \
            PT_LAUNCH_CAND2(ST, DF, BV, true, NL);                                                                     \
        else                                                                                                           \
            PT_LAUNCH_CAND2(ST, DF, BV, false, NL);                                                                    \
    } while (0)
// (more than 64 KB of dynamic LDS - wide deep trees with hundreds of pixels per stream - has to be asked for)
#define PT_LAUNCH_CAND2(ST, DF, BV, PR, NL)                                                                               \
    do {                                                                                                               \
        if (lds > 64u * 1024u) {                                                                                       \
            const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pass_cand<ST, DF, BV, PR, NL>), \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
            if (ea != hipSuccess) {                                                                                    \
                set_error("k_pass_cand needs " + std::to_string(lds) + " bytes of LDS per workgroup for this scene: " +  \
                          hipGetErrorString(ea));                                                                      \
                return ea;                                                                                             \
            }                                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL((k_pass_cand<ST, DF, BV, PR, NL>), dim3(K), dim3(kBlock), lds, st, S2, F, q0, q1, cap, s0, s_here, m, acc, blk_rays, flags); \
    } while (0)
hipError_t launch_pass_cand_flat(hipStream_t st, uint32_t K, const DevScene &S2, const FrameParams &F, const RayQueue &q0,
                                 const RayQueue &q1, uint32_t cap, uint32_t s0, uint32_t s_here, uint32_t m, unsigned long long *acc,
                                 unsigned long long *blk_rays, uint32_t *flags, size_t lds, bool staged, bool defer);
#ifdef PT_TU_FLAT
// the stand-alone intersect step with the candidate scan (PT_FLAG_SEPARATE_KERNELS, scenes without BVH meshes)
void launch_intersect_cand(hipStream_t st, uint32_t K, const DevScene &S, const RayQueue &q, float2 *hit, const uint32_t *cnt,
                           uint32_t cap, unsigned long long *blk_rays) {
    const size_t rec = (size_t)S.n_cand_pairs * sizeof(CandPairRec);
    if (intersect_cand_lds_bytes() + rec <= 160u * 1024u / PT_ISECT_WAVES)
        hipLaunchKernelGGL(k_intersect_cand<true>, dim3(K), dim3(kBlock), intersect_cand_lds_bytes() + rec, st, S, q, hit, cnt,
                           cap, blk_rays);
    else
        hipLaunchKernelGGL(k_intersect_cand<false>, dim3(K), dim3(kBlock), intersect_cand_lds_bytes(), st, S, q, hit, cnt, cap,
                           blk_rays);
}
hipError_t launch_pass_cand_flat(hipStream_t st, uint32_t K, const DevScene &S2, const FrameParams &F, const RayQueue &q0,
                                 const RayQueue &q1, uint32_t cap, uint32_t s0, uint32_t s_here, uint32_t m, unsigned long long *acc,
                                 unsigned long long *blk_rays, uint32_t *flags, size_t lds, bool staged, bool defer) {
    // (diagnosis: PT_LDS_PAD=n asks for n bytes of LDS the kernel does not use - where does the fifth workgroup of a CU stop fitting?)
    static const size_t lds_pad = getenv("PT_LDS_PAD") ? (size_t)atol(getenv("PT_LDS_PAD")) : 0u;
    static bool said = false;
    if (getenv("PT_LDS_PAD") && !said) {
        said = true;
        fprintf(stderr, "k_pass_cand: %zu bytes of LDS per workgroup (+ %zu of padding), m = %u\n", lds, lds_pad, m);
    }
    lds += lds_pad;
    if (staged && defer)
        PT_LAUNCH_CAND(true, true, false, false);
    else if (staged)
        PT_LAUNCH_CAND(true, false, false, false);
    else if (defer)
        PT_LAUNCH_CAND(false, true, false, false);
    else
        PT_LAUNCH_CAND(false, false, false, false);
    return hipSuccess;
}
#else
hipError_t launch_pass(hipStream_t st, uint32_t K, const DevScene &S, const FrameParams &F, const RayQueue &q0,
                       const RayQueue &q1, uint32_t cap, uint32_t s0, uint32_t s_here, uint32_t m, unsigned long long *acc,
                       unsigned long long *blk_rays, uint32_t *flags) {
    // The deferral buffers are 24 KB per workgroup: worth it while 5-6 workgroups still fit a CU's 160 KB of LDS (the
    // accumulators of a stream take 28 B per pixel); frames so large that a stream owns hundreds of pixels (4096^2:
    // 1024) shade every material in place instead.
    const size_t lds_plain = pass_lds_defer_offset(m);
    const size_t lds_defer = lds_plain + (size_t)(kBlock / 64u) * 3u * kDeferCap * sizeof(float4);
    if (S.cand_scan) {
        // candidate scan: ray slots, keys and ring per wave + the workgroup's copy of the candidate and shading records while
        // as many workgroups still fit a CU's 160 KiB as the kernel is built to run waves per SIMD (with walks four: 40 KiB each)
        const bool bvh = S.n_bvh_nodes != 0u;
        // (measured with PT_LDS_PAD: four workgroups of 40 928 B share a CU, five of 32 144 B do, five of 32 400 B do not)
        const size_t budget = 160u * 1024u / (bvh ? PT_CAND_BVH_WAVES : PT_CAND_WAVES) - (bvh ? 0u : 512u);
        DevScene S2 = S;
        S2.bvh_in_lds &= ~1u;  // (nodes from global memory: PT_BVH_LDS asks for the staged k_intersect, not for this kernel)
        const size_t rec_cand = (size_t)S.n_cand_pairs * sizeof(CandPairRec);
        const size_t rec_surf = (size_t)(S.n_objs + S.n_tris) * sizeof(SurfRec);
        // glass deferral: not with walks (their queues take its place in LDS; a walked ray is shaded in place anyway)
        // (Without levels the deferral no longer pays: a chunk mixes rays of every depth and nearly every trip shades some glass
        // anyway - shading it in place, 46.5 against 46.05 G bounces/s on cornell, builds alternated; PT_GLASS_DEFER=1 brings
        // the buffers back for that comparison.)
        const bool defer = !bvh && S.glass_defer_ok;  // (the scene has glass and the context holds parking areas: pt_api.hip)
        // walks: the nodes of a small tree are staged in LDS beside (smaller) walk queues when they fit with the candidate
        // records (mesh.json: 171 nodes, 10.9 KB: up to 24 pixels per stream).  Measured: no gain and no loss against the
        // gathers from L2 (26.74 / 26.72 G bounces/s) - a box-test batch waits for its turn at the SIMD, not for its node -
        // so streams are not shortened to make room for it
        bool nodes_lds = false;
        if (bvh && S.nodes_in_lds_ok) {
            DevScene S3 = S2;
            S3.bvh_in_lds |= 4u;
            nodes_lds = pass_lds_cand_offset(m, false) + pass_lds_cand_bytes() + pass_cand_bvh_bytes(S3) + rec_cand <= budget;
            if (nodes_lds) S2 = S3;
        }
        const size_t walk = bvh ? pass_cand_bvh_bytes(S2) : 0u;
        const size_t before = pass_lds_cand_offset(m, defer) + pass_lds_cand_bytes() + walk;
        S2.surf_staged = before + rec_cand + rec_surf <= budget ? 1u : 0u;
        // (without walks the records are staged whole or not at all; with walks the candidate records alone may be)
        const bool staged = bvh ? (S2.surf_staged || before + rec_cand <= budget + 8u * 1024u) : S2.surf_staged != 0u;
        // (walks, table too large: as many leading ranks as still fit - the objects visited first, the room of mesh.json)
        S2.surf_head = 0u;
        if (bvh && staged && !S2.surf_staged && before + rec_cand < budget) {
            const size_t fit = (budget - before - rec_cand) / sizeof(SurfRec);
            const size_t n_ranks = (size_t)S.n_objs + S.n_tris;
            S2.surf_head = (uint32_t)(fit < n_ranks ? fit : n_ranks);
        }
        size_t lds = before + (staged ? rec_cand + (S2.surf_staged ? rec_surf : (size_t)S2.surf_head * sizeof(SurfRec)) : 0u);
        if (bvh && getenv("PT_LDS_PAD")) lds += (size_t)atol(getenv("PT_LDS_PAD"));
        if (!bvh) return launch_pass_cand_flat(st, K, S2, F, q0, q1, cap, s0, s_here, m, acc, blk_rays, flags, lds, staged, defer);
        static bool said = false;
        if (getenv("PT_LDS_PAD") && !said) {
            said = true;
            fprintf(stderr, "k_pass_cand<BVH>: %zu bytes of LDS per workgroup (+ %ld of padding), m = %u, staged %d nodes_lds %d surf_head %u\n", lds, atol(getenv("PT_LDS_PAD")), m, (int)staged, (int)nodes_lds, S2.surf_head);
        }
        if (staged && nodes_lds)
            PT_LAUNCH_CAND(true, false, true, true);
        else if (staged)
            PT_LAUNCH_CAND(true, false, true, false);
        else
            PT_LAUNCH_CAND(false, false, true, false);
        return hipSuccess;
    }
    if (lds_defer <= 32u * 1024u) {
        if (F.probe)
            hipLaunchKernelGGL((k_pass<true, true>), dim3(K), dim3(kBlock), lds_defer, st, S, F, q0, q1, cap, s0, s_here, m, acc, blk_rays, flags);
        else
            hipLaunchKernelGGL((k_pass<true, false>), dim3(K), dim3(kBlock), lds_defer, st, S, F, q0, q1, cap, s0, s_here, m, acc, blk_rays, flags);
    } else {
        if (F.probe)
            hipLaunchKernelGGL((k_pass<false, true>), dim3(K), dim3(kBlock), lds_plain, st, S, F, q0, q1, cap, s0, s_here, m, acc, blk_rays, flags);
        else
            hipLaunchKernelGGL((k_pass<false, false>), dim3(K), dim3(kBlock), lds_plain, st, S, F, q0, q1, cap, s0, s_here, m, acc, blk_rays, flags);
    }
    return hipSuccess;
}
#endif  // PT_TU_FLAT
#undef PT_LAUNCH_CAND
#undef PT_LAUNCH_CAND2
#ifndef PT_TU_FLAT
void launch_pass_bvh(hipStream_t st, uint32_t K, const DevScene &S, const FrameParams &F, const RayQueue &q0,
                     const RayQueue &q1, uint32_t cap, uint32_t s0, uint32_t s_here, uint32_t m, unsigned long long *acc,
                     unsigned long long *blk_rays, uint32_t *flags) {
    if (F.probe)
        hipLaunchKernelGGL(k_pass_bvh<true>, dim3(K), dim3(kBlock), pass_bvh_lds_bytes(S, m), st, S, F, q0, q1, cap, s0, s_here, m,
                           acc, blk_rays, flags);
    else
        hipLaunchKernelGGL(k_pass_bvh<false>, dim3(K), dim3(kBlock), pass_bvh_lds_bytes(S, m), st, S, F, q0, q1, cap, s0, s_here, m,
                           acc, blk_rays, flags);
}
void launch_scatter_chunks(hipStream_t st, const float *src, float *dst, uint32_t npix, uint32_t C, uint32_t n,
                           uint32_t j) {
    hipLaunchKernelGGL(k_scatter_chunks, dim3((npix + kBlock - 1) / kBlock), dim3(kBlock), 0, st, src, dst, npix, C, n, j);
}
void launch_resolve(hipStream_t st, const unsigned long long *acc, float *out, uint32_t npix, uint32_t spp,
                    uint32_t n_streams, uint32_t m, bool clamp) {
    hipLaunchKernelGGL(k_resolve, dim3((npix + kBlock - 1) / kBlock), dim3(kBlock), 0, st, acc, out, npix, spp,
                       n_streams, m, clamp ? 1u : 0u);
}
void launch_mega(hipStream_t st, uint32_t grid, const DevScene &S, const FrameParams &F, unsigned long long *acc,
                 uint32_t s_begin, uint32_t s_end, uint32_t lane_spp, uint32_t n_split, unsigned long long *total_rays, char *stack_mem) {
    const size_t rec = ((size_t)S.n_cand_pairs * sizeof(CandPairRec) + 15) & ~(size_t)15;
    DevScene S2 = S;
    S2.bvh_in_lds &= ~5u;  // (the candidate forms read the nodes from global memory, with full-size walk queues)
    const size_t walk = S.n_bvh_nodes != 0u ? pass_cand_queues_bytes(S2) + (size_t)(kBlock / 64u) * kCandWalkKeyBytes : 0u;
    if (mega_uses_cand(S) && stack_mem) {  // the candidate scan, two paths per lane (k_mega_cand)
        const size_t rec_surf = (size_t)(S.n_objs + S.n_tris) * sizeof(SurfRec);
        const size_t base = intersect_cand_lds_bytes() + rec + walk;
        const size_t budget = 40u * 1024u;  // four workgroups per CU
        const uint32_t depth = base + mega_spare_bytes(4) + kMegaOwnerBytes <= budget ? 4u : 2u;
        const uint32_t spare_off = (uint32_t)base;
        const size_t after = base + mega_spare_bytes(depth) + kMegaOwnerBytes;
        const uint32_t surf_off = after + rec_surf <= budget ? (uint32_t)after : 0u;
        const size_t lds_c = after + (surf_off ? rec_surf : 0u);
        MegaStack stk;
        stk.lanes = grid * kBlock;
        stk.a = reinterpret_cast<float4 *>(stack_mem);
        stk.b = stk.a + (size_t)kMegaStackEntries * stk.lanes;
        stk.c = reinterpret_cast<float2 *>(stk.b + (size_t)kMegaStackEntries * stk.lanes);
        if (S.n_bvh_nodes != 0u && F.probe)
            hipLaunchKernelGGL((k_mega_cand<true, true>), dim3(grid), dim3(kBlock), lds_c, st, S2, F, acc, s_begin, s_end, lane_spp, n_split, total_rays, stk, spare_off, depth, surf_off);
        else if (S.n_bvh_nodes != 0u)
            hipLaunchKernelGGL((k_mega_cand<true, false>), dim3(grid), dim3(kBlock), lds_c, st, S2, F, acc, s_begin, s_end, lane_spp, n_split, total_rays, stk, spare_off, depth, surf_off);
        else if (F.probe)
            hipLaunchKernelGGL((k_mega_cand<false, true>), dim3(grid), dim3(kBlock), lds_c, st, S2, F, acc, s_begin, s_end, lane_spp, n_split, total_rays, stk, spare_off, depth, surf_off);
        else
            hipLaunchKernelGGL((k_mega_cand<false, false>), dim3(grid), dim3(kBlock), lds_c, st, S2, F, acc, s_begin, s_end, lane_spp, n_split, total_rays, stk, spare_off, depth, surf_off);
        return;
    }
    const size_t lds = S.n_bvh_nodes != 0u ? bvh_lds_bytes(S, kBlock) : 0u;
    if (S.n_bvh_nodes != 0u && F.probe)
        hipLaunchKernelGGL((k_mega<true, true>), dim3(grid), dim3(kBlock), lds, st, S, F, acc, s_begin, s_end, lane_spp, n_split, total_rays);
    else if (S.n_bvh_nodes != 0u)
        hipLaunchKernelGGL((k_mega<true, false>), dim3(grid), dim3(kBlock), lds, st, S, F, acc, s_begin, s_end, lane_spp, n_split, total_rays);
    else if (F.probe)
        hipLaunchKernelGGL((k_mega<false, true>), dim3(grid), dim3(kBlock), lds, st, S, F, acc, s_begin, s_end, lane_spp, n_split, total_rays);
    else
        hipLaunchKernelGGL((k_mega<false, false>), dim3(grid), dim3(kBlock), lds, st, S, F, acc, s_begin, s_end, lane_spp, n_split, total_rays);
}
// does the megakernel run the candidate scan for this scene (and need the split stacks: mega_stack_bytes)?
bool mega_uses_cand(const DevScene &S) {
    const size_t rec = ((size_t)S.n_cand_pairs * sizeof(CandPairRec) + 15) & ~(size_t)15;
    DevScene S2 = S;
    S2.bvh_in_lds &= ~5u;
    const size_t walk = S.n_bvh_nodes != 0u ? pass_cand_queues_bytes(S2) + (size_t)(kBlock / 64u) * kCandWalkKeyBytes : 0u;
    return S.cand_scan && intersect_cand_lds_bytes() + rec + walk + mega_spare_bytes(2) + kMegaOwnerBytes <= 40u * 1024u;
}
size_t mega_stack_mem_bytes(uint32_t grid) { return mega_stack_bytes(grid * kBlock); }
void launch_query(hipStream_t st, const DevScene &S, const float *o, const float *d, uint32_t n, float *t,
                  int32_t *object_id, int32_t *tri_id, float *x, float *nrm) {
    uint32_t grid = (n + kBlock - 1) / kBlock;
    if (grid > 4096u) grid = 4096u;
    if (grid == 0u) grid = 1u;
    hipLaunchKernelGGL(k_query, dim3(grid), dim3(kBlock), bvh_lds_bytes(S, kBlock), st, S, o, d, n, t, object_id, tri_id, x, nrm);
}
void launch_bounds(hipStream_t st, const DevScene &S, const TriPairRec *boxes, uint32_t mode, uint32_t object, const float *o,
                   const float *d, uint32_t n, int32_t *hit, float *t, float *x, float *nrm, int32_t *object_id) {
    uint32_t grid = (n + kBlock - 1) / kBlock;
    if (grid > 4096u) grid = 4096u;
    if (grid == 0u) grid = 1u;
    hipLaunchKernelGGL(k_bounds, dim3(grid), dim3(kBlock), 0, st, S, boxes, mode, object, o, d, n, hit, t, x, nrm, object_id);
}
void launch_numerics_sweep(hipStream_t st, unsigned long long *out4) {
    hipLaunchKernelGGL(k_numerics_sweep, dim3(8192), dim3(256), 0, st, out4);
}
void launch_numerics(hipStream_t st, const float *in, uint32_t n, float *out_sin, float *out_cos, float *out_sqrt,
                     float *out_rcp, uint32_t *out_philox) {
    hipLaunchKernelGGL(k_numerics, dim3((n + 255u) / 256u), dim3(256), 0, st, in, n, out_sin, out_cos, out_sqrt,
                       out_rcp, out_philox);
}

#endif  // PT_TU_FLAT
}  // namespace pt
