// pt_kernels.h — launch interface between the C-ABI layer (pt_api.hip) and the kernels (pt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "pt_device.h"

namespace pt {

constexpr uint32_t kBlockBvh = 256;  // workgroup of the intersect kernel of scenes with a BVH
// waves per SIMD (= workgroups per compute unit) k_pass_cand is compiled for, without walks (pt_kernels_flat.hip) and with: its
// __launch_bounds__, launch_pass's LDS budget per workgroup and plan_pass's rounds of resident workgroups follow them
#ifndef PT_CAND_WAVES
#define PT_CAND_WAVES 5
#endif
#ifndef PT_ISECT_WAVES
#define PT_ISECT_WAVES 4  // k_intersect_cand (the flat unit's too)
#endif
#ifndef PT_ISECT_PREFETCH
#define PT_ISECT_PREFETCH 1  // k_intersect_cand loads the next chunk's rays a trip ahead (0.318 -> 0.324 of the HBM peak, 74 VGPRs: six waves)
#endif
#ifndef PT_MEGA_WAVES
#define PT_MEGA_WAVES 4  // k_mega_cand without walks
#endif
#ifndef PT_SAMPLE_MAJOR
#define PT_SAMPLE_MAJOR 1  // k_pass_cand: a trip's 64 primary rays are consecutive samples of one pixel (0: one sample of 64 pixels)
#endif
#ifndef PT_CAND_BVH_WAVES
#define PT_CAND_BVH_WAVES 4
#endif
constexpr uint32_t kLevels = 13;    // ray depths 0..11 plus the (always empty) level written by the last shade
// (kBlock, kMaxStreamPixels, kRayBytes, the wave-stack sizes and queue_bytes: pt_device.h - the pass planner of pt_host.cpp
// needs them without the HIP headers)

// The ray queue: K stream slices of cap * 40 bytes, each slice three arrays of `cap` slots one after the other -
//   [od0: float4 origin xyz, direction x][tp: float4 throughput rgb, bookkeeping word (pack_word)][od1: float2 direction yz]
// - so that a workgroup addresses its stream with ONE scalar base and 32-bit byte offsets (see StreamSlice, pt_kernels.hip).
struct RayQueue {
    char *buf;
};


// one whole pass of a scene without BVH meshes in one launch (see k_pass)
hipError_t launch_pass(hipStream_t st, uint32_t K, const DevScene &S, const FrameParams &F, const RayQueue &q0,
                 const RayQueue &q1, uint32_t cap, uint32_t s0, uint32_t s_here, uint32_t m, unsigned long long *acc,
                 unsigned long long *blk_rays, uint32_t *flags);
// the same for scenes with BVH meshes (nodes read from global memory: DevScene.bvh_in_lds bit 0 clear)
void launch_pass_bvh(hipStream_t st, uint32_t K, const DevScene &S, const FrameParams &F, const RayQueue &q0,
                     const RayQueue &q1, uint32_t cap, uint32_t s0, uint32_t s_here, uint32_t m, unsigned long long *acc,
                     unsigned long long *blk_rays, uint32_t *flags);
void launch_generate(hipStream_t st, uint32_t K, const FrameParams &F, const RayQueue &q, uint32_t *cnt0,
                     uint32_t cap, uint32_t s0, uint32_t s_here, uint32_t m);
void launch_intersect_cand(hipStream_t st, uint32_t K, const DevScene &S, const RayQueue &q, float2 *hit, const uint32_t *cnt,
                           uint32_t cap, unsigned long long *blk_rays);  // (pt_kernels_flat.hip)
void launch_intersect(hipStream_t st, uint32_t K, const DevScene &S, const RayQueue &q, float2 *hit,
                      const uint32_t *cnt, uint32_t cap, unsigned long long *blk_rays);
void launch_shade(hipStream_t st, uint32_t K, const DevScene &S, const FrameParams &F, const RayQueue &qin,
                  const RayQueue &qout, const float2 *hit, const uint32_t *cnt_in, uint32_t *cnt_out, uint32_t cap,
                  unsigned long long *acc, uint32_t *flags, uint32_t m, uint32_t s0);
void launch_scatter_chunks(hipStream_t st, const float *src, float *dst, uint32_t npix, uint32_t C, uint32_t n,
                           uint32_t j);
// acc is stream-major: pixel p = slot (p % n_streams) * m + p / n_streams of each colour plane (megakernel: 1, npix)
void launch_resolve(hipStream_t st, const unsigned long long *acc, float *out, uint32_t npix, uint32_t spp,
                    uint32_t n_streams, uint32_t m, bool clamp = true);
// one round of the megakernel: samples [s_begin, s_end) of every pixel, n_split lanes of lane_spp samples per pixel
// (total_rays: [0] the ray counter, [1] an overflow flag of k_mega_cand's split stacks, [7] its item counter, which the caller
// zeroes before every launch; stack_mem: mega_stack_mem_bytes(grid) bytes for the split stacks when mega_uses_cand(S), else unused)
void launch_mega(hipStream_t st, uint32_t grid, const DevScene &S, const FrameParams &F, unsigned long long *acc,
                 uint32_t s_begin, uint32_t s_end, uint32_t lane_spp, uint32_t n_split, unsigned long long *total_rays, char *stack_mem);
bool mega_uses_cand(const DevScene &S);
size_t mega_stack_mem_bytes(uint32_t grid);
void launch_query(hipStream_t st, const DevScene &S, const float *o, const float *d, uint32_t n, float *t,
                  int32_t *object_id, int32_t *tri_id, float *x, float *nrm);
// mode 0: intersect_bounds of one object; mode 1: get_orbit_point (see k_bounds)
void launch_bounds(hipStream_t st, const DevScene &S, const TriPairRec *boxes, uint32_t mode, uint32_t object, const float *o,
                   const float *d, uint32_t n, int32_t *hit, float *t, float *x, float *nrm, int32_t *object_id);
void launch_numerics_sweep(hipStream_t st, unsigned long long *out4);
// sincos_f32 on all 2^24 reachable arguments against host tables (bit patterns); out2 = {mismatches, compared}
void launch_sincos_sweep(hipStream_t st, const uint32_t *want_sin, const uint32_t *want_cos, unsigned long long *out2);
// render_pixel's per-sample ray for n (framebuffer index, sample) pairs; form 0: primary_ray, 1: primary_ray_at
void launch_primary_rays(hipStream_t st, const FrameParams &F, const uint32_t *pixel, const uint32_t *sample, uint32_t n, uint32_t form,
                         float *o, float *d);
void launch_numerics(hipStream_t st, const float *in, uint32_t n, float *out_sin, float *out_cos, float *out_sqrt,
                     float *out_rcp, uint32_t *out_philox);

}  // namespace pt
