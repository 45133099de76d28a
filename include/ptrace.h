/*
 * ptrace.h — C ABI of libptrace_hip.so, the MI355X (gfx950) implementation of the
 * per-pixel radiance() path-tracing loop of filippo-orru/path-tracer-rust.
 *
 * The reference has no FFI/plugin interface (it is safe Rust only).  The narrowest seam is
 * the parallel section of render() — src/render/mod.rs:1017-1024 — which fills
 * `pixels: Vec<Vec3>` (index (H-1-y)*W+x, mod.rs:805-806) for a RenderConfig
 * (mod.rs:859-864) under a cancel flag (mod.rs:943,1003) and a progress counter
 * (mod.rs:960,850).  pt_render() replaces exactly that loop; everything in this header is
 * what a Rust `extern "C"` block for that seam would bind (INTEGRATION.md shows the shim).
 *
 * Conventions: plain pointers and sizes, caller owns every buffer, no unwinding, every
 * entry point returns an int status (PT_OK or a negative PT_ERR_*), message through
 * pt_last_error().  There is no CPU fallback: without a HIP device every compute entry
 * point fails with PT_ERR_NO_DEVICE.
 */
#ifndef PTRACE_H
#define PTRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 5

/* status codes (reference behaviour: unwrap() panics, mod.rs:96,309,1032,1042,1093) */
#define PT_OK 0
#define PT_ERR_INVALID (-1)    /* bad argument / malformed scene */
#define PT_ERR_NO_DEVICE (-2)  /* no HIP device (the product never falls back to the CPU) */
#define PT_ERR_HIP (-3)        /* a HIP runtime call failed; text in pt_last_error() */
#define PT_CANCELLED (-4)      /* *cancel became non-zero; framebuffer = the samples accumulated so far / their count */
#define PT_ERR_OVERFLOW (-5)   /* a ray queue overflowed (cannot happen with the sizes pt_render picks) */
#define PT_ERR_IO (-6)         /* file could not be read / written */
#define PT_ERR_PARSE (-7)      /* scene JSON / OFF syntax or shape error */
#define PT_ERR_COMM (-8)       /* an RCCL call failed (pt_comm_*); text in pt_last_error() */

/* ReflectType — enum order of src/render/mod.rs:71-76 */
#define PT_DIFFUSE 0u
#define PT_SPECULAR 1u
#define PT_REFRACT 2u

/* SceneObject kind — src/render/mod.rs:326-335 */
#define PT_SPHERE 0u
#define PT_MESH 1u

/* backends of the hot path (both run on the GPU) */
#define PT_BACKEND_WAVEFRONT 0u /* SoA ray queues in HBM, generate/intersect/shade kernels per bounce */
#define PT_BACKEND_MEGAKERNEL 1u /* persistent threads: whole render_pixel loop per lane (two interleaved paths), no ray queues */

/* pt_config.flags */
#define PT_FLAG_NO_BVH 1u /* meshes are scanned triangle by triangle as the reference does (mod.rs:558) */
/* Wavefront backend: generate / intersect / shade as separate kernels per depth even where a whole pass could run as
 * one launch (scenes without BVH meshes).  Same image, bit for bit; for A/B checks and per-step profiling. */
#define PT_FLAG_SEPARATE_KERNELS 2u
/* Concurrent pipelines (wavefront backend): bits 8..11 of flags = n (2..8).  The call's pixels are dealt chunk by
 * chunk to n independent wavefront pipelines on n HIP streams of the same GPU, so the VALU-bound intersect kernels
 * of one pipeline overlap the HBM-bound shade kernels of another (cornell: +15 % with 2; 3 or more only help when
 * the runtime exposes enough hardware queues, GPU_MAX_HW_QUEUES=8).  Same
 * image, bit for bit.  Per-kernel timings (pt_stats.ms_intersect, rocprof) then describe kernels that share the
 * machine, which is why it is opt-in: the default single pipeline keeps per-kernel roofline numbers meaningful. */
#define PT_FLAG_PIPELINES(n) (((uint32_t)(n) & 15u) << 8)

/* CameraData — src/render/mod.rs:162-176.  `direction` is used as stored (not renormalised). */
typedef struct pt_camera {
    float position[3];
    float direction[3];
    float focal_length;
    float sensor_width;
    float aspect_ratio;
} pt_camera;

/* Triangle — src/render/mod.rs:538-543, object-local vertices. */
typedef struct pt_triangle {
    float a[3];
    float b[3];
    float c[3];
} pt_triangle;

/* SceneObjectData + Material flattened — src/render/mod.rs:253-258, 78-83, 326-335, 440-448.
 * For PT_MESH, triangles [tri_offset, tri_offset+tri_count) of the triangle array belong to the
 * object and (bs_center, bs_radius) is Mesh.bounding_sphere exactly as stored/computed
 * (object-local centre; mod.rs:268 adds `position`). */
typedef struct pt_object {
    uint32_t kind;
    float position[3];
    float radius; /* PT_SPHERE only */
    float color[3];
    float emission[3]; /* the reference spells it `emmission` */
    uint32_t reflect_type;
    uint32_t tri_offset;
    uint32_t tri_count;
    float bs_center[3];
    float bs_radius;
} pt_object;

/* RenderConfig + Resolution (src/render/mod.rs:859-870) plus what the GPU path needs. */
typedef struct pt_config {
    uint32_t width;
    uint32_t height;
    uint32_t spp;      /* samples_per_pixel */
    uint32_t backend;  /* PT_BACKEND_* */
    uint64_t seed;     /* key of the counter-based RNG that stands in for rand::random (mod.rs:53) */
    uint32_t idx_begin; /* framebuffer-index band [idx_begin, idx_end) to render; 0,0 = whole frame */
    uint32_t idx_end;
    uint32_t rays_per_pass; /* wavefront: primary rays per pass (megakernel: primary samples per round); 0 = the library's own:
                             * passes sized by measured time, at most 512 Mi primary rays - cancel and progress are looked
                             * at between passes */
    uint32_t flags;
    /* Interleaved partition of the band for load balance across ranks (the cost of a pixel varies over the
     * image: contiguous eighths of cornell.json differ by up to 1.31x).  The band is cut into chunks of
     * chunk_pixels framebuffer indices; this call renders chunks chunk_first, chunk_first+chunk_step, ... and
     * writes them back to back into the output.  chunk_step = 0 or 1: the whole band (the other two ignored). */
    uint32_t chunk_pixels;
    uint32_t chunk_first;
    uint32_t chunk_step;
    /* Minimum interval between two progress callbacks, in milliseconds.  0 = 500, the reference's RenderUpdate cadence
     * (mod.rs:965-982); PT_PROGRESS_EVERY_PASS = at every pass boundary (a few milliseconds apart).  The cancel byte is
     * read at every pass boundary whatever this says (the reference polls it every 100 ms, mod.rs:947-958). */
    uint32_t progress_ms;
} pt_config;
#define PT_PROGRESS_EVERY_PASS 0xffffffffu

typedef struct pt_stats {
    uint64_t ray_bounces;        /* number of intersect_scene evaluations (mod.rs:663), exact */
    uint64_t samples;            /* primary samples traced */
    uint64_t intersect_rays;     /* rays processed by the dominant kernel (== ray_bounces for wavefront) */
    uint32_t intersect_launches; /* launches of the dominant kernel: k_pass (one per pass) or, for BVH scenes, k_intersect */
    uint32_t passes;
    double ms_total;     /* wall time of the call */
    double ms_device;    /* HIP-event time from first to last kernel of the call */
    double ms_intersect; /* HIP-event time summed over those launches (only if PT profiling on) */
} pt_stats;

/* Progress callback: fraction in [0,1], between passes (and between the parts of a very large call), at most every
 * pt_config.progress_ms, and once with 1.0 when the frame is complete and in the output buffer.  pt_ctx_render invokes it
 * on the calling thread; pt_render_multi and PT_FLAG_PIPELINES render on worker threads and invoke it from the worker of
 * rank / pipeline 0 - a GUI host has to marshal it - except for the final 1.0, which comes from the calling thread after
 * every rank / pipeline has finished.  It may raise the
 * cancel byte: the render then stops at that boundary.  It may call pt_ctx_snapshot on the context it was given to
 * (not under PT_FLAG_PIPELINES, where the accumulators live in child contexts: the snapshot reports an error). */
typedef void (*pt_progress_fn)(void *user, float fraction);

typedef struct pt_ctx pt_ctx;

const char *pt_version(void);
/* the back-end (-mllvm) switches the library was built with, "<general set> | flat: <set of the pass kernel without walks>": the
 * Makefile probes each against the compiler and drops the ones it rejects (they only steer instruction placement: same images
 * with any subset) */
const char *pt_build_flags(void);
/* the first 16 hex digits of sha256 over the device assembly the library's kernels were built from (Makefile: pt_kernels.s of
 * the same compile): the _traffic.json files under profiles/ name the hash of the library they were measured on, bench.py compares */
const char *pt_kernel_isa_hash(void);
const char *pt_last_error(void);
int pt_abi_version(void);
int pt_device_count(void);

/* a3 — CameraData::{lens_center, orthogonals} (mod.rs:211-232), host arithmetic in f32. */
int pt_camera_basis(const pt_camera *cam, float lens_center[3], float su[3], float sv[3]);

/* a10 — Mesh::new bounding sphere (mod.rs:450-499), including its `min + max*0.5` centre. */
int pt_mesh_bounding_sphere(const pt_triangle *tris, uint32_t n_tris, float center[3], float *radius);

/* One context = one GPU, one stream, device copies of one scene and the ray queues. */
int pt_ctx_create(int device, pt_ctx **out);
void pt_ctx_destroy(pt_ctx *ctx);
int pt_ctx_set_scene(pt_ctx *ctx, const pt_camera *cam, const pt_object *objs, uint32_t n_objs,
                     const pt_triangle *tris, uint32_t n_tris);

/* Number of pixels a call with this config renders (the band, or this rank's chunks of it); 0 on a bad config. */
uint32_t pt_config_pixels(const pt_config *cfg);

/* Render the band [idx_begin, idx_end) (or this rank's chunks of it) into DEVICE memory: d_out_rgb holds
 * pt_config_pixels(cfg)*3 floats, pixel k of the call at element k*3+c, linear, clamped to [0,1] — for an
 * un-chunked band the memory image of the reference's Vec<Vec3> slice (mod.rs:1013-1014, 852-856).
 * `hip_stream` is a hipStream_t (NULL = the context's own stream).  Blocking.
 * Cancel (both backends): *cancel is read between passes (wavefront) / rounds (megakernel), which the library sizes by MEASURED
 * time when rays_per_pass is 0 - a tiny timed first pass of a scene, then as many samples as fit 100-120 ms, the rate kept with
 * the context - so a cancel comes back within about a tenth of a second whatever a ray of the scene costs (the reference polls
 * its flag every 100 ms, mod.rs:947-958); with an explicit rays_per_pass a pass is as long as asked for.  On PT_CANCELLED the
 * buffer holds every pixel averaged over the samples that were accumulated (stats->samples / pixels of the call), all zero if
 * none - the picture pt_ctx_snapshot would have given. */
int pt_ctx_render(pt_ctx *ctx, const pt_config *cfg, void *d_out_rgb, void *hip_stream,
                  const volatile uint8_t *cancel, pt_progress_fn cb, void *user, pt_stats *stats);

/* Device buffers for hosts that have no GPU allocator of their own (a Rust/C host driving pt_ctx_render;
 * tests).  Plain hipMalloc / hipFree / hipMemcpy on `device`. */
int pt_device_malloc(int device, size_t bytes, void **out);
int pt_device_free(int device, void *p);
int pt_device_download(int device, void *dst_host, const void *src_device, size_t bytes);

/* Size limit of one scene's BVHs (pt_ctx_set_scene fails with PT_ERR_INVALID beyond it): the walkers pack a node index or a
 * leaf code (first pair record << 1 | records - 1: a leaf is one or two pair records) into 26 bits of a queue entry - 2^26
 * nodes, 2^25 pair records (two triangles each) over all BVH meshes of the scene.  1 = fits. */
int pt_bvh_refs_fit(uint64_t n_bvh_nodes, uint64_t n_pair_records);

/* Device memory the wavefront backend may take for its ray queues in this context (bytes; 0 = the default: 85 % of what
 * the device reports free, divided among the contexts one call creates on it).  The default kernel (k_pass_cand) keeps
 * each wave's waiting rays on a stack of at most 1024 slots: K streams x 4 waves x 40 KB, whatever the pass holds (5.9 GB
 * for the 35 747 streams of a 1024x768 pass of 683 samples; small passes need less); the level-by-level forms (PT_FLAG_SEPARATE_KERNELS,
 * PT_FLAG_NO_BVH, PT_CAND_SCAN=0) hold rays_per_pass primary rays at 352 B each, 36 GB at their default.  A pass that does
 * not fit is halved until it does (a failed allocation does the same), which changes how the samples are batched and
 * nothing in the image.  A figure set here also bounds an explicit pt_config.rays_per_pass (the budget wins); without one
 * an explicit rays_per_pass is taken as given and only a failed allocation halves it. */
int pt_ctx_set_memory_budget(pt_ctx *ctx, size_t bytes);

/* Enable HIP-event timing of every launch of the dominant kernel (fills pt_stats.ms_intersect). */
int pt_ctx_set_profiling(pt_ctx *ctx, int enabled);
/* Name of the kernel the wavefront backend launches for this context's scene with these pt_config.flags - the one
 * pt_stats.intersect_launches / ms_intersect describe: "k_pass_cand" (one launch per pass, candidate scan: scenes
 * without BVH meshes), "k_pass_cand_bvh" (the same kernel's form for scenes with BVH meshes: candidate scan + parked
 * walks), "k_pass" (every triangle tested per ray), "k_pass_bvh" (scan + depth-first parked walks: PT_CAND_BVH=0),
 * or, with separate kernels, "k_intersect_cand" (the stand-alone intersect step with the candidate scan: scenes without BVH
 * meshes) / "k_intersect" (every triangle per ray, or scan + walks for BVH scenes).  For profilers and bench.py; NULL without
 * a scene. */
const char *pt_ctx_pass_kernel(const pt_ctx *ctx, uint32_t flags);

/* radiance(&ray, depth, &scene) (mod.rs:661-792) for ONE given ray, averaged over n_samples independent evaluations:
 * what the reference's own test_radiance does (src/render/test.rs:146-183: the sum of 10 000 calls with depth 0 divided
 * by their number), and the entry point through which hand-derived cases reach roulette / emission / specular / refract /
 * Fresnel on the device (tests/kats_shading.py; `depth` 2 and 5 put the first hit behind mod.rs:760 and mod.rs:677).
 * Sample i draws from the RNG stream (seed; counter = pixel, i, ...) exactly as sample i of framebuffer index `pixel`
 * would - `pixel` is only that counter, no frame is involved - so the oracle's pto_radiance_mean_at with the same
 * arguments walks the same paths.  depth < 12 (MAX_DEPTH, mod.rs:661).  backend / flags as in pt_config (PT_FLAG_PIPELINES
 * is refused).  out_rgb = the mean, NOT clamped (the reference clamps in render_pixel, mod.rs:852-856, not in radiance);
 * stats->ray_bounces = intersect_scene evaluations, exact.  Host pointers; blocking. */
int pt_ctx_radiance(pt_ctx *ctx, const float o[3], const float d[3], uint32_t depth, uint32_t n_samples, uint64_t seed,
                    uint32_t pixel, uint32_t backend, uint32_t flags, float out_rgb[3], pt_stats *stats);

/* Single-ray queries through the same device intersection code (a6): the callers are object
 * picking / click-debug / orbit pivot (src/views/viewport_tab.rs:240-246, render_tab.rs:177-205).
 * Host arrays: o,d = n*3 floats; outputs may be NULL.  object_id = -1 on a miss
 * (intersect_scene -> None), tri_id = index into the object's triangle list or -1 for spheres.
 * Bit-for-bit agreement with the reference's intersect_scene is guaranteed for rays like the path tracer's own:
 * direction of unit length (the reference normalises every direction it casts) and origin where its rays start - inside the bounding box of the scene's objects and of the camera given to
 * pt_ctx_set_scene: the error bounds behind the BVH boxes and the bounding-sphere shortcuts assume that distance
 * scale (pt_host.cpp).  A picking ray from a camera far outside the scene still gets the nearest hit, but a hit that
 * only exists through f32 round-off of the reference's arithmetic at that distance may be resolved differently. */
int pt_ctx_intersect(pt_ctx *ctx, const float *o, const float *d, uint32_t n, float *t,
                     int32_t *object_id, int32_t *tri_id, float *x, float *normal);

/* Diagnostics: intersect_scene (mod.rs:631-659) for n rays through the INTERSECT STEP OF THE WAVEFRONT PIPELINE itself - the
 * rays are laid out as ray streams and run through the kernel PT_FLAG_SEPARATE_KERNELS launches per depth (candidate scan:
 * conservative filters, per-wave ring, dense exact batches; with PT_FLAG_NO_BVH every triangle per ray; scenes with BVH
 * meshes: scan + parked walks) - where pt_ctx_intersect above goes through the single-ray query kernel.  t[i] = the hit
 * distance (+inf on a miss), id[i] = -1 (miss), the object index of a sphere, or n_objs + the flattened triangle index.
 * For ray-by-ray parity tests of the scan forms (rays that start ON a triangle included). */
int pt_ctx_intersect_streams(pt_ctx *ctx, const float *o, const float *d, uint32_t n, uint32_t flags, float *t, int32_t *id);

/* SceneObjectData::intersect_bounds (mod.rs:282-290) of object `object` for n rays: a sphere is tested itself
 * (intersect_sphere), a mesh through Triangle::intersect over the 12 triangles of Mesh.bounding_box.  hit[i] = 1/0;
 * t / x / normal as Hit holds them (zero on a miss).  Outputs may be NULL. */
int pt_ctx_intersect_bounds(pt_ctx *ctx, uint32_t object, const float *o, const float *d, uint32_t n, int32_t *hit,
                            float *t, float *x, float *normal);
/* get_orbit_point (src/views/viewport_tab.rs:401-431): objects from the last to the first; an object whose bounds are
 * hit contributes its real hit if it has one, else the bounds hit; the nearest (strict <) wins.  found[i] = 1/0, point =
 * hit.intersection, object_id = the object that supplied it (-1), t = its distance. */
int pt_ctx_orbit_point(pt_ctx *ctx, const float *o, const float *d, uint32_t n, int32_t *found, float *point,
                       int32_t *object_id, float *t);
/* Mesh.bounding_box of a mesh object (12 object-local triangles).  pt_ctx_set_scene computes it as Mesh::new does
 * (mod.rs:452-476, 501-536); an inline Mesh of a scene file carries its own (deserialised verbatim, mod.rs:440-448) -
 * pass that one here (pt_scene_bounding_box) when it may differ. */
int pt_ctx_set_mesh_bounds(pt_ctx *ctx, uint32_t object, const pt_triangle box[12]);
/* bounding_box_to_triangles over the AABB of the triangles, as Mesh::new stores it (mod.rs:452-476, 501-536) */
int pt_mesh_bounding_box(const pt_triangle *tris, uint32_t n_tris, pt_triangle out[12]);

/* Diagnostics: evaluate the device's numerics contract (sin, cos, sqrt, 1/x on in[i]; Philox block for
 * counter (i, bits(in[i]), (i<<8)|(i&15), 0), key 0x0123456789abcdef) so tests can compare it bit for bit
 * with the host.  Host arrays of n (out_philox: 4n). */
int pt_ctx_numerics_probe(pt_ctx *ctx, const float *in, uint32_t n, float *out_sin, float *out_cos,
                          float *out_sqrt, float *out_rcp, uint32_t *out_philox);

/* Diagnostics, exhaustive: the device's f_sqrt against the compiler's IEEE square root on all 2^32 binary32 bit patterns
 * and its f_rcp against IEEE 1/d on every normal d with 2^-126 <= |d| <= 2^126 (the domain its callers keep to).
 * out[0], out[1] = inputs whose results differ in bits (NaN == NaN) - both must be 0; out[2], out[3] = inputs compared.
 * About a second of GPU time. */
int pt_ctx_numerics_sweep(pt_ctx *ctx, uint64_t out[4]);

/* Diagnostics, exhaustive: the device's sincos_f32 - the one transcendental of the path, cos / sin of r1 = 2 pi rand01()
 * in the diffuse bounce (mod.rs:691,703) - on ALL 2^24 arguments that expression can take (rand01() = k * 2^-24, rand
 * 0.8.5's f32 mapping) against the host instantiation of the same source (pt_host_sincos, which tests/test_abi.py holds to
 * the platform libm on the same arguments).  out[0] = arguments whose sine or cosine differs in bits (must be 0),
 * out[1] = arguments compared (2^24).  Uploads two 64 MB tables; well under a second of GPU time. */
int pt_ctx_sincos_sweep(pt_ctx *ctx, uint64_t out[2]);

/* Diagnostics: the per-sample part of render_pixel (mod.rs:805-843) - y = H-1 - idx/W, x = idx%W, the (s%2, (s/2)%2)
 * sub-pixel, the tent filter of two rand01() draws, sx / sy, sensor_pos = (position + su*sx) + sv*sy, direction =
 * (lens_center - sensor_pos).normalize(), origin = lens_center - ON THE DEVICE, through the functions the frame kernels
 * call, for n (framebuffer index, sample) pairs of a width x height frame with the context's camera; the two draws are
 * words 0 and 1 of the RNG block (seed; pixel, sample, tag 0).  form 0: as k_generate / k_mega / k_pass make it (column and
 * row by division), form 1: as k_pass_cand makes it (column and row handed in).  The oracle's pto_primary_ray is the
 * counterpart; tests/kats_camera.py holds both to an independent restatement, bit for bit.  Host arrays: o, d = n*3. */
int pt_ctx_primary_rays(pt_ctx *ctx, uint32_t width, uint32_t height, uint64_t seed, const uint32_t *pixel,
                        const uint32_t *sample, uint32_t n, uint32_t form, float *o, float *d);

/* The host instantiation of the shared numerics header's sincos (the same source the kernels compile). */
void pt_host_sincos(float y, float *s, float *c);

/* The drop-in for mod.rs:1017-1024: host buffers in, host framebuffer out (whole W*H*3 floats,
 * only the band is written).  Uses device 0 (or PT_DEVICE env).  Blocking. */
int pt_render(const pt_config *cfg, const pt_camera *cam, const pt_object *objs, uint32_t n_objs,
              const pt_triangle *tris, uint32_t n_tris, float *out_rgb,
              const volatile uint8_t *cancel, pt_progress_fn cb, void *user, pt_stats *stats);

/* The same frame dealt to n_ranks ranks row by row - rank r renders image rows r, r+n_ranks, ... of the band (the
 * interleaved partition of pt_config.chunk_*: contiguous parts of a picture differ in cost) - one host thread and one
 * context per rank, rank r on device r mod pt_device_count(): single-process multi-GPU for hosts that want the image
 * in HOST memory (the CLI).  Each rank downloads its rows and copies them to their places in out_rgb, so no
 * device-to-device collective is involved; a host that keeps the framebuffer on the GPUs runs one process per GPU over
 * pt_ctx_render and gathers with pt_comm_gather_frame (RCCL) below.  The image is bit-identical for every n_ranks.
 * The progress callback comes from rank 0's worker thread. */
int pt_render_multi(const pt_config *cfg, uint32_t n_ranks, const pt_camera *cam, const pt_object *objs,
                    uint32_t n_objs, const pt_triangle *tris, uint32_t n_tris, float *out_rgb,
                    const volatile uint8_t *cancel, pt_progress_fn cb, void *user, pt_stats *stats);

/* Progressive preview (RenderUpdate, mod.rs:881-885, sent every 500 ms by mod.rs:965-982): callable from the
 * progress callback of pt_ctx_render on the same thread.  Resolves what has been accumulated so far into
 * d_out_rgb (the call's layout, pt_config_pixels(cfg)*3 floats) and reports how many samples per pixel it holds.  The
 * reference's snapshot is a random subset of finished pixels; this one is every pixel at partial spp.  (A wavefront call
 * of more than 1.5 M pixels is rendered in parts of 2^20 pixels: the snapshot then shows the finished parts final, the
 * part in progress at partial spp - spp_done speaks of that part - and the parts not started black.) */
int pt_ctx_snapshot(pt_ctx *ctx, void *d_out_rgb, uint32_t *spp_done);

/* ---- the one collective of the path: the framebuffer gather over RCCL (xGMI) ---------------------------------
 * One process (or thread) per GPU renders its rows with pt_ctx_render (chunk_first = rank, chunk_step = n_ranks) into
 * device memory; pt_comm_gather_frame then gives EVERY rank the whole frame in device memory: one in-place
 * ncclAllGather of the rank buffers (padded to the largest) and one kernel that puts the rows back in framebuffer order.
 * librccl is loaded on first use (dlopen; PT_RCCL_LIB overrides the name), so hosts that never gather do not need it. */
typedef struct pt_comm pt_comm;
#define PT_COMM_ID_BYTES 128
/* rank 0: a fresh id (ncclGetUniqueId); the host carries the bytes to the other ranks by its own means */
int pt_comm_unique_id(uint8_t id[PT_COMM_ID_BYTES]);
/* every rank, collectively: ncclCommInitRank on `device` */
int pt_comm_create(int device, int rank, int n_ranks, const uint8_t id[PT_COMM_ID_BYTES], pt_comm **out);
void pt_comm_destroy(pt_comm *comm);
/* cfg: the frame (width, height, band) and the chunk size the ranks rendered with (chunk_pixels; chunk_first /
 * chunk_step are taken from the communicator).  d_local_rgb: this rank's pt_ctx_render output; d_frame_rgb: band
 * pixels * 3 floats on this rank's device.  `hip_stream`: NULL = the communicator's own stream.  Blocking. */
int pt_comm_gather_frame(pt_comm *comm, const pt_config *cfg, const void *d_local_rgb, void *d_frame_rgb, void *hip_stream);

/* Image.hash (mod.rs:897-926): Rust's DefaultHasher (SipHash-1-3, zero key) over the f32 bit patterns of the
 * pixels in order; the GUI uses it to invalidate its canvas cache (src/views/render_tab.rs:248-256). */
uint64_t pt_image_hash(const float *rgb, size_t n_floats);
/* the underlying SipHash-c-d (k0, k1 = key) so that the implementation can be pinned on the published
 * SipHash-2-4 test vector */
uint64_t pt_siphash(uint32_t c_rounds, uint32_t d_rounds, uint64_t k0, uint64_t k1, const uint8_t *data, size_t len);

/* ---- formats either side of the path (host only, no GPU needed) ------------------------- */

typedef struct pt_scene pt_scene;

/* SceneDescriptor::load + to_data (mod.rs:92-110, 304-318) and load_off (load_off.rs:8-85).
 * `path` is the JSON file; MeshFile paths are resolved against `base_dir` (the reference
 * resolves them against the process CWD; pass "." for that behaviour). */
int pt_scene_load(const char *path, const char *base_dir, pt_scene **out);
/* flags = PT_LOAD_TRIANGULATE additionally accepts OFF faces with more than 3 vertices (meshes/hdodec.off has
 * pentagons) and fan-triangulates them.  An extension: the reference's load_off rejects such files
 * (load_off.rs:73-76), so there is no reference behaviour to match beyond "a triangle stays a triangle". */
#define PT_LOAD_TRIANGULATE 1u
int pt_scene_load_ex(const char *path, const char *base_dir, uint32_t flags, pt_scene **out);
/* SceneData::to_descriptor + SceneDescriptor::save (mod.rs:112-117, 127-149): the same bytes
 * serde_json::to_string_pretty writes (MeshFile objects keep their path/scale, inline meshes their
 * bounding_sphere / bounding_box). */
int pt_scene_save(const pt_scene *s, const char *path);
int pt_scene_set_camera(pt_scene *s, const pt_camera *cam);
/* setup_scenes (src/render/scenes.rs:43-318): the scenes the reference builds in code and saves when scenes/ holds
 * no *.json (load_scene_ids, scenes.rs:28-38) - "single-sphere", "cartesian", "two-spheres", "three-spheres",
 * "cornell", "mesh" in that order.  `base_dir` is where "mesh" finds meshes/mctri.off. */
uint32_t pt_builtin_scene_count(void);
const char *pt_builtin_scene_id(uint32_t i);
int pt_scene_builtin(const char *id, const char *base_dir, pt_scene **out);
void pt_scene_free(pt_scene *s);
const char *pt_scene_id(const pt_scene *s);
const pt_camera *pt_scene_camera(const pt_scene *s);
const pt_object *pt_scene_objects(const pt_scene *s, uint32_t *n);
const pt_triangle *pt_scene_triangles(const pt_scene *s, uint32_t *n);
/* Mesh.bounding_box of object i as the scene file stored it, or as Mesh::new computes it for MeshFile objects and meshes
 * built through the API; NULL for spheres.  12 triangles, valid until the scene is freed. */
const pt_triangle *pt_scene_bounding_box(const pt_scene *s, uint32_t object);

/* load_off (load_off.rs:8-85): returns a malloc'ed triangle array (free with pt_free). */
int pt_load_off(const char *path, float scale, pt_triangle **tris, uint32_t *n_tris);
int pt_load_off_ex(const char *path, float scale, uint32_t flags, pt_triangle **tris, uint32_t *n_tris);
void pt_free(void *p);

/* gamma (mod.rs:57-63) and the P3 writer (mod.rs:1043-1076). */
float pt_gamma_correction(float x);
uint32_t pt_to_int_with_gamma_correction(float x);
int pt_write_ppm(const char *path, const float *rgb, uint32_t width, uint32_t height, uint32_t spp,
                 const char *scene_id, uint64_t seconds);

#ifdef __cplusplus
}
#endif
#endif /* PTRACE_H */
