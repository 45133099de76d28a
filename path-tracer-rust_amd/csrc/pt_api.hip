// pt_api.hip — C ABI of the compute path (include/ptrace.h): contexts, scene flattening, the pass loop
// of the wavefront pipeline, the megakernel launch, single-ray queries.  No CPU fallback exists here:
// without a HIP device every entry point returns PT_ERR_NO_DEVICE.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ptrace.h"
#include "pt_host.h"
#include "pt_kernels.h"

namespace pt {

thread_local std::string g_last_error;

void set_error(const std::string &m) { g_last_error = m; }

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                              \
            return PT_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

// DevBuf::ensure's "out of device memory" (never leaves the library: render_wavefront retries with smaller passes and
// reports PT_ERR_HIP when even the smallest does not fit)
constexpr int PT_ERR_NOMEM_INTERNAL = -1000;

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    size_t bytes() const { return p ? n * sizeof(T) : 0; }
    int ensure(size_t count, bool tell_oom = false) {
        if (count <= n && p) return PT_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        hipError_t e = hipMalloc((void **)&p, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();  // (the error is reported through the return value; do not leave it sticky)
            set_error(std::string("hipMalloc of ") + std::to_string((count ? count : 1) * sizeof(T)) + " bytes: " + hipGetErrorString(e));
            return (tell_oom && e == hipErrorOutOfMemory) ? PT_ERR_NOMEM_INTERNAL : PT_ERR_HIP;
        }
        n = count;
        return PT_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace pt

using namespace pt;

// Tuning switches (A/B runs, profiling).  Read from the environment ONCE, when a context is created, and kept with the
// context: a frame never sees two different answers.  PT_DEBUG (ablation switches that change the image) exists only in
// builds made with -DPT_ALLOW_DEBUG; the shipped library ignores it.
struct Tuning {
    bool pass_kernel = true;   // PT_PASS_KERNEL=0: generate / intersect / shade as separate kernels
    bool pass_bvh = true;      // PT_PASS_BVH=0: BVH scenes through the separate kernels
    bool bvh_lds = false;      // PT_BVH_LDS=1: stage BVH nodes in LDS (separate kernels only)
    bool cand_scan = true;     // PT_CAND_SCAN=0: k_pass scans every triangle per ray (the round-1 form) instead of candidates
    uint32_t walk_queue_cap = 0;  // PT_WALK_QUEUE_CAP=n: the walk queue of k_pass_cand holds n entries (>= 128) instead of what its
                                  // LDS area allows - small values exercise the depth-first second walk (tests)
    bool cand_bvh = true;      // PT_CAND_BVH=0: scenes with BVH meshes run k_pass_bvh (scan + parked walks) instead of
                               // the candidate scan with parked walks (k_pass_cand<.., BVH>)
    uint32_t leaf_quorum = 12; // PT_LEAF_QUORUM: lanes on a leaf that send a walking wave to the triangle code
    uint64_t streams = 0;      // PT_STREAMS: ray streams per pass (0 = derived from the frame)
    uint32_t per_stream = 0;   // PT_PER_STREAM: primary rays per stream and pass that k_pass_cand's stream count aims at (0 = default)
    uint64_t rays_per_pass = 0;  // PT_RAYS_PER_PASS: the default of pt_config.rays_per_pass (probes; 0 = the library's)
    bool glass_defer = false;    // PT_GLASS_DEFER=1: k_pass_cand collects glass hits per wave and shades them 64 at a time (A/B: it
                                 // paid with levels, it does not without)
    bool nodes_lds = true;       // PT_NODES_LDS=0: k_pass_cand with walks reads the BVH nodes from global memory even when they
                                 // would fit its LDS (A/B)
    uint32_t wave_stack = 0;     // PT_WAVE_STACK=n: k_pass_cand's stacks hold n slots (a power of two, 512 <= n < kWaveStackMax)
                                 // instead of kWaveStackMax - the waves then have to hold their primaries back (tests)
    uint32_t mega_items = 0;     // PT_MEGA_ITEMS=n: the megakernel cuts a round into n items per lane the chip holds (0 = default)
    uint32_t debug = 0;
};
static Tuning read_tuning() {
    Tuning t;
    auto num = [](const char *name, long long dflt) {
        const char *e = getenv(name);
        return e ? atoll(e) : dflt;
    };
    t.pass_kernel = num("PT_PASS_KERNEL", 1) != 0;
    t.pass_bvh = num("PT_PASS_BVH", 1) != 0;
    t.bvh_lds = num("PT_BVH_LDS", 0) != 0;
    t.cand_scan = num("PT_CAND_SCAN", 1) != 0;
    t.cand_bvh = num("PT_CAND_BVH", 1) != 0;
    t.walk_queue_cap = (uint32_t)num("PT_WALK_QUEUE_CAP", 0);
    t.leaf_quorum = (uint32_t)num("PT_LEAF_QUORUM", 12);
    const long long st = num("PT_STREAMS", 0);
    t.streams = st > 0 ? (uint64_t)st : 0;
    t.per_stream = (uint32_t)num("PT_PER_STREAM", 0);
    t.rays_per_pass = (uint64_t)num("PT_RAYS_PER_PASS", 0);
    t.nodes_lds = num("PT_NODES_LDS", 1) != 0;
    t.glass_defer = num("PT_GLASS_DEFER", 0) != 0;
    {
        const long long mi = num("PT_MEGA_ITEMS", 0);
        t.mega_items = mi > 0 && mi <= 4096 ? (uint32_t)mi : 0u;
    }
    {
        const long long ws = num("PT_WAVE_STACK", 0);
        if (ws >= 512 && ws < (long long)kWaveStackMax && (ws & (ws - 1)) == 0) t.wave_stack = (uint32_t)ws;
    }
#ifdef PT_ALLOW_DEBUG
    t.debug = (uint32_t)num("PT_DEBUG", 0);
#endif
    return t;
}

struct pt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool has_scene = false;
    bool profiling = false;
    Tuning tune;
    pt_camera cam{};
    DevScene scene{};
    DevBuf<ObjRec> d_objs;
    DevBuf<ObjPairRec> d_opairs;
    DevBuf<TriPairRec> d_tris;
    DevBuf<MatRec> d_mats;
    DevBuf<TriShade> d_tshade;
    DevBuf<BvhNode> d_nodes;
    DevBuf<BvhNode4> d_nodes4;
    DevBuf<SphPairRec> d_sph;
    DevBuf<FlatPairRec> d_flat;
    DevBuf<CandPairRec> d_cand;
    DevBuf<uint32_t> d_rank_id;
    DevBuf<uint32_t> d_tri_rank;
    DevBuf<BvhMeshRec> d_bvh_meshes;
    DevBuf<SurfRec> d_surf;
    bool cand_ok = false;
    uint32_t n_bvh_nodes = 0;
    uint32_t n_cus = 0;  // compute units of the device (stream-count rounding)
    // Mesh.bounding_box of every object (12 object-local triangles each; Mesh::new's unless pt_ctx_set_mesh_bounds gave
    // the stored ones) and their device form (6 pair records per object), for intersect_bounds / orbit-point queries
    std::vector<pt_triangle> h_boxes;
    std::vector<pt_object> h_objs;
    DevBuf<TriPairRec> d_boxes;
    bool boxes_dirty = true;
    // scratch of the single-ray query entry points (kept across calls: a picking caller sends one ray per click)
    DevBuf<float> q_o, q_d, q_t, q_x, q_n;
    DevBuf<int32_t> q_oid, q_tid;
    // wavefront queues
    uint32_t K = 0, cap = 0;
    DevBuf<char> q_buf[2];  // the two ray-queue containers: K slices of cap * 40 bytes each (RayQueue, pt_kernels.h)
    DevBuf<float2> hit;
    DevBuf<uint32_t> cnt, flags;
    DevBuf<unsigned long long> blk_rays, acc, total_rays;
    std::vector<hipEvent_t> ev_pool;
    // state of the frame being rendered (for pt_ctx_snapshot from the progress callback)
    uint32_t live_npix = 0, live_spp_issued = 0, live_streams = 1, live_m = 0;
    hipStream_t live_stream = nullptr;
    // a large call is rendered in parts: pixels [0, live_k0) of it are final in live_out, [live_k0, live_k0 + live_npix) are
    // the part in progress, the rest has not been started (live_total pixels in all)
    float *live_out = nullptr;
    uint32_t live_k0 = 0, live_total = 0;
    // concurrent pipelines (PT_FLAG_PIPELINES): child contexts that borrow this context's scene tables
    std::vector<pt_ctx *> pipes;
    std::vector<DevBuf<float>> pipe_out;
    bool borrowed_scene = false;
    // memory-aware pass sizing: contexts that will hold ray queues on this device at the same time (pipelines of one call,
    // ranks of pt_render_multi that share a device) and an explicit cap on the queue memory of this context (0 = 85 % of
    // what hipMemGetInfo reports free, divided by `mem_share`)
    uint32_t mem_share = 1;
    size_t mem_budget = 0;
    double cb_last_ms = 0.0;  // time of the last progress callback of the call in progress (throttle: pt_config.progress_ms)
    // Passes sized by TIME (k_pass_cand, megakernel rounds): primary samples per millisecond the last timed pass / round of this
    // scene went through, per backend (0: not measured yet - the next frame starts with a short timed pass).  pt_ctx_set_scene
    // forgets them.
    double pass_rate = 0.0, round_rate = 0.0;
    const char *pass_rate_kernel = nullptr;  // the kernel pass_rate was measured on (flags choose other kernels)
};

namespace {

int device_count_quiet() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int check_cfg(const pt_config *cfg, uint32_t *idx_begin, uint32_t *idx_end) {
    if (!cfg) {
        set_error("cfg is NULL");
        return PT_ERR_INVALID;
    }
    if (cfg->width == 0 || cfg->height == 0 || cfg->spp == 0) {
        set_error("width, height and spp must be positive");
        return PT_ERR_INVALID;
    }
    const uint64_t npix = (uint64_t)cfg->width * cfg->height;
    if (npix > 0x7fffffffull) {
        set_error("width*height exceeds 2^31-1");
        return PT_ERR_INVALID;
    }
    if (cfg->spp > (1u << 24)) {
        set_error("spp exceeds 2^24");
        return PT_ERR_INVALID;
    }
    uint32_t b = cfg->idx_begin, e = cfg->idx_end;
    if (b == 0 && e == 0) e = (uint32_t)npix;
    if (b >= e || e > npix) {
        set_error("band [idx_begin, idx_end) is empty or outside the frame");
        return PT_ERR_INVALID;
    }
    if (cfg->backend != PT_BACKEND_WAVEFRONT && cfg->backend != PT_BACKEND_MEGAKERNEL) {
        set_error("unknown backend");
        return PT_ERR_INVALID;
    }
    if (cfg->chunk_step > 1u && (cfg->chunk_pixels == 0u || cfg->chunk_first >= cfg->chunk_step)) {
        set_error("chunk_pixels must be positive and chunk_first < chunk_step");
        return PT_ERR_INVALID;
    }
    *idx_begin = b;
    *idx_end = e;
    return PT_OK;
}

// pixels of the band [b, e) that fall into chunks first, first+step, ... (all of them when step <= 1)
uint32_t owned_pixels(const pt_config *cfg, uint32_t b, uint32_t e) {
    const uint64_t span = e - b;
    if (cfg->chunk_step <= 1u) return (uint32_t)span;
    const uint64_t C = cfg->chunk_pixels, n_chunks = (span + C - 1) / C;
    uint64_t total = 0;
    for (uint64_t c = cfg->chunk_first; c < n_chunks; c += cfg->chunk_step) {
        const uint64_t lo = c * C, hi = (lo + C < span) ? lo + C : span;
        total += hi - lo;
    }
    return (uint32_t)total;
}

FrameParams make_frame(const pt_ctx *ctx, const pt_config *cfg, uint32_t idx_begin, uint32_t idx_end) {
    FrameParams F{};
    float lens[3], su[3], sv[3];
    host::camera_basis(ctx->cam, lens, su, sv);
    F.width = cfg->width;
    F.height = cfg->height;
    F.spp = cfg->spp;
    F.idx_begin = idx_begin;
    F.npix = owned_pixels(cfg, idx_begin, idx_end);
    F.chunk_pixels = cfg->chunk_pixels;
    F.chunk_first = cfg->chunk_first;
    F.chunk_step = cfg->chunk_step;
    F.n_streams = 1;  // set by the wavefront renderer
    F.seed_lo = (uint32_t)cfg->seed;
    F.seed_hi = (uint32_t)(cfg->seed >> 32);
    F.cam_px = ctx->cam.position[0];
    F.cam_py = ctx->cam.position[1];
    F.cam_pz = ctx->cam.position[2];
    F.lens_x = lens[0];
    F.lens_y = lens[1];
    F.lens_z = lens[2];
    F.su_x = su[0];
    F.su_y = su[1];
    F.su_z = su[2];
    F.sv_x = sv[0];
    F.sv_y = sv[1];
    F.sv_z = sv[2];
    F.debug = ctx->tune.debug;
    F.k_begin = 0;
    return F;
}

RayQueue queue_of(pt_ctx *c, int which) {
    RayQueue q;
    q.buf = c->q_buf[which].p;
    return q;
}

hipEvent_t get_event(pt_ctx *c, size_t i) {
    while (c->ev_pool.size() <= i) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[i];
}

// does a frame with these flags run the candidate scan (k_pass_cand)?  Scenes with BVH meshes: with parked walks, unless
// their nodes are staged in LDS (PT_BVH_LDS=1) or PT_CAND_BVH=0 asks for k_pass_bvh.
static uint32_t cand_scan_for(const pt_ctx *c, uint32_t flags) {
    if (!c->tune.cand_scan || !c->cand_ok || (flags & PT_FLAG_NO_BVH)) return 0u;
    if (c->n_bvh_nodes != 0u && (!c->tune.cand_bvh || (c->scene.bvh_in_lds & 1u))) return 0u;
    return 1u;
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int render_wavefront(pt_ctx *c, const pt_config *cfg, const FrameParams &frame, hipStream_t st,
                     const volatile uint8_t *cancel, pt_progress_fn cb, void *user, pt_stats *stats) {
    FrameParams F = frame;
    const uint64_t npix = F.npix;
    // Level-by-level forms (k_pass, k_pass_bvh, the separate kernels): 96 Mi primary rays per pass by default, 36 GB of ray
    // queues (two containers x 4 slots per primary ray x 40 B) of the 288 GB of HBM - fewer, longer launches: cornell 1024x768
    // @4096 spp 32 Mi 35.7, 48 Mi 35.7, 64 Mi 36.2, 96 Mi 36.4 G bounces/s (a launch ends with its slowest streams).
    // k_pass_cand (`stack_form`) keeps a wave's waiting rays on a stack of at most kWaveStackMax slots whatever the pass holds:
    // its passes are sized by TIME - 512 Mi primary rays, about 0.1 s between two looks at the cancel flag (the reference
    // polls it every 100 ms, mod.rs:947-958) - and its memory is the streams' (K x 4 waves x stack x 40 B: 4.0 GB for the 24 576
    // streams of the bench frame's pass; small passes need less: 4 x pow2(primaries per wave) slots per stream).
    // The default is what the DEVICE can give: 85 % of the free memory (plus what this context's queues hold already),
    // divided by the contexts that share the device in this call (PT_FLAG_PIPELINES, ranks of pt_render_multi on one GPU),
    // or pt_ctx_set_memory_budget's figure - at 352 B per primary ray for the level-by-level forms (queues + hit records of
    // the three-kernel form), by the streams' stacks for k_pass_cand; and whatever was asked for, a failed allocation halves
    // the pass and tries again: passes only change how the samples are batched, never the image.
    const bool bvh_ok = c->scene.n_bvh_nodes == 0u || (!(c->scene.bvh_in_lds & 1u) && c->tune.pass_bvh);
    const bool one_kernel = bvh_ok && c->tune.pass_kernel && !(cfg->flags & PT_FLAG_SEPARATE_KERNELS);
    const bool needs_hits = !one_kernel;
    const bool stack_form = one_kernel && c->scene.cand_scan != 0u;
    // container 1: the waves' parked rays (scenes with walks) or deferred glass hits
    const bool stack_park = stack_form && (c->scene.n_bvh_nodes != 0u || c->scene.glass_defer_ok != 0u);
    uint64_t want = cfg->rays_per_pass ? cfg->rays_per_pass : c->tune.rays_per_pass;
    size_t stack_budget = 0;  // stack_form, default pass size: what the streams' stacks may take
    if (!want) {
        const uint64_t dflt = stack_form ? (512u << 20) : (96u << 20);
        want = dflt;
        size_t held = c->hit.bytes();
        for (int w = 0; w < 2; ++w) held += c->q_buf[w].bytes();
        size_t avail = c->mem_budget;
        if (!avail) {
            size_t mem_free = 0, mem_total = 0;
            if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess) avail = (size_t)((double)(mem_free + held) * 0.85) / (c->mem_share ? c->mem_share : 1u);
        }
        if (avail && !stack_form) {
            const uint64_t fit = avail / 352u;
            if (fit < want) want = fit;
        }
        if (stack_form) stack_budget = avail;
        if (c->mem_share > 1u && want > dflt / c->mem_share) want = dflt / c->mem_share;  // (co-resident contexts also share the chip)
        if (want < npix) want = npix;  // one sample per pixel and pass at least
    } else if (c->mem_budget) {
        // an explicit pass size under an explicit budget (pt_ctx_set_memory_budget): the budget wins - the pass is cut to what
        // it allows (level-by-level forms: 352 B per primary ray; k_pass_cand: its streams' stacks, plan_pass retries with
        // smaller passes).  Without a budget an explicit size is taken as given and only a failed allocation halves it.
        if (!stack_form) {
            const uint64_t fit = c->mem_budget / 352u;
            if (fit < want) want = fit;
        } else {
            stack_budget = c->mem_budget;
        }
        if (want < npix) want = npix;
    }
    uint32_t spp_pass = 0, m = 0, K = 0, cap = 0;
    for (;;) {
        host::PassPlanIn pin;
        pin.npix = npix;
        pin.spp = cfg->spp;
        pin.want = want;
        pin.want_is_default = !cfg->rays_per_pass;
        pin.stack_form = stack_form;
        pin.stack_park = stack_park;
        pin.cand_scan = c->scene.cand_scan != 0u;
        pin.has_bvh = c->scene.n_bvh_nodes != 0u;
        pin.streams = c->tune.streams;
        pin.per_stream = c->tune.per_stream;
        pin.wave_stack = c->tune.wave_stack;
        pin.n_cus = c->n_cus;
        pin.groups_per_cu = !stack_form ? 4u : (c->scene.n_bvh_nodes == 0u ? (uint32_t)PT_CAND_WAVES : (uint32_t)PT_CAND_BVH_WAVES);
        pin.stack_budget = stack_budget;
        host::PassPlan plan;
        uint64_t want_next = want;
        const int pr = host::plan_pass(pin, plan, &want_next);  // (the arithmetic and its measurements: pt_host.cpp)
        if (pr == host::kPlanRetry) {
            want = want_next;
            continue;
        }
        if (pr != host::kPlanOk) {
            set_error("rays per pass too large");
            return PT_ERR_INVALID;
        }
        spp_pass = plan.spp_pass;
        m = plan.m;
        K = plan.K;
        cap = plan.cap;
        const size_t slots = (size_t)K * cap;
        int rc = PT_OK;
        rc = c->q_buf[0].ensure(plan.bytes0, true);
        if (!rc && plan.bytes1) rc = c->q_buf[1].ensure(plan.bytes1, true);
        // scenes without BVH meshes run a pass as one launch (k_pass), BVH scenes as k_pass_bvh unless their nodes are staged
        // in LDS; PT_FLAG_SEPARATE_KERNELS / PT_PASS_KERNEL=0 / PT_PASS_BVH=0 keep the three-kernel form (A/B, profiling).
        // Only that form needs the hit records: k_pass keeps hits in registers.
        if (!rc && needs_hits) rc = c->hit.ensure(slots, true);
        if (!rc) rc = c->cnt.ensure((size_t)kLevels * K, true);
        if (!rc) rc = c->flags.ensure(1, true);
        if (!rc) rc = c->blk_rays.ensure(K, true);
        if (!rc) rc = c->acc.ensure(3 * (size_t)K * m, true);
        if (!rc) break;
        if (rc != PT_ERR_NOMEM_INTERNAL) return rc;
        // out of device memory: give back what this attempt took and try passes of half the size
        for (int w = 0; w < 2; ++w) c->q_buf[w].release();
        c->hit.release();
        if (spp_pass <= 1u) return PT_ERR_HIP;  // (the message names the allocation that failed)
        want = (uint64_t)npix * (spp_pass / 2u ? spp_pass / 2u : 1u);
    }
    c->K = K;
    c->cap = cap;
    F.n_streams = K;  // stream b owns pixels b, b+K, ...; accumulators are stream-major (K*m slots per channel)
    c->live_streams = K;
    c->live_m = m;
    HIP_TRY(hipMemsetAsync(c->acc.p, 0, 3 * (size_t)K * m * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(c->blk_rays.p, 0, K * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(c->flags.p, 0, sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(c->cnt.p, 0, (size_t)kLevels * K * sizeof(uint32_t), st));

    const uint32_t n_pass = (cfg->spp + spp_pass - 1) / spp_pass;
    const int n_depth = kMaxDepth;  // rays of depth 0..11 exist
    // PASSES THAT FOLLOW THE SCENE (k_pass_cand at the library's own pass size).  The reference looks at its stop flag every
    // 100 ms (mod.rs:947-958); here the flag is read between passes, so a pass must not take much longer than that WHATEVER a
    // ray of the scene costs - 512 Mi primary rays are 0.1 s on cornell.json, 0.15 s on mesh.json, and a scene of 392
    // unfiltered candidate records inside an emitting sphere (tests) is fifty times dearer per primary ray.  Nothing is known
    // about a scene's cost before its first rays have been traced, so the first pass of a scene is TINY (kProbeRays primary
    // rays: a fraction of a millisecond on the bench scene) and timed (HIP events around the launch); every pass is timed,
    // and the next one gets as many samples as the measured rate fits into kPassTargetMs (stretched by up to a fifth where
    // that saves a pass) - at most sixteen times the pass before (a short pass measures launch overheads too, and its
    // streams are too short to be efficient: the estimate errs towards short passes), never more than the planned spp_pass
    // (the stacks, the sample field of the bookkeeping word), the rest of the frame cut into equal passes.  The rate is kept
    // with the context (pt_ctx.pass_rate), so the following frames of the same scene start at full length: the bench frame
    // is six passes of 683 samples as before, and the first frame of a scene pays three short passes (a few milliseconds).
    // Frames of at most kAdaptiveMinRays primary rays are one pass; an explicit rays_per_pass is taken as given.  Passes
    // only batch the samples: the image does not depend on them.
    constexpr uint64_t kProbeRays = 1ull << 20, kAdaptiveMinRays = 4ull << 20;
    constexpr double kPassTargetMs = 100.0;
    const bool adaptive = stack_form && !cfg->rays_per_pass && !c->tune.rays_per_pass && n_pass > 0u &&
                          (uint64_t)npix * cfg->spp > kAdaptiveMinRays;
    const char *const rate_key = pt_ctx_pass_kernel(c, cfg->flags);
    double rate = (adaptive && c->pass_rate_kernel == rate_key) ? c->pass_rate : 0.0;
    size_t ev_i = 0;
    hipEvent_t ev_begin = get_event(c, ev_i++), ev_end = get_event(c, ev_i++);
    hipEvent_t pass_done[2] = {get_event(c, ev_i++), get_event(c, ev_i++)};
    hipEvent_t pass_begin[2] = {get_event(c, ev_i++), get_event(c, ev_i++)};
    if (!ev_begin || !ev_end || !pass_done[0] || !pass_done[1] || !pass_begin[0] || !pass_begin[1]) {
        set_error("hipEventCreate failed");
        return PT_ERR_HIP;
    }
    const size_t ev_prof0 = ev_i;
    size_t n_prof = 0;
    HIP_TRY(hipEventRecord(ev_begin, st));
    bool cancelled = false;
    uint32_t passes_done = 0;
    // RenderUpdate cadence (mod.rs:965-982): the reference reports every 500 ms.  Passes are much shorter than that, so
    // the callback is throttled to pt_config.progress_ms (0 = 500 ms, PT_PROGRESS_EVERY_PASS = every pass boundary);
    // the cancel byte is read at EVERY pass boundary (the reference polls it every 100 ms, mod.rs:947-958).
    const double cb_every_ms = cfg->progress_ms == PT_PROGRESS_EVERY_PASS ? 0.0 : (cfg->progress_ms ? (double)cfg->progress_ms : 500.0);
    double &cb_last_ms = c->cb_last_ms;  // (set when the call began: pt_ctx_render; a call rendered in parts keeps one clock)
    uint32_t s_next = 0u, s_prev = 0u;  // samples of a pixel issued so far / in the pass before
    for (uint32_t p = 0; s_next < cfg->spp; ++p) {
        // keep two passes in flight; k_pass_cand's long passes (0.1 s) one - the cancel flag is looked at when a pass ends, and
        // the few microseconds between two launches are nothing against such a pass
        if (stack_form && p >= 1) {
            HIP_TRY(hipEventSynchronize(pass_done[(p - 1) & 1]));
            if (adaptive) {
                float ms = 0.0f;
                HIP_TRY(hipEventElapsedTime(&ms, pass_begin[(p - 1) & 1], pass_done[(p - 1) & 1]));
                if (ms > 0.0f) rate = (double)npix * s_prev / ms;
            }
        }
        if (p >= 2) HIP_TRY(hipEventSynchronize(pass_done[p & 1]));
        if (cancel && *cancel) {
            cancelled = true;
            break;
        }
        if (cb && p >= (stack_form ? 1u : 2u)) {
            const double t_now = now_ms();
            if (t_now - cb_last_ms >= cb_every_ms) {
                cb_last_ms = t_now;
                // (the samples known to be done)
                cb(user, stack_form ? (float)s_next / (float)cfg->spp : (float)(p - 1u) / (float)n_pass);
                if (cancel && *cancel) {  // raised from inside the callback
                    cancelled = true;
                    break;
                }
            }
        }
        const uint32_t s0 = s_next;
        uint32_t s_here = (cfg->spp - s0) < spp_pass ? (cfg->spp - s0) : spp_pass;
        if (adaptive) s_here = host::next_pass_samples(rate, kPassTargetMs, npix, kProbeRays, s_prev, cfg->spp - s0, spp_pass);
        s_next = s0 + s_here;
        s_prev = s_here;
        c->live_spp_issued = s0 + s_here;
        if (adaptive) HIP_TRY(hipEventRecord(pass_begin[p & 1], st));
        if (one_kernel) {  // the whole pass in one launch (k_pass)
            hipEvent_t a = nullptr, b = nullptr;
            if (c->profiling) {
                a = get_event(c, ev_prof0 + 2 * n_prof), b = get_event(c, ev_prof0 + 2 * n_prof + 1);
                if (!a || !b) {
                    set_error("hipEventCreate failed");
                    return PT_ERR_HIP;
                }
                HIP_TRY(hipEventRecord(a, st));
            }
            if (c->scene.n_bvh_nodes != 0u && !c->scene.cand_scan)
                launch_pass_bvh(st, K, c->scene, F, queue_of(c, 0), queue_of(c, 1), cap, s0, s_here, m, c->acc.p,
                                c->blk_rays.p, c->flags.p);
            else if (launch_pass(st, K, c->scene, F, queue_of(c, 0), queue_of(c, 1), cap, s0, s_here, m, c->acc.p, c->blk_rays.p,
                                 c->flags.p) != hipSuccess)
                return PT_ERR_HIP;  // (the message says how much LDS the scene's kernel asked for)
            if (c->profiling) {
                HIP_TRY(hipEventRecord(b, st));
                ++n_prof;
            }
            HIP_TRY(hipEventRecord(pass_done[p & 1], st));
            ++passes_done;
            continue;
        }
        launch_generate(st, K, F, queue_of(c, 0), c->cnt.p, cap, s0, s_here, m);
        for (int d = 0; d < n_depth; ++d) {
            const RayQueue qin = queue_of(c, d & 1), qout = queue_of(c, (d + 1) & 1);
            if (c->profiling) {
                hipEvent_t a = get_event(c, ev_prof0 + 2 * n_prof), b = get_event(c, ev_prof0 + 2 * n_prof + 1);
                if (!a || !b) {
                    set_error("hipEventCreate failed");
                    return PT_ERR_HIP;
                }
                HIP_TRY(hipEventRecord(a, st));
                launch_intersect(st, K, c->scene, qin, c->hit.p, c->cnt.p + (size_t)d * K, cap, c->blk_rays.p);
                HIP_TRY(hipEventRecord(b, st));
                ++n_prof;
            } else {
                launch_intersect(st, K, c->scene, qin, c->hit.p, c->cnt.p + (size_t)d * K, cap, c->blk_rays.p);
            }
            launch_shade(st, K, c->scene, F, qin, qout, c->hit.p, c->cnt.p + (size_t)d * K,
                         c->cnt.p + (size_t)(d + 1) * K, cap, c->acc.p, c->flags.p, m, s0);
        }
        HIP_TRY(hipEventRecord(pass_done[p & 1], st));
        ++passes_done;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev_end, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (adaptive && passes_done != 0u) {  // the last pass counts too (a frame of one probe and one long pass would otherwise only know the probe)
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, pass_begin[(passes_done - 1u) & 1u], pass_done[(passes_done - 1u) & 1u]));
        if (ms > 0.0f && (double)npix * s_prev >= 16.0 * (double)kProbeRays) rate = (double)npix * s_prev / ms;
        c->pass_rate = rate;
        c->pass_rate_kernel = rate_key;
    }
    std::vector<unsigned long long> rays(K);
    HIP_TRY(hipMemcpy(rays.data(), c->blk_rays.p, K * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    uint32_t flags = 0;
    HIP_TRY(hipMemcpy(&flags, c->flags.p, sizeof flags, hipMemcpyDeviceToHost));
    if (stats) {
        unsigned long long total = 0;
        for (auto v : rays) total += v;
        stats->ray_bounces = total;
        stats->intersect_rays = total;
        stats->intersect_launches = one_kernel ? passes_done : passes_done * (uint32_t)n_depth;
        stats->passes = passes_done;
        stats->samples = npix * (uint64_t)s_next;  // (every pass that was issued has run: the stream is synchronised)
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev_begin, ev_end));
        stats->ms_device = ms;
        double mi = 0.0;
        for (size_t i = 0; i < n_prof; ++i) {
            float e = 0.0f;
            HIP_TRY(hipEventElapsedTime(&e, c->ev_pool[ev_prof0 + 2 * i], c->ev_pool[ev_prof0 + 2 * i + 1]));
            mi += e;
        }
        stats->ms_intersect = mi;
    }
    if (flags & 3u) {
        set_error((flags & 2u) ? "ray stream slices too small for k_pass_cand's wave stacks" : "ray stream overflow");
        return PT_ERR_OVERFLOW;
    }
    if (cancelled) {
        set_error("cancelled");
        return PT_CANCELLED;
    }
    return PT_OK;
}

int render_mega(pt_ctx *c, const pt_config *cfg, const FrameParams &F, hipStream_t st, const volatile uint8_t *cancel,
                pt_progress_fn cb, void *user, pt_stats *stats) {
    const uint64_t npix = F.npix;
    int rc;
    if ((rc = c->acc.ensure(3 * npix)) || (rc = c->total_rays.ensure(16))) return rc;
    c->live_streams = 1;  // accumulators in pixel order
    c->live_m = (uint32_t)npix;
    HIP_TRY(hipMemsetAsync(c->acc.p, 0, 3 * npix * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(c->total_rays.p, 0, 16 * sizeof(unsigned long long), st));
    // The frame is cut into ROUNDS: one round = every pixel of the call x round_spp consecutive samples, one launch (a
    // lane = one pixel's samples of the round, walked one after the other).  A round is sized to about a tenth of a
    // second of work, so the cancel byte and the progress callback are served between launches (the reference polls
    // cancel every 100 ms, mod.rs:947-958) and every pixel holds the same number of samples at every boundary - which is
    // what pt_ctx_snapshot and a cancelled frame resolve.  Small frames get several lanes per pixel inside a round
    // (n_split) so that a launch still fills the chip.
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    const uint64_t lanes = (uint64_t)prop.multiProcessorCount * 2048u;
    const uint64_t round_budget = cfg->rays_per_pass ? cfg->rays_per_pass : (256ull << 20);  // primary samples per launch
    uint64_t round_spp64 = round_budget / npix;
    if (round_spp64 == 0) round_spp64 = 1;
    if (round_spp64 > cfg->spp) round_spp64 = cfg->spp;
    const uint32_t round_spp = (uint32_t)round_spp64;
    uint32_t n_split = 1;  // lanes per pixel within a round
    // (k_mega_cand hands its items out dynamically: finer ones - 8 per lane the chip holds, cornell 41.3 G bounces/s; 4: 39.1,
    // 16: 40.5, 32: 38.4 - so that a launch's last items are a small part of it; PT_MEGA_ITEMS for A/B runs and tests)
    const uint64_t item_mult = c->tune.mega_items ? c->tune.mega_items : (mega_uses_cand(c->scene) ? 8u : 4u);
    const uint64_t want_items = item_mult * lanes;
    while ((uint64_t)npix * n_split < want_items && n_split < round_spp) n_split *= 2;
    if (n_split > round_spp) n_split = round_spp;
    const uint32_t n_rounds = (cfg->spp + round_spp - 1) / round_spp;
    // rounds that follow the scene, as the wavefront's passes do (render_wavefront): a short timed first round, then as many
    // samples per round as the measured rate fits into 100 ms (+ a fifth) (at most 16 x the round before, at most round_spp), the rest
    // of the frame in equal rounds, one launch in flight; the rate stays with the context for the next frame
    constexpr uint64_t kProbeSamples = 1ull << 20, kAdaptiveMinSamples = 4ull << 20;
    constexpr double kRoundTargetMs = 100.0;
    const bool adaptive = !cfg->rays_per_pass && (uint64_t)npix * cfg->spp > kAdaptiveMinSamples;
    double rate = adaptive ? c->round_rate : 0.0;
    hipEvent_t ev_begin = get_event(c, 0), ev_end = get_event(c, 1);
    hipEvent_t round_done[2] = {get_event(c, 2), get_event(c, 3)};
    hipEvent_t round_begin[2] = {get_event(c, 4), get_event(c, 5)};
    if (!ev_begin || !ev_end || !round_done[0] || !round_done[1] || !round_begin[0] || !round_begin[1]) {
        set_error("hipEventCreate failed");
        return PT_ERR_HIP;
    }
    HIP_TRY(hipEventRecord(ev_begin, st));
    const double cb_every_ms = cfg->progress_ms == PT_PROGRESS_EVERY_PASS ? 0.0 : (cfg->progress_ms ? (double)cfg->progress_ms : 500.0);
    double &cb_last_ms = c->cb_last_ms;  // (set when the call began: pt_ctx_render; a call rendered in parts keeps one clock)
    bool cancelled = false;
    uint32_t rounds_done = 0;
    uint64_t samples = 0;
    uint32_t s_next = 0u, s_prev = 0u;
    for (uint32_t r = 0; s_next < cfg->spp; ++r) {
        if (adaptive && r >= 1) {  // one launch in flight: its time sizes the next
            HIP_TRY(hipEventSynchronize(round_done[(r - 1) & 1]));
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, round_begin[(r - 1) & 1], round_done[(r - 1) & 1]));
            if (ms > 0.0f) rate = (double)npix * s_prev / ms;
        }
        if (r >= 2) HIP_TRY(hipEventSynchronize(round_done[r & 1]));  // two launches in flight
        if (cancel && *cancel) {
            cancelled = true;
            break;
        }
        if (cb && r >= (adaptive ? 1u : 2u)) {
            const double t_now = now_ms();
            if (t_now - cb_last_ms >= cb_every_ms) {
                cb_last_ms = t_now;
                cb(user, adaptive ? (float)s_next / (float)cfg->spp : (float)(r - 1) / (float)n_rounds);
                if (cancel && *cancel) {
                    cancelled = true;
                    break;
                }
            }
        }
        const uint32_t s0 = s_next;
        uint32_t s_here = (cfg->spp - s0) < round_spp ? (cfg->spp - s0) : round_spp;
        if (adaptive) {
            s_here = host::next_pass_samples(rate, kRoundTargetMs, npix, kProbeSamples, s_prev, cfg->spp - s0, round_spp);
            HIP_TRY(hipEventRecord(round_begin[r & 1], st));
        }
        s_next = s0 + s_here;
        s_prev = s_here;
        const uint32_t split = n_split < s_here ? n_split : s_here;
        const uint32_t lane_spp = (s_here + split - 1) / split;
        const uint64_t items = npix * split;
        const uint64_t grid64 = (items + kBlock - 1) / kBlock;
        const uint64_t max_grid = (uint64_t)prop.multiProcessorCount * 8u;
        const uint32_t grid = (uint32_t)(grid64 < max_grid ? grid64 : max_grid);
        if (mega_uses_cand(c->scene) && (rc = c->q_buf[0].ensure(mega_stack_mem_bytes(grid ? grid : 1u)))) return rc;  // split stacks
        HIP_TRY(hipMemsetAsync(c->total_rays.p + 7, 0, sizeof(unsigned long long), st));  // k_mega_cand's item counter
        launch_mega(st, grid ? grid : 1u, c->scene, F, c->acc.p, s0, s0 + s_here, lane_spp, split, c->total_rays.p, c->q_buf[0].p);
        HIP_TRY(hipEventRecord(round_done[r & 1], st));
        c->live_spp_issued = s0 + s_here;
        samples += npix * s_here;
        ++rounds_done;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev_end, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (adaptive && rounds_done != 0u) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, round_begin[(rounds_done - 1u) & 1u], round_done[(rounds_done - 1u) & 1u]));
        if (ms > 0.0f && (double)npix * s_prev >= 16.0 * (double)kProbeSamples) rate = (double)npix * s_prev / ms;
        c->round_rate = rate;
    }
    if (stats) {
        unsigned long long total2[16] = {0};
        HIP_TRY(hipMemcpy(total2, c->total_rays.p, sizeof total2, hipMemcpyDeviceToHost));
#ifdef PT_MEGA_STATS
        fprintf(stderr, "mega stats: trips %llu, started per trip %.2f, finished per trip %.2f, maker iterations %llu (per trip %.3f) at %.1f lanes\n",
                total2[2], (double)total2[3] / (double)(total2[2] ? total2[2] : 1), (double)total2[4] / (double)(total2[2] ? total2[2] : 1), total2[5],
                (double)total2[5] / (double)(total2[2] ? total2[2] : 1), (double)total2[6] / (double)(total2[5] ? total2[5] : 1));
        fprintf(stderr, "mega stats: lanes with an item %.2f, of them dry (no ray to start, samples used up) %.2f, lanes told no more %.2f per trip\n",
                (double)total2[8] / (double)(total2[2] ? total2[2] : 1), (double)total2[9] / (double)(total2[2] ? total2[2] : 1),
                (double)total2[10] / (double)(total2[2] ? total2[2] : 1));
#endif
        const unsigned long long total = total2[0];
        if (total2[1]) {
            set_error("megakernel: a lane's split stack overflowed");
            return PT_ERR_OVERFLOW;
        }
        stats->ray_bounces = total;
        stats->intersect_rays = 0;
        stats->intersect_launches = 0;
        stats->passes = rounds_done;
        stats->samples = samples;
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev_begin, ev_end));
        stats->ms_device = ms;
        stats->ms_intersect = 0.0;
    }
    if (cancelled) {
        set_error("cancelled");
        return PT_CANCELLED;
    }
    return PT_OK;
}

}  // namespace

extern "C" {

const char *pt_version(void) { return "ptrace-hip 0.3 (gfx950)"; }
#ifndef PT_BUILD_FLAGS
#define PT_BUILD_FLAGS ""
#endif
const char *pt_build_flags(void) { return PT_BUILD_FLAGS; }
const char *pt_last_error(void) { return g_last_error.c_str(); }
int pt_abi_version(void) { return PT_ABI_VERSION; }
int pt_device_count(void) { return device_count_quiet(); }

uint32_t pt_config_pixels(const pt_config *cfg) {
    uint32_t b = 0, e = 0;
    if (check_cfg(cfg, &b, &e) != PT_OK) return 0;
    return owned_pixels(cfg, b, e);
}

int pt_camera_basis(const pt_camera *cam, float lens_center[3], float su[3], float sv[3]) {
    if (!cam || !lens_center || !su || !sv) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    host::camera_basis(*cam, lens_center, su, sv);
    return PT_OK;
}

int pt_mesh_bounding_sphere(const pt_triangle *tris, uint32_t n_tris, float center[3], float *radius) {
    if (!tris || !center || !radius || n_tris == 0) {
        set_error("NULL argument or empty mesh");
        return PT_ERR_INVALID;
    }
    host::mesh_bounding_sphere(tris, n_tris, center, radius);
    return PT_OK;
}

int pt_ctx_create(int device, pt_ctx **out) {
    if (!out) {
        set_error("out is NULL");
        return PT_ERR_INVALID;
    }
    *out = nullptr;
    const int n = device_count_quiet();
    if (n <= 0) {
        set_error("no HIP device: libptrace_hip has no CPU fallback");
        return PT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device index out of range");
        return PT_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device));
    pt_ctx *c = new pt_ctx();
    c->device = device;
    c->tune = read_tuning();
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_cus = (uint32_t)cus;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e));
        delete c;
        return PT_ERR_HIP;
    }
    *out = c;
    return PT_OK;
}

void pt_ctx_destroy(pt_ctx *c) {
    if (!c) return;
    for (pt_ctx *p : c->pipes) pt_ctx_destroy(p);  // children own their queues, not the scene tables
    c->pipes.clear();
    for (auto &b : c->pipe_out) b.release();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    c->d_objs.release();
    c->d_opairs.release();
    c->d_tris.release();
    c->d_mats.release();
    c->d_tshade.release();
    c->d_nodes.release();
    c->d_nodes4.release();
    c->d_sph.release();
    c->d_flat.release();
    c->d_cand.release();
    c->d_rank_id.release();
    c->d_tri_rank.release();
    c->d_bvh_meshes.release();
    c->d_surf.release();
    c->d_boxes.release();
    c->q_o.release();
    c->q_d.release();
    c->q_t.release();
    c->q_x.release();
    c->q_n.release();
    c->q_oid.release();
    c->q_tid.release();
    for (int w = 0; w < 2; ++w) c->q_buf[w].release();
    c->hit.release();
    c->cnt.release();
    c->flags.release();
    c->blk_rays.release();
    c->acc.release();
    c->total_rays.release();
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int pt_ctx_set_scene(pt_ctx *c, const pt_camera *cam, const pt_object *objs, uint32_t n_objs,
                     const pt_triangle *tris, uint32_t n_tris) {
    if (!c || !cam || (!objs && n_objs) || (!tris && n_tris)) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    host::FlatScene fs;
    std::string err;
    if (!host::flatten_scene(*cam, objs, n_objs, tris, n_tris, fs, err)) {
        set_error(err);
        return PT_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->d_objs.ensure(fs.objs.size())) || (rc = c->d_opairs.ensure(fs.obj_pairs.size())) || (rc = c->d_tris.ensure(fs.tri_pairs.size())) ||
        (rc = c->d_mats.ensure(fs.mats.size())) || (rc = c->d_tshade.ensure(fs.tri_shade.size())) ||
        (rc = c->d_nodes.ensure(fs.bvh_nodes.size())) || (rc = c->d_nodes4.ensure(fs.bvh_nodes4.size())) ||
        (rc = c->d_sph.ensure(fs.sph_pairs.size())) ||
        (rc = c->d_flat.ensure(fs.flat_pairs.size())) || (rc = c->d_cand.ensure(fs.cand_pairs.size())) ||
        (rc = c->d_rank_id.ensure(fs.rank_id.size())) || (rc = c->d_surf.ensure(fs.surf.size())) ||
        (rc = c->d_tri_rank.ensure(fs.tri_rank.size())) || (rc = c->d_bvh_meshes.ensure(fs.bvh_meshes.size())))
        return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!fs.objs.empty())
        HIP_TRY(hipMemcpy(c->d_objs.p, fs.objs.data(), fs.objs.size() * sizeof(ObjRec), hipMemcpyHostToDevice));
    if (!fs.obj_pairs.empty())
        HIP_TRY(hipMemcpy(c->d_opairs.p, fs.obj_pairs.data(), fs.obj_pairs.size() * sizeof(ObjPairRec),
                          hipMemcpyHostToDevice));
    if (!fs.tri_pairs.empty())
        HIP_TRY(hipMemcpy(c->d_tris.p, fs.tri_pairs.data(), fs.tri_pairs.size() * sizeof(TriPairRec),
                          hipMemcpyHostToDevice));
    if (!fs.mats.empty())
        HIP_TRY(hipMemcpy(c->d_mats.p, fs.mats.data(), fs.mats.size() * sizeof(MatRec), hipMemcpyHostToDevice));
    if (!fs.tri_shade.empty())
        HIP_TRY(hipMemcpy(c->d_tshade.p, fs.tri_shade.data(), fs.tri_shade.size() * sizeof(TriShade),
                          hipMemcpyHostToDevice));
    if (!fs.bvh_nodes.empty())
        HIP_TRY(hipMemcpy(c->d_nodes.p, fs.bvh_nodes.data(), fs.bvh_nodes.size() * sizeof(BvhNode),
                          hipMemcpyHostToDevice));
    if (!fs.bvh_nodes4.empty())
        HIP_TRY(hipMemcpy(c->d_nodes4.p, fs.bvh_nodes4.data(), fs.bvh_nodes4.size() * sizeof(BvhNode4), hipMemcpyHostToDevice));
    c->scene.bvh_nodes4 = c->d_nodes4.p;
    c->scene.n_bvh_nodes4 = (uint32_t)fs.bvh_nodes4.size();
    if (!fs.sph_pairs.empty())
        HIP_TRY(hipMemcpy(c->d_sph.p, fs.sph_pairs.data(), fs.sph_pairs.size() * sizeof(SphPairRec), hipMemcpyHostToDevice));
    if (!fs.flat_pairs.empty())
        HIP_TRY(hipMemcpy(c->d_flat.p, fs.flat_pairs.data(), fs.flat_pairs.size() * sizeof(FlatPairRec), hipMemcpyHostToDevice));
    if (!fs.cand_pairs.empty())
        HIP_TRY(hipMemcpy(c->d_cand.p, fs.cand_pairs.data(), fs.cand_pairs.size() * sizeof(CandPairRec), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_rank_id.p, fs.rank_id.data(), fs.rank_id.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->scene.sph_pairs = c->d_sph.p;
    c->scene.flat_pairs = c->d_flat.p;
    c->scene.cand_pairs = c->d_cand.p;
    HIP_TRY(hipMemcpy(c->d_surf.p, fs.surf.data(), fs.surf.size() * sizeof(SurfRec), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_tri_rank.p, fs.tri_rank.data(), fs.tri_rank.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->scene.rank_id = c->d_rank_id.p;
    c->scene.tri_rank = c->d_tri_rank.p;
    if (!fs.bvh_meshes.empty())
        HIP_TRY(hipMemcpy(c->d_bvh_meshes.p, fs.bvh_meshes.data(), fs.bvh_meshes.size() * sizeof(BvhMeshRec), hipMemcpyHostToDevice));
    c->scene.bvh_meshes = c->d_bvh_meshes.p;
    c->scene.n_bvh_meshes = (uint32_t)fs.bvh_meshes.size();
    c->scene.surf = c->d_surf.p;
    c->scene.n_sph_pairs = (uint32_t)fs.sph_pairs.size();
    c->scene.n_flat_pairs = (uint32_t)fs.flat_pairs.size();
    c->scene.n_cand_pairs = (uint32_t)fs.cand_pairs.size();
    c->scene.n_other_pairs = fs.n_other_pairs;
    c->scene.n_flat_exact = fs.n_flat_exact;
    c->scene.nodes_in_lds_ok = c->tune.nodes_lds ? 1u : 0u;
    {  // glass deferral only where there is glass to defer
        bool has_glass = false;
        for (uint32_t i = 0; i < n_objs; ++i) has_glass = has_glass || objs[i].reflect_type == PT_REFRACT;
        c->scene.glass_defer_ok = (c->tune.glass_defer && has_glass) ? 1u : 0u;
    }
    c->cand_ok = fs.cand_ok;
    c->scene.cand_staged = 0u;
    c->scene.surf_staged = 0u;
    c->scene.surf_head = 0u;
    c->scene.walk_queue_cap = c->tune.walk_queue_cap;
#ifdef PT_WALK_STATS
    if (!c->scene.stats) {
        HIP_TRY(hipMalloc((void **)&c->scene.stats, 16 * sizeof(unsigned long long)));  // instrumented builds only: never freed
        HIP_TRY(hipMemset(c->scene.stats, 0, 16 * sizeof(unsigned long long)));
    }
#endif
#ifdef PT_PHASE_STATS
    if (!c->scene.phase_stats) {  // instrumented builds only: never freed
        HIP_TRY(hipMalloc((void **)&c->scene.phase_stats, (kPhCount * 3 + 2) * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(c->scene.phase_stats, 0, (kPhCount * 3 + 2) * sizeof(unsigned long long)));
    }
#endif
    c->n_bvh_nodes = (uint32_t)fs.bvh_nodes.size();
    c->scene.bvh_nodes = c->d_nodes.p;
    c->scene.n_bvh_nodes = c->n_bvh_nodes;
    c->scene.planar = 1u;
    {
        // bit 1: node and leaf references fit 16 bits (u16 traversal stacks); bit 0: nodes staged in LDS by every
        // workgroup.  With the walks done in dense waves (k_intersect<true>) occupancy is worth more than LDS-resident
        // nodes - mesh.json: 7.7 G bounces/s staged, 8.6 G from L1/L2 - so staging is opt-in (PT_BVH_LDS=1).
        const bool ref16 = c->n_bvh_nodes < 0x8000u && ((uint64_t)fs.bvh_pair_span << kBvhLeafBits) < 0x8000u;
        const bool stage = ref16 && c->n_bvh_nodes <= kBvhMaxLdsNodes && c->tune.bvh_lds;
        c->scene.bvh_in_lds = (ref16 ? 2u : 0u) | (stage ? 1u : 0u);
    }
    c->scene.cand_scan = cand_scan_for(c, 0u);
    c->scene.bvh_pair_base = fs.bvh_pair_base;
    c->scene.bvh_stack = fs.bvh_stack;
    c->scene.leaf_quorum = c->tune.leaf_quorum;  // mesh.json: 65 (all) 11.7, 32 12.5, 16 12.8, 8 12.8, 1 11.0 G bounces/s
    c->scene.objs = c->d_objs.p;
    c->scene.obj_pairs = c->d_opairs.p;
    c->scene.tri_pairs = c->d_tris.p;
    c->scene.mats = c->d_mats.p;
    c->scene.tri_shade = c->d_tshade.p;
    c->scene.n_objs = n_objs;
    c->scene.n_tris = n_tris;
    c->cam = *cam;
    c->h_objs.assign(objs, objs + n_objs);
    c->h_boxes.assign((size_t)12 * n_objs, pt_triangle{});
    for (uint32_t i = 0; i < n_objs; ++i)
        if (objs[i].kind == PT_MESH && objs[i].tri_count != 0u)
            host::mesh_bounding_box(tris + objs[i].tri_offset, objs[i].tri_count, &c->h_boxes[(size_t)12 * i]);
    c->boxes_dirty = true;
    c->has_scene = true;
    c->pass_rate = c->round_rate = 0.0;  // (another scene: the passes' length is measured again)
    c->pass_rate_kernel = nullptr;
    return PT_OK;
}

int pt_device_malloc(int device, size_t bytes, void **out) {
    if (!out) {
        set_error("out is NULL");
        return PT_ERR_INVALID;
    }
    *out = nullptr;
    if (device_count_quiet() <= 0) {
        set_error("no HIP device");
        return PT_ERR_NO_DEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    const hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // reported here: a refused allocation must not show up again as the next frame's error
        *out = nullptr;
        set_error(std::string("hipMalloc of ") + std::to_string(bytes) + " bytes: " + hipGetErrorString(e));
        return PT_ERR_HIP;
    }
    return PT_OK;
}

int pt_device_free(int device, void *p) {
    if (!p) return PT_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(p));
    return PT_OK;
}

int pt_device_download(int device, void *dst_host, const void *src_device, size_t bytes) {
    if (!dst_host || !src_device) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

const char *pt_ctx_pass_kernel(const pt_ctx *c, uint32_t flags) {
    if (!c || !c->has_scene) return nullptr;
    const bool no_bvh = (flags & PT_FLAG_NO_BVH) != 0u;
    const uint32_t n_nodes = no_bvh ? 0u : c->n_bvh_nodes;
    const bool bvh_ok = n_nodes == 0u || (!(c->scene.bvh_in_lds & 1u) && c->tune.pass_bvh);
    const bool one_kernel = bvh_ok && c->tune.pass_kernel && !(flags & PT_FLAG_SEPARATE_KERNELS);
    if (!one_kernel) return (n_nodes == 0u && cand_scan_for(c, flags)) ? "k_intersect_cand" : "k_intersect";
    if (cand_scan_for(c, flags)) return n_nodes != 0u ? "k_pass_cand_bvh" : "k_pass_cand";  // (k_pass_cand<.., BVH = true>)
    return n_nodes != 0u ? "k_pass_bvh" : "k_pass";
}

int pt_bvh_refs_fit(uint64_t n_bvh_nodes, uint64_t n_pair_records) { return host::bvh_refs_fit(n_bvh_nodes, n_pair_records) ? 1 : 0; }

int pt_ctx_set_memory_budget(pt_ctx *c, size_t bytes) {
    if (!c) {
        set_error("ctx is NULL");
        return PT_ERR_INVALID;
    }
    c->mem_budget = bytes;
    return PT_OK;
}

int pt_ctx_set_profiling(pt_ctx *c, int enabled) {
    if (!c) {
        set_error("ctx is NULL");
        return PT_ERR_INVALID;
    }
    c->profiling = enabled != 0;
    return PT_OK;
}

// progress relay for pipeline 0 / rank 0 of a call that renders on several contexts: its fractions are passed on, its
// "1.0" is not - the call is complete when EVERY pipeline has finished and the frame is assembled, and the parent says so
struct Below1 {
    pt_progress_fn cb;
    void *user;
    static void fn(void *self, float f) {
        Below1 *r = (Below1 *)self;
        if (f < 1.0f) r->cb(r->user, f);
    }
};

// n concurrent wavefront pipelines over the pixels of one call (PT_FLAG_PIPELINES)
static int render_pipelined(pt_ctx *c, const pt_config *cfg, uint32_t n, uint32_t ib, uint32_t ie, float *d_out,
                            const volatile uint8_t *cancel, pt_progress_fn cb, void *user, pt_stats *stats) {
    // the caller's partition of [ib, ie): chunks first, first+step, ... of C pixels (whole band: rows of the image)
    const uint32_t C = cfg->chunk_step > 1u ? cfg->chunk_pixels : cfg->width;
    const uint32_t first = cfg->chunk_step > 1u ? cfg->chunk_first : 0u;
    const uint32_t step = cfg->chunk_step > 1u ? cfg->chunk_step : 1u;
    while (c->pipes.size() < n) {
        pt_ctx *p = nullptr;
        int rc = pt_ctx_create(c->device, &p);
        if (rc) return rc;
        p->borrowed_scene = true;
        c->pipes.push_back(p);
        c->pipe_out.emplace_back();
    }
    std::vector<pt_config> cfgs(n, *cfg);
    std::vector<pt_stats> sts(n);
    std::vector<int> rcs(n, PT_OK);
    std::vector<std::string> errs(n);
    std::vector<uint32_t> own(n, 0);
    for (uint32_t j = 0; j < n; ++j) {
        pt_ctx *p = c->pipes[j];
        p->scene = c->scene;  // device pointers of the parent's scene tables (read-only)
        p->cam = c->cam;
        p->n_bvh_nodes = c->n_bvh_nodes;
        p->cand_ok = c->cand_ok;
        p->has_scene = true;
        p->profiling = c->profiling;
        p->mem_share = n;  // n sets of ray queues on this device at once
        p->mem_budget = c->mem_budget / n;
        cfgs[j].flags &= ~PT_FLAG_PIPELINES(15);
        cfgs[j].idx_begin = ib;
        cfgs[j].idx_end = ie;
        cfgs[j].chunk_pixels = C;
        cfgs[j].chunk_first = first + step * j;
        cfgs[j].chunk_step = step * n;
        own[j] = owned_pixels(&cfgs[j], ib, ie);
        int rc = c->pipe_out[j].ensure((size_t)own[j] * 3);
        if (rc) return rc;
    }
    std::vector<std::thread> th;
    Below1 relay{cb, user};
    for (uint32_t j = 0; j < n; ++j) {
        if (own[j] == 0u) continue;
        th.emplace_back([&, j]() {
            rcs[j] = pt_ctx_render(c->pipes[j], &cfgs[j], c->pipe_out[j].p, nullptr, cancel, (j == 0 && cb) ? &Below1::fn : nullptr,
                                   &relay, &sts[j]);
            if (rcs[j] != PT_OK) errs[j] = g_last_error;
        });
    }
    for (auto &t : th) t.join();
    HIP_TRY(hipSetDevice(c->device));
    int worst = PT_OK;
    for (uint32_t j = 0; j < n; ++j) {
        if (own[j] == 0u) continue;
        if (rcs[j] != PT_OK && rcs[j] != PT_CANCELLED) {
            set_error("pipeline " + std::to_string(j) + ": " + errs[j]);
            return rcs[j];
        }
        if (rcs[j] == PT_CANCELLED) worst = PT_CANCELLED;
        // pipeline j's k-th chunk is the call's (k*n + j)-th chunk
        launch_scatter_chunks(c->stream, c->pipe_out[j].p, d_out, own[j], C, n, j);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (stats) {
        for (uint32_t j = 0; j < n; ++j) {
            stats->ray_bounces += sts[j].ray_bounces;
            stats->samples += sts[j].samples;
            stats->intersect_rays += sts[j].intersect_rays;
            stats->intersect_launches += sts[j].intersect_launches;
            stats->passes += sts[j].passes;
            stats->ms_device = sts[j].ms_device > stats->ms_device ? sts[j].ms_device : stats->ms_device;
            stats->ms_intersect += sts[j].ms_intersect;
        }
    }
    if (worst == PT_CANCELLED) set_error("cancelled");
    return worst;
}

int pt_ctx_render(pt_ctx *c, const pt_config *cfg, void *d_out_rgb, void *hip_stream, const volatile uint8_t *cancel,
                  pt_progress_fn cb, void *user, pt_stats *stats) {
    if (!c || !d_out_rgb) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    if (!c->has_scene) {
        set_error("no scene set");
        return PT_ERR_INVALID;
    }
    uint32_t ib = 0, ie = 0;
    int rc = check_cfg(cfg, &ib, &ie);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t n_pipes = (cfg->flags >> 8) & 15u;
    if (n_pipes > 1u && cfg->backend == PT_BACKEND_WAVEFRONT) {
        if (n_pipes > 8u) {
            set_error("at most 8 concurrent pipelines");
            return PT_ERR_INVALID;
        }
        if (stats) memset(stats, 0, sizeof *stats);
        const double t0p = now_ms();
        c->scene.n_bvh_nodes = (cfg->flags & PT_FLAG_NO_BVH) ? 0u : c->n_bvh_nodes;
        c->scene.planar = (cfg->flags & PT_FLAG_NO_BVH) ? 0u : 1u;
        rc = render_pipelined(c, cfg, n_pipes, ib, ie, (float *)d_out_rgb, cancel, cb, user, stats);
        c->scene.n_bvh_nodes = c->n_bvh_nodes;
        c->scene.planar = 1u;
        if (cb && rc == PT_OK) cb(user, 1.0f);  // every pipeline has finished and the chunks are in place
        if (stats) stats->ms_total = now_ms() - t0p;
        return rc;
    }
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const FrameParams F = make_frame(c, cfg, ib, ie);
    if (stats) memset(stats, 0, sizeof *stats);
    if (F.npix == 0u) return PT_OK;  // this rank owns no chunk of the band
    // PT_FLAG_NO_BVH: scan meshes triangle by triangle as the reference does (same result, for A/B checks)
    c->scene.n_bvh_nodes = (cfg->flags & PT_FLAG_NO_BVH) ? 0u : c->n_bvh_nodes;
    c->scene.planar = (cfg->flags & PT_FLAG_NO_BVH) ? 0u : 1u;
    c->scene.cand_scan = cand_scan_for(c, cfg->flags);
    const double t0 = now_ms();
    // Parts.  The wavefront kernels are tuned for streams of a few dozen pixels with a couple of thousand rays per pass
    // (accumulators, ray slots and deferral buffers share 40 KB of LDS per workgroup); a call of many millions of pixels
    // (4096^2: BASELINE config 5) would need streams of 1024 pixels or passes of hundreds of GB.  Such a call is
    // rendered as consecutive parts of about a million call-local pixels, each exactly like a frame of that size - the
    // RNG is keyed on the global pixel index, so the image is the same bits.  Progress counts finished parts; a
    // cancelled call leaves the parts it did not start black (the reference's unrendered pixels are black too,
    // mod.rs:1003-1016) and the part in progress averaged over its accumulated samples.
    const uint32_t total = F.npix;
    const bool in_parts = cfg->backend == PT_BACKEND_WAVEFRONT && total > (3u << 19);
    const uint32_t part_px = in_parts ? (1u << 20) : total;
    const uint32_t n_parts = (total + part_px - 1u) / part_px;
    struct Relay {
        pt_progress_fn cb;
        void *user;
        float base, scale;
        static void fn(void *self, float f) {
            Relay *r = (Relay *)self;
            if (f < 1.0f) r->cb(r->user, r->base + r->scale * f);  // a part's completion is reported by the next part / the end
        }
    } relay{cb, user, 0.0f, 1.0f / (float)n_parts};
    c->live_out = (float *)d_out_rgb;
    c->live_total = total;
    c->live_stream = st;
    c->cb_last_ms = now_ms();
    const double cb_every_ms = cfg->progress_ms == PT_PROGRESS_EVERY_PASS ? 0.0 : (cfg->progress_ms ? (double)cfg->progress_ms : 500.0);
    rc = PT_OK;
    for (uint32_t part = 0; part < n_parts && rc == PT_OK; ++part) {
        // a part boundary is a progress point of its own (a part of one or two passes makes no callback from inside); a cancel
        // raised there is seen by the part's first pass, which leaves it and the parts behind it black
        if (cb && part > 0u) {
            const double t_now = now_ms();
            if (t_now - c->cb_last_ms >= cb_every_ms) {
                c->cb_last_ms = t_now;
                cb(user, (float)part / (float)n_parts);
            }
        }
        FrameParams Fp = F;
        Fp.k_begin = part * part_px;
        Fp.npix = (total - Fp.k_begin) < part_px ? (total - Fp.k_begin) : part_px;
        float *out_p = (float *)d_out_rgb + (size_t)Fp.k_begin * 3;
        c->live_k0 = Fp.k_begin;
        c->live_npix = Fp.npix;
        c->live_spp_issued = 0;
        relay.base = (float)part / (float)n_parts;
        pt_stats ps;
        memset(&ps, 0, sizeof ps);
        pt_progress_fn part_cb = cb ? (n_parts > 1u ? &Relay::fn : cb) : nullptr;
        void *part_user = n_parts > 1u ? (void *)&relay : user;
        if (cfg->backend == PT_BACKEND_WAVEFRONT)
            rc = render_wavefront(c, cfg, Fp, st, cancel, part_cb, part_user, &ps);
        else
            rc = render_mega(c, cfg, Fp, st, cancel, part_cb, part_user, &ps);
        if (rc == PT_OK || rc == PT_CANCELLED) {
            // A cancelled part resolves what was accumulated by the samples per pixel that were accumulated
            // (live_spp_issued, also reported through stats->samples): every pixel at full brightness over fewer
            // samples - the same picture pt_ctx_snapshot gives.  (The reference's partial image has finished pixels at
            // full spp and the rest black; a GPU pass covers every pixel, so "fewer samples everywhere" is its
            // counterpart.)  Nothing accumulated yet: all zero, as the reference's untouched `pixels` vector.
            const uint32_t spp_done = rc == PT_OK ? cfg->spp : c->live_spp_issued;
            if (spp_done != 0u)
                launch_resolve(st, c->acc.p, out_p, Fp.npix, spp_done, c->live_streams, c->live_m);
            else
                HIP_TRY(hipMemsetAsync(out_p, 0, (size_t)Fp.npix * 3 * sizeof(float), st));
            if (rc == PT_CANCELLED && Fp.k_begin + Fp.npix < total)  // the parts that were never started
                HIP_TRY(hipMemsetAsync(out_p + (size_t)Fp.npix * 3, 0, (size_t)(total - Fp.k_begin - Fp.npix) * 3 * sizeof(float), st));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));
        }
        if (stats) {
            stats->ray_bounces += ps.ray_bounces;
            stats->samples += ps.samples;
            stats->intersect_rays += ps.intersect_rays;
            stats->intersect_launches += ps.intersect_launches;
            stats->passes += ps.passes;
            stats->ms_device += ps.ms_device;
            stats->ms_intersect += ps.ms_intersect;
        }
    }
    if (cb && rc == PT_OK) cb(user, 1.0f);
    if (stats) stats->ms_total = now_ms() - t0;
    c->scene.n_bvh_nodes = c->n_bvh_nodes;
    c->scene.planar = 1u;
    c->scene.cand_scan = cand_scan_for(c, 0u);
    c->live_npix = 0;
    c->live_out = nullptr;
    return rc;
}

int pt_ctx_radiance(pt_ctx *c, const float o[3], const float d[3], uint32_t depth, uint32_t n_samples, uint64_t seed,
                    uint32_t pixel, uint32_t backend, uint32_t flags, float out_rgb[3], pt_stats *stats) {
    if (!c || !o || !d || !out_rgb) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    if (!c->has_scene) {
        set_error("no scene set");
        return PT_ERR_INVALID;
    }
    if (n_samples == 0u || n_samples > (1u << 24) || depth >= (uint32_t)kMaxDepth ||
        (backend != PT_BACKEND_WAVEFRONT && backend != PT_BACKEND_MEGAKERNEL) || ((flags >> 8) & 15u) > 1u) {
        set_error("n_samples outside 1..2^24, depth >= MAX_DEPTH, unknown backend, or PT_FLAG_PIPELINES (not offered for a one-ray call)");
        return PT_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->device));
    // a frame of ONE pixel whose samples all start with the given ray: the pixel index is only the RNG counter
    pt_config cfg{};
    cfg.width = 1;
    cfg.height = 1;
    cfg.spp = n_samples;
    cfg.backend = backend;
    cfg.seed = seed;
    cfg.flags = flags;
    FrameParams F = make_frame(c, &cfg, 0u, 1u);
    F.idx_begin = pixel;
    F.npix = 1;
    F.probe = 1u;
    F.depth0 = depth;
    F.probe_ox = o[0];
    F.probe_oy = o[1];
    F.probe_oz = o[2];
    F.probe_dx = d[0];
    F.probe_dy = d[1];
    F.probe_dz = d[2];
    if (stats) memset(stats, 0, sizeof *stats);
    c->scene.n_bvh_nodes = (flags & PT_FLAG_NO_BVH) ? 0u : c->n_bvh_nodes;
    c->scene.planar = (flags & PT_FLAG_NO_BVH) ? 0u : 1u;
    c->scene.cand_scan = cand_scan_for(c, flags);
    const double t0 = now_ms();
    pt_stats ps;
    memset(&ps, 0, sizeof ps);
    hipStream_t st = c->stream;
    int rc = backend == PT_BACKEND_WAVEFRONT ? render_wavefront(c, &cfg, F, st, nullptr, nullptr, nullptr, &ps)
                                             : render_mega(c, &cfg, F, st, nullptr, nullptr, nullptr, &ps);
    c->scene.n_bvh_nodes = c->n_bvh_nodes;
    c->scene.planar = 1u;
    c->scene.cand_scan = cand_scan_for(c, 0u);
    if (rc != PT_OK) return rc;
    DevBuf<float> d_out;
    if ((rc = d_out.ensure(3))) return rc;
    launch_resolve(st, c->acc.p, d_out.p, 1u, n_samples, c->live_streams, c->live_m, false);  // the mean, not clamped
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(out_rgb, d_out.p, 3 * sizeof(float), hipMemcpyDeviceToHost);
    d_out.release();
    if (e != hipSuccess) {
        set_error(std::string("pt_ctx_radiance: ") + hipGetErrorString(e));
        return PT_ERR_HIP;
    }
    if (stats) {
        *stats = ps;
        stats->ms_total = now_ms() - t0;
    }
    return PT_OK;
}

int pt_ctx_intersect_streams(pt_ctx *c, const float *o, const float *d, uint32_t n, uint32_t flags, float *t, int32_t *id) {
    if (!c || !o || !d || !t || !id) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    if (!c->has_scene) {
        set_error("no scene set");
        return PT_ERR_INVALID;
    }
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(c->device));
    // the rays as ray streams of 4096 slots (the queue layout of the wavefront pipeline), one workgroup per stream
    const uint32_t cap = 4096u, K = (n + cap - 1u) / cap;
    std::vector<char> hq(queue_bytes(K, cap), 0);  // slice b: [od0: cap x 16][tp: cap x 16][od1: cap x 8]
    std::vector<uint32_t> hc(K);
    for (uint32_t i = 0; i < n; ++i) {
        char *slice = hq.data() + (size_t)(i / cap) * cap * kRayBytes;
        const uint32_t j = i % cap;
        const float4 a = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i]);
        const float2 bq = make_float2(d[3 * i + 1], d[3 * i + 2]);
        memcpy(slice + (size_t)j * 16u, &a, sizeof a);
        memcpy(slice + (size_t)cap * 32u + (size_t)j * 8u, &bq, sizeof bq);
    }
    for (uint32_t b = 0; b < K; ++b) hc[b] = (n - b * cap) < cap ? (n - b * cap) : cap;
    DevBuf<char> d0;
    DevBuf<float2> dh;
    DevBuf<uint32_t> dc;
    DevBuf<unsigned long long> dr;
    int rc;
    if ((rc = d0.ensure(hq.size())) || (rc = dh.ensure((size_t)K * cap)) || (rc = dc.ensure(K)) || (rc = dr.ensure(K))) return rc;
    hipStream_t st = c->stream;
    hipError_t e = hipMemcpyAsync(d0.p, hq.data(), hq.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dc.p, hc.data(), K * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(dr.p, 0, K * sizeof(unsigned long long), st);
    if (e == hipSuccess) {
        DevScene S = c->scene;
        S.n_bvh_nodes = (flags & PT_FLAG_NO_BVH) ? 0u : c->n_bvh_nodes;
        S.planar = (flags & PT_FLAG_NO_BVH) ? 0u : 1u;
        S.cand_scan = cand_scan_for(c, flags);
        RayQueue q;
        q.buf = d0.p;
        launch_intersect(st, K, S, q, dh.p, dc.p, cap, dr.p);
        e = hipGetLastError();
    }
    std::vector<float2> hh((size_t)K * cap);
    if (e == hipSuccess) e = hipMemcpyAsync(hh.data(), dh.p, hh.size() * sizeof(float2), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    d0.release();
    dh.release();
    dc.release();
    dr.release();
    if (e != hipSuccess) {
        set_error(std::string("pt_ctx_intersect_streams: ") + hipGetErrorString(e));
        return PT_ERR_HIP;
    }
    for (uint32_t i = 0; i < n; ++i) {
        t[i] = hh[i].x;
        memcpy(&id[i], &hh[i].y, sizeof(int32_t));
    }
    return PT_OK;
}

int pt_ctx_intersect(pt_ctx *c, const float *o, const float *d, uint32_t n, float *t, int32_t *object_id,
                     int32_t *tri_id, float *x, float *normal) {
    if (!c || !o || !d) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    if (!c->has_scene) {
        set_error("no scene set");
        return PT_ERR_INVALID;
    }
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->q_o.ensure(3 * (size_t)n)) || (rc = c->q_d.ensure(3 * (size_t)n)) || (rc = c->q_t.ensure(n)) ||
        (rc = c->q_x.ensure(3 * (size_t)n)) || (rc = c->q_n.ensure(3 * (size_t)n)) || (rc = c->q_oid.ensure(n)) ||
        (rc = c->q_tid.ensure(n)))
        return rc;
    hipStream_t st = c->stream;
    HIP_TRY(hipMemcpyAsync(c->q_o.p, o, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->q_d.p, d, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    launch_query(st, c->scene, c->q_o.p, c->q_d.p, n, c->q_t.p, c->q_oid.p, c->q_tid.p, c->q_x.p, c->q_n.p);
    HIP_TRY(hipGetLastError());
    if (t) HIP_TRY(hipMemcpyAsync(t, c->q_t.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
    if (object_id) HIP_TRY(hipMemcpyAsync(object_id, c->q_oid.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (tri_id) HIP_TRY(hipMemcpyAsync(tri_id, c->q_tid.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (x) HIP_TRY(hipMemcpyAsync(x, c->q_x.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
    if (normal) HIP_TRY(hipMemcpyAsync(normal, c->q_n.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PT_OK;
}

// device form of the bounding boxes (rebuilt after pt_ctx_set_scene / pt_ctx_set_mesh_bounds)
static int upload_boxes(pt_ctx *c) {
    if (!c->boxes_dirty) return PT_OK;
    const size_t n_objs = c->h_objs.size();
    std::vector<TriPairRec> recs(6 * (n_objs ? n_objs : 1), TriPairRec{});
    for (size_t i = 0; i < n_objs; ++i)
        if (c->h_objs[i].kind == PT_MESH) host::box_pair_records(&c->h_boxes[12 * i], c->h_objs[i].position, &recs[6 * i]);
    int rc = c->d_boxes.ensure(recs.size());
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->d_boxes.p, recs.data(), recs.size() * sizeof(TriPairRec), hipMemcpyHostToDevice));
    c->boxes_dirty = false;
    return PT_OK;
}

static int bounds_query(pt_ctx *c, uint32_t mode, uint32_t object, const float *o, const float *d, uint32_t n, int32_t *hit,
                        float *t, float *x, float *normal, int32_t *object_id) {
    if (!c || !o || !d) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    if (!c->has_scene) {
        set_error("no scene set");
        return PT_ERR_INVALID;
    }
    if (mode == 0u && object >= c->h_objs.size()) {
        set_error("object index out of range");
        return PT_ERR_INVALID;
    }
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = upload_boxes(c);
    if (rc) return rc;
    if ((rc = c->q_o.ensure(3 * (size_t)n)) || (rc = c->q_d.ensure(3 * (size_t)n)) || (rc = c->q_t.ensure(n)) ||
        (rc = c->q_x.ensure(3 * (size_t)n)) || (rc = c->q_n.ensure(3 * (size_t)n)) || (rc = c->q_oid.ensure(n)) ||
        (rc = c->q_tid.ensure(n)))
        return rc;
    hipStream_t st = c->stream;
    HIP_TRY(hipMemcpyAsync(c->q_o.p, o, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->q_d.p, d, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    launch_bounds(st, c->scene, c->d_boxes.p, mode, object, c->q_o.p, c->q_d.p, n, c->q_tid.p, c->q_t.p, c->q_x.p, c->q_n.p,
                  c->q_oid.p);
    HIP_TRY(hipGetLastError());
    if (hit) HIP_TRY(hipMemcpyAsync(hit, c->q_tid.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (t) HIP_TRY(hipMemcpyAsync(t, c->q_t.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
    if (x) HIP_TRY(hipMemcpyAsync(x, c->q_x.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
    if (normal) HIP_TRY(hipMemcpyAsync(normal, c->q_n.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
    if (object_id) HIP_TRY(hipMemcpyAsync(object_id, c->q_oid.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PT_OK;
}

int pt_ctx_intersect_bounds(pt_ctx *c, uint32_t object, const float *o, const float *d, uint32_t n, int32_t *hit, float *t,
                            float *x, float *normal) {
    return bounds_query(c, 0u, object, o, d, n, hit, t, x, normal, nullptr);
}

int pt_ctx_orbit_point(pt_ctx *c, const float *o, const float *d, uint32_t n, int32_t *found, float *point,
                       int32_t *object_id, float *t) {
    return bounds_query(c, 1u, 0u, o, d, n, found, t, point, nullptr, object_id);
}

int pt_ctx_set_mesh_bounds(pt_ctx *c, uint32_t object, const pt_triangle box[12]) {
    if (!c || !box) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    if (!c->has_scene || object >= c->h_objs.size() || c->h_objs[object].kind != PT_MESH) {
        set_error("no scene set, or object is not a mesh of it");
        return PT_ERR_INVALID;
    }
    memcpy(&c->h_boxes[(size_t)12 * object], box, 12 * sizeof(pt_triangle));
    c->boxes_dirty = true;
    return PT_OK;
}

int pt_mesh_bounding_box(const pt_triangle *tris, uint32_t n_tris, pt_triangle out[12]) {
    if (!tris || !out || n_tris == 0) {
        set_error("NULL argument or empty mesh");
        return PT_ERR_INVALID;
    }
    host::mesh_bounding_box(tris, n_tris, out);
    return PT_OK;
}

void pt_host_sincos(float y, float *s, float *c) { sincos_f32(y, s, c); }

int pt_ctx_snapshot(pt_ctx *c, void *d_out_rgb, uint32_t *spp_done) {
    if (!c || !d_out_rgb) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    if (c->live_npix == 0 || c->live_spp_issued == 0) {
        set_error("no frame in progress on this context (call from the progress callback of pt_ctx_render; under "
                  "PT_FLAG_PIPELINES the accumulators live in child contexts and no snapshot is offered)");
        return PT_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->device));
    // stream order: the resolve runs after every pass issued so far, i.e. over live_spp_issued samples per pixel.  A call
    // rendered in parts: the finished parts are copied from the call's own output, the part in progress is resolved,
    // the parts not started are black.
    float *snap = (float *)d_out_rgb;
    hipStream_t st = c->live_stream;
    if (c->live_k0 != 0u && c->live_out && c->live_out != snap)
        HIP_TRY(hipMemcpyAsync(snap, c->live_out, (size_t)c->live_k0 * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_resolve(st, c->acc.p, snap + (size_t)c->live_k0 * 3, c->live_npix, c->live_spp_issued, c->live_streams, c->live_m);
    const uint32_t done = c->live_k0 + c->live_npix;
    if (done < c->live_total)
        HIP_TRY(hipMemsetAsync(snap + (size_t)done * 3, 0, (size_t)(c->live_total - done) * 3 * sizeof(float), st));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    if (spp_done) *spp_done = c->live_spp_issued;
    return PT_OK;
}

int pt_ctx_numerics_probe(pt_ctx *c, const float *in, uint32_t n, float *out_sin, float *out_cos, float *out_sqrt,
                          float *out_rcp, uint32_t *out_philox) {
    if (!c || !in || !out_sin || !out_cos || !out_sqrt || !out_rcp || !out_philox || n == 0) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->device));
    DevBuf<float> d_in, d_s, d_c, d_q, d_r;
    DevBuf<uint32_t> d_p;
    int rc;
    if ((rc = d_in.ensure(n)) || (rc = d_s.ensure(n)) || (rc = d_c.ensure(n)) || (rc = d_q.ensure(n)) ||
        (rc = d_r.ensure(n)) || (rc = d_p.ensure(4 * (size_t)n)))
        return rc;
    hipError_t e = hipMemcpy(d_in.p, in, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_numerics(c->stream, d_in.p, n, d_s.p, d_c.p, d_q.p, d_r.p, d_p.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out_sin, d_s.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_cos, d_c.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_sqrt, d_q.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_rcp, d_r.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_philox, d_p.p, 4 * (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    d_in.release();
    d_s.release();
    d_c.release();
    d_q.release();
    d_r.release();
    d_p.release();
    if (e != hipSuccess) {
        set_error(std::string("numerics probe: ") + hipGetErrorString(e));
        return PT_ERR_HIP;
    }
    return PT_OK;
}

int pt_ctx_numerics_sweep(pt_ctx *c, uint64_t out[4]) {
    if (!c || !out) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->device));
    DevBuf<unsigned long long> d;
    int rc = d.ensure(4);
    if (rc) return rc;
    hipError_t e = hipMemsetAsync(d.p, 0, 4 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) {
        launch_numerics_sweep(c->stream, d.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    unsigned long long h[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost);
    d.release();
    if (e != hipSuccess) {
        set_error(std::string("numerics sweep: ") + hipGetErrorString(e));
        return PT_ERR_HIP;
    }
    for (int i = 0; i < 4; ++i) out[i] = h[i];
    return PT_OK;
}

int pt_ctx_sincos_sweep(pt_ctx *c, uint64_t out[2]) {
    if (!c || !out) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->device));
    // the host instantiation of the shared numerics header on the 2^24 arguments shade_surface can form (mod.rs:691,703)
    const uint32_t n = 1u << 24;
    std::vector<uint32_t> hs(n), hc(n);
    for (uint32_t k = 0; k < n; ++k) {
        const float r1 = (2.0f * 3.141592653589793f) * unit_f32(k << 8);
        float s, co;
        sincos_f32(r1, &s, &co);
        memcpy(&hs[k], &s, 4);
        memcpy(&hc[k], &co, 4);
    }
    DevBuf<uint32_t> ds, dc;
    DevBuf<unsigned long long> d;
    int rc;
    if ((rc = ds.ensure(n)) || (rc = dc.ensure(n)) || (rc = d.ensure(2))) return rc;
    hipError_t e = hipMemcpy(ds.p, hs.data(), (size_t)n * 4u, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dc.p, hc.data(), (size_t)n * 4u, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(d.p, 0, 2 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) {
        launch_sincos_sweep(c->stream, ds.p, dc.p, d.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    unsigned long long h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost);
    ds.release();
    dc.release();
    d.release();
    if (e != hipSuccess) {
        set_error(std::string("sincos sweep: ") + hipGetErrorString(e));
        return PT_ERR_HIP;
    }
    out[0] = h[0];
    out[1] = h[1];
    return PT_OK;
}

int pt_ctx_primary_rays(pt_ctx *c, uint32_t width, uint32_t height, uint64_t seed, const uint32_t *pixel, const uint32_t *sample,
                        uint32_t n, uint32_t form, float *o, float *d) {
    if (!c || !pixel || !sample || !o || !d || n == 0 || form > 1u) {
        set_error("NULL argument, n == 0 or an unknown form");
        return PT_ERR_INVALID;
    }
    if (!c->has_scene) {
        set_error("no scene (the camera comes with it): call pt_ctx_set_scene first");
        return PT_ERR_INVALID;
    }
    pt_config cfg{};
    cfg.width = width;
    cfg.height = height;
    cfg.spp = 1;
    cfg.seed = seed;
    uint32_t ib = 0, ie = 0;
    int rc = check_cfg(&cfg, &ib, &ie);
    if (rc) return rc;
    for (uint32_t i = 0; i < n; ++i)
        if (pixel[i] >= (uint64_t)width * height || sample[i] >= (1u << 24)) {
            set_error("pixel index outside the frame or sample >= 2^24");
            return PT_ERR_INVALID;
        }
    HIP_TRY(hipSetDevice(c->device));
    const FrameParams F = make_frame(c, &cfg, ib, ie);
    DevBuf<uint32_t> dp, dsm;
    DevBuf<float> d_o, d_d;
    if ((rc = dp.ensure(n)) || (rc = dsm.ensure(n)) || (rc = d_o.ensure(3 * (size_t)n)) || (rc = d_d.ensure(3 * (size_t)n))) return rc;
    hipError_t e = hipMemcpy(dp.p, pixel, (size_t)n * 4u, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dsm.p, sample, (size_t)n * 4u, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_primary_rays(c->stream, F, dp.p, dsm.p, n, form, d_o.p, d_d.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(o, d_o.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(d, d_d.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    dp.release();
    dsm.release();
    d_o.release();
    d_d.release();
    if (e != hipSuccess) {
        set_error(std::string("primary rays: ") + hipGetErrorString(e));
        return PT_ERR_HIP;
    }
    return PT_OK;
}

// one band on one device into the host framebuffer
static int render_band_to_host(int dev, const pt_config *cfg, const pt_camera *cam, const pt_object *objs,
                               uint32_t n_objs, const pt_triangle *tris, uint32_t n_tris, float *out_rgb,
                               const volatile uint8_t *cancel, pt_progress_fn cb, void *user, pt_stats *stats,
                               std::string *err, uint32_t mem_share = 1) {
    uint32_t ib = 0, ie = 0;
    int rc = check_cfg(cfg, &ib, &ie);
    if (stats) memset(stats, 0, sizeof *stats);
    const uint32_t own = rc ? 0u : owned_pixels(cfg, ib, ie);
    if (!rc && own == 0u) return PT_OK;  // this rank owns no chunk of the band: nothing was created yet
    pt_ctx *c = nullptr;
    if (!rc) rc = pt_ctx_create(dev, &c);
    if (!rc) c->mem_share = mem_share;
    if (!rc) rc = pt_ctx_set_scene(c, cam, objs, n_objs, tris, n_tris);
    float *d_out = nullptr;
    const size_t nfl = (size_t)own * 3;
    if (!rc) {
        hipError_t e = hipMalloc((void **)&d_out, nfl * sizeof(float));
        if (e != hipSuccess) {
            set_error(std::string("hipMalloc(out): ") + hipGetErrorString(e));
            rc = PT_ERR_HIP;
        }
    }
    if (!rc) {
        rc = pt_ctx_render(c, cfg, d_out, nullptr, cancel, cb, user, stats);
        if (rc == PT_OK || rc == PT_CANCELLED) {
            hipError_t e = hipSuccess;
            if (cfg->chunk_step <= 1u) {
                e = hipMemcpy(out_rgb + (size_t)ib * 3, d_out, nfl * sizeof(float), hipMemcpyDeviceToHost);
            } else {  // chunks go back to their places in the frame
                std::vector<float> tmp(nfl);
                e = hipMemcpy(tmp.data(), d_out, nfl * sizeof(float), hipMemcpyDeviceToHost);
                const uint64_t span = ie - ib, C = cfg->chunk_pixels;
                size_t at = 0;
                for (uint64_t ck = cfg->chunk_first; e == hipSuccess && ck * C < span; ck += cfg->chunk_step) {
                    const uint64_t lo = ck * C, hi = (lo + C < span) ? lo + C : span;
                    memcpy(out_rgb + ((size_t)ib + lo) * 3, tmp.data() + at, (size_t)(hi - lo) * 3 * sizeof(float));
                    at += (size_t)(hi - lo) * 3;
                }
            }
            if (e != hipSuccess) {
                set_error(std::string("hipMemcpy(out): ") + hipGetErrorString(e));
                rc = PT_ERR_HIP;
            }
        }
    }
    if (rc && err) *err = g_last_error;  // the message lives in this thread's slot
    if (d_out) (void)hipFree(d_out);
    if (c) pt_ctx_destroy(c);
    return rc;
}

int pt_render(const pt_config *cfg, const pt_camera *cam, const pt_object *objs, uint32_t n_objs,
              const pt_triangle *tris, uint32_t n_tris, float *out_rgb, const volatile uint8_t *cancel,
              pt_progress_fn cb, void *user, pt_stats *stats) {
    if (!out_rgb) {
        set_error("out_rgb is NULL");
        return PT_ERR_INVALID;
    }
    int dev = 0;
    if (const char *e = getenv("PT_DEVICE")) dev = atoi(e);
    return render_band_to_host(dev, cfg, cam, objs, n_objs, tris, n_tris, out_rgb, cancel, cb, user, stats, nullptr);
}

int pt_render_multi(const pt_config *cfg, uint32_t n_ranks, const pt_camera *cam, const pt_object *objs,
                    uint32_t n_objs, const pt_triangle *tris, uint32_t n_tris, float *out_rgb,
                    const volatile uint8_t *cancel, pt_progress_fn cb, void *user, pt_stats *stats) {
    if (!out_rgb || n_ranks == 0 || n_ranks > 64) {
        set_error("out_rgb is NULL or n_ranks outside 1..64");
        return PT_ERR_INVALID;
    }
    uint32_t ib = 0, ie = 0;
    int rc = check_cfg(cfg, &ib, &ie);
    if (rc) return rc;
    const int n_dev = device_count_quiet();
    if (n_dev <= 0) {
        set_error("no HIP device: libptrace_hip has no CPU fallback");
        return PT_ERR_NO_DEVICE;
    }
    // rank r renders the chunks r, r+n_ranks, ... of [ib, ie) (one chunk = one image row) on device r mod n_dev:
    // interleaving evens out the cost per rank; pixels are independent and the RNG is keyed on the global pixel
    // index, so the image does not depend on n_ranks (mod.rs:1021-1023)
    std::vector<pt_config> cfgs(n_ranks, *cfg);
    std::vector<pt_stats> sts(n_ranks);
    std::vector<int> rcs(n_ranks, PT_OK);
    std::vector<std::string> errs(n_ranks);
    std::vector<std::thread> th;
    const double t0 = now_ms();
    Below1 relay{cb, user};
    for (uint32_t r = 0; r < n_ranks; ++r) {
        cfgs[r].idx_begin = ib;
        cfgs[r].idx_end = ie;
        if (n_ranks > 1) {
            cfgs[r].chunk_pixels = cfg->width;
            cfgs[r].chunk_first = r;
            cfgs[r].chunk_step = n_ranks;
        }
        if (owned_pixels(&cfgs[r], ib, ie) == 0u) continue;  // more ranks than chunks
        // ranks r, r + n_dev, ... share device r % n_dev: each takes its share of that device's memory for its ray queues
        const uint32_t dev = r % (uint32_t)n_dev;
        const uint32_t on_dev = (n_ranks - dev + (uint32_t)n_dev - 1u) / (uint32_t)n_dev;
        th.emplace_back([&, r, dev, on_dev]() {
            rcs[r] = render_band_to_host((int)dev, &cfgs[r], cam, objs, n_objs, tris, n_tris, out_rgb, cancel,
                                         (r == 0 && cb) ? &Below1::fn : nullptr, &relay, &sts[r], &errs[r], on_dev);
        });
    }
    for (auto &t : th) t.join();
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (uint32_t r = 0; r < n_ranks; ++r) {
            stats->ray_bounces += sts[r].ray_bounces;
            stats->samples += sts[r].samples;
            stats->intersect_rays += sts[r].intersect_rays;
            stats->intersect_launches += sts[r].intersect_launches;
            stats->passes += sts[r].passes;
            stats->ms_device = sts[r].ms_device > stats->ms_device ? sts[r].ms_device : stats->ms_device;
            stats->ms_intersect += sts[r].ms_intersect;
        }
        stats->ms_total = now_ms() - t0;
    }
    for (uint32_t r = 0; r < n_ranks; ++r)
        if (rcs[r] != PT_OK) {
            set_error("rank " + std::to_string(r) + ": " + errs[r]);
            return rcs[r];
        }
    if (cb) cb(user, 1.0f);  // every rank has finished and its rows are in out_rgb
    return PT_OK;
}

#ifdef PT_WALK_STATS
// instrumented builds only (tools/walk_stats.py): read and clear the walk counters
int pt_debug_walk_stats(pt_ctx *c, unsigned long long *out16) {
    if (!c || !c->scene.stats) return PT_ERR_INVALID;
    hipDeviceSynchronize();
    hipMemcpy(out16, c->scene.stats, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    hipMemset(c->scene.stats, 0, 16 * sizeof(unsigned long long));
    return PT_OK;
}
#endif

#ifdef PT_PHASE_STATS
// instrumented builds only (tools/phase_budget.py): read and clear the phase counters; returns the number of phases
int pt_debug_phase_stats(pt_ctx *c, unsigned long long *out, uint32_t cap) {
    if (!c || !c->scene.phase_stats || cap < kPhCount * 3 + 2) return PT_ERR_INVALID;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out, c->scene.phase_stats, (kPhCount * 3 + 2) * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipMemset(c->scene.phase_stats, 0, (kPhCount * 3 + 2) * sizeof(unsigned long long));
    return (int)kPhCount;
}
#endif

}  // extern "C"
