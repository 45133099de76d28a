#!/usr/bin/env python3
"""bench.py — ray-bounces/s of the radiance() hot path on MI355X.

One step = one whole frame of BASELINE.json's metric workload (scenes/cornell.json, 1024x768, 4096 spp by
default) rendered by the wavefront HIP pipeline through the C ABI, the scene already resident in HBM.
With N ranks (one process per GPU) rank r renders image rows r, r+N, ..., and one RCCL all-gather assembles the image:
total work is fixed, so scaling is "strong".  value = ray bounces of all ranks / max-over-ranks time.
`python bench.py --gpus N` starts the N ranks itself (a child `python -m torch.distributed.run --nproc-per-node N bench.py
...`; the parent never touches a GPU and only relays the children's output and return code); launched under
torch.distributed.run (WORLD_SIZE set) it is one of the ranks.  --gather torch (default): the collective is
torch.distributed's all_gather_into_tensor (backend "nccl" = RCCL); --gather abi: the C ABI's own pt_comm_gather_frame
(ncclAllGather + un-permute kernel inside libptrace_hip.so), torch.distributed only carrying the 128-byte id.

The JSON line also carries
  roofline      the dominant kernel, timed live with HIP events around each of its launches in the timed region,
                against the 8 TB/s HBM peak.  Scenes without BVH meshes (the default) run a whole pass per launch
                (k_pass_cand): its algorithmic traffic is the ray queue - every ray of depth >= 1 is written once and
                read once, 40 B each way (origin, direction, throughput, bookkeeping word); primary rays are made in
                registers and hit records never leave them (k_pass_bvh, scenes with a BVH: the primaries go through the
                queue too).  k_pass_cand keeps that queue as one small stack per wave, which stays in L2: `traffic` -
                the HBM bytes the PMC counters saw - is then far BELOW the algorithmic bytes (9.4 against 70.8 B per
                bounce on the bench scene).  With PT_FLAG_SEPARATE_KERNELS the figure is k_intersect's (24 B ray in +
                8 B hit out).
                "binds": false - HBM is not what limits these kernels;
  valu_roofline what does: VALU instruction issue, priced with the per-class cycle costs measured on the box
                (tools/valu_issue_bench.hip -> profiles/r02_valu_issue_costs.json), the kernel's dynamic instruction count
                (PMC) and its static class mix (ISA) - profiles/r02_<kernel>_traffic.json;
  cpu_baseline  the oracle (CPU port of the reference's rayon loop; the Rust reference cannot be built in
                this image) timed on this box's host cores on a bounded sample of the same workload;
  variants      the same frame through the persistent megakernel backend, through separate generate / intersect /
                shade kernels (with that intersect kernel's own roofline) and through two concurrent pipelines.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
L2_PEAK_GBS = 34500.0  # the eight XCDs' L2s together (MI355X_MICROARCH.md, "L2 (per XCD)": about 34.5 TB/s)
LOADED_ISA_HASH = [None]  # pt_kernel_isa_hash() of the library in use (set in main)
PROFILE_ROUNDS = ("r04", "r03", "r02")  # newest first: profiles/<round>_<kernel>_traffic.json
VALU_CYCLES_PEAK = 256 * 4 * 2.4e9  # SIMD-cycles per second: 256 CUs x 4 SIMDs at the 2.4 GHz peak clock
INTERSECT_BYTES_PER_RAY = 32  # k_intersect: 24 B (o, d) read + 8 B (t, id) written
QUEUE_BYTES_PER_STORED_RAY = 80  # k_pass: a ray of depth >= 1 is appended (40 B) and read back (40 B) exactly once


def host_cpu_share(limit):
    """Threads this process may really use: min(OpenMP default, affinity mask, cgroup cpu quota)."""
    n = limit
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    env = os.environ.get("PT_CPU_THREADS")
    if env:
        n = int(env)
    return max(1, n)


def cpu_baseline(width, height, seed, budget_spp):
    """Time the oracle (tests-only CPU restatement) on all host cores on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ptlib

    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    threads = host_cpu_share(ptlib.oracle().pto_max_threads())
    _, cnt, secs = ptlib.oracle_render(sc, width, height, budget_spp, seed, threads=threads)
    _, cnt1, secs1 = ptlib.oracle_render(sc, width, height, 1, seed, threads=1)
    return {
        "value": cnt.ray_bounces / secs,
        "unit": "ray-bounces/s",
        "cores": threads,
        "kind": "port",
        "sample": "cornell.json %dx%d @%dspp (same scene and resolution, spp cut from the GPU run's; cost is "
                  "linear in spp), OpenMP dynamic over shuffled pixels = the reference's rayon loop" %
                  (width, height, budget_spp),
        "single_core_value": cnt1.ray_bounces / secs1,
        "seconds": secs,
        "note": "C restatement of the reference's CPU path (no Rust toolchain in this image)",
    }


def load_profile(kernel):
    """profiles/<round>_<kernel>_traffic.json of the newest round that has one (tools/make_traffic_json.py), or None."""
    for r in PROFILE_ROUNDS:
        q = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (r, kernel))
        if os.path.exists(q):
            try:
                with open(q) as f:
                    tr = json.load(f)
                tr["_path"] = os.path.relpath(q, ROOT)
                tr["_round"] = r
                return tr
            except (OSError, ValueError) as e:
                print("bench.py: warning: cannot read %s: %s" % (q, e), file=sys.stderr)
    return None


def valu_figures(tr, kernel, rays_per_s, matches):
    """The resource that binds these kernels: VALU instruction issue.  Ceiling = 1024 SIMDs x the clock; a wave-instruction
    costs its class's measured cycles (profiles/r02_valu_issue_costs.json: 2 / 4 / 8, additive in real code).  The instruction
    count is dynamic (PMC SQ_INSTS_VALU); so is the mix where the profile has it (SQ_INSTS_VALU_* categories, each priced with
    the average class cost of its instructions in the ISA).  The counts come from a committed profile: they are only used
    when that profile was measured on the library that is loaded now (kernel_isa_hash) - otherwise None, with the reason.
    frac is given against the nominal 2.4 GHz and against the clock the chip held in this kernel (in-kernel s_memtime /
    s_memrealtime stamps of a diagnostic build: profiles/<round>_<kernel>_phase_budget.json) where that was measured."""
    valu = tr.get("valu") or {}
    mix = valu.get("static_mix")
    if not mix:
        return None
    if not matches:
        print("bench.py: warning: %s was measured on another build of the kernels (profile %s, library %s): no valu_roofline"
              % (tr["_path"], tr.get("kernel_isa_hash"), LOADED_ISA_HASH[0]), file=sys.stderr)
        return None
    try:
        dyn = valu.get("dynamic_mix")
        avg_cost = dyn["avg_cost"] if dyn else mix["avg_cost"]
        insts = valu["insts_per_ray"]
        cyc_per_ray = insts / 64.0 * avg_cost
        used = cyc_per_ray * rays_per_s
        peak_nominal = 256 * 4 * 2.4e9
        fig = {
            "kernel": kernel, "bound": "valu", "binds": True,
            "achieved": used, "peak": peak_nominal, "unit": "SIMD-cycles/s", "frac": used / peak_nominal,
            "clock_ghz": 2.4, "clock_source": "nominal peak clock",
            "max_rays_per_s": peak_nominal / cyc_per_ray,  # the issue ceiling at this instruction count and mix
            "valu_insts_per_ray": insts, "avg_cycles_per_wave_inst": avg_cost,
            "mix": "dynamic (SQ_INSTS_VALU_* categories)" if dyn else "static (whole-kernel ISA)",
            "dynamic_category_share": dyn["share"] if dyn else None,
            "active_lanes_of_64": (dyn or {}).get("active_lanes"),
            "class_counts_static": {k: mix[k] for k in ("A", "B", "C")},
            "class_cycles": {"A": 2, "B": 4, "C": 8},
            "issue_slots_frac_pmc": valu.get("issue_slots_frac", valu.get("busy_frac")),
            "profile": tr["_path"],
            "note": "frac = share of the SIMDs' cycles spent issuing VALU instructions, <= 1 by construction of the model; "
                    "issue_slots_frac_pmc = SQ_ACTIVE_INST_VALU per SIMD-quad-cycle, which ticks once per instruction whatever "
                    "its class and so exceeds frac by 4 / avg_cycles_per_wave_inst"}
        for r in PROFILE_ROUNDS:
            pb = os.path.join(ROOT, "profiles", "%s_%s_phase_budget.json" % (r, kernel))
            if os.path.exists(pb):
                with open(pb) as f:
                    ghz = json.load(f)["in_kernel_clock_ghz"]
                fig["frac_at_measured_clock"] = used / (256 * 4 * ghz * 1e9)
                fig["measured_clock_ghz"] = ghz
                fig["measured_clock_source"] = "in-kernel s_memtime / s_memrealtime of a -DPT_PHASE_STATS build, " + os.path.relpath(pb, ROOT)
                break
        return fig
    except (KeyError, TypeError, ValueError, OSError) as e:
        print("bench.py: warning: %s is malformed (%r): no valu_roofline" % (tr["_path"], e), file=sys.stderr)
        return None


def mesh_variant(pkg, torch, dev, dev_index, seed, W=1024, H=768, spp=1024, steps=2):
    scene = pkg.Scene(os.path.join(ROOT, "scenes", "mesh.json"), ROOT)
    ctx = pkg.Context(dev_index)
    ctx.set_scene(scene)
    ctx.set_profiling(True)
    buf = torch.zeros((W * H, 3), dtype=torch.float32, device=dev)
    ctx.render(buf.data_ptr(), W, H, spp, seed=seed)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bounces = launches = 0
    ms = 0.0
    for _ in range(steps):
        st = ctx.render(buf.data_ptr(), W, H, spp, seed=seed)
        bounces += st.ray_bounces
        launches += st.intersect_launches
        ms += st.ms_intersect
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel = ctx.pass_kernel()
    fig = {"value": bounces / dt, "unit": "ray-bounces/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "workload": "scenes/mesh.json %dx%d @%dspp, wavefront HIP backend" % (W, H, spp), "kernel": kernel,
           "avg_launch_ms": ms / max(1, launches), "launches": launches, "ray_bounces_per_frame": bounces // steps,
           "image_hash": "%016x" % pkg.image_hash(buf)}
    tr = load_profile(kernel)
    if tr:
        match = tr.get("kernel_isa_hash") == pkg.kernel_isa_hash()
        fig["profile"] = tr["_path"]
        fig["profile_matches_binary"] = match
        fig["valu_roofline"] = valu_figures(tr, kernel, bounces / (ms * 1e-3) if ms > 0 else bounces / dt, match)
    ctx.close()
    scene.close()
    return fig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=4096)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rays-per-pass", type=int, default=0)
    ap.add_argument("--backend", default="wavefront", choices=["wavefront", "megakernel"])
    ap.add_argument("--pipelines", type=int, default=1,
                    help="concurrent wavefront pipelines per GPU (opt-in; per-kernel roofline is only reported for 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--separate-kernels", action="store_true",
                    help="the main run with generate / intersect / shade as separate kernels per depth (PT_FLAG_SEPARATE_KERNELS): "
                         "for profiling the stand-alone intersect kernel (tools/pmc_run.sh)")
    ap.add_argument("--cpu-spp", type=int, default=128)  # ~15-20 s of CPU work on 16 host cores
    ap.add_argument("--gather", default="torch", choices=["torch", "abi"],
                    help="who runs the framebuffer all-gather: torch.distributed (RCCL) or the C ABI's pt_comm_gather_frame")
    ap.add_argument("--force-collective", action="store_true",
                    help="with one rank: initialise the process group / communicator and run the all-gather anyway "
                         "(checks the RCCL path on a one-GPU box)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Not under a launcher: start the ranks as children.  This process has not touched the GPU (no torch.cuda call,
        # no HIP call) and never will: it relays the children's JSON line and return code.
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch

    pkg = importlib.import_module("path-tracer-rust_amd")
    LOADED_ISA_HASH[0] = pkg.kernel_isa_hash()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher started a different number of ranks" % (args.gpus, world))
    dist_backend = os.environ.get("PT_BENCH_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if n_dev == 0:
        raise SystemExit("rank %d: bench.py needs a GPU: the product has no CPU path" % rank)
    if dist_backend == "nccl" and local_rank >= n_dev:
        raise SystemExit("rank %d: --gpus %d needs %d devices (one rank per GPU over RCCL), this box has %d"
                         % (rank, args.gpus, args.gpus, n_dev))
    # PT_BENCH_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices and
    # the band gather goes through host memory); the measured configuration is one rank per GPU over RCCL ("nccl")
    dev_index = local_rank if dist_backend == "nccl" else local_rank % max(1, n_dev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    use_collective = world > 1 or args.force_collective
    abi_gather = args.gather == "abi" and use_collective
    if use_collective:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if abi_gather:
            # the data plane is libptrace_hip's own RCCL communicator; torch.distributed (gloo) only carries the id,
            # the barrier and the scalar reductions of the timing
            dist.init_process_group(backend="gloo")
        elif dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=dist_backend)
    coll_dev = dev if (dist_backend == "nccl" and not abi_gather) else torch.device("cpu")
    comm = None
    if abi_gather:
        # (libptrace_hip.so finds the RCCL that belongs to the HIP runtime it is bound to - here torch's, loaded first - by
        # itself: csrc/pt_comm.hip)
        ident = [pkg.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        comm = pkg.Comm(dev_index, rank, world, ident[0])

    W, H, spp = args.width, args.height, args.spp
    npix = W * H
    scene = pkg.Scene(os.path.join(ROOT, "scenes", args.scene + ".json"), ROOT)
    ctx = pkg.Context(dev_index)
    ctx.set_scene(scene)  # scene tables resident in HBM before the timed region
    # rank r renders image rows r, r+N, r+2N, ... (chunk = one row of framebuffer indices): contiguous eighths of
    # this image differ by up to 1.31x in ray bounces, interleaved rows by < 1 %
    chunk = W
    counts = pkg.chunk_counts(npix, world, chunk)
    chunks = (chunk, rank, world) if world > 1 else None
    local = torch.zeros((counts[rank], 3), dtype=torch.float32, device=dev)
    frame = torch.zeros((npix, 3), dtype=torch.float32, device=dev) if abi_gather else None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(backend, profile, pipelines=1, separate=False):
        ctx.set_profiling(profile)
        t0 = time.perf_counter()
        st = ctx.render(local.data_ptr(), W, H, spp, seed=args.seed, backend=backend, chunks=chunks,
                        rays_per_pass=args.rays_per_pass, pipelines=pipelines, separate_kernels=separate)
        t1 = time.perf_counter()  # (pt_ctx_render is blocking: the rank's rows are in `local`)
        if abi_gather:
            comm.gather_frame(local.data_ptr(), frame.data_ptr(), W, H, chunk)
            full = frame
        else:
            full = pkg.gather_chunks(local if dist_backend == "nccl" else local.cpu(), npix, rank, world, chunk, dist,
                                     force_collective=args.force_collective)
        if use_collective:
            torch.cuda.synchronize()  # the gather's end on this rank (it waits for the slowest rank's rows: skew shows here)
        t2 = time.perf_counter()
        return st, full, 1e3 * (t1 - t0), 1e3 * (t2 - t1)

    def timed(backend, steps, warmup, profile, pipelines=1, separate=False):
        for _ in range(warmup):
            step(backend, profile, pipelines, separate)
        barrier()
        t0 = time.perf_counter()
        bounces = isect_rays = samples = passes = 0
        isect_ms = render_ms = gather_ms = 0.0
        launches = 0
        for _ in range(steps):
            st, full, r_ms, g_ms = step(backend, profile, pipelines, separate)
            bounces += st.ray_bounces
            isect_rays += st.intersect_rays
            isect_ms += st.ms_intersect
            launches += st.intersect_launches
            samples += st.samples
            passes += st.passes
            render_ms += r_ms
            gather_ms += g_ms
        barrier()
        dt = time.perf_counter() - t0
        per_rank = None
        if dist is not None:
            # one all-gather of (render ms, gather ms, bounces) per rank AFTER the timed region: with it a scaling figure
            # below N says whether skew between the ranks' rows, the launches or the gather took the rest
            mine = torch.tensor([render_ms / steps, gather_ms / steps, float(bounces) / steps], dtype=torch.float64, device=coll_dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            rows = [[float(v) for v in t.cpu()] for t in every]
            per_rank = {k: {"min": min(r[i] for r in rows), "max": max(r[i] for r in rows), "mean": sum(r[i] for r in rows) / world}
                        for i, k in enumerate(("render_ms", "gather_ms", "ray_bounces"))}
            per_rank["note"] = ("per step; render_ms = this rank's pt_ctx_render wall time (its rows, blocking), gather_ms = from there "
                                "to the end of the all-gather on this rank (includes waiting for the slowest rank)")
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            b = torch.tensor([bounces], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(b, op=dist.ReduceOp.SUM)
            bounces = int(b.item())
        return dict(dt=dt, bounces=bounces, isect_rays=isect_rays, isect_ms=isect_ms, launches=launches, image=full,
                    samples=samples, passes=passes, per_rank=per_rank, render_ms=render_ms / steps, gather_ms=gather_ms / steps)

    main_run = timed(args.backend, args.steps, args.warmup, profile=(args.backend == "wavefront" and args.pipelines == 1),
                     pipelines=args.pipelines, separate=args.separate_kernels)
    value = main_run["bounces"] / main_run["dt"]
    out = {
        "metric": "ray_bounces_per_sec",
        "value": value,
        "unit": "ray-bounces/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * main_run["dt"] / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "scenes/%s.json %dx%d @%dspp, %s HIP backend, %d band(s)%s"
                        % (args.scene, W, H, spp, args.backend, world,
                           " (interleaved rows) + one RCCL all-gather of the framebuffer" if world > 1 else ""),
            "backend": args.backend,
            "collective": (("pt_comm_gather_frame (C ABI: ncclAllGather + un-permute kernel)" if abi_gather else
                            "RCCL all_gather_into_tensor" if dist_backend == "nccl" else dist_backend + " (rehearsal)")
                           if use_collective else None),
            "partition": "rows interleaved over ranks (chunk = %d pixels)" % chunk if world > 1 else "whole frame",
            "width": W, "height": H, "spp": spp, "seed": args.seed,
            "build_flags": pkg.build_flags(),  # the -mllvm switches the compiler accepted (path-tracer-rust_amd/Makefile)
            "build_flags_complete": pkg.build_flags_complete(),  # false: the library lost some of the tuned switches
            "kernel_isa_hash": pkg.kernel_isa_hash(),  # of the device assembly this library's kernels were built from
            "ray_bounces_per_frame": main_run["bounces"] // max(1, args.steps),
            # Image.hash of the assembled frame (mod.rs:916-926: SipHash-1-3 of the f32 bits): equal across rank counts,
            # backends and gather paths
            "image_hash": "%016x" % pkg.image_hash(main_run["image"]),
        },
    }
    if not pkg.build_flags_complete():
        print("bench.py: warning: libptrace_hip.so was built without some of the tuned -mllvm switches (%r)" % pkg.build_flags(),
              file=sys.stderr)
    out["per_rank"] = main_run["per_rank"]
    out["render_ms"] = main_run["render_ms"]  # rank 0, per step: the blocking pt_ctx_render call
    out["gather_ms"] = main_run["gather_ms"] if use_collective else None  # rank 0, per step: the all-gather, synchronised
    if args.backend == "wavefront" and main_run["isect_ms"] > 0:
        # rank 0's launches of the dominant kernel (every rank runs the same kernel on its own rows)
        kernel = ctx.pass_kernel(separate_kernels=args.separate_kernels)
        rays, ms, launches = main_run["isect_rays"], main_run["isect_ms"], max(1, main_run["launches"])
        if kernel in ("k_pass", "k_pass_cand", "k_pass_cand_bvh"):
            # the primary rays are made in registers; every ray of depth >= 1 is appended once and read back once
            stored = rays - main_run["samples"]
        elif kernel == "k_pass_bvh":
            stored = rays  # the primaries go through the queue like every level
        else:
            stored = None
        if stored is not None:
            alg_bytes = QUEUE_BYTES_PER_STORED_RAY * stored
            per_unit = {"bytes_per_stored_ray": QUEUE_BYTES_PER_STORED_RAY, "stored_rays_per_launch": stored / launches,
                        "bytes_per_ray_bounce": alg_bytes / rays}
        else:
            alg_bytes = INTERSECT_BYTES_PER_RAY * rays
            per_unit = {"bytes_per_ray": INTERSECT_BYTES_PER_RAY}
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        out["roofline"] = {
            "kernel": kernel,
            "bound": "hbm",  # the roofline the north star names: algorithmic queue bytes against the HBM peak ...
            "binds": False,  # ... which is NOT what limits this kernel: see valu_roofline
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes / launches,
            "rays_per_launch": rays / launches,
            "avg_launch_ms": ms / launches,
            "rays_per_s": rays / (ms * 1e-3),
            "launches": launches,
        }
        out["roofline"].update(per_unit)
        lt = out["roofline"]["avg_launch_ms"] * 1e-3
        out["roofline"]["l2_frac"] = achieved / L2_PEAK_GBS  # the algorithmic bytes go through L2 (the waves' stacks): against its ~34.5 TB/s
        out["roofline"]["l2_peak"] = L2_PEAK_GBS
        tr = load_profile(kernel)
        out["roofline"]["profile"] = tr and tr["_path"]
        out["roofline"]["profile_matches_binary"] = bool(tr) and tr.get("kernel_isa_hash") == pkg.kernel_isa_hash()
        if tr:
            # PMC bytes per ray (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate rocprofv3 passes of this command
            # at reduced spp, committed under profiles/) times the rays one launch processes.  hbm_counter_frac is what
            # the counters say about the HBM interface: those bytes / the live launch time / the peak - `frac` above prices
            # ALGORITHMIC bytes, which for k_pass_cand never leave L2
            out["roofline"]["traffic"] = tr["hbm_bytes_per_ray"] * out["roofline"]["rays_per_launch"]
            out["roofline"]["traffic_bytes_per_ray"] = tr["hbm_bytes_per_ray"]
            out["roofline"]["traffic_source"] = tr.get("source")
            out["roofline"]["hbm_counter_gbs"] = out["roofline"]["traffic"] / lt / 1e9
            out["roofline"]["hbm_counter_frac"] = out["roofline"]["hbm_counter_gbs"] / HBM_PEAK_GBS
            out["valu_roofline"] = valu_figures(tr, kernel, out["roofline"]["rays_per_s"], out["roofline"]["profile_matches_binary"])
    if rank == 0 and world == 1 and not args.no_variants:
        other = "megakernel" if args.backend == "wavefront" else "wavefront"
        v = timed(other, max(1, min(args.steps, 2)), 1, profile=False)
        same = bool(torch.equal(v["image"], main_run["image"]))
        out["variants"] = {other: {"value": v["bounces"] / v["dt"], "unit": "ray-bounces/s",
                                   "ms_per_step": 1e3 * v["dt"] / max(1, min(args.steps, 2)),
                                   "image_identical_to_main_backend": same}}
        if args.backend == "wavefront" and args.pipelines == 1:
            # opt-in mode: 2 independent wavefront pipelines on 2 streams of this GPU (VALU-bound intersect of one
            # overlaps HBM-bound shade of another); not the headline because per-kernel timings lose their meaning
            if main_run["launches"] == main_run["passes"]:
                # the same frame with generate / intersect / shade as separate kernels per depth
                # (PT_FLAG_SEPARATE_KERNELS), which has an intersect kernel of its own to put on the roofline:
                # 24 B ray in + 8 B hit out per ray
                vs = timed("wavefront", max(1, min(args.steps, 2)), 1, profile=True, separate=True)
                ach = INTERSECT_BYTES_PER_RAY * vs["isect_rays"] / (vs["isect_ms"] * 1e-3) / 1e9
                out["variants"]["wavefront_separate_kernels"] = {
                    "value": vs["bounces"] / vs["dt"], "unit": "ray-bounces/s",
                    "ms_per_step": 1e3 * vs["dt"] / max(1, min(args.steps, 2)),
                    "image_identical_to_main_backend": bool(torch.equal(vs["image"], main_run["image"])),
                    "roofline_k_intersect": {"kernel": ctx.pass_kernel(separate_kernels=True), "bound": "hbm", "achieved": ach,
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": ach / HBM_PEAK_GBS, "bytes_per_ray": INTERSECT_BYTES_PER_RAY,
                                             "avg_launch_ms": vs["isect_ms"] / max(1, vs["launches"]),
                                             "rays_per_s": vs["isect_rays"] / (vs["isect_ms"] * 1e-3)}}
                # why the north star's ">= 60 % of the HBM roofline in the intersect kernel" is out of reach of exact f32
                # arithmetic on this scene: the kernel's own ceiling is VALU issue - its instructions per ray (PMC, a
                # committed profile of THIS build) at the measured class costs give the most rays per second the 1024 SIMDs
                # can issue, and that many rays move 32 B each
                rk = out["variants"]["wavefront_separate_kernels"]["roofline_k_intersect"]
                trk = load_profile(rk["kernel"])
                rk["profile"] = trk and trk["_path"]
                rk["profile_matches_binary"] = bool(trk) and trk.get("kernel_isa_hash") == pkg.kernel_isa_hash()
                vk = valu_figures(trk, rk["kernel"], rk["rays_per_s"], rk["profile_matches_binary"]) if trk else None
                rk["valu_ceiling"] = vk and {
                    "valu_insts_per_ray": vk["valu_insts_per_ray"], "avg_cycles_per_wave_inst": vk["avg_cycles_per_wave_inst"],
                    "max_rays_per_s": vk["max_rays_per_s"], "frac_of_valu_issue": vk["frac"],
                    "hbm_frac_at_valu_ceiling": vk["max_rays_per_s"] * INTERSECT_BYTES_PER_RAY / 1e9 / HBM_PEAK_GBS,
                    "note": "max_rays_per_s = 1024 SIMDs x 2.4 GHz / (instructions per ray / 64 x cycles per wave-instruction); "
                            "hbm_frac_at_valu_ceiling = that rate x 32 B / 8 TB/s: below 0.6 whatever the memory system does"}
            v3 = timed("wavefront", max(1, min(args.steps, 2)), 1, profile=False, pipelines=2)
            out["variants"]["wavefront_2_concurrent_pipelines"] = {
                "value": v3["bounces"] / v3["dt"], "unit": "ray-bounces/s",
                "ms_per_step": 1e3 * v3["dt"] / max(1, min(args.steps, 2)),
                "image_identical_to_main_backend": bool(torch.equal(v3["image"], main_run["image"]))}
        if args.backend == "wavefront" and args.scene == "cornell" and args.pipelines == 1:
            # BASELINE.json's config 4 beside the headline: scenes/mesh.json (meshes/mctri.off, 810 triangles: candidate scan +
            # parked BVH walks, k_pass_cand_bvh) 1024x768 @1024 spp, so that the second kernel form gets a driver-timed number
            out["variants"]["mesh_json"] = mesh_variant(pkg, torch, dev, dev_index, args.seed)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W, H, args.seed, args.cpu_spp)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
