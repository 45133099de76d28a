#!/usr/bin/env python3
"""Turn the three PMC passes of tools/pmc_run.sh <tag> into profiles/<tag>_pmc_<what>.json and, per dominant kernel,
profiles/<round>_<kernel>_traffic.json (read by bench.py for roofline.traffic and for the VALU issue figure):
  hbm_bytes_per_ray   (2 x FETCH_SIZE + WRITE_SIZE) / rays - the gfx950 correction of MI355X_MICROARCH.md's HBM section
  valu                dynamic VALU wave-instructions per ray x 64 (SQ_INSTS_VALU), VALU busy (SQ_ACTIVE_INST_VALU /
                      SQ_BUSY_CYCLES / 8), and the kernel's static issue-class mix (tools/isa_mix.py over
                      path-tracer-rust_amd/pt_kernels.s, classes priced by profiles/r02_valu_issue_costs.json)
usage: python tools/make_traffic_json.py <tag> <what, e.g. cornell_1024x768_128spp> "<bench args of the run>" """
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise
import isa_mix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, what, bench_args = sys.argv[1], sys.argv[2], sys.argv[3]
out = {k: summarise(os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, k))) for k in ("sq", "fetch", "write", "mix", "mix2")
       if os.path.isdir(os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, k)))}
line = [l for l in open(os.path.join(ROOT, "gpurun_out", "pmc_%s_sq.log" % tag)) if l.startswith("{")][-1]
rays = json.loads(line)["config"]["ray_bounces_per_frame"]
# the library the counters were collected on (bench.py prints its pt_kernel_isa_hash): bench.py prices a live rate with these
# counts only when the library it has loaded carries the same hash
isa_hash = json.loads(line)["config"].get("kernel_isa_hash")
ROUND = tag[:3]  # "r04a" -> profiles/r04_<kernel>_traffic.json
name = "%s_pmc_%s.json" % (tag, what)
with open(os.path.join(ROOT, "profiles", name), "w") as f:
    json.dump({"command": "tools/pmc_run.sh %s %s" % (tag, bench_args), "rays_per_frame": rays, "passes": out}, f, indent=1)


def kernel(d, frag):
    return next((v for k, v in d.items() if frag in k), None)


NOTE = ("insts_per_ray = SQ_INSTS_VALU x 64 / rays (dynamic); issue_slots_frac = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / 8: the "
        "counter ticks one quad-cycle per VALU instruction whatever its class (it equals SQ_INSTS_VALU within 2 %), so it is the "
        "share of 4-cycle issue SLOTS taken, not of time, and reads above 1 for a mix cheaper than 4 cycles; "
        "dynamic_mix = the instructions by SQ_INSTS_VALU_* category (a PMC pass of its own), each category priced with the "
        "average class cost of that category's instructions in the ISA: avg_cost is the dynamic counterpart of static_mix's; "
        "static_mix = instruction classes of the kernel's ISA (A: 2 cycles per wave-instruction per SIMD, B: 4, C: 8 - "
        "measured, profiles/r02_valu_issue_costs.json; the classes add up in real code: tools/valu_issue_bench.hip "
        "and the no-packed-instruction A/B of DESIGN section 4); avg_cost = sum(count x cost) / count")
ASM = os.path.join(ROOT, "path-tracer-rust_amd", "pt_kernels.s")
# kernel name in the profile -> symbol fragment in the listing (most specific first)
# (k_pass_cand<STAGED, DEFER, BVH, PROBE, NLDS>: "true, false, false, false, false" is the bench scene's (glass in place),
# "true, false, true, false, true" mesh.json's (nodes staged): pt_ctx_pass_kernel
# calls the latter k_pass_cand_bvh, and bench.py looks its traffic up under that name)
KERNELS = [("k_pass_cand<true, false, true, false, true>", "k_pass_candILb1ELb0ELb1ELb0ELb1", "k_pass_cand_bvh"),
           ("k_pass_cand<true, false, true, false, false>", "k_pass_candILb1ELb0ELb1ELb0ELb0", "k_pass_cand_bvh"),
           ("k_pass_cand<true, false, false, false, false>", "k_pass_candILb1ELb0ELb0ELb0ELb0", "k_pass_cand"),
           ("k_pass_cand<true, true, false, false, false>", "k_pass_candILb1ELb1ELb0ELb0ELb0", "k_pass_cand"), ("k_pass_bvh", "k_pass_bvhILb0", None),
           ("k_pass<", "k_passILb1ELb0", None), ("k_intersect_cand", "k_intersect_candILb1", None),
           ("k_intersect<", "k_intersectILb0", "k_intersect"), ("k_mega_cand<false, false>", "k_mega_candILb0ELb0", "k_mega_cand"),
           ("k_mega<", "k_megaILb0ELb0", "k_mega")]
done = {}
for kname, sym, label in KERNELS:
    sq = kernel(out["sq"], kname)
    if sq is None:
        continue
    fetch = kernel(out["fetch"], kname)["FETCH_SIZE"] * 1024.0  # counter is in KiB
    write = kernel(out["write"], kname)["WRITE_SIZE"] * 1024.0
    clean = label or kname.rstrip("<")
    static = isa_mix.mix(ASM, sym) if os.path.exists(ASM) else None
    dynamic = None
    mixp = kernel(out.get("mix", {}), kname)
    if mixp and static:
        by_cat = isa_mix.mix_by_category(ASM, sym)
        dyn = {"add_f32": mixp["SQ_INSTS_VALU_ADD_F32"], "mul_f32": mixp["SQ_INSTS_VALU_MUL_F32"], "fma_f32": mixp["SQ_INSTS_VALU_FMA_F32"],
               "trans_f32": mixp["SQ_INSTS_VALU_TRANS_F32"], "int32": mixp["SQ_INSTS_VALU_INT32"], "int64": mixp["SQ_INSTS_VALU_INT64"],
               "cvt": mixp["SQ_INSTS_VALU_CVT"]}
        m2 = kernel(out.get("mix2", {}), kname)
        if m2:
            dyn["f64"] = m2["SQ_INSTS_VALU_ADD_F64"] + m2["SQ_INSTS_VALU_MUL_F64"] + m2["SQ_INSTS_VALU_FMA_F64"]
        dyn["other"] = max(0.0, mixp["SQ_INSTS_VALU"] - sum(dyn.values()))
        total = sum(dyn.values())
        cyc = sum(v * (by_cat.get(k, {}).get("avg_cost", static["avg_cost"])) for k, v in dyn.items())
        dynamic = {"wave_insts": dyn, "share": {k: v / total for k, v in dyn.items()},
                   "static_avg_cost_by_category": {k: v["avg_cost"] for k, v in by_cat.items()}, "avg_cost": cyc / total}
        if m2:
            dynamic["active_lanes"] = m2["SQ_THREAD_CYCLES_VALU"] / max(1.0, sq["SQ_ACTIVE_INST_VALU"])
            dynamic["wave_cycles_waiting_frac"] = m2["SQ_WAIT_ANY"] / m2["SQ_WAVE_CYCLES"]
    tr = {
        "kernel": clean,
        "kernel_isa_hash": isa_hash,
        "rays": rays,
        "dispatches": sq["dispatches"],
        "FETCH_SIZE_bytes": fetch,
        "WRITE_SIZE_bytes": write,
        "hbm_bytes_per_ray": (2.0 * fetch + write) / rays,
        "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests of wide coalesced reads at 64 B, "
                      "MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; both in KiB",
        "source": "profiles/%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py "
                  "--steps 1 --warmup 0 --no-cpu-baseline --no-variants %s`" % (name, bench_args),
        "valu": {
            "insts_per_ray": sq["SQ_INSTS_VALU"] * 64.0 / rays,
            "issue_slots_frac": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_BUSY_CYCLES"] / 8.0,
            "static_mix": static,
            "dynamic_mix": dynamic,
            "note": NOTE,
        },
    }
    done[clean] = tr
    with open(os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (ROUND, clean)), "w") as f:
        json.dump(tr, f, indent=1)
print(json.dumps(done, indent=1))
