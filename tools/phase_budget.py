"""Dynamic phase budget of k_pass_cand from a -DPT_PHASE_STATS build of the library (`make -C path-tracer-rust_amd phase`,
never the shipped one) on one GPU:
  PT_LIB=scratch/libptrace_phase.so python tools/phase_budget.py [scene] [spp] [out.json]
Per phase: share of the waves' lifetime (in-kernel s_memtime stamps), entries, active lanes at entry; the clock the
chip held (s_memtime / s_memrealtime ticks of every wave); the frame's rate with the stamps in (they cost time: compare
with the shipped library's rate, printed by bench.py)."""
import ctypes as C
import json
import os
import sys


def correct(path, shipped_rate, pmc_lanes=None):
    """Add the stamp-corrected budget to a budget file: the shipped library renders the same frame at `shipped_rate` ray
    bounces per second, i.e. in fewer wave-cycles per bounce than the stamped build; the difference, divided by the number
    of stamps, is what one stamp costs a wave, and a phase's corrected cycles are its stamped cycles minus its entries
    times that.  python tools/phase_budget.py --correct <budget.json> <shipped bounces/s>"""
    res = json.load(open(path))
    # SIMDs x waves per SIMD: k_pass_cand runs five waves per SIMD without walks, four with (csrc/pt_kernels.h)
    waves_per_chip = 256 * 4 * (4 if "bvh" in res.get("kernel", "") else 5)
    shipped_cyc = waves_per_chip * res["in_kernel_clock_ghz"] * 1e9 / shipped_rate * 64.0  # wave-cycles per 64 bounces
    stamps = sum(p["entries_per_kbounce"] for p in res["phases"].values()) * 0.064
    per_stamp = (res["wave_cycles_per_bounce"] - shipped_cyc) / stamps
    tot = 0.0
    for p in res["phases"].values():
        p["corrected_wave_cycles_per_bounce"] = max(0.0, p["wave_cycles_per_bounce"] - per_stamp * p["entries_per_kbounce"] * 0.064)
        tot += p["corrected_wave_cycles_per_bounce"]
    for p in res["phases"].values():
        p["corrected_share"] = p["corrected_wave_cycles_per_bounce"] / tot
    # THE EXEC BUDGET: a phase's corrected share of the waves' lifetime, weighted with the lanes that were active when it was
    # entered - the kernel is bound by VALU issue, so time is a proxy for instructions - is what the PMC pair
    # SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU measures as "lanes active per VALU instruction" (a third argument names it)
    lanes = sum(p["corrected_share"] * p["lanes_at_entry"] for p in res["phases"].values())
    res["time_weighted_active_lanes"] = lanes
    if pmc_lanes is not None:
        res["pmc_active_lanes_per_valu_instruction"] = pmc_lanes
    res["exec_budget"] = {n: {"share_of_time": p["corrected_share"], "lanes": p["lanes_at_entry"],
                              "lane_slots_lost_share": p["corrected_share"] * (64.0 - p["lanes_at_entry"]) / 64.0}
                          for n, p in sorted(res["phases"].items(), key=lambda kv: -kv[1]["corrected_share"] * (64.0 - kv[1]["lanes_at_entry"]))}
    res["shipped_bounces_per_s"] = shipped_rate
    res["shipped_wave_cycles_per_bounce"] = shipped_cyc
    res["stamp_cost_wave_cycles"] = per_stamp
    res["corrected_total_wave_cycles_per_bounce"] = tot
    json.dump(res, open(path, "w"), indent=1)
    print("%s: shipped %.2f G bounces/s = %.0f wave-cycles per 64 bounces; a stamp costs %.0f; corrected sum %.0f" %
          (path, shipped_rate / 1e9, shipped_cyc, per_stamp, tot))
    for n, p in sorted(res["phases"].items(), key=lambda kv: -kv[1]["corrected_share"]):
        print("  %-22s %5.1f %%  (stamped %5.1f %%)  %5.1f lanes  -> %4.1f %% of the lane slots idle here" %
              (n, 100 * p["corrected_share"], 100 * p["share"], p["lanes_at_entry"], 100 * p["corrected_share"] * (64.0 - p["lanes_at_entry"]) / 64.0))
    print("  time-weighted active lanes %.1f of 64%s" % (lanes, "" if pmc_lanes is None else "  (PMC, per VALU instruction: %.1f)" % pmc_lanes))


if len(sys.argv) > 1 and sys.argv[1] == "--correct":
    correct(sys.argv[2], float(sys.argv[3]), float(sys.argv[4]) if len(sys.argv) > 4 else None)
    sys.exit(0)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ptlib
from ptlib import PtConfig, PtStats

NAMES = ["other", "load_ray", "spheres", "filter_push", "exact_batch", "finish_bookkeeping", "surface_fetch", "rng_roulette",
         "diffuse", "specular", "glass", "append", "glass_defer", "barrier", "bvh_wants", "walk_gate", "walk_box_batch",
         "walk_leaf_batch", "primary_ray", "emit", "push_to_ring", "append_second_ray", "roulette_rescale", "emit_fixed_point_adds"]

ptlib.PRODUCT_SO = os.environ["PT_LIB"]
L = ptlib.product()
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out_path = sys.argv[3] if len(sys.argv) > 3 else None
W, H = 1024, 768
sc = ptlib.load_scene_py(ptlib.scene_path(scene))
ctx = C.c_void_p()
assert L.pt_ctx_create(0, C.byref(ctx)) == 0
assert L.pt_ctx_set_scene(ctx, C.byref(sc.cam), sc.objs, sc.n_objs, sc.tris, sc.n_tris) == 0
cfg = PtConfig(W, H, spp, 0, 1, 0, 0, 512 << 20, 0, 0, 0, 0, 0)  # (the default pass size, given explicitly: no timed short passes)
d = C.c_void_p()
assert L.pt_device_malloc(0, W * H * 12, C.byref(d)) == 0
L.pt_debug_phase_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32]
buf = (C.c_ulonglong * 128)()
st = PtStats()
for rep in range(3):  # two warm frames (clocks settle), the third is read
    assert L.pt_ctx_render(ctx, C.byref(cfg), d, None, None, None, None, C.byref(st)) == 0, L.pt_last_error()
    n_ph = L.pt_debug_phase_stats(ctx, buf, 128)
    assert n_ph == len(NAMES), n_ph
v = list(buf)
tot_cyc = sum(v[3 * k] for k in range(n_ph))
life, real = v[3 * n_ph], v[3 * n_ph + 1]
ghz = life / real * 0.1 if real else 0.0
bounces = st.ray_bounces
res = {"scene": scene, "width": W, "height": H, "spp": spp, "kernel": L.pt_ctx_pass_kernel(ctx, 0).decode(),
       "ray_bounces": bounces, "ms_device": st.ms_device, "bounces_per_s_instrumented": bounces / (st.ms_device * 1e-3),
       "in_kernel_clock_ghz": ghz, "wave_lifetime_cycles": life, "stamped_cycles": tot_cyc,
       "wave_cycles_per_bounce": life / bounces * 64.0,
       "note": "cycles = s_memtime ticks of wave lifetime (4 waves share a SIMD: a phase's share of lifetime is its share of the "
               "SIMD's time when every phase is issue-bound alike); lanes = popcount(EXEC) when the phase is entered",
       "phases": {}}
for k in range(n_ph):
    cyc, ent, lanes = v[3 * k], v[3 * k + 1], v[3 * k + 2]
    if ent == 0 and cyc == 0:
        continue
    res["phases"][NAMES[k]] = {"share": cyc / tot_cyc, "cycles_per_entry": cyc / max(ent, 1), "entries_per_kbounce": 1e3 * ent / bounces,
                               "lanes_at_entry": lanes / max(ent, 1), "wave_cycles_per_bounce": cyc / bounces * 64.0}
print("%s %dx%d @%d spp: %s, %.2f G bounces/s with stamps, clock %.3f GHz, %.0f wave-cycles per 64 bounces" %
      (scene, W, H, spp, res["kernel"], res["bounces_per_s_instrumented"] / 1e9, ghz, res["wave_cycles_per_bounce"]))
for n, p in sorted(res["phases"].items(), key=lambda kv: -kv[1]["share"]):
    print("  %-20s %5.1f %%   %7.0f cyc/entry   %6.2f entries/k-bounce   %4.1f lanes" %
          (n, 100 * p["share"], p["cycles_per_entry"], p["entries_per_kbounce"], p["lanes_at_entry"]))
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
