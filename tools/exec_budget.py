#!/usr/bin/env python3
"""The EXEC budget of a corrected phase budget as a file of its own:
  python tools/exec_budget.py profiles/r04_k_pass_cand_phase_budget.json profiles/r04_k_pass_cand_exec_budget.json
(after `tools/phase_budget.py --correct <budget.json> <shipped bounces/s> <PMC lanes>`)."""
import json
import sys

res = json.load(open(sys.argv[1]))
out = {
    "kernel": res["kernel"], "scene": res["scene"], "spp": res["spp"],
    "what": "share of the waves' lifetime per phase (in-kernel s_memtime stamps of a -DPT_PHASE_STATS build, the stamps' own cost "
            "subtracted) and popcount(EXEC) when the phase - or the divergent block inside it - is entered; "
            "lane_slots_lost_share = share x (64 - lanes) / 64",
    "time_weighted_active_lanes": res["time_weighted_active_lanes"],
    "pmc_active_lanes_per_valu_instruction": res.get("pmc_active_lanes_per_valu_instruction"),
    "why_the_two_differ": "the stamps weight a phase with its TIME, the PMC pair SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU with its "
                          "VALU INSTRUCTIONS: the dense phases that wait (the filters' scalar record loads, the ray loads, the LDS "
                          "gathers of the batches) take more time per instruction than the divergent blocks, which are pure arithmetic",
    "in_kernel_clock_ghz": res["in_kernel_clock_ghz"],
    "shipped_bounces_per_s": res["shipped_bounces_per_s"],
    "phases": res["exec_budget"],
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print("%s: %.1f lanes by time, %s by instructions (PMC)" % (sys.argv[2], out["time_weighted_active_lanes"], out["pmc_active_lanes_per_valu_instruction"]))
