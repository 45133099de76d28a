#!/bin/bash
# After tools/profile_round.sh <tag> and tools/final_lines.sh ran on the GPU box and gpurun_out/ was merged back:
#   tools/collect_profiles.sh r04 <shipped cornell@256 bounces/s> <shipped mesh@128 bounces/s>
# writes profiles/<tag>?_pmc_*.json, <tag>_<kernel>_traffic.json, the kernel stats, the phase / EXEC budgets and the bench lines.
set -e
T=$1; RC=$2; RM=$3
cd "$(dirname "$0")/.."
python3 tools/make_traffic_json.py ${T}a cornell_1024x768_683spp "--spp 683" > /dev/null
python3 tools/make_traffic_json.py ${T}m mesh_1024x768_512spp "--scene mesh --spp 512" > /dev/null
python3 tools/make_traffic_json.py ${T}g cornell_megakernel_1024x768_341spp "--backend megakernel --spp 341" > /dev/null
python3 tools/make_traffic_json.py ${T}s cornell_separate_kernels_128spp "--separate-kernels --spp 128" > /dev/null
for t in a m g a2 m2; do f=$(ls gpurun_out/stats_${T}$t/*/*kernel_stats.csv | head -1); cp $f profiles/${T}${t}_kernel_stats.csv; done
lanes() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = next(k for k in d["passes"]["mix2"] if sys.argv[2] in k)
print("%.1f" % (d["passes"]["mix2"][k]["SQ_THREAD_CYCLES_VALU"] / d["passes"]["sq"][k]["SQ_ACTIVE_INST_VALU"]))
PY
}
LC=$(lanes profiles/${T}a_pmc_cornell_1024x768_683spp.json k_pass_cand); LM=$(lanes profiles/${T}m_pmc_mesh_1024x768_512spp.json k_pass_cand)
for k in k_pass_cand k_pass_cand_bvh; do cp gpurun_out/${T}_${k}_phase_budget.json profiles/; done
python3 tools/phase_budget.py --correct profiles/${T}_k_pass_cand_phase_budget.json $RC $LC > profiles/${T}_k_pass_cand_phase_budget.txt
python3 tools/phase_budget.py --correct profiles/${T}_k_pass_cand_bvh_phase_budget.json $RM $LM > profiles/${T}_k_pass_cand_bvh_phase_budget.txt
python3 tools/exec_budget.py profiles/${T}_k_pass_cand_phase_budget.json profiles/${T}_k_pass_cand_exec_budget.json
python3 tools/exec_budget.py profiles/${T}_k_pass_cand_bvh_phase_budget.json profiles/${T}_k_pass_cand_bvh_exec_budget.json
cp gpurun_out/${T}_walk_stats_mesh_512spp.txt gpurun_out/${T}_bigmesh_400.txt gpurun_out/${T}_bigmesh_96.txt gpurun_out/${T}_rounding_search.json profiles/
for f in ${T}_final_cornell4096 ${T}_final_cornell1024 ${T}_final_mesh1024 ${T}_final_config5_4096x4096_16384spp_1gpu; do cp gpurun_out/$f.json profiles/bench_lines/; done
cp gpurun_out/${T}_size_*.json profiles/bench_lines/
python3 - $T <<'PY'
import json, glob, sys
T = sys.argv[1]
for k in ("k_pass_cand", "k_pass_cand_bvh", "k_mega_cand", "k_intersect_cand"):
    d = json.load(open("profiles/%s_%s_traffic.json" % (T, k))); v = d["valu"]
    print("%-18s hash %s  %.1f instructions/bounce  issue slots %.3f  HBM %.2f B/bounce" % (k, d["kernel_isa_hash"], v["insts_per_ray"], v["issue_slots_frac"], d["hbm_bytes_per_ray"]))
for f in sorted(glob.glob("profiles/bench_lines/%s_final_*.json" % T)) + sorted(glob.glob("profiles/bench_lines/%s_size_*.json" % T)):
    d = json.load(open(f))
    print("%-60s %6.2f G  %9.2f ms  %s  matches %s" % (f.split("/")[-1], d["value"] / 1e9, d["ms_per_step"], d["config"].get("image_hash"), d.get("roofline", {}).get("profile_matches_binary")))
PY
