#!/bin/bash
# The profile set of a round, on the GPU box: usage tools/profile_round.sh <round tag, e.g. r03>
#   PMC (separate passes: SQ counters, FETCH_SIZE, WRITE_SIZE, the SQ_INSTS_VALU_* categories) for the bench scene, mesh.json
#   and the megakernel; rocprofv3 --kernel-trace --stats of the default bench command and of the mesh bench command; the
#   in-kernel phase budgets of a -DPT_PHASE_STATS build (scratch/libptrace_phase.so: `make -C path-tracer-rust_amd phase`);
#   the exhaustive sqrt / reciprocal search (tools/rounding_search); the stand-alone intersect kernel (separate kernels), the walk
#   statistics of a -DPT_WALK_STATS build, the large-mesh probe.
# Everything lands in gpurun_out/; tools/make_traffic_json.py and copies of the *_kernel_stats.csv go to profiles/.
set -e
T=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ -f scratch/libptrace_phase.so ]; then
  PT_LIB=scratch/libptrace_phase.so python3 tools/phase_budget.py cornell 256 gpurun_out/${T}_k_pass_cand_phase_budget.json > gpurun_out/${T}_phase_cornell.txt 2>&1
  PT_LIB=scratch/libptrace_phase.so python3 tools/phase_budget.py mesh 128 gpurun_out/${T}_k_pass_cand_bvh_phase_budget.json > gpurun_out/${T}_phase_mesh.txt 2>&1
fi
if [ -x tools/rounding_search ]; then ./tools/rounding_search > gpurun_out/${T}_rounding_search.json; fi
$R/tools/pmc_run.sh ${T}a --spp 683
$R/tools/pmc_run.sh ${T}m --scene mesh --spp 512
$R/tools/pmc_run.sh ${T}g --backend megakernel --spp 341
PMC_NO_RPP=1 $R/tools/pmc_run.sh ${T}s --separate-kernels --spp 128
if [ -f scratch/libptrace_walk.so ]; then PT_LIB=scratch/libptrace_walk.so python3 tools/walk_stats.py mesh 512 > gpurun_out/${T}_walk_stats_mesh_512spp.txt 2>&1; fi
python3 tools/bigmesh_probe.py 400 64 > gpurun_out/${T}_bigmesh_400.txt 2>&1
python3 tools/bigmesh_probe.py 96 64 > gpurun_out/${T}_bigmesh_96.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${T}a -- python3 $R/bench.py --no-cpu-baseline --no-variants > $R/gpurun_out/stats_${T}a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${T}m -- python3 $R/bench.py --no-cpu-baseline --no-variants --scene mesh --spp 1024 > $R/gpurun_out/stats_${T}m.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${T}g -- python3 $R/bench.py --no-cpu-baseline --no-variants --backend megakernel --spp 1024 --steps 2 > $R/gpurun_out/stats_${T}g.log 2>&1
echo profiled
