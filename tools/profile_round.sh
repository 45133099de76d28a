#!/bin/bash
# The profile set of a round, on the GPU box: usage tools/profile_round.sh <round tag, e.g. r02>
#   PMC (three separate passes each: SQ counters, FETCH_SIZE, WRITE_SIZE) for the bench scene, mesh.json and the megakernel,
#   rocprofv3 --kernel-trace --stats of the default bench command and of the mesh bench command.
# Everything lands in gpurun_out/; tools/make_traffic_json.py and a copy of the *_kernel_stats.csv go to profiles/.
set -e
T=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
$R/tools/pmc_run.sh ${T}a --spp 128
$R/tools/pmc_run.sh ${T}m --scene mesh --spp 64
$R/tools/pmc_run.sh ${T}g --backend megakernel --spp 128
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${T}a -- python3 $R/bench.py --no-cpu-baseline --no-variants > $R/gpurun_out/stats_${T}a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${T}m -- python3 $R/bench.py --no-cpu-baseline --no-variants --scene mesh --spp 1024 > $R/gpurun_out/stats_${T}m.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${T}g -- python3 $R/bench.py --no-cpu-baseline --no-variants --backend megakernel --spp 1024 --steps 2 > $R/gpurun_out/stats_${T}g.log 2>&1
echo profiled
