#!/bin/bash
# Collect rocprofv3 PMC counters for one short bench run (separate passes per counter group).
# usage: tools/pmc_run.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>_{sq,fetch,write}/
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-variants $@"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_sq -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_write -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_write.log 2>&1
echo done
