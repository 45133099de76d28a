#!/bin/bash
# Collect rocprofv3 PMC counters for one short bench run (separate passes per counter group).
# usage: tools/pmc_run.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>_{sq,fetch,write}/
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
# (--rays-per-pass: the default pass size given explicitly, so that the one profiled frame has no short timed first pass)
# (PMC_NO_RPP=1: not for the level-by-level forms, whose queues hold 352 B per primary ray of a pass)
RPP="--rays-per-pass 536870912"; if [ -n "$PMC_NO_RPP" ]; then RPP=""; fi
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-variants $RPP $@"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_sq -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_write -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_write.log 2>&1
# dynamic instruction mix: the VALU instructions by the categories the SQ tallies (tools/make_traffic_json.py prices them)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_mix -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_mix.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_mix2 -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_mix2.log 2>&1
echo done
