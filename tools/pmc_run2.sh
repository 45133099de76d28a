#!/bin/bash
# usage: tools/pmc_run2.sh <tag> "<counters>" [bench args...]
set -e
TAG=$1; shift; CNT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG} -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants --rays-per-pass 536870912 "$@" > $R/gpurun_out/pmc_${TAG}.log 2>&1
echo done
