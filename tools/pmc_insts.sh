#!/bin/bash
# Dynamic VALU instructions per ray bounce of one build of the library (one short frame through tools/one_frame.py):
#   tools/pmc_insts.sh <tag> <lib.so> [scene] [spp] [backend]   ->  gpurun_out/pmci_<tag>.txt
# (SQ_INSTS_VALU x 64 / bounces of the dominant kernel; SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = lanes active per instruction)
set -e
TAG=$1; LIB=$2; SCENE=${3:-cornell}; SPP=${4:-128}; BACKEND=${5:-0}
R=${GRAFT_REPO_ROOT:-/root/repo}
export PT_LIB=$(realpath $LIB)
# the default pass / round size, given explicitly: no short timed first passes in a one-frame profile
if [ "$BACKEND" = "1" ]; then export PT_ONE_FRAME_RPP=268435456; else export PT_ONE_FRAME_RPP=536870912; fi
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmci_${TAG}
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmci_${TAG} -- python3 $R/tools/one_frame.py $SCENE $SPP 1 $BACKEND > $R/gpurun_out/pmci_${TAG}.log 2>&1
cd $R
python3 - <<PY > gpurun_out/pmci_${TAG}.txt
import re, sys
sys.path.insert(0, "tools")
from pmc_summary import summarise
log = open("gpurun_out/pmci_${TAG}.log").read()
bounces = int(re.search(r"(\d+) bounces", log).group(1))
s = summarise("gpurun_out/pmci_${TAG}")
for k, v in sorted(s.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:2]:
    iv = v.get("SQ_INSTS_VALU", 0)
    print("%s  %s: VALU insts/bounce %.1f  SALU/bounce %.1f  lanes/inst %.1f  issue slots %.3f  waiting %.3f  (%d bounces)" % (
        "${TAG}", k[:60], iv * 64.0 / bounces, v.get("SQ_INSTS_SALU", 0) * 64.0 / bounces,
        v.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, v.get("SQ_ACTIVE_INST_VALU", 1)),
        v.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, v.get("SQ_BUSY_CYCLES", 1)) / 8.0,
        v.get("SQ_WAIT_ANY", 0) / max(1.0, v.get("SQ_WAVE_CYCLES", 1)), bounces))
PY
cat gpurun_out/pmci_${TAG}.txt
