#!/bin/bash
# SQ counters of the prepared MSM at 2^L (3 MSMs):  bash tools/msm_counters.sh <tag> <L> [VAR=value ...]   -> gpurun_out/<tag>_msm_sq.txt
set -e
TAG=$1; L=$2; shift; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/${TAG}_sq_a -- python3 $R/tools/prof_msm.py $L 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES -d $O/${TAG}_sq_b -- python3 $R/tools/prof_msm.py $L 3 > /dev/null 2>&1
cd $R
python3 tools/summarize_prof.py sq $O/${TAG}_sq_a $O/${TAG}_sq_b $O/${TAG}_msm_sq.txt
rm -rf $O/${TAG}_sq_a $O/${TAG}_sq_b
