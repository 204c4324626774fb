#!/bin/bash
# SQ issue / wait counters of any workload:  bash tools/sq_counters.sh <tag> <script.py> [args ...]   -> gpurun_out/<tag>_sq.txt
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/${TAG}_sq_a -- python3 $R/"$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES -d $O/${TAG}_sq_b -- python3 $R/"$@" > /dev/null 2>&1
cd $R
python3 tools/summarize_prof.py sq $O/${TAG}_sq_a $O/${TAG}_sq_b $O/${TAG}_sq.txt
rm -rf $O/${TAG}_sq_a $O/${TAG}_sq_b
