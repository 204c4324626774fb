#!/bin/bash
# Collects everything profiles/README.md lists for the final state, on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <tag>        -> gpurun_out/<tag>_*
# rocprofv3 passes: kernel stats of the bench command, FETCH_SIZE / WRITE_SIZE (separate passes) of the bench command and of the NTT at 2^22 / 2^24,
# SQ issue / wait counters of the MSM + NTT workload (separate passes, --kernel-trace only), the roctx marker trace, and LAST (the long one) the
# kernel stats of the full bench.  Raw rocprofv3 directories are summarised and removed as soon as their summaries are written, so that a call that
# runs out of time still leaves small files under gpurun_out/.
set -e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH_MSM="python3 $R/bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline --no-general-path"     # the bench line's own steps / warm-up
BENCH_PMC="python3 $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-general-path"
S="python3 $R/tools/summarize_prof.py"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_msm -- $BENCH_MSM > $O/${TAG}_prof_msm.log 2>&1
(cd $R && $S stats $O/${TAG}_prof_msm $O/${TAG}_kernel_stats_bench_msm2p20.csv); rm -rf $O/${TAG}_prof_msm
echo "stats msm done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -- $BENCH_PMC > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -- $BENCH_PMC > $O/${TAG}_pmc_write.log 2>&1
(cd $R && $S pmc $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_hbm_bytes_bench.txt $O/${TAG}_pmc_traffic.json); rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
echo "pmc hbm done"
# NTT passes (round 4): FETCH / WRITE per k_ntt_pass launch at 2^22 and 2^24, one size per pass so that the per-kernel averages are per size
for L in 22 24; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_pmc_ntt${L}_fetch -- python3 $R/tools/ntt_prof.py $L > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_pmc_ntt${L}_write -- python3 $R/tools/ntt_prof.py $L > /dev/null 2>&1
done
(cd $R && $S pmc_ntt $O/${TAG}_pmc_traffic_ntt.json 22 $O/${TAG}_pmc_ntt22_fetch $O/${TAG}_pmc_ntt22_write 24 $O/${TAG}_pmc_ntt24_fetch $O/${TAG}_pmc_ntt24_write)
rm -rf $O/${TAG}_pmc_ntt22_fetch $O/${TAG}_pmc_ntt22_write $O/${TAG}_pmc_ntt24_fetch $O/${TAG}_pmc_ntt24_write
echo "pmc ntt done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/${TAG}_pmc_sq_a -- python3 $R/tools/prof_msm.py 20 3 24 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES -d $O/${TAG}_pmc_sq_b -- python3 $R/tools/prof_msm.py 20 3 24 > /dev/null 2>&1
(cd $R && $S sq $O/${TAG}_pmc_sq_a $O/${TAG}_pmc_sq_b $O/${TAG}_pmc_sq_issue.txt); rm -rf $O/${TAG}_pmc_sq_a $O/${TAG}_pmc_sq_b
echo "pmc sq done"
# roctx ranges of the C ABI (one per call; opt-in since round 4: ZKHIP_ROCTX=1): marker trace + kernel trace of a short MSM / NTT workload
export ZKHIP_ROCTX=1
rocprofv3 --marker-trace --kernel-trace --stats -d $O/${TAG}_marker -- python3 $R/tools/prof_msm.py 20 2 22 > $O/${TAG}_marker.log 2>&1 || true
python3 $R/tools/marker_summary.py $O/${TAG}_marker $O/${TAG}_marker_trace.txt || true
unset ZKHIP_ROCTX
rm -rf $O/${TAG}_marker
echo "marker done"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_full -- python3 $R/bench.py --no-cpu-baseline > $O/${TAG}_prof_full.log 2>&1
(cd $R && $S stats $O/${TAG}_prof_full $O/${TAG}_kernel_stats_bench_full.csv); rm -rf $O/${TAG}_prof_full
echo "stats full done; summaries written"
