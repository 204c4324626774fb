#!/bin/bash
# Round-end evidence in one call on one box (run through gpurun from the repo root): the rocprofv3 summaries of tools/collect_profiles.sh, copied
# into profiles/ on the box so that bench.py's `roofline.frac_rocprof` is read from the profile of THIS box, then the default bench.
#   bash tools/final_collect.sh <tag> ["<commit and what it holds>"]     -> gpurun_out/<tag>_* and gpurun_out/<tag>_bench_final.{json,err}
set -e
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
bash $R/tools/collect_profiles.sh $TAG
for f in kernel_stats_bench_msm2p20.csv kernel_stats_bench_full.csv pmc_hbm_bytes_bench.txt pmc_sq_issue.txt marker_trace.txt; do
  [ -f $O/${TAG}_$f ] && cp $O/${TAG}_$f $R/profiles/${TAG}_rocprofv3_$f
done
for f in pmc_traffic.json pmc_traffic_ntt.json; do [ -f $O/${TAG}_$f ] && cp $O/${TAG}_$f $R/profiles/${TAG}_$f; done
# the traffic files say which state of the tree they were collected on (bench.py quotes it in roofline.traffic_source); there is no .git on the box
if [ -n "$2" ]; then
  python3 - "$2" $O/${TAG}_pmc_traffic.json $O/${TAG}_pmc_traffic_ntt.json $R/profiles/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic_ntt.json <<'PY'
import json, sys
for path in sys.argv[2:]:
    d = json.load(open(path)); d["commit"] = sys.argv[1]; json.dump(d, open(path, "w"), indent=1)
PY
fi
cd $R && python3 bench.py > $O/${TAG}_bench_final.json 2> $O/${TAG}_bench_final.err
tail -c 600 $O/${TAG}_bench_final.json
