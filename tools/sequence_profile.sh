#!/bin/bash
# rocprofv3 kernel statistics of the C99 prover sequence (tests/cpp/prover_sequence.c --device-only: the whole device side of a proof, three
# proofs over the same buffers after setup / keygen in the same process) at the wrapper's shape and at the voter's.  Run through gpurun from the
# repo root:   bash tools/sequence_profile.sh <tag>   -> gpurun_out/<tag>_kernel_stats_prover_sequence_k22.csv, ..._k13.csv
set -e
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
gcc -std=c99 -O2 -I include tests/cpp/prover_sequence.c -o /tmp/prover_sequence -L zksnap_circuits_halo2_amd -lzkhip -Wl,-rpath,$R/zksnap_circuits_halo2_amd
python3 -c "
from zksnap_circuits_halo2_amd import evaluation as E
open('/tmp/p22.bin','wb').write(E.export_prover_programs(22, 4, 1, seed=22))
open('/tmp/p13.bin','wb').write(E.export_prover_programs(13, 256, 1, seed=13))
"
cd /tmp && export TMPDIR=/tmp
for K in 22 13; do
  rocprofv3 --kernel-trace --stats -d /tmp/seqprof_$K -- /tmp/prover_sequence /tmp/p$K.bin --device-only 3 > $O/${TAG}_prover_sequence_k$K.log 2>&1
  (cd $R && python3 tools/summarize_prof.py stats /tmp/seqprof_$K $O/${TAG}_kernel_stats_prover_sequence_k$K.csv)
  rm -rf /tmp/seqprof_$K
  grep "sequence_ms\|shape" $O/${TAG}_prover_sequence_k$K.log
done
