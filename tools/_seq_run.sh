set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
gcc -std=c99 -O2 -I include tests/cpp/prover_sequence.c -o /tmp/prover_sequence -L zksnap_circuits_halo2_amd -lzkhip -Wl,-rpath,$R/zksnap_circuits_halo2_amd
python3 -c "
from zksnap_circuits_halo2_amd import evaluation as E
open('/tmp/p13.bin','wb').write(E.export_prover_programs(13, 256, 1, seed=13))
open('/tmp/p15.bin','wb').write(E.export_prover_programs(15, 64, 1, seed=15))
open('/tmp/p22.bin','wb').write(E.export_prover_programs(22, 4, 1, seed=22))
"
timeout -k 10 300 /tmp/prover_sequence /tmp/p13.bin --device-only 5 | tail -2
timeout -k 10 300 /tmp/prover_sequence /tmp/p15.bin --device-only 5 | tail -2
timeout -k 10 300 /tmp/prover_sequence /tmp/p22.bin --device-only 3 | tail -2
