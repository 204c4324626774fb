#!/bin/bash
# single-lane pyramid steps as one resident round of looping waves (workgroup cap) against one addition per thread: alternating processes, same box
for round in 1 2; do
  for q in 0 1536 3072 6144; do
    ZKHIP_PYR_PERSIST=$q timeout -k 10 120 python3 tools/ab_time_msm.py 20 100 "persist=$q" || exit 1
  done
done
for q in 0 3072; do
  ZKHIP_PYR_PERSIST=$q timeout -k 10 120 python3 tools/ab_time_msm.py 22 40 "22persist=$q" || exit 1
done
