#!/bin/bash
# pyramid quad threshold A/B: alternating processes, same box
for round in 1 2; do
  for q in 65536 40000 24576 12288; do
    ZKHIP_PYR_QUAD_MAX=$q timeout -k 10 120 python3 tools/ab_time_msm.py 20 100 "q=$q" || exit 1
  done
done
for q in 65536 40000 24576; do
  ZKHIP_PYR_QUAD_MAX=$q timeout -k 10 120 python3 tools/ab_time_msm.py 22 40 "22q=$q" || exit 1
done
