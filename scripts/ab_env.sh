#!/bin/bash
# Same-box A/B of bench.py over values of one environment variable: scripts/ab_env.sh VAR v1 v2 ...   (3 rounds)
# The library's own switches (ENF_ZFOLD, ENF_ZFOLD_BWD, ENF_SIDE_STREAM, ENF_TAIL_LA2, ENF_WZ_GRID) exist only in a build with
# -DENF_AB_SWITCHES: scripts/build_variant.sh ab -DENF_AB_SWITCHES, then ENF_HIP_LIB=variants/libenf_ab.so scripts/ab_env.sh ...
VAR=$1; shift
for r in 1 2 3; do
  for v in "$@"; do
    env $VAR=$v timeout -k 10 120 python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], 'fit', d['split']['ms_fit'], 'decode', d['split']['ms_decode'])"
  done
done
