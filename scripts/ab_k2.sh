#!/bin/bash
# A/B the K2 launch time of several library variants on one box: scripts/ab_k2.sh base asm ...
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "$@"; do
    ENF_HIP_LIB=$PWD/variants/libenf_$v.so timeout -k 10 120 python scripts/probe_k2.py $v || exit 1
  done
done
