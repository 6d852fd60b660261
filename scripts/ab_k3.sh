#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "$@"; do
    ENF_HIP_LIB=$PWD/variants/libenf_$v.so timeout -k 10 120 python scripts/probe_k3.py $v || exit 1
  done
done
