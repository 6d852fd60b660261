#!/bin/bash
# enf_backward_weights (K3 with the activation store + K4) at the bench shape, 16 and 32 signals: working tree against variants/libenf_$1.so
for r in 1 2 3; do for v in $1 default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  echo "$v: $(ENF_HIP_LIB=$L timeout -k 10 200 python scripts/probe_wgrad.py 16 32 2>/dev/null | cut -c1-48 | tr '\n' ' ')"
done; done
