#!/bin/bash
# K3's fit_bwd leg: the packed gelu polynomial (gelu0 = off), the packed four-term dots (dot0 = off), the one-fma gelu' (dg0 = off), all three off (gd0)
O=gpurun_out/r03
mkdir -p $O
for r in 1 2 3; do
for v in gd0 gelu0 dot0 dg0 default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  ENF_HIP_LIB=$L timeout -k 10 200 python bench.py --roofline-only --kernel-iters 100 > $O/k3pk_$v.json 2>$O/k3pk_$v.err || { echo "bench $v failed"; tail -5 $O/k3pk_$v.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/k3pk_$v.json').readline()); print('$v', {k:v['launch_ms'] for k,v in d['roofline_kernels'].items()})"
done; done
