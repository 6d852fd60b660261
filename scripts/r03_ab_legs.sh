#!/bin/bash
# per-kernel legs (bench.py --roofline-only --kernel-iters 100), three interleaved rounds: variants/libenf_$1.so against the working tree
for r in 1 2 3; do for v in $1 default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  ENF_HIP_LIB=$L timeout -k 10 200 python bench.py --roofline-only --kernel-iters 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v', {k:v['launch_ms'] for k,v in d['roofline_kernels'].items()})"
done; done
