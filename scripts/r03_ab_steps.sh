#!/bin/bash
# whole steps, three interleaved rounds: variants/libenf_$1.so against the working tree (+ backward / reentrancy parity of the variant first)
O=gpurun_out/r03
mkdir -p $O
ENF_HIP_LIB=variants/libenf_$1.so timeout -k 10 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_reentrancy.py -m gpu -x -q > $O/st_tests.log 2>&1; rc=$?; echo "tests($1) rc=$rc"; tail -2 $O/st_tests.log
[ $rc = 0 ] || exit 1
for r in 1 2 3; do for v in $1 default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  ENF_HIP_LIB=$L timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-meta --no-ode --no-roofline --no-accuracy --events-steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], d['timing']['events']['ms_median'])"
done; done
