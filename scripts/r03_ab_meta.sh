#!/bin/bash
# the outer (meta) step: working tree against variants/libenf_$1.so, weight-gradient parity first
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_weight_grads.py tests/test_gpu_trainer.py tests/test_gpu_bf16_contract.py tests/test_gpu_configs.py -m gpu -x -q > $O/meta_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/meta_tests.log
[ $rc = 0 ] || exit 1
for r in 1 2 3; do for v in $1 default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  ENF_HIP_LIB=$L timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ode --events-steps 0 --no-roofline --no-accuracy > $O/meta_$v.json 2>$O/meta_$v.err || { echo "bench $v failed"; tail -5 $O/meta_$v.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/meta_$v.json').readline()); print('$v', 'meta', d['meta_step']['ms_per_step'])"
done; done
