#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_weight_grads.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do
for l in variants/libenf_nostorespec.so -; do
  if [ "$l" = "-" ]; then unset ENF_HIP_LIB; else export ENF_HIP_LIB=$PWD/$l; fi
  timeout -k 10 200 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-ode --events-steps 0 --no-accuracy 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$l meta', d['meta_step']['ms_per_step'])"
done
done 2>&1 | tee $O/c13_ab.log
