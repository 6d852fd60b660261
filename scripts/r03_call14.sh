#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1200 python -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py tests/test_gpu_weight_grads.py tests/test_gpu_golden.py tests/test_gpu_ball.py tests/test_gpu_narrow.py tests/test_gpu_layers.py -m gpu -x -q > $O/c14_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/c14_tests.log
for r in 1 2 3; do
for l in variants/libenf_oldpro.so -; do
  if [ "$l" = "-" ]; then unset ENF_HIP_LIB; else export ENF_HIP_LIB=$PWD/$l; fi
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-meta --no-ode --events-steps 0 --no-accuracy 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$l', d['ms_per_step'], d['split']['ms_fit'], d['split']['ms_decode'], d['final_fit_loss'])"
done
done 2>&1 | tee $O/c14_ab.log
