#!/bin/bash
# enf_fit_inputs: parity, then the step with (default) and without (ENF_FIT_INPUTS=0) the one-launch setup, interleaved on one box
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_sgd_update.py tests/test_gpu_configs.py tests/test_gpu_trainer.py -m gpu -x -q > $O/fi_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/fi_tests.log
[ $rc = 0 ] || exit 1
for r in 1 2 3; do for v in 0 1; do
  ENF_FIT_INPUTS=$v timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-meta --no-ode --no-roofline --no-accuracy --events-steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('fit_inputs=$v', d['ms_per_step'], d['timing']['events']['ms_median'])"
done; done
