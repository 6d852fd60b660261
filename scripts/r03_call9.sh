#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_golden.py tests/test_gpu_weight_grads.py tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | tail -2
AB_ROUNDS=3 timeout -k 10 600 python scripts/ab_kernels.py variants/libenf_k2nospec.so - variants/libenf_ns2.so 2>&1 | tee $O/c9_ab.log
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-ode --events-steps 0 --no-accuracy 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('meta', d['meta_step'])"
bash scripts/pmc_k3.sh r03 > $O/c9_pmc.log 2>&1; tail -30 gpurun_out/pmc_r03/summary.txt | head -40
