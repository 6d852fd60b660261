#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_golden.py -m gpu -x -q -k "fit_step or inner_loop or config1" 2>&1 | tail -3
for r in 1 2 3; do
for v in "0:-" "1:-" "1:variants/libenf_wz192.so" "1:variants/libenf_wz96.so"; do
  f=${v%%:*}; l=${v#*:}
  if [ "$l" = "-" ]; then unset ENF_HIP_LIB; else export ENF_HIP_LIB=$PWD/$l; fi
  ENF_FIT_STEP=$f timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-meta --no-ode --events-steps 0 --no-accuracy 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$f lib=$l', d['ms_per_step'], d['split']['ms_fit'], d['split']['ms_decode'], d['final_fit_loss'])"
done
done 2>&1 | tee $O/c11_ab.log
