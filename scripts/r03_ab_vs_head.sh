#!/bin/bash
# the working tree (default) against the previous commit (variant base, built from a worktree of HEAD): parity subset, then the headline bench
# with its per-kernel legs, interleaved on one box
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py tests/test_gpu_golden.py tests/test_gpu_reentrancy.py tests/test_gpu_weight_grads.py tests/test_gpu_bf16_contract.py tests/test_gpu_layers.py tests/test_gpu_narrow.py tests/test_gpu_ball.py -m gpu -x -q > $O/pk_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pk_tests.log
[ $rc = 0 ] || exit 1
for v in base default base default base default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  ENF_HIP_LIB=$L timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-meta --no-ode --events-steps 0 --kernel-iters 60 > $O/pk_$v.json 2>$O/pk_$v.err || { echo "bench $v failed"; tail -5 $O/pk_$v.err; exit 1; }
  python3 - $O/pk_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(sys.argv[2], 'ms/step', d['ms_per_step'], 'mse', '%.2e' % d['accuracy']['mse_vs_oracle'], {k:(v['launch_ms'],v['frac']) for k,v in d['roofline_kernels'].items()})
PY
done
