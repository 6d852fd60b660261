#!/bin/bash
# stream-K z-fold forward: parity (forward / golden / variant tests) then config 3's per-kernel legs, runs of 72 against runs of 43 (= the
# three equal parts per tile of earlier in the round) on one box
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_golden.py tests/test_gpu_reentrancy.py tests/test_gpu_layers.py -m gpu -x -q > $O/c18_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/c18_tests.log
[ $rc = 0 ] || exit 1
for v in sk43 default sk43 default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  ENF_HIP_LIB=$L timeout -k 10 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-meta --no-ode --events-steps 0 --no-accuracy > $O/c18_$v.json 2>$O/c18_$v.err || { echo "bench $v failed"; tail -5 $O/c18_$v.err; exit 1; }
  python3 - $O/c18_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline())
ks=d.get('kernels') or d.get('roofline',{}).get('kernels') or {}
print(sys.argv[2], 'ms/step', d['ms_per_step'], 'value', d['value'])
def walk(o,pre=''):
    if isinstance(o,dict):
        if 'frac' in o and ('launch_ms' in o or 'ms' in o): print('   ',pre, {k:o[k] for k in o if k in ('frac','launch_ms','ms','variant')})
        for k,v in o.items(): walk(v,pre+'/'+k)
walk(d)
PY
done
