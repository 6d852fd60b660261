#!/bin/bash
# quick same-box A/B of the working tree against the previous commit (variants/libenf_base.so): forward / golden parity, then steps + a kernel trace
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_golden.py tests/test_gpu_backward.py tests/test_gpu_weight_grads.py tests/test_gpu_layers.py -m gpu -x -q > $O/q_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/q_tests.log
[ $rc = 0 ] || exit 1
for v in base default base default base default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  ENF_HIP_LIB=$L timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-meta --no-ode --events-steps 0 --no-roofline --no-accuracy > $O/q_$v.json 2>$O/q_$v.err || { echo "bench $v failed"; tail -5 $O/q_$v.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/q_$v.json').readline()); print('$v', d['ms_per_step'])"
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in base default; do
  L=$R/variants/libenf_$v.so; [ $v = default ] && L=
  rm -rf $R/$O/q_prof_$v
  ENF_HIP_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/q_prof_$v -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-meta --no-ode --events-steps 0 --no-accuracy > /dev/null 2>&1
  python3 - $R/$O/q_prof_$v $v <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1]+'/p_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('tail','pair_fwd','prologue','pair_bwd')): print(sys.argv[2], r['Name'].split('(')[0][-62:], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
