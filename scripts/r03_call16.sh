#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_weight_grads.py tests/test_gpu_ball.py tests/test_gpu_narrow.py -m gpu -x -q > $O/c16_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/c16_tests.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/c16_new -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-meta --no-ode --events-steps 0 --no-accuracy > /dev/null 2>&1
python3 - $R/$O/c16_new <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('prologue',)):
        print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1))
PY
