#!/bin/bash
# prologue kernels: VALU (pro0) / MFMA both (pro3) / default (bwd only), at the headline config and config 4, one box
O=gpurun_out/r03
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in 2 4 3; do
for v in pro0 pro3; do
  rm -rf $R/$O/prologue_$v
  ENF_HIP_LIB=$R/variants/libenf_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prologue_$v -o p -- python3 $R/bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-meta --no-ode --events-steps 0 --no-accuracy > /dev/null 2>&1
  echo "cfg $cfg $v"
  python3 - $R/$O/prologue_$v <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'prologue' in r['Name']:
        print('  ', r['Name'].split('::')[-1][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1))
PY
done; done
cd $R
for v in pro0 pro3 default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  for i in 1 2; do ENF_HIP_LIB=$L python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-meta --no-ode --events-steps 0 --no-accuracy | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'])"; done
done
