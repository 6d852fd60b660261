#!/bin/bash
# kernel durations of config 3's step (stream-K z-fold forward, merge, fold kernel)
O=gpurun_out/r03
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/c19
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/c19 -o p -- python3 $R/bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-meta --no-ode --events-steps 0 --no-accuracy > /dev/null 2>&1
python3 - $R/$O/c19 <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'].split('(')[0][-70:], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1), r['Percentage'])
PY
