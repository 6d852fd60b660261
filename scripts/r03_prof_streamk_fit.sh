#!/bin/bash
# kernel durations of config 2's step with the fit-shape forward as runs of 16 latent steps (variant sk16) and with the default library
O=gpurun_out/r03
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in sk16 default; do
rm -rf $R/$O/c21_$v
L=$R/variants/libenf_$v.so; [ $v = default ] && L=
ENF_HIP_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/c21_$v -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-meta --no-ode --events-steps 0 --no-accuracy > /dev/null 2>&1
echo "== $v"
python3 - $R/$O/c21_$v <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'].split('(')[0][-70:], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1), r['Percentage'])
PY
done
