#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for l in old new; do
  if [ "$l" = "new" ]; then unset ENF_HIP_LIB; else export ENF_HIP_LIB=$GRAFT_REPO_ROOT/variants/libenf_oldpro.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/c15_$l -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-meta --no-ode --events-steps 0 --no-accuracy > /dev/null 2>&1
  python3 - $GRAFT_REPO_ROOT/$O/c15_$l <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('prologue','tail','wz','sgd')):
        print(sys.argv[1][-3:], r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1))
PY
done
