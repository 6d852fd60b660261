#!/bin/bash
# the fit + decode step as a captured hipGraph (bench.py --graph-leg) beside the eager step, one box
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 400 python bench.py --graph-leg --steps 20 --warmup 5 --no-cpu-baseline --no-meta --no-ode --no-roofline --no-accuracy > $O/graph_leg.json 2> $O/graph_leg.err
echo "rc=$?"; tail -3 $O/graph_leg.err
python3 -c "
import json; d=json.loads(open('$O/graph_leg.json').readline()); print('wall', d['ms_per_step'], 'events median', d['timing'].get('events',{}).get('ms_median'), 'graph', d['timing'].get('graph'))"
