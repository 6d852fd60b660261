#!/bin/bash
O=gpurun_out/r03_bench
mkdir -p $O
timeout -k 10 600 python bench.py --config 4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err || { tail -5 $O/bench_c4.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench_c4.json').readline()); print('config 4', d['value'], d['ms_per_step'], 'ms', {k:(v.get('variant'),v['launch_ms'],v['frac']) for k,v in d.get('roofline_kernels',{}).items()}, (d.get('meta_step') or {}).get('ms_per_step'))"
