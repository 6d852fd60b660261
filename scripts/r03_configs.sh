#!/bin/bash
# every other BASELINE.json config end to end on one GPU + the outer step in f32 mode (the accuracy route of DESIGN.md item 7)
O=gpurun_out/r03_bench
mkdir -p $O
for c in 1 3 4 5; do
  timeout -k 10 600 python bench.py --config $c --no-cpu-baseline > $O/bench_c$c.json 2> $O/bench_c$c.err || { echo "config $c failed"; tail -5 $O/bench_c$c.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/bench_c$c.json').readline()); print('config $c', d['value'], d['unit'], d['ms_per_step'], 'ms', {k:(v.get('variant'),v['launch_ms'],v['frac']) for k,v in d.get('roofline_kernels',{}).items()}, (d.get('meta_step') or {}).get('ms_per_step'))"
done
timeout -k 10 600 python bench.py --precision f32 --steps 5 --warmup 2 --no-cpu-baseline --no-ode --events-steps 0 > $O/bench_c2_f32.json 2> $O/bench_c2_f32.err || { echo "f32 failed"; tail -5 $O/bench_c2_f32.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench_c2_f32.json').readline()); print('f32', d['value'], d['ms_per_step'], 'ms', d.get('meta_step'), d.get('accuracy'))"
