#!/bin/bash
# config 2: the fit-shape forward as 256 runs of 16 latent steps (variant sk16) against the latent-split kernel (default), one box
O=gpurun_out/r03
mkdir -p $O
for v in sk16 default sk16 default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  ENF_HIP_LIB=$L timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-meta --no-ode --events-steps 0 > $O/c20_$v.json 2>$O/c20_$v.err || { echo "bench $v failed"; tail -5 $O/c20_$v.err; exit 1; }
  python3 - $O/c20_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(sys.argv[2], 'ms/step', d['ms_per_step'], 'acc', d.get('accuracy'), {k:(v.get('variant'),v['launch_ms'],v['frac']) for k,v in d['roofline_kernels'].items()})
PY
done
