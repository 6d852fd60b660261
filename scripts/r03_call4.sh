#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
bash scripts/k3_race/form_probe.sh 4000
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/c4_bench.json 2> $O/c4_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/c4_bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','timing','accuracy')}); print(d['roofline']['launch_ms'], d['roofline']['frac'], d['meta_step'])
PY
