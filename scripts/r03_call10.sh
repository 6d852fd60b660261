#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_backward.py -m gpu -x -q > $O/c10_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/c10_tests.log
timeout -k 10 400 python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-meta > $O/c10_bench_c3.json 2>$O/c10_bench_c3.err; echo "bench c3 rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/c10_bench_c3.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['split']); print({k:(v['launch_ms'],v['frac'],v['variant']) for k,v in d['roofline_kernels'].items()}); print(d.get('accuracy'))
PY
