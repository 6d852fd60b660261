#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_bf16_contract.py tests/test_gpu_ode_trainer.py tests/test_gpu_layers.py tests/test_gpu_reentrancy.py -m gpu -x -q > $O/c7_trainer.log 2>&1; echo "trainer rc=$?"; tail -8 $O/c7_trainer.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --events-steps 0 --no-accuracy > $O/c7_bench.json 2> $O/c7_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/c7_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['meta_step'])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/c7_prof -o meta -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-ode --events-steps 0 --no-accuracy > $GRAFT_REPO_ROOT/$O/c7_prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; f=$(find $O/c7_prof -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernels:",len(rows),"total ms",tot/1e6, "calls", sum(int(r['Calls']) for r in rows))
print("Cijk kernels:", [(r['Name'][:40], r['Calls']) for r in rows if r['Name'].startswith('Cijk')])
for r in rows[:22]: print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
cp "$f" $O/c7_meta_kernel_stats.csv
