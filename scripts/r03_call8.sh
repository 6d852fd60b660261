#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_weight_grads.py -m gpu -x -q -k "native or invariants" 2>&1 | tail -2
for r in 1 2; do
for m in 0 1; do
  ENF_TRAIN_COMPOSED=$m timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-ode --events-steps 0 --no-accuracy 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('composed=$m', d['ms_per_step'], d['meta_step']['ms_per_step'], d['meta_step']['loss'])"
done
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/c8_prof -o meta -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-ode --events-steps 0 --no-accuracy > $GRAFT_REPO_ROOT/$O/c8_prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; f=$(find $O/c8_prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/c8_meta_kernel_stats.csv; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernels:",len(rows),"total ms",round(tot/1e6,1), "calls", sum(int(r['Calls']) for r in rows))
print("Cijk kernels:", [(r['Name'][:40], r['Calls']) for r in rows if r['Name'].startswith('Cijk')])
for r in rows[:16]: print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
