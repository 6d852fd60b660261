#!/bin/bash
# The round's closing measurement, ONE box: smoke, default bench, the same bench under rocprofv3 --kernel-trace --stats, the per-kernel legs alone
# under the profiler (so that a kernel's average there is over exactly the launches bench.py's events time).  Usage: scripts/r03_final.sh [suite]
O=gpurun_out/r03_final
mkdir -p $O
R=$PWD
if [ "$1" = "suite" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; tail -3 $O/gpu_tests.log > $O/gpu_tests_tail.txt; cat $O/gpu_tests_tail.txt
  [ $rc = 0 ] || exit 1
fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -2 $O/smoke.log &&
python bench.py > $O/bench.json 2> $O/bench.err && cut -c1-400 $O/bench.json &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/bench_profiled.json 2> $R/$O/bench_profiled.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_roofline -o p -- python3 $R/bench.py --roofline-only > $R/$O/roofline_only.json 2> $R/$O/roofline_only.err &&
cd $R && cp $O/prof_bench/p_kernel_stats.csv $O/bench_kernel_stats.csv && cp $O/prof_roofline/p_kernel_stats.csv $O/roofline_only_kernel_stats.csv &&
python3 - <<'PY'
import csv, json
d = json.loads(open('gpurun_out/r03_final/roofline_only.json').readline())
print('roofline-only legs:', {k: v['launch_ms'] for k, v in d['roofline_kernels'].items()})
for r in csv.DictReader(open('gpurun_out/r03_final/roofline_only_kernel_stats.csv')):
    if 'pair_' in r['Name']: print(r['Name'].split('(')[0][-60:], r['Calls'], round(float(r['AverageNs']) / 1e6, 4), 'ms')
PY
