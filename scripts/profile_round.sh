#!/bin/bash
# Round profile set (run on the GPU box through gpurun): scripts/profile_round.sh TAG
#   1. rocprofv3 --kernel-trace --stats of bench.py           -> gpurun_out/prof_TAG/bench_kernel_stats.csv, bench.json
#   2. HBM traffic of the forward pair kernel (separate --pmc passes: FETCH_SIZE, WRITE_SIZE)
#   3. SQ counters of the forward pair kernel (scripts/pmc_k2.sh)
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_profiled.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/bench_profiled.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/hbm_$c -o p -- python3 $REPO/scripts/prof_fwd.py fwd > $OUT/hbm_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 $OUT/hbm_$c.log; exit 1; }
done
cd $REPO
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
acc = collections.defaultdict(list)
for f in glob.glob(out + "/hbm_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "enf_pair_fwd_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
print("median per launch:", med, {k: len(v) for k, v in acc.items()})
json.dump(med, open(out + "/hbm_counters.json", "w"))
PY
